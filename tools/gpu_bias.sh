#!/bin/bash
TAG=${1:-bias}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_wgrad256.py tests/test_gpu_rowblock.py -q --timeout 120 -p no:cacheprovider > $OUT/pytest_w.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_w.log | cut -c1-250 | head -30
timeout 900 python -m pytest tests -m gpu -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest all exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | cut -c1-250 | head -30
for v in 0 1; do
OTR_NO_FUSED_BIAS_COLSUM=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$v.log 2>&1; echo "bench NO_FUSED_BIAS=$v exit $?"; grep -v amdgpu.ids $OUT/bench_$v.log | tail -1 | cut -c1-330
done
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_rb -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py /tmp/prof_rb/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; head -40 $OUT/kernel_summary_graph.txt | cut -c1-150
