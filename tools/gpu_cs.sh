#!/bin/bash
TAG=${1:-cs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_wgrad256.py tests/test_gpu_rowblock.py tests/test_gpu_dp.py -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | cut -c1-250 | head -30
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-330
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_rb -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py /tmp/prof_rb/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; grep -E "colsum|total kernel|wgrad256|grouped|rb_linear|64, 64, true, true" $OUT/kernel_summary_graph.txt | cut -c1-150
