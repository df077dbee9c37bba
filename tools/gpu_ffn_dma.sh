#!/bin/bash
TAG=${1:-ffndma}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python tools/pytest_dbg.py 5=16 tests/test_gpu_ffn_fused.py tests/test_gpu_model.py -q --timeout 200 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | cut -c1-250 | head -20
for kv in "5=4" "5=16" "5=4" "5=16"; do timeout 200 python tools/bench_dbg.py $kv --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200 | sed "s/^/[$kv] /"; done
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_rb -o graph -- python $R/tools/bench_dbg.py 5=16 --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py /tmp/prof_rb/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; grep -E "ffn_|total kernel" $OUT/kernel_summary_graph.txt | cut -c1-150
