#!/bin/bash
# bench + graph profile only (no tests): quick perf iteration
TAG=${1:-bp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-330
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py $OUT/prof/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; grep -E "${2:-add_ln}" $OUT/kernel_summary_graph.txt | cut -c1-150
