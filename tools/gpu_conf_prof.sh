#!/bin/bash
TAG=${1:-cp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o graph -- python $R/bench.py --model conformer --steps 4 --warmup 2 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
grep -v amdgpu.ids $OUT/rocprof.log | tail -1 | cut -c1-200
python tools/prof_summary.py $OUT/prof/graph_results.db 4 > $OUT/kernel_summary_graph.txt 2>&1; head -42 $OUT/kernel_summary_graph.txt | cut -c1-175
