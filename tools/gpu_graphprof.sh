#!/bin/bash
TAG=${1:-gp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
ls $OUT/prof | head; python tools/graph_gaps.py $OUT/prof/graph_results.db
python tools/prof_summary.py $OUT/prof/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; head -30 $OUT/kernel_summary_graph.txt | cut -c1-200
grep -v amdgpu.ids $OUT/rocprof.log | tail -2 | cut -c1-300
