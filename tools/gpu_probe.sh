#!/bin/bash
TAG=${1:-p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== graph probe c1"; timeout 300 python tools/graph_probe.py c1 > $OUT/probe_c1.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/probe_c1.log | tail -25
echo "== graph probe c2"; timeout 600 python tools/graph_probe.py c2 > $OUT/probe_c2.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/probe_c2.log | tail -40
echo "== rocprof eager"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o eager -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1; echo "exit $?"
cd $GRAFT_REPO_ROOT; tail -2 $OUT/rocprof.log | cut -c1-400; find $OUT/prof -name "*stats*" | head; 
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
