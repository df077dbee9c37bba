#!/bin/bash
# everything the round needs from one box: GPU test-suite, the bench line (with cpu_baseline), rocprofv3 kernel trace of
# the same command; logs under gpurun_out/full/
OUT=gpurun_out/full
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/test.log 2>&1; echo "pytest exit $?"
tail -6 $OUT/test.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -2 | cut -c1-6000
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py $OUT/prof/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; head -45 $OUT/kernel_summary_graph.txt | cut -c1-230
rm -rf $OUT/prof
