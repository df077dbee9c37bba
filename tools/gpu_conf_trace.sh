#!/bin/bash
# kernel trace of the conformer bench step: gpurun_out/$1/kernel_summary_conformer.txt + step sequence
TAG=${1:-conf}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o graph -- python $R/bench.py --model conformer --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $R/$OUT/rocprof_conf.log 2>&1; echo "rocprof exit $?")
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python tools/graph_gaps.py $DB > $OUT/graph_gaps_conformer.txt 2>&1
python tools/prof_summary.py $DB 8 > $OUT/kernel_summary_conformer.txt 2>&1; head -50 $OUT/kernel_summary_conformer.txt | cut -c1-200
