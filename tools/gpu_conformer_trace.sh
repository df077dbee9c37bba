#!/bin/bash
# kernel trace of the Conformer training step (bench.py --model conformer): per-kernel totals + the launch sequence of one step
TAG=${1:-ctrace}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; grep '^{' $OUT/bench_conformer.log | tail -1 > $OUT/bench_conformer.json; cut -c1-300 $OUT/bench_conformer.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o conf -- python $R/bench.py --model conformer --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python tools/graph_gaps.py $DB $OUT/step_kernels.json > $OUT/graph_gaps.txt 2>&1
python tools/prof_summary.py $DB 9 > $OUT/conformer_kernels.txt 2>&1; head -60 $OUT/conformer_kernels.txt | cut -c1-200
