#!/bin/bash
# kernel trace of the C5 decode bench (bench.py --task decode): per-kernel totals of the cached step
TAG=${1:-dtrace}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout 300 python bench.py --task decode --no-cpu-baseline > $OUT/decode.log 2>&1; grep '^{' $OUT/decode.log | tail -1 > $OUT/decode.json; cut -c1-400 $OUT/decode.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o dec -- python $R/bench.py --task decode --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
python tools/prof_summary.py $DB 1 > $OUT/decode_kernels.txt 2>&1; head -50 $OUT/decode_kernels.txt | cut -c1-170
