#!/bin/bash
# A/B on one box: bench with and without an env switch, alternating
TAG=$1; VAR=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
  for v in 0 1; do
    env $VAR=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_${v}_$i.log 2>&1
    echo "$VAR=$v run $i: $(grep -v amdgpu.ids $OUT/bench_${v}_$i.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "utt/s", round(d["ms_per_step"],3), "ms")')"
  done
done
