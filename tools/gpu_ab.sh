#!/bin/bash
# A/B on one box: the timed region of bench.py under two values of an environment switch, alternating
# usage: gpu_ab.sh TAG VAR [VALUE_A VALUE_B]     e.g.  gpu_ab.sh ab1 OTR_DEBUG_SET 13=0 13=1
TAG=$1; VAR=$2; VA=${3:-0}; VB=${4:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  for v in "$VA" "$VB"; do
    f=$OUT/bench_$(echo "$v" | tr '/=' '__')_$i.log
    env $VAR=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $f 2>&1
    echo "$VAR=$v run $i: $(grep '^{' $f | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "utt/s", round(d["ms_per_step"],3), "ms")')"
  done
done
