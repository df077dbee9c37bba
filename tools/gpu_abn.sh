#!/bin/bash
# A/B/C... on one box: the timed region of bench.py under several values of ONE environment variable, two alternating rounds
# usage: gpu_abn.sh TAG VAR VALUE...     e.g.  gpu_abn.sh s1 OTR_SWITCHES ops._DEC_TOUCH=1 ops._DEC_TOUCH=0 ops._DEC_FFN_SLICES=16
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
  for v in "$@"; do
    f=$OUT/bench_$(echo "$v" | tr '/=,' '___')_$i.log
    env $VAR=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $f 2>&1
    echo "$VAR=$v run $i: $(grep '^{' $f | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "utt/s", round(d["ms_per_step"],3), "ms")')"
  done
done
