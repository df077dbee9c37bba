#!/bin/bash
# what the dropout masks cost: the timed region of bench.py with residual_dropout 0.1 (the metric) and 0.0, alternating, same box
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-drop}; mkdir -p $OUT
for i in 1 2 3; do for p in 0.1 0.0; do
  ms=$(timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --dropout $p 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "dropout $p run $i: $ms ms" | tee -a $OUT/dropout_ab.txt
done; done
