#!/bin/bash
TAG=${1:-cf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -n 2 --timeout 300 -p no:cacheprovider -k "conformer or dwconv or bn_swish or c4 or relpos or attention" > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | head -20
timeout 600 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -1 | cut -c1-330
