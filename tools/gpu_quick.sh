#!/bin/bash
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | sed -e 's/ - .*//' | head -40
echo "== bench (hipGraph)"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -3 | cut -c1-1600
