#!/bin/bash
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest $@"
timeout 900 python -m pytest "$@" -m gpu -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | head -60
