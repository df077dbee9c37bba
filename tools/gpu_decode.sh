#!/bin/bash
# decode-focused GPU visit: decode parity tests, then the C5 decode benchmark
TAG=${1:-d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q -x --timeout 300 -p no:cacheprovider > $OUT/pytest_decode.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_decode.log | sed -e 's/ - .*//' | head -40
timeout 600 python tools/decode_bench.py --batch 8 ${DECODE_ARGS} > $OUT/decode_bench.log 2>&1; echo "decode bench exit $?"
grep -v amdgpu.ids $OUT/decode_bench.log | tail -5 | cut -c1-2000
