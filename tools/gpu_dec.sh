#!/bin/bash
TAG=${1:-dec}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_headline.py -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | cut -c1-250 | head -30
for v in 1 0; do
OTR_NO_ROWBLOCK=$v timeout 600 python tools/decode_bench.py --batch 8 > $OUT/decode_bench_norb$v.log 2>&1; echo "decode NO_ROWBLOCK=$v exit $?"; grep -v amdgpu.ids $OUT/decode_bench_norb$v.log | tail -1 | cut -c1-900
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-2500
