#!/bin/bash
TAG=${1:-w256g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_wgrad256.py -q --timeout 120 -p no:cacheprovider > $OUT/pytest_w256.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/pytest_w256.log | cut -c1-220
timeout 200 python tools/wgrad256_bench.py --grids 0,-248,0 --ablate 48,16 > $OUT/w256.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/w256.log | tail -1 | cut -c1-1200
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-200; done
