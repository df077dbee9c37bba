#!/bin/bash
TAG=${1:-w256d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_wgrad256.py -q --timeout 120 -p no:cacheprovider > $OUT/pytest_w256.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/pytest_w256.log | cut -c1-220
timeout 200 python tools/wgrad256_bench.py --grids 0,-248,-232 --ablate 1,2,3,4,5,6,7 > $OUT/w256_ablate.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/w256_ablate.log | tail -3 | cut -c1-1800
bash tools/gpu_pmc3.sh $TAG '%wgrad256_kernel%' tools/wgrad256_one.py -248 3 2>&1 | grep -v "^$" | cut -c1-200
