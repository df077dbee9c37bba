#!/bin/bash
TAG=${1:-g}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -n 2 --timeout 300 -p no:cacheprovider -k "linear or gemm or conv" > $OUT/pytest_gemm.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gemm.log | sed -e 's/ - .*//' | head -20
timeout 300 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/gemm_bench.log | head -20
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-400
