#!/bin/bash
# perf-focused GPU visit: selected tests, sweep, bench
TAG=${1:-p}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | head -40
timeout 600 python tools/gemm_sweep.py > $OUT/gemm_sweep.log 2>&1; echo "sweep exit $?"; grep -v amdgpu.ids $OUT/gemm_sweep.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -2 | cut -c1-1200
