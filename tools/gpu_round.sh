#!/bin/bash
# full GPU test-suite + bench line; logs under gpurun_out/round/
mkdir -p gpurun_out/round
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/round/test.log 2>&1; echo "pytest exit $?"
tail -15 gpurun_out/round/test.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/round/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids gpurun_out/round/bench.log | tail -3 | cut -c1-3000
