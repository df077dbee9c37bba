#!/bin/bash
TAG=${1:-sm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; grep -v amdgpu.ids $OUT/smoke.log | tail -4
timeout 300 python bench.py --mode fp32 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_fp32.log 2>&1; echo "fp32 bench exit $?"; grep -v amdgpu.ids $OUT/bench_fp32.log | tail -1 | cut -c1-300
timeout 300 python bench.py --no-graph --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_eager.log 2>&1; echo "eager bench exit $?"; grep -v amdgpu.ids $OUT/bench_eager.log | tail -1 | cut -c1-300
