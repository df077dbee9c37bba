#!/bin/bash
# numbers for the docs: full bench (with cpu baseline), conformer bench, decode bench, gemm bench
TAG=${1:-docs}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-2600
timeout 600 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -1 | cut -c1-700
timeout 600 python tools/decode_bench.py --batch 8 --cpu-baseline > $OUT/decode_bench.log 2>&1; echo "decode exit $?"; grep -v amdgpu.ids $OUT/decode_bench.log | tail -1 | cut -c1-1500
timeout 300 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1; grep -v amdgpu.ids $OUT/gemm_bench.log | head -18
