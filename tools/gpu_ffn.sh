#!/bin/bash
# fused FFN: unit tests + microbenchmark
mkdir -p gpurun_out/ffn
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ffn_fused.py -x -q > gpurun_out/ffn/test.log 2>&1; echo "pytest exit $?"
tail -25 gpurun_out/ffn/test.log | cut -c1-400
timeout 300 python tools/ffn_bench.py > gpurun_out/ffn/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids gpurun_out/ffn/bench.log | tail -5 | cut -c1-1200
