#!/bin/bash
TAG=${1:-dg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 python tools/conv2_dgrad_bench.py > $OUT/dgrad_bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/dgrad_bench.log | tail -3 > $OUT/conv2_dgrad_bench.json; cat $OUT/conv2_dgrad_bench.json
