#!/bin/bash
TAG=${1:-conf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -1 | cut -c1-400
bash tools/gpu_conf_prof.sh $TAG 2>&1 | head -50 | cut -c1-160
