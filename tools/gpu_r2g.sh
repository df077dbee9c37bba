#!/bin/bash
TAG=${1:-r2g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -n 3 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | cut -c1-300 | head -30
timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -1 > $OUT/bench_conformer.json; cut -c1-300 $OUT/bench_conformer.json
bash tools/gpu_conf_prof.sh $TAG > $OUT/conformer_kernels.txt 2>&1; head -30 $OUT/conformer_kernels.txt | cut -c1-130
