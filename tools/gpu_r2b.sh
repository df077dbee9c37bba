#!/bin/bash
# second-batch kernels: their tests, the copy-origin profile, headline and conformer bench lines
TAG=${1:-r2b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_round2b.py tests/test_gpu_ops.py tests/test_gpu_dropout.py -m gpu -q -x -n 3 --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | sed -e 's/ - .*//' | head -30
timeout 200 python tools/copy_origin.py > $OUT/copy_origin.log 2>&1; echo "copy_origin exit $?"; grep -v amdgpu.ids $OUT/copy_origin.log | tail -70 | cut -c1-260
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-300
timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -1 | cut -c1-300
