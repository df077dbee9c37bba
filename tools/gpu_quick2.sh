#!/bin/bash
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "conformer_full" --timeout 500 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed|^E  " $OUT/pytest.log | cut -c1-400 | head
timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -2 | cut -c1-900
