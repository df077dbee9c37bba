#!/bin/bash
TAG=${1:-r2f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_round2b.py "tests/test_gpu_ops.py::test_conv_subsample" -m gpu -q -n 3 --timeout 200 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | sed -e 's/ - .*//' | cut -c1-300 | head -20
timeout 120 python tools/conv2_dgrad_bench.py > $OUT/dgrad_bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/dgrad_bench.log | tail -1 > $OUT/conv2_dgrad_bench.json; cat $OUT/conv2_dgrad_bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-300
