#!/bin/bash
# last check of a HEAD: full GPU suite, smoke, bench line (no CPU leg)
TAG=${1:-fin}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -q -n 3 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | cut -c1-300 | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-300
