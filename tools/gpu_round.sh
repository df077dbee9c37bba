#!/bin/bash
# One GPU-box visit: GPU parity tests (crash-isolated by xdist), smoke, GEMM microbench, bench (hipGraph), rocprof.
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -n 2 --timeout 300 -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | head -40
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
echo "== gemm bench"
timeout 300 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/gemm_bench.log
echo "== bench (hipGraph)"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -3 | cut -c1-1700
echo "== rocprof (eager, 3 steps)"
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o eager -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py $OUT/prof/eager_results.db 4 > $OUT/kernel_summary.txt 2>&1; head -40 $OUT/kernel_summary.txt
