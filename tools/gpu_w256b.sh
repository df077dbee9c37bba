#!/bin/bash
# second visit: parity tests (ragged shapes, XCD-major chunks), microbench sweep, PMC passes, bench A/B, kernel trace
TAG=${1:-w256b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_wgrad256.py -q --timeout 120 -p no:cacheprovider > $OUT/pytest_w256.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest_w256.log | cut -c1-220
timeout 200 python tools/wgrad256_bench.py --grids ${2:-0,-256,-248,-232,208,192} > $OUT/w256_bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/w256_bench.log | tail -3 | cut -c1-1800
bash tools/gpu_pmc2.sh $TAG/pmc '%wgrad256_kernel%' python tools/wgrad256_bench.py --reps 2 --grids 0 2>&1 | grep -v "^$" | cut -c1-200
for v in 1; do
  OTR_WGRAD256=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$v.log 2>&1; echo "bench.py OTR_WGRAD256=$v exit $?"; grep -v amdgpu.ids $OUT/bench_$v.log | tail -1 | cut -c1-400
done
R=$PWD
(cd /tmp && OTR_WGRAD256=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py $OUT/prof/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; head -30 $OUT/kernel_summary_graph.txt | cut -c1-170
