#!/bin/bash
# first visit of the 256-wide weight-gradient kernel: probe + parity tests, microbench, bench A/B, kernel trace
TAG=${1:-w256}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_wgrad256.py -q -x --timeout 120 -p no:cacheprovider > $OUT/pytest_w256.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/pytest_w256.log | cut -c1-220
timeout 200 python tools/wgrad256_bench.py --grids ${2:-0,248,224,192} > $OUT/w256_bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/w256_bench.log | tail -3 | cut -c1-1200
for v in 0 1; do
  OTR_WGRAD256=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$v.log 2>&1; echo "bench.py OTR_WGRAD256=$v exit $?"; grep -v amdgpu.ids $OUT/bench_$v.log | tail -1 | cut -c1-400
done
R=$PWD
(cd /tmp && OTR_WGRAD256=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/prof_summary.py $OUT/prof/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; head -14 $OUT/kernel_summary_graph.txt | cut -c1-170
