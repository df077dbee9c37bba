#!/bin/bash
# closing visit of the second batch: full GPU suite, smoke, bench with the CPU baseline, hipGraph kernel trace, Conformer bench + per-step table
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -n 3 --timeout 300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | cut -c1-300 | head -30
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 > $OUT/bench.json; cut -c1-700 $OUT/bench.json
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/graph_gaps.py /tmp/prof_fin/graph_results.db | tail -1
python tools/prof_summary.py /tmp/prof_fin/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; head -8 $OUT/kernel_summary_graph.txt | cut -c1-150
timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"; grep -v amdgpu.ids $OUT/bench_conformer.log | tail -1 > $OUT/bench_conformer.json; cut -c1-300 $OUT/bench_conformer.json
timeout 120 python tools/conv2_dgrad_bench.py > $OUT/dgrad_bench.log 2>&1; grep -v amdgpu.ids $OUT/dgrad_bench.log | tail -1 > $OUT/conv2_dgrad_bench.json; cat $OUT/conv2_dgrad_bench.json
bash tools/gpu_conf_prof.sh $TAG > $OUT/conformer_kernels.txt 2>&1; head -34 $OUT/conformer_kernels.txt | cut -c1-130
