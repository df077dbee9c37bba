#!/bin/bash
TAG=${1:-r2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_round2b.py "tests/test_gpu_ops.py::test_conv_subsample" tests/test_gpu_dropout.py tests/test_gpu_model.py -m gpu -q -n 3 --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest.log | sed -e 's/ - .*//' | cut -c1-300 | head -30
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench.log | tail -1 | cut -c1-300
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python tools/graph_gaps.py /tmp/prof_fin/graph_results.db | tail -1
python tools/prof_summary.py /tmp/prof_fin/graph_results.db 6 > $OUT/kernel_summary_graph.txt 2>&1; grep -E "conv|sqnorm|relu|col2im|adam|pack_frags|transpose" $OUT/kernel_summary_graph.txt | cut -c1-150
