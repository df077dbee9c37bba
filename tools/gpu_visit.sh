#!/bin/bash
# one GPU-box visit: [tests] [bench] [pmc] [trace] selected by words in $2..; results under gpurun_out/$1/
# usage: gpu_visit.sh TAG tests bench pmc trace ffn
TAG=${1:-v}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
for what in "$@"; do
case $what in
tests)
  timeout 900 python -m pytest tests -m gpu -q -n 3 --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_gpu.log | sed -e 's/ - .*//' | cut -c1-300 | head -30
  cp gpurun_out/parity_*.json gpurun_out/decode_validity_*.json $OUT/ 2>/dev/null ;;
newtests)
  timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_wgrad256.py tests/test_gpu_rowblock.py -m gpu -q -n 3 --timeout 600 -p no:cacheprovider > $OUT/pytest_new.log 2>&1
  echo "pytest(new) exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_new.log | sed -e 's/ - .*//' | cut -c1-400 | head -30
  cp gpurun_out/parity_*.json gpurun_out/decode_validity_*.json $OUT/ 2>/dev/null ;;
smoke)
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; grep -v amdgpu.ids $OUT/smoke.log | tail -3 ;;
bench)
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?"
  grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err | cut -c1-300 ;;
benchq)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-line > $OUT/benchq.log 2> $OUT/benchq.err; echo "benchq exit $?"
  grep '^{' $OUT/benchq.log | tail -1 > $OUT/benchq.json; cut -c1-300 $OUT/benchq.json; tail -3 $OUT/benchq.err | cut -c1-300 ;;
pmc)
  bash tools/gpu_pmc_step.sh $TAG ;;
trace)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o graph -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
  DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
  python tools/graph_gaps.py $DB $OUT/step_kernels.json > $OUT/graph_gaps.txt 2>&1
  python tools/prof_summary.py $DB 9 > $OUT/kernel_summary_graph.txt 2>&1; head -45 $OUT/kernel_summary_graph.txt | cut -c1-180 ;;
ffn)
  timeout 600 python -m pytest tests/test_gpu_ffn_fused.py -x -q -p no:cacheprovider > $OUT/ffn_test.log 2>&1; echo "ffn pytest exit $?"
  tail -15 $OUT/ffn_test.log | cut -c1-300
  timeout 300 python tools/ffn_bench.py --mode fp16 > $OUT/ffn_bench.log 2>&1; echo "ffn bench exit $?"; grep -v amdgpu.ids $OUT/ffn_bench.log | tail -3 | cut -c1-1500 ;;
conformer)
  timeout 300 python bench.py --model conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_conformer.log 2>&1; echo "conformer exit $?"
  grep '^{' $OUT/bench_conformer.log | tail -1 | cut -c1-300 ;;
*) echo "unknown step $what" ;;
esac
done
