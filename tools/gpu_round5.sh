#!/bin/bash
# usage: gpu_round5.sh TAG COMMIT [steps...]
# round-5 evidence visit: everything the bench line and DESIGN.md cite, into gpurun_out/$TAG/ (copied to profiles/r05_* afterwards)
TAG=${1:-r5}; COMMIT=${2:-working-tree}; shift; shift
WHAT=${@:-tests smoke bench trace pmc two conformer decode}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/tolerance_cases.jsonl
for w in $WHAT; do
case $w in
pmc)
  bash tools/gpu_pmc_step.sh $TAG > $OUT/pmc_stdout.txt 2>&1
  python - <<PY
import json
f = '$OUT/pmc_step.json'
d = json.load(open(f))
d['_meta'] = {'commit': '$COMMIT', 'command': 'tools/gpu_pmc_step.sh (bench.py --steps 2 --warmup 1 --no-graph, three --pmc passes); records keyed by kernel, grid size and duration class'}
json.dump(d, open(f, 'w'), indent=1)
PY
  head -30 $OUT/pmc_step.txt | cut -c1-160 ;;
tests|smoke|bench|benchq|trace|conformer|newtests)
  bash tools/gpu_visit.sh $TAG $w
  cp gpurun_out/tolerance_cases.jsonl gpurun_out/decode_eos_live_*.json $OUT/ 2>/dev/null ;;
two)
  # the N > 1 control flow of bench.py end to end on ONE GPU: two ranks on cuda:0, collectives over gloo (numbers mean nothing)
  OTR_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 4 --warmup 2 --calib-steps 3 --no-cpu-baseline > $OUT/bench_2rank_onegpu_gloo.log 2>&1; echo "two-rank exit $?"
  grep '^{' $OUT/bench_2rank_onegpu_gloo.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('value','n_gpus','ms_per_step','step_breakdown','INVALID')})[:1500])" ;;
decode)
  timeout 300 python bench.py --task decode --no-cpu-baseline > $OUT/decode.log 2>&1; grep '^{' $OUT/decode.log | tail -1 > $OUT/decode.json; cut -c1-300 $OUT/decode.json ;;
ctrace)
  bash tools/gpu_conformer_trace.sh ${TAG}_c > $OUT/ctrace_stdout.txt 2>&1; cp gpurun_out/${TAG}_c/conformer_kernels.txt $OUT/ 2>/dev/null
  # (r06: straight to its own name -- copied as graph_gaps.txt first it replaced the TRAIN step's sequence of the `trace` step of the same visit)
  cp gpurun_out/${TAG}_c/graph_gaps.txt $OUT/conformer_step_sequence.txt 2>/dev/null; cp gpurun_out/${TAG}_c/bench_conformer.log $OUT/ 2>/dev/null; head -12 $OUT/conformer_kernels.txt | cut -c1-160 ;;
dtrace)
  bash tools/gpu_decode_trace.sh ${TAG}_d > $OUT/dtrace_stdout.txt 2>&1; cp gpurun_out/${TAG}_d/decode_kernels.txt gpurun_out/${TAG}_d/decode.json $OUT/ 2>/dev/null; cut -c1-300 $OUT/decode.json; head -14 $OUT/decode_kernels.txt | cut -c1-160 ;;
seqtests)
  # exactly what the driver runs at round end: sequential, stop at the first failure
  timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest(sequential) exit $?"; tail -3 $OUT/pytest_gpu.log | cut -c1-200
  cp gpurun_out/parity_*.json gpurun_out/decode_validity_*.json gpurun_out/tolerance_cases.jsonl gpurun_out/decode_eos_live_*.json $OUT/ 2>/dev/null ;;
ffn)
  timeout 300 python tools/ffn_bench.py --mode fp16 > $OUT/ffn_bench.log 2>&1; grep '^{' $OUT/ffn_bench.log | tail -1 > $OUT/ffn_bench.json; cut -c1-300 $OUT/ffn_bench.json ;;
esac
done
