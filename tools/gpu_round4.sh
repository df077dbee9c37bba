#!/bin/bash
# usage: gpu_round4.sh TAG COMMIT [steps...]
# round-4 evidence visit: everything the bench line and DESIGN.md cite, into gpurun_out/$TAG/ (copied to profiles/r04_* afterwards)
TAG=${1:-r4final}; COMMIT=${2:-working-tree}; shift; shift
WHAT=${@:-pmc tests smoke bench trace conformer decode ffn dectrace eatrace ffntrace}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for w in $WHAT; do
case $w in
pmc)
  bash tools/gpu_pmc_step.sh $TAG > $OUT/pmc_stdout.txt 2>&1
  python - <<PY
import json
f = '$OUT/pmc_step.json'
d = json.load(open(f))
d['_meta'] = {'commit': '$COMMIT', 'command': 'tools/gpu_pmc_step.sh (bench.py --steps 2 --warmup 1 --no-graph, three --pmc passes); records keyed by kernel and grid size'}
json.dump(d, open(f, 'w'), indent=1)
PY
  cp $OUT/pmc_step.json profiles/r04_pmc_step.json; head -30 $OUT/pmc_step.txt | cut -c1-160 ;;
tests|smoke|bench|trace|conformer)
  bash tools/gpu_visit.sh $TAG $w ;;
decode)
  timeout 300 python bench.py --task decode --no-cpu-baseline > $OUT/decode.log 2>&1; grep '^{' $OUT/decode.log | tail -1 > $OUT/decode.json; cut -c1-300 $OUT/decode.json ;;
ffn)
  timeout 300 python tools/ffn_bench.py --mode fp16 > $OUT/ffn_bench.log 2>&1; grep '^{' $OUT/ffn_bench.log | tail -1 > $OUT/ffn_bench.json; cut -c1-300 $OUT/ffn_bench.json ;;
eatrace)
  timeout 300 python tools/encattn_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/encattn_trace.txt; cat $OUT/encattn_trace.txt ;;
ffntrace)
  timeout 300 python tools/ffn3_trace.py slab save 2>&1 | grep -v amdgpu.ids > $OUT/ffn3_slab_trace.txt; head -8 $OUT/ffn3_slab_trace.txt; tail -5 $OUT/ffn3_slab_trace.txt ;;
dectrace)
  timeout 300 python tools/dec_trace.py > $OUT/dec_trace.txt 2>&1; tail -40 $OUT/dec_trace.txt ;;
esac
done
cp gpurun_out/decode_eos_live_*.json $OUT/ 2>/dev/null
