#!/bin/bash
# usage: gpu_round3.sh TAG COMMIT
# round-3 evidence visit: everything the bench line and DESIGN.md cite, into gpurun_out/$1/ (copied to profiles/r03_* afterwards)
TAG=${1:-r3final}; COMMIT=${2:-working-tree}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_pmc_step.sh $TAG
python - <<PY
import json, subprocess
f = '$OUT/pmc_step.json'
d = json.load(open(f))
d['_meta'] = {'commit': '$COMMIT',
              'command': 'tools/gpu_pmc_step.sh (bench.py --steps 2 --warmup 1 --no-graph, three --pmc passes)'}
json.dump(d, open(f, 'w'), indent=1)
PY
cp $OUT/pmc_step.json profiles/r03_pmc_step.json        # the bench line below reads it
bash tools/gpu_visit.sh $TAG tests smoke bench trace conformer
timeout 300 python tools/ffn_bench.py --mode fp16 > $OUT/ffn_bench.log 2>&1; grep '^{' $OUT/ffn_bench.log | tail -1 > $OUT/ffn_bench.json
timeout 300 python bench.py --task decode --no-cpu-baseline > $OUT/decode.log 2>&1; grep '^{' $OUT/decode.log | tail -1 > $OUT/decode.json; cut -c1-300 $OUT/decode.json
