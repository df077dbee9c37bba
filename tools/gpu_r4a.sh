#!/bin/bash
# round-4 first visit: full -m gpu suite (incl. the new round-4 tests), quick bench
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_visit.sh $TAG tests benchq
cp gpurun_out/decode_eos_live_*.json $OUT/ 2>/dev/null
