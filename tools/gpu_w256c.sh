#!/bin/bash
# ablations + PMC of the 256-wide weight-gradient kernel
TAG=${1:-w256c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python tools/wgrad256_bench.py --grids -248 --ablate 1,2,3,4,5,6,7 > $OUT/w256_ablate.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/w256_ablate.log | tail -3 | cut -c1-1800
bash tools/gpu_pmc2.sh $TAG/pmc_plain '%wgrad256_kernel%' tools/wgrad256_bench.py --reps 2 --grids -248 2>&1 | grep -v "^$" | cut -c1-200
bash tools/gpu_pmc2.sh $TAG/pmc_aligned '%wgrad256_kernel%' tools/wgrad256_bench.py --reps 2 --grids 0 2>&1 | grep -v "^$" | cut -c1-200
