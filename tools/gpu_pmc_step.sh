#!/bin/bash
# PMC evidence for the hot kernels INSIDE the training step: separate rocprofv3 --kernel-trace --pmc passes (SQ set, FETCH_SIZE,
# WRITE_SIZE) over an eager bench.py run -> gpurun_out/$TAG/pmc_step.json + .txt (copy to profiles/rNN_pmc_step.*).
# usage: gpu_pmc_step.sh TAG [bench.py flags...]
TAG=${1:-pmc}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
run() { # pass name, counters
  rm -rf /tmp/pmc_$1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmc_$1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras "${@:3}" > $R/$OUT/$1.log 2>&1; echo "$1 exit $?")
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "$@"
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
python tools/pmc_summary.py $OUT/pmc_step.json /tmp/pmc_sq /tmp/pmc_fetch /tmp/pmc_write | tee $OUT/pmc_step.txt | head -40
