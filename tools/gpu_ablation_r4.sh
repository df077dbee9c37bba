#!/bin/bash
# same-box ablation of the round-4 changes: each line switches ONE of them off (bench.py, 20 steps, no extras)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-abl}; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  ms=$(env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$name $ms" | tee -a $OUT/ablation.txt
}
: > $OUT/ablation.txt
run "all_on(1)" X=1
run "encoder_ffn_exchange_form(slab_off)" OTR_SWITCHES=ops._FFN_SLAB=0
run "generic_attention_backward" OTR_DEBUG_SET=21=0
run "conv2_only_on_new_kernel" OTR_DEBUG_SET=22=2
run "conv_generic_paths" OTR_DEBUG_SET=22=0
run "per_operator_decoder" OTR_SWITCHES=ops._DEC_FUSED=0
run "all_on(2)" X=1
run "all_round4_switches_off" OTR_SWITCHES=ops._FFN_SLAB=0,ops._DEC_FUSED=0 OTR_DEBUG_SET=21=0,22=0
