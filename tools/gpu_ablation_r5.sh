#!/bin/bash
# same-box ablation of the round-5 changes that have a switch: each line switches ONE of them off (bench.py, 30 steps, no extras)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-abl5}; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  ms=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$name $ms" | tee -a $OUT/ablation.txt
}
: > $OUT/ablation.txt
run "all_on(1)" X=1
run "frontend_linear_fp32_gradient(g16_off)" OTR_SWITCHES=ops._G16=0
run "three_kernel_loss+scale_launches(ls_fused_off)" OTR_SWITCHES=ops._LS_FUSED=0
run "dec_sum_launch(embed_sink_off)" OTR_SWITCHES=ops._EMBED_SINK=0
run "all_on(2)" X=1
run "all_three_off" OTR_SWITCHES=ops._G16=0,ops._LS_FUSED=0,ops._EMBED_SINK=0
run "all_on(3)" X=1
