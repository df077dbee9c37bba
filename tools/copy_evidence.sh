#!/bin/bash
# copy one evidence visit (gpurun_out/TAG, made by tools/gpu_round5.sh) into profiles/rNN_*     usage: copy_evidence.sh TAG [rNN]
T=gpurun_out/${1:-r5ev}; R=${2:-r05}; P=profiles
c() { [ -s "$1" ] && cp "$1" "$2"; }
c $T/bench.json $P/${R}_bench.json
[ -f $T/bench_conformer.log ] && grep '^{' $T/bench_conformer.log | tail -1 > $P/${R}_bench_conformer.json
c $T/decode.json $P/${R}_decode_bench_c5.json
c $T/kernel_summary_graph.txt $P/${R}_kernel_trace_graph.txt
c $T/graph_gaps.txt $P/${R}_step_sequence.txt
c $T/step_kernels.json $P/${R}_step_kernels.json
c $T/pmc_step.json $P/${R}_pmc_step.json; c $T/pmc_step.txt $P/${R}_pmc_step.txt
c $T/ffn_bench.json $P/${R}_ffn_bench.json
c $T/conformer_kernels.txt $P/${R}_conformer_kernels.txt; c $T/conformer_step_sequence.txt $P/${R}_conformer_step_sequence.txt
c $T/decode_kernels.txt $P/${R}_decode_kernels.txt
c $T/tolerance_cases.jsonl $P/${R}_tolerance_cases.jsonl
c $T/bench_2rank_onegpu_gloo.log $P/${R}_bench_2rank_onegpu_gloo.log
c $T/bench_1400.json $P/${R}_bench_1400_frames.json; c $T/bench_1400_generic_attention_bwd.json $P/${R}_bench_1400_frames_generic_attention_bwd.json
[ -f $T/pytest_gpu.log ] && grep -E "passed|failed" $T/pytest_gpu.log | tail -1 > $P/${R}_pytest_gpu.txt
[ -f $T/smoke.log ] && grep -v amdgpu.ids $T/smoke.log | tail -4 > $P/${R}_smoke.txt
for f in parity_c4_engine_fp16 parity_c4_engine_bf16 parity_headline_engine_fp16 parity_headline_engine_bf16 parity_headline_fp16 parity_headline_bf16 parity_headline_fp32 parity_c2ctc_fp16 parity_c2ctc_bf16 parity_c2ctc_fp32 parity_c4_fp16 parity_c4_bf16 parity_c4_fp32 parity_report decode_validity_fp16 decode_validity_bf16 decode_eos_live_fp16 decode_eos_live_fp32 decode_eos_live_bf16; do
  c $T/$f.json $P/${R}_$f.json
done
ls $P | grep ${R}_ | tr '\n' ' '
