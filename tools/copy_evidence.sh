#!/bin/bash
# copy one evidence visit (gpurun_out/TAG, made by tools/gpu_round4.sh) into profiles/r04_*
T=gpurun_out/${1:-r4ev}; P=profiles
cp $T/bench.json $P/r04_bench.json
grep '^{' $T/bench_conformer.log | tail -1 > $P/r04_bench_conformer.json
cp $T/decode.json $P/r04_decode_bench_c5.json
cp $T/kernel_summary_graph.txt $P/r04_kernel_trace_graph.txt
cp $T/graph_gaps.txt $P/r04_step_sequence.txt
cp $T/pmc_step.json $P/r04_pmc_step.json; cp $T/pmc_step.txt $P/r04_pmc_step.txt
cp $T/ffn_bench.json $P/r04_ffn_bench.json
cp $T/dec_trace.txt $P/r04_dec_trace.txt
cp $T/encattn_trace.txt $P/r04_encattn_trace.txt
cp $T/ffn3_slab_trace.txt $P/r04_ffn3_slab_trace.txt
grep -E "passed|failed" $T/pytest_gpu.log | tail -1 > $P/r04_pytest_gpu.txt
grep -v amdgpu.ids $T/smoke.log | tail -4 > $P/r04_smoke.txt
for f in parity_headline_fp16 parity_headline_bf16 parity_headline_fp32 parity_c2ctc_fp16 parity_c2ctc_bf16 parity_c2ctc_fp32 parity_c4_fp16 parity_c4_bf16 parity_c4_fp32 parity_report decode_validity_fp16 decode_validity_bf16 decode_eos_live_fp16 decode_eos_live_fp32; do
  [ -f $T/$f.json ] && cp $T/$f.json $P/r04_$f.json
done
ls $P | grep r04_
