export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_ops.py tests/test_gpu_round2b.py -x -q -m gpu -p no:cacheprovider -k "conv" 2>&1 | tail -3
bash tools/gpu_conformer_trace.sh r6c5 > /dev/null 2>&1
grep -i "conv\|col2im\|wgrad256\|w256\|relu_bwd" gpurun_out/r6c5/conformer_kernels.txt | cut -c1-150
