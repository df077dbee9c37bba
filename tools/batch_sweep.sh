export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 8 16 32 64 128; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, 'utt/s %.0f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'model TF/s %.0f' % d['model_tflops_per_s'])"
done
