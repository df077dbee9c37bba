#!/bin/bash
# One GPU-box visit: device facts, GPU parity tests (crash-isolated by xdist), smoke, short bench.
# Everything is logged under gpurun_out/ (merged back by gpurun).  Usage: tools/gpu_round.sh [tag]
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{
  echo "== device"; python - <<'PY'
import torch, os
print('torch', torch.__version__, 'devices', torch.cuda.device_count(), 'cpus', os.cpu_count())
for i in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(i); print(i, p.name, p.total_memory // 2**30, 'GiB', p.multi_processor_count, 'CUs')
PY
} > $OUT/device.log 2>&1
cat $OUT/device.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -n 2 --timeout 600 -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | sed -e 's/ - .*//' | sort | uniq -c | sort -rn | head -100
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -5 $OUT/smoke.log
echo "== bench eager"
timeout 600 python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $OUT/bench_eager.log 2>&1; echo "exit $?"; tail -3 $OUT/bench_eager.log
echo "== bench graph"
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench_graph.log 2>&1; echo "exit $?"; tail -3 $OUT/bench_graph.log
