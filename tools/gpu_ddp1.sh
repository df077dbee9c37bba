#!/bin/bash
TAG=${1:-ddp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_torchrun.log 2>&1; echo "torchrun exit $?"; grep -v amdgpu.ids $OUT/bench_torchrun.log | tail -3 | cut -c1-400
# force the collective path with a world of 1
timeout 300 python - > $OUT/nccl_ws1.log 2>&1 <<'PY'
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29512', RANK='0', WORLD_SIZE='1')
dist.init_process_group('nccl', init_method='env://')
torch.cuda.set_device(0)
x = torch.ones(36_600_000, device='cuda')
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    dist.all_reduce(x)
torch.cuda.synchronize()
print('rccl all_reduce world=1 146MB: %.3f ms' % ((time.perf_counter() - t0) * 100), float(x[0]))
dist.barrier()
dist.destroy_process_group()
PY
echo "nccl exit $?"; grep -v amdgpu.ids $OUT/nccl_ws1.log | tail -3
