#!/bin/bash
TAG=${1:-m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
echo "== torchrun world=1 (nccl init path of bench.py)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_torchrun.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $OUT/bench_torchrun.log | tail -2 | cut -c1-600
echo "== pmc FETCH_SIZE"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/pf -o p -- python $R/tools/gemm_one.py fwd 7968 4096 256 bf bf bf 3 > $R/$OUT/pf.log 2>&1; echo "exit $?")
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/pw -o p -- python $R/tools/gemm_one.py fwd 7968 4096 256 bf bf bf 3 > $R/$OUT/pw.log 2>&1; echo "exit $?")
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob('$OUT/p[fw]/*.db')):
    db = sqlite3.connect(f)
    print(f, db.execute("select counter_name, avg(counter_value), count(*) from pmc_events where name like '%gemm_kernel%' group by counter_name").fetchall())
PY
