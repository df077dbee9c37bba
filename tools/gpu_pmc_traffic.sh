#!/bin/bash
# HBM traffic of the FFN w_1 forward GEMM: FETCH_SIZE and WRITE_SIZE in separate passes (guide: MI355X_MICROARCH.md)
TAG=${1:-pmct}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/$c -o p -- python $R/tools/gemm_one.py fwd 7968 4096 256 bf bf bf 4 > $R/$OUT/$c.log 2>&1; echo "$c exit $?")
done
python - <<PY
import sqlite3, glob
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('$OUT/%s/*.db' % c):
        db = sqlite3.connect(f)
        rows = db.execute("select counter_name, dispatch_id, sum(value), count(*) from counters_collection where kernel_name like '%gemm_kernel%' group by counter_name, dispatch_id").fetchall()
        for r in rows: print(c, 'dispatch', r[1], 'sum %.6g' % r[2], 'instances', r[3])
PY
