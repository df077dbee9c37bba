#!/bin/bash
# PMC passes over one harness; profiler output stays in /tmp on the GPU box, only the per-kernel sums come back.
# usage: gpu_pmc3.sh TAG 'kernel name LIKE pattern' tools/xxx.py args...
TAG=$1; PAT=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
CMD=(python "${@/#tools/$R/tools}")
run() { # name, pmc list
  rm -rf /tmp/pmc_$1
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmc_$1 -o p -- "${CMD[@]}" > $R/$OUT/$1.log 2>&1; echo "$1 exit $?")
  python - <<PY | tee -a $OUT/pmc_summary.txt
import sqlite3, glob
for f in glob.glob('/tmp/pmc_$1/**/*.db', recursive=True):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select counter_name, count(distinct dispatch_id), sum(value) from counters_collection where kernel_name like '$PAT' group by counter_name").fetchall()
        for r in rows: print('  $1', r[0], 'dispatches=%d' % r[1], 'per_dispatch=%.6g' % (r[2] / max(r[1], 1)))
    except Exception as e:
        print('query failed', e, [r[0] for r in db.execute("select name from sqlite_master").fetchall()][:40])
PY
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"

run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE"

