#!/bin/bash
# PMC passes over one harness; profiler output stays in /tmp on the GPU box, only the per-kernel sums come back.
# usage: gpu_pmc4.sh TAG 'kernel name LIKE pattern' tools/xxx.py args...
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
        for kn, in db.execute("select distinct kernel_name from counters_collection where kernel_name like '$PAT'").fetchall():
            dur = db.execute("select avg(end-start), count(*) from kernels where name = ?", (kn,)).fetchone()
            print('== %s  avg %.1f us over %d launches' % (kn[:90], (dur[0] or 0) / 1e3, dur[1]))
            rows = db.execute("select counter_name, count(distinct dispatch_id), sum(value) from counters_collection where kernel_name = ? group by counter_name", (kn,)).fetchall()
            for r in rows: print('  $1 %-28s per_dispatch=%.6g' % (r[0], r[2] / max(r[1], 1)))
    except Exception as e:
        print('query failed', e)
PY
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
run p2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE"
