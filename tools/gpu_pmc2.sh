#!/bin/bash
# PMC passes over one harness.  usage: gpu_pmc2.sh TAG 'kernel name LIKE pattern' python tools/xxx.py args...
TAG=$1; PAT=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
run() { # name, pmc list
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $2 -d $R/$OUT/$1 -o p -- "${CMD[@]}" > $R/$OUT/$1.log 2>&1; echo "$1 exit $?")
  python - <<PY
import sqlite3, glob
for f in glob.glob('$OUT/$1/*.db'):
    db = sqlite3.connect(f)
    rows = db.execute("select k.name, p.name, avg(e.value), count(*), avg(k.duration) from pmc_events e join pmc_info p on e.pmc_id = p.id join kernels k on k.dispatch_id = e.dispatch_id where k.name like '$PAT' group by k.name, p.name").fetchall()
    for r in rows: print('  ', r[0][:60], r[1], '%.5g' % r[2], 'n=%d' % r[3], 'dur_us=%.1f' % (r[4] / 1e3))
PY
}
CMD=(python "${@/#tools/$R/tools}")
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"
run p2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
run p3 "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum"
run p4 "WRITE_SIZE GRBM_GUI_ACTIVE GRBM_COUNT"
