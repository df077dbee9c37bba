#!/bin/bash
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -ciE "^\s*(Name|counter)" $OUT/counters.txt; grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|OccupancyPercent|MeanOccupancyPerCU|LdsBankConflict|L2CacheHit|MemUnitStalled)\b" $OUT/counters.txt | sort -u | tr '\n' ' ' | cut -c1-6000
echo
run() { # name, pmc list
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $2 -d $R/$OUT/$1 -o p -- python $R/tools/gemm_one.py fwd 7968 4096 256 bf bf bf 6 > $R/$OUT/$1.log 2>&1; echo "$1 exit $?")
  python - <<PY
import sqlite3, glob
for f in glob.glob('$OUT/$1/*.db'):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select k.name, p.name, avg(e.value), count(*) from pmc_events e join pmc_info p on e.pmc_id = p.id join kernels k on k.dispatch_id = e.dispatch_id where k.name like '%gemm_kernel%' group by k.name, p.name").fetchall()
        for r in rows: print('  ', r[1], '%.4g' % r[2], 'n=%d' % r[3])
    except Exception as ex:
        print('query failed', ex)
        print([r[1] for r in db.execute("pragma table_info(pmc_events)")])
        print([r[1] for r in db.execute("pragma table_info(pmc_info)")])
PY
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"
run p2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
run p3 "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum"
run p4 "WRITE_SIZE GRBM_GUI_ACTIVE GRBM_COUNT"
run p5 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
