#!/bin/bash
TAG=${1:-w256f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PWD_R=$PWD
timeout 300 python -m pytest tests/test_gpu_wgrad256.py -q --timeout 120 -p no:cacheprovider > $OUT/pytest_w256.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/pytest_w256.log | cut -c1-220
timeout 200 python tools/wgrad256_bench.py --grids 0,256,240,232,-248 --ablate 8 > $OUT/w256_rounds.log 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/w256_rounds.log | tail -3 | cut -c1-1800
pmc() { # tag, args, counters
  rm -rf /tmp/pmc_x; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $3 -d /tmp/pmc_x -o p -- python $PWD_R/tools/wgrad256_one.py $2 > $PWD_R/$OUT/$1.log 2>&1; echo "$1 exit $?")
  python - <<PY | tee -a $OUT/pmc_summary.txt
import sqlite3, glob
for f in glob.glob('/tmp/pmc_x/**/*.db', recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select counter_name, count(distinct dispatch_id), sum(value) from counters_collection where kernel_name like '%wgrad256_kernel%' group by counter_name").fetchall()
    for r in rows: print('  $1', r[0], 'dispatches=%d' % r[1], 'per_dispatch=%.6g' % (r[2] / max(r[1], 1)))
PY
}
pmc fetch_rounds "0 3" FETCH_SIZE
pmc write_rounds "0 3" WRITE_SIZE
pmc hit_rounds "0 3" "TCC_HIT_sum TCC_MISS_sum"
OTR_WGRAD256=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_1.log 2>&1; echo "bench.py exit $?"; grep -v amdgpu.ids $OUT/bench_1.log | tail -1 | cut -c1-300
