#!/bin/bash
# HBM traffic of the hot kernels INSIDE the training step: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over an
# eager bench.py run, averaged per kernel -> gpurun_out/pmc/r02_pmc_traffic.json (copy to profiles/).  Per the guide
# (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950 for wide coalesced loads -> doubled.
OUT=gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $R/$OUT/$c.log 2>&1; echo "$c exit $?")
done
python - <<PY
import sqlite3, glob, json
names = {'linear_wgrad_grouped': 'gemm_grouped_kernel', 'ffn_ln_fwd': 'ffn_ln_fwd_kernel', 'ffn_bwd': 'ffn_bwd_kernel',
         'attn_fwd': 'attn_fwd_kernel', 'add_ln_bwd': 'add_ln_bwd_kernel'}
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('$OUT/%s/*.db' % c):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        for key, pat in names.items():
            try:
                rows = db.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like ? and counter_name = ? group by dispatch_id", ('%' + pat + '%', c)).fetchall()
            except Exception as e:
                print('query failed', e, tabs[:10]); rows = []
            if rows:
                vals = sorted(r[1] for r in rows)
                res.setdefault(key, {})[c + '_KB_median'] = vals[len(vals) // 2]
                res[key]['dispatches_' + c] = len(vals)
for key, d in res.items():
    if 'FETCH_SIZE_KB_median' in d and 'WRITE_SIZE_KB_median' in d:
        d['hbm_bytes_per_launch'] = int(2 * d['FETCH_SIZE_KB_median'] * 1024 + d['WRITE_SIZE_KB_median'] * 1024)
res['_comment'] = ('HBM bytes per launch inside the training step: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over '
                   '`bench.py --steps 2 --warmup 1 --no-graph`; median over the dispatches of each kernel; FETCH_SIZE doubled (gfx950 tallies '
                   '128-B requests at 64 B, MI355X_MICROARCH.md); tools/gpu_pmc_step.sh')
json.dump(res, open('$OUT/r02_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
