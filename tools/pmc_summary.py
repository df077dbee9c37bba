#!/usr/bin/env python3
"""Per-kernel PMC summary of rocprofv3 --kernel-trace --pmc passes (one result directory per pass: SQ counters, FETCH_SIZE,
WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes).  For every kernel above `min_share` of the kernel time:
launches, median duration, the counter sums per launch, and the derived figures

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)   the matrix pipes' busy share at PEAK clock
                   (what 2.5 PFLOP/s is quoted at); mfma_busy_of_sq_busy relates it to SQ_BUSY_CYCLES x 4 SIMDs x (CUs per SE)
  hbm_bytes      = 2 x FETCH_SIZE KB + WRITE_SIZE KB (FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950 for wide loads)

usage: pmc_summary.py OUT.json DIR [DIR ...]"""
import glob
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\bvoid ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return name.split('(')[0][:90]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = {}
    for d in dirs:
        for f in glob.glob(d + '/**/*.db', recursive=True):
            db = sqlite3.connect(f)
            try:
                dur = db.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()
            except Exception as e:                                         # noqa: BLE001
                print('no kernel table in', f, e)
                continue
            tot = sum(r[2] for r in dur) or 1
            for name, n, t in dur:
                k = res.setdefault(short(name), {})
                k.setdefault('launches', n)
                k.setdefault('avg_us', t / n / 1e3)
                k.setdefault('share_of_kernel_time', t / tot)
            try:
                rows = db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
                                  "group by kernel_name, counter_name").fetchall()
            except Exception as e:                                         # noqa: BLE001
                print('no counters in', f, e)
                continue
            for name, c, nd, v in rows:
                k = res.setdefault(short(name), {})
                k[c] = v / max(nd, 1)
                k['dispatches_' + c] = nd
    keep = {}
    for name, k in res.items():
        if k.get('share_of_kernel_time', 0) < 0.004:
            continue
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in k:
            k['mfma_busy_frac'] = k['SQ_VALU_MFMA_BUSY_CYCLES'] / (k['avg_us'] * 1e-6 * 2.4e9 * 1024)
            if k.get('SQ_BUSY_CYCLES'):
                k['mfma_busy_cycles_per_sq_busy_cycle'] = k['SQ_VALU_MFMA_BUSY_CYCLES'] / k['SQ_BUSY_CYCLES']
        if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
            k['hbm_bytes_per_launch'] = int(2 * k['FETCH_SIZE'] * 1024 + k['WRITE_SIZE'] * 1024)
        keep[name] = k
    keep = dict(sorted(keep.items(), key=lambda kv: -kv[1].get('share_of_kernel_time', 0)))
    json.dump(keep, open(out, 'w'), indent=1)
    for name, k in keep.items():
        print('%-70s n=%4d avg %8.1f us share %5.1f%%  mfma_busy %s  hbm %s' % (
            name[:70], k.get('launches', 0), k.get('avg_us', 0), 100 * k.get('share_of_kernel_time', 0),
            '%.3f' % k['mfma_busy_frac'] if 'mfma_busy_frac' in k else '   - ',
            '%.1f MB' % (k['hbm_bytes_per_launch'] / 1e6) if 'hbm_bytes_per_launch' in k else '-'))


if __name__ == '__main__':
    main()
