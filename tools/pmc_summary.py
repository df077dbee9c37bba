#!/usr/bin/env python3
"""Per-kernel PMC summary of rocprofv3 --kernel-trace --pmc passes (one result directory per pass: SQ counters, FETCH_SIZE,
WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes).  For every kernel above `min_share` of the kernel time:
launches, median duration, the counter sums per launch, and the derived figures

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)   the matrix pipes' busy share at PEAK clock
                   (what 2.5 PFLOP/s is quoted at); mfma_busy_of_sq_busy relates it to SQ_BUSY_CYCLES x 4 SIMDs x (CUs per SE)
  hbm_bytes      = 2 x FETCH_SIZE KB + WRITE_SIZE KB (FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950 for wide loads)

A kernel that is launched with several grid sizes (wgrad256_kernel: one launch per row count) gets one record per grid size,
keyed "<kernel> [grid N]".

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


def cols(db, table):
    try:
        return [r[1] for r in db.execute('pragma table_info(%s)' % table).fetchall()]
    except Exception:                                                  # noqa: BLE001
        return []


def grid_expr(columns, prefix=''):
    """SQL expression for the launch's total grid size (work-items) from whatever the rocpd view offers"""
    for c in ('grid_size',):
        if c in columns:
            return prefix + c
    if all(c in columns for c in ('grid_x', 'grid_y', 'grid_z')):
        return '(%sgrid_x * %sgrid_y * %sgrid_z)' % (prefix, prefix, prefix)
    if all(c in columns for c in ('grid_size_x', 'grid_size_y', 'grid_size_z')):
        return '(%sgrid_size_x * %sgrid_size_y * %sgrid_size_z)' % (prefix, prefix, prefix)
    return None


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = {}
    schema_note = {}
    for d in dirs:
        for f in glob.glob(d + '/**/*.db', recursive=True):
            db = sqlite3.connect(f)
            kc, cc = cols(db, 'kernels'), cols(db, 'counters_collection')
            schema_note = {'kernels': kc, 'counters_collection': cc}
            kg = grid_expr(kc)
            try:
                dur = db.execute("select name, %s, count(*), sum(end-start) from kernels group by 1, 2" % (kg or '0')).fetchall()
            except Exception as e:                                         # noqa: BLE001
                print('no kernel table in', f, e)
                continue
            tot = sum(r[3] for r in dur) or 1
            for name, grid, n, t in dur:
                k = res.setdefault((short(name), int(grid or 0)), {})
                k.setdefault('launches', n)
                k.setdefault('avg_us', t / n / 1e3)
                k.setdefault('share_of_kernel_time', t / tot)
            # counters keyed by (kernel, grid) as well: a kernel launched on two problem sizes (wgrad256_kernel: the M = 7968 group
            # and the decoder's M = 480 group) must not have its FETCH / WRITE averaged over both
            cg = grid_expr(cc)
            try:
                if cg is not None:
                    rows = db.execute("select kernel_name, %s, counter_name, count(distinct dispatch_id), sum(value) from "
                                      "counters_collection group by 1, 2, 3" % cg).fetchall()
                elif kg is not None and 'dispatch_id' in kc and 'dispatch_id' in cc:
                    rows = db.execute("select c.kernel_name, %s, c.counter_name, count(distinct c.dispatch_id), sum(c.value) from "
                                      "counters_collection c join kernels k on k.dispatch_id = c.dispatch_id group by 1, 2, 3"
                                      % grid_expr(kc, 'k.')).fetchall()
                else:
                    rows = [(a, 0, b, c, d_) for a, b, c, d_ in db.execute(
                        "select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
                        "group by 1, 2").fetchall()]
            except Exception as e:                                         # noqa: BLE001
                print('no counters in', f, e)
                continue
            for name, grid, c, nd, v in rows:
                k = res.setdefault((short(name), int(grid or 0)), {})
                k[c] = v / max(nd, 1)
                k['dispatches_' + c] = nd
    names = {}
    for (name, grid) in res:
        names.setdefault(name, set()).add(grid)
    keep = {}
    for (name, grid), k in res.items():
        if k.get('share_of_kernel_time', 0) < 0.004:
            continue
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in k and 'avg_us' in k:
            k['mfma_busy_frac'] = k['SQ_VALU_MFMA_BUSY_CYCLES'] / (k['avg_us'] * 1e-6 * 2.4e9 * 1024)
            if k.get('SQ_BUSY_CYCLES'):
                k['mfma_busy_cycles_per_sq_busy_cycle'] = k['SQ_VALU_MFMA_BUSY_CYCLES'] / k['SQ_BUSY_CYCLES']
        if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
            k['hbm_bytes_per_launch'] = int(2 * k['FETCH_SIZE'] * 1024 + k['WRITE_SIZE'] * 1024)
        k['grid_work_items'] = grid
        keep[name if len(names[name]) == 1 else '%s [grid %d]' % (name, grid)] = k
    keep = dict(sorted(keep.items(), key=lambda kv: -kv[1].get('share_of_kernel_time', 0)))
    keep['_schema'] = schema_note
    json.dump(keep, open(out, 'w'), indent=1)
    for name, k in keep.items():
        if name.startswith('_'):
            continue
        print('%-78s n=%4d avg %8.1f us share %5.1f%%  mfma_busy %s  hbm %s' % (
            name[:78], k.get('launches', 0), k.get('avg_us', 0), 100 * k.get('share_of_kernel_time', 0),
            '%.3f' % k['mfma_busy_frac'] if 'mfma_busy_frac' in k else '   - ',
            '%.1f MB' % (k['hbm_bytes_per_launch'] / 1e6) if 'hbm_bytes_per_launch' in k else '-'))


if __name__ == '__main__':
    main()
