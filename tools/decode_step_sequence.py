"""From a rocprofv3 kernel trace of `bench.py --task decode`: the launch sequence of one replayed cached decode step (between two
consecutive beam_prune launches late in the run): index, start us, duration us, gap to the previous kernel's end, name.
usage: decode_step_sequence.py DB"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith('beam_prune_kernel')]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]
t0 = seg[0][1]
print('one cached decode step: %d kernels, wall %.1f us, sum of durations %.1f us' % (len(seg), (seg[-1][2] - seg[0][1]) / 1e3, sum(e - s for _, s, e in seg) / 1e3))
prev = None
for i, (n, s, e) in enumerate(seg):
    short = n.replace('(anonymous namespace)::', '').replace('at::native::', '')[:100]
    print('%3d %8.1f %6.1f %6.1f  %s' % (i, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, short))
    prev = e
