"""Summarise a rocprofv3 --kernel-trace result DB: per-kernel count / total / avg, % of GPU time."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print('total kernel time %.3f ms over %d kernel launches; /%d steps = %.3f ms/step'
      % (tot / 1e6, sum(r[1] for r in rows), steps, tot / 1e6 / steps))
print('%7s %6s %10s %10s %10s  %s' % ('%time', 'calls', 'avg_us', 'min_us', 'max_us', 'kernel'))
for r in rows[:60]:
    name = re.sub(r'\bvoid ', '', r[0])[:120]
    print('%6.2f%% %6d %10.1f %10.1f %10.1f  %s' % (100 * r[2] / tot, r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, name))
