"""From a rocprofv3 kernel trace of the hipGraph bench: per-step busy time vs wall time (inter-kernel gaps)."""
import json
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith('adam_kernel')]
print('adam launches', len(idx), 'kernels', len(rows))
for a, b in zip(idx[-4:-1], idx[-3:]):
    seg = rows[a + 1:b + 1]
    wall = seg[-1][2] - seg[0][1]
    busy = sum(r[2] - r[1] for r in seg)
    # union of intervals (concurrency-aware)
    cur_s, cur_e, union = seg[0][1], seg[0][2], 0
    for _, s, e in seg[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    gaps = sorted((seg[i + 1][1] - seg[i][2]) for i in range(len(seg) - 1))
    print('step: %d kernels, wall %.2f ms, sum of durations %.2f ms, covered %.2f ms, idle %.2f ms; gap median %.2f us, p90 %.2f us'
          % (len(seg), wall / 1e6, busy / 1e6, union / 1e6, (wall - union) / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3))
# the launch sequence of the last whole step (which kernels sit next to the copy / fill nodes, what runs concurrently)
if len(idx) >= 2:
    seg = rows[idx[-2] + 1:idx[-1] + 1]
    t0 = seg[0][1]
    print('--- sequence of the last step: index, start us, duration us, name')
    for i, (n, s, e) in enumerate(seg):
        short = n.replace('(anonymous namespace)::', '').replace('at::native::', '')[:90]
        print('%4d %9.1f %7.1f  %s' % (i, (s - t0) / 1e3, (e - s) / 1e3, short))

# machine-readable: every launch of the last three whole steps by kernel name -> profiles/rNN_step_kernels.json; bench.py prints the
# in-step duration of a graded kernel from it beside its live graph-replay timing       usage: graph_gaps.py DB [OUT.json]
if len(sys.argv) > 2 and len(idx) >= 4:
    per = {}
    nsteps = 0
    for a, b in zip(idx[-4:-1], idx[-3:]):
        nsteps += 1
        for n, s, e in rows[a + 1:b + 1]:
            k = re.sub(r'\bvoid ', '', n).replace('(anonymous namespace)::', '').split('(')[0]
            per.setdefault(k, []).append((e - s) / 1e3)
    out = {k: {'launches_per_step': len(v) / nsteps, 'avg_us': sum(v) / len(v), 'max_us': max(v), 'min_us': min(v)} for k, v in per.items()}
    out['_meta'] = {'steps': nsteps, 'source': 'rocprofv3 --kernel-trace of bench.py (hipGraph replay), the last three whole steps'}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
