#!/bin/bash
TAG=${1:-cp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_cp -o graph -- python $R/bench.py --model conformer --steps 4 --warmup 2 --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1; echo "rocprof exit $?")
python - <<PY
import sqlite3,re
from collections import Counter, defaultdict
db=sqlite3.connect('/tmp/prof_cp/graph_results.db')
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if r[0].startswith('adam_kernel')]
a,b=idx[-2],idx[-1]
seg=rows[a+1:b+1]
c=Counter(); t=defaultdict(float)
for r in seg:
    n=re.sub(r'at::native::|void |unsigned short','',r[0]); n=re.sub(r'\(.*','',n)[:72]
    c[n]+=1; t[n]+=(r[2]-r[1])/1000
tot=sum(t.values()); print(len(seg),'kernels',round(tot),'us')
for n,v in sorted(t.items(), key=lambda kv:-kv[1])[:26]:
    print('%7.0f us %5.1f%% %4d x %6.1f  %s'%(v,100*v/tot,c[n],v/c[n],n))
PY
