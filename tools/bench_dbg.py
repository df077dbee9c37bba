"""bench.py with library tuning hooks preset: python tools/bench_dbg.py KEY=VALUE[,KEY=VALUE...] [bench.py args]  (otr_debug_set keys)"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import _lib   # noqa: E402

pairs = [kv.split('=') for kv in sys.argv[1].split(',') if kv]
for kind in ('bf16', 'fp16'):
    lib = _lib.load(kind)
    for k, v in pairs:
        lib.otr_debug_set(int(k), int(v))
sys.argv = ['bench.py'] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
