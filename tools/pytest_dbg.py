"""pytest with library tuning hooks preset: python tools/pytest_dbg.py KEY=VALUE[,KEY=VALUE...] [pytest args]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest   # noqa: E402
from opentransformer_amd import _lib   # noqa: E402

pairs = [kv.split('=') for kv in sys.argv[1].split(',') if kv]
for kind in ('bf16', 'fp16'):
    lib = _lib.load(kind)
    for k, v in pairs:
        lib.otr_debug_set(int(k), int(v))
sys.exit(pytest.main(sys.argv[2:]))
