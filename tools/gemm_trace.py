"""Per-workgroup phase timeline of one GEMM launch (otr_debug_trace): where does a 4-k-step block spend ~10 us?"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib

bf, f32 = torch.bfloat16, torch.float32


def trace(name, m, n, k, ydt, kind='fwd'):
    lib = _lib.load()
    x = torch.randn(m, k, device='cuda').to(bf)
    w = (torch.randn(n, k, device='cuda') / 16).to(bf)
    b = torch.randn(n, device='cuda')
    dy = torch.randn(m, n, device='cuda').to(bf)
    fn = (lambda: ops.linear_fwd_raw(x, w, b, ydt)) if kind == 'fwd' else (lambda: ops.linear_wgrad_raw(dy, x, w))
    for _ in range(3):
        fn()
    buf = torch.zeros(1 << 16, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    lib.otr_debug_trace(C.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    lib.otr_debug_trace(None)
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    rel = (t - t0).astype(np.float64)
    dur = rel[:, 3] - rel[:, 0]
    order = np.argsort(rel[:, 0])
    print('%s %s M=%d N=%d K=%d: %d workgroups traced, event time %.1f us (incl. trace stores)' % (name, kind, m, n, k, len(t), e0.elapsed_time(e1) * 1e3))
    print('  clock ticks (units of the shader cycle counter): span %.0f, wg duration mean %.0f min %.0f max %.0f' %
          (rel[:, 3].max(), dur.mean(), dur.min(), dur.max()))
    print('  phase means: stage-in %.0f | k-loop %.0f | epilogue %.0f' %
          ((rel[:, 1] - rel[:, 0]).mean(), (rel[:, 2] - rel[:, 1]).mean(), (rel[:, 3] - rel[:, 2]).mean()))
    st = np.sort(rel[:, 0])
    q = [0, len(st) // 8, len(st) // 4, len(st) // 2, 3 * len(st) // 4, len(st) - 1]
    print('  start-time quantiles:', ' '.join('%.0f' % st[i] for i in q))
    print('  first 6 by start:', [tuple(int(v) for v in rel[i]) for i in order[:6]])
    print('  last 3 by start:', [tuple(int(v) for v in rel[i]) for i in order[-3:]])
    return e0.elapsed_time(e1) * 1e3 / rel[:, 3].max()


if __name__ == '__main__':
    ops.set_compute_dtype('bf16')
    r = trace('out', 7968, 256, 256, f32)
    print('  => us per tick ~ %.5f' % r)
    trace('w1', 7968, 4096, 256, bf)
    trace('qkv', 7968, 768, 256, bf)
    trace('w2', 7968, 256, 2048, f32)
    trace('w1', 7968, 4096, 256, bf, 'wgrad')
