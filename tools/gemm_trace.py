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
    if kind == 'ffn_fwd':        # n = 2F
        F = n // 2
        h = torch.empty(m, n, device='cuda', dtype=bf)
        u = torch.empty(m, F, device='cuda', dtype=bf)
        fn = lambda: lib.otr_ffn_glu_fwd(ops._p(x), k, ops._p(w), k, ops._p(b), ops._p(h), ops._p(u), m, F, k, ops._stream())
    elif kind == 'ffn_bwd':      # n = F: dy [m, k=d], w2t [F, d]
        F = n
        h = torch.randn(m, 2 * F, device='cuda').to(bf)
        dh = torch.empty_like(h)
        part = torch.empty((m + 63) // 64, 2 * F, device='cuda')
        rows = C.c_int32(0)
        fn = lambda: lib.otr_ffn_glu_bwd(ops._p(x), 1, k, ops._p(w), k, ops._p(h), 1, ops._p(dh), ops._p(part), part.shape[0],
                                         C.byref(rows), m, F, k, ops._stream())
    else:
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
    trace('w1', 7968, 4096, 256, bf)
    trace('w1+glu', 7968, 4096, 256, bf, 'ffn_fwd')
    trace('du', 7968, 2048, 256, bf)
    trace('du+glu_bwd', 7968, 2048, 256, bf, 'ffn_bwd')
    trace('qkv', 7968, 768, 256, bf)
