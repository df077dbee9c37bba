#!/usr/bin/env python3
"""Per-workgroup shader-clock timeline of ffn3_fwd_kernel (otr_debug_trace): prologue, the 3 phases of every chunk, exchange steps."""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L
ops.set_compute_dtype('fp16')
dev = 'cuda'
M, d, F = 7968, 256, 2048
hdt = ops.act_dtype()
torch.manual_seed(0)
w1 = torch.randn(2 * F, d, device=dev) / math.sqrt(d); w2 = torch.randn(d, F, device=dev) / math.sqrt(F)
b1, b2 = torch.randn(2 * F, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
x = torch.randn(M, d, device=dev); x16 = x.to(hdt)
P = ops.ffn_packs(w1, w2)
y, y16, z = torch.empty_like(x), torch.empty_like(x16), torch.empty_like(x)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
seed = ops.rng_seed_tensor(dev)
lib = L.load(); p, st = ops._p, ops._stream
nb = lib.otr_ffn_split_scratch_bytes(M)
scratch = torch.empty(nb // 4, device=dev)
sync = ops._ffn_sync(torch.device('cuda', torch.cuda.current_device()))
hsave = torch.zeros(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=hdt, device=dev)
usave = torch.zeros(lib.otr_ffn_split_padded_rows(M), F, dtype=hdt, device=dev)
save = 'save' in sys.argv[1:]
slab = 'slab' in sys.argv[1:]
slabs = torch.empty(4, M, d, dtype=hdt, device=dev)
def run():
    if slab:
        L.check(lib.otr_ffn_fwd_split_slab(p(x16), p(P[0]), p(b1), p(P[1]), p(hsave) if save else None, p(usave) if save else None, p(slabs),
                                           M, F, d, st()), 'fwd3 slab')
        return
    L.check(lib.otr_ffn_ln_fwd_split(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0, 1e-5, p(y), p(y16),
                                     p(z), p(mean), p(rstd), p(hsave) if save else None, p(usave) if save else None, p(scratch), nb, p(sync), sync.numel(), M, F, d, st()), 'fwd3')
for _ in range(3): run()
tr = torch.zeros(256 * 48 + 512, dtype=torch.int64, device=dev)
lib.otr_debug_set(4, 16); lib.otr_debug_trace(p(tr)); run(); torch.cuda.synchronize(); lib.otr_debug_trace(None); lib.otr_debug_set(4, 0)
rt = tr.cpu().numpy()[256 * 48:].reshape(256, 2)
t = tr.cpu().numpy()[:256 * 48].reshape(256, 48)
live = t[:, 0] > 0
t = t[live]
n = int((t[0] > 0).sum())
dt = np.diff(t[:, :n], axis=1).astype(np.float64)
names = ['prologue'] + ['ph%d%s' % (i // 3, 'ABG'[i % 3]) for i in range(24)] + (['closing', 'drain+bar', 'slab store'] if slab else []) + ['closing', 'drain+bar', 'send', 'hoisted loads', 'store drain+bar', 'arrive wait', 'recv+sum', 'LN+out']
print('workgroups', t.shape[0], 'stamps', n, 'total cycles median', np.median(t[:, n - 1] - t[:, 0]), '(100 MHz s_memtime ticks?)')
for i in range(min(n - 1, len(names))):
    print('%-16s median %8.0f  p10 %8.0f  p90 %8.0f' % (names[i], np.median(dt[:, i]), np.percentile(dt[:, i], 10), np.percentile(dt[:, i], 90)))
print('start spread (first stamp max-min):', t[:, 0].max() - t[:, 0].min())

if slab:
    r = rt[rt[:, 0] > 0].astype(np.float64)
    s0, e0 = r[:, 0].min(), r[:, 1].max()
    print('100 MHz clock: first workgroup starts at 0, the last one starts %.2f us later (median start %.2f us); ends: first %.2f us, median %.2f us, last %.2f us'
          % ((r[:, 0].max() - s0) / 100, (np.median(r[:, 0]) - s0) / 100, (r[:, 1].min() - s0) / 100, (np.median(r[:, 1]) - s0) / 100, (e0 - s0) / 100))
