#!/usr/bin/env python3
"""Per-workgroup shader-clock timeline of ffn3_bwd_kernel<SLAB> (otr_debug_trace): prologue, D / XA / XB of every chunk, closing, slab store."""
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
b1 = torch.randn(2 * F, device=dev) * 0.1
x16 = torch.randn(M, d, device=dev).to(hdt)
da = (torch.randn(M, d, device=dev) * 0.1).to(hdt)
P = ops.ffn_packs(w1, w2)
lib = L.load(); p, st = ops._p, ops._stream
hsave = torch.zeros(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=hdt, device=dev)
usave = torch.zeros(lib.otr_ffn_split_padded_rows(M), F, dtype=hdt, device=dev)
slabs = torch.empty(4, M, d, dtype=hdt, device=dev)
L.check(lib.otr_ffn_fwd_split_slab(p(x16), p(P[0]), p(b1), p(P[1]), p(hsave), p(usave), p(slabs), M, F, d, st()), 'fwd')
dh = torch.empty(usave.shape[0], 2 * F, dtype=hdt, device=dev)
bsl = torch.empty(4, M, d, dtype=hdt, device=dev)
def run():
    L.check(lib.otr_ffn_bwd_split_slab(p(da), p(hsave), p(P[2]), p(P[3]), p(dh), p(bsl), M, F, d, st()), 'bwd')
for _ in range(3): run()
tr = torch.zeros(256 * 48, dtype=torch.int64, device=dev)
lib.otr_debug_set(4, 16); lib.otr_debug_trace(p(tr)); run(); torch.cuda.synchronize(); lib.otr_debug_trace(None); lib.otr_debug_set(4, 0)
t = tr.cpu().numpy().reshape(256, 48)
t = t[t[:, 0] > 0]
n = int((t[0] > 0).sum())
dt = np.diff(t[:, :n], axis=1).astype(np.float64)
names = ['prologue (DMA x2, dy rows, first tiles)'] + ['ch%d %s' % (i // 3, 'D XA XB'.split()[i % 3]) for i in range(24)] + ['closing XA XB + drain', 'slab store']
print('workgroups', t.shape[0], 'stamps', n, 'total cycles median', np.median(t[:, n - 1] - t[:, 0]))
for i in range(min(n - 1, len(names))):
    print('%-40s median %8.0f  p10 %8.0f  p90 %8.0f' % (names[i], np.median(dt[:, i]), np.percentile(dt[:, i], 10), np.percentile(dt[:, i], 90)))
