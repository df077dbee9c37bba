#!/usr/bin/env python3
"""Shader-clock timeline of the encoder-shape attention backward launch (csrc/encattn.hip, otr_debug_trace) at the bench shape
(B = 32, H = 4, T = 249): phases per orientation, median over workgroups, plus the launch's duration by events."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L
ops.set_compute_dtype('fp16')
B, T, H, d = 32, 249, 4, 256
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B, T, 3 * d, generator=g) * 0.7).cuda().half().requires_grad_(True)
lens = torch.randint(T // 2, T + 1, (B,), generator=g); lens[0] = T
km = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).cuda().to(torch.uint8)
gy = torch.randn(B, T, d, generator=g).cuda().half()
lib = L.load()
def step():
    out = ops.SelfAttentionFn.apply(qkv, km, H, False)
    return torch.autograd.grad(out, qkv, gy)
for _ in range(3):
    step()
torch.cuda.synchronize()
tr = torch.zeros(16384 + 8 * 256 * 16, dtype=torch.int64, device='cuda')
lib.otr_debug_trace(ops._p(tr))
step()
torch.cuda.synchronize()
lib.otr_debug_trace(None)
t = tr.cpu().numpy()[16384:].reshape(8, 256, 16)
ph = ['issue', 'first chunk staged', 'chunks + tiles (wave 0)', 'wait for the other waves', 'store']
for k, nm in ((6, 'lane = query (dq)'), (7, 'lane = key (dk, dv)')):
    a = t[k]
    a = a[a[:, 0] > 0]
    dd = np.diff(a[:, :6], axis=1).astype(np.float64)
    tot = a[:, 5] - a[:, 0]
    print('%-20s workgroups %3d  total cycles median %7.0f (p90 %7.0f); start spread %d' % (nm, a.shape[0], np.median(tot), np.percentile(tot, 90), a[:, 0].max() - a[:, 0].min()))
    for i, p_ in enumerate(ph):
        print('    %-26s median %7.0f   p90 %7.0f' % (p_, np.median(dd[:, i]), np.percentile(dd[:, i], 90)))
