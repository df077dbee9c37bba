#!/usr/bin/env python3
"""Do the row-block projection launches (q|k|v 768x256, out-proj 256x256 + LayerNorm) lose time to COLD weight packs, and does a
touch by the launch before them bring it back?  Between measured launches 1 GB of other traffic goes through the chip (so nothing of
the operands is in the L2s or the memory-side cache: the situation inside the training step); "touched" reads the pack once right
before the launch (what a prefetch from the previous kernel would do).  Events bracket the measured launch alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L
ops.set_compute_dtype('fp16')
dev = 'cuda'
M, d = 7968, 256
hdt = ops.act_dtype()
torch.manual_seed(0)
x = torch.randn(M, d, device=dev); x16 = x.to(hdt)
c16 = torch.randn(M, d, device=dev).to(hdt)
wq = torch.randn(768, d, device=dev) / 16; bq = torch.zeros(768, device=dev)
wo = torch.randn(d, d, device=dev) / 16; bo = torch.zeros(d, device=dev)
g, be = torch.ones(d, device=dev), torch.zeros(d, device=dev)
pq, po = ops.lin_packs(wq), ops.lin_packs(wo)
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)       # 1 GB
def flush():
    big.add_(1.0)
def qkv():
    ops.rb_linear_raw(x16, pq[0], 768, bq, hdt)
xa = ops.attach_lp(x, x16)
def proj():
    ops.proj_add_layernorm(xa, c16, wo, bo, g, be, 0.0, 1e-5, po)
def touch(t):
    t.view(torch.int16).max()
def measure(fn, pre):
    evs = []
    for _ in range(12):
        flush()
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[2:])
    return v[len(v) // 2]
def warm(fn):
    evs = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[2:])
    return v[len(v) // 2]
with torch.no_grad():
    for _ in range(3):
        qkv(); proj()
    print('q|k|v projection   warm %.1f us | cold %.1f | pack touched %.1f | pack + rows touched %.1f' % (warm(qkv), measure(qkv, None), measure(qkv, lambda: touch(pq[0])), measure(qkv, lambda: (touch(pq[0]), touch(x16)))))
    print('out-proj + LN      warm %.1f us | cold %.1f | pack touched %.1f | pack + rows touched %.1f' % (warm(proj), measure(proj, None), measure(proj, lambda: touch(po[0])), measure(proj, lambda: (touch(po[0]), touch(c16), touch(x)))))
