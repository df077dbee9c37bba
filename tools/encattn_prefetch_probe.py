#!/usr/bin/env python3
"""Encoder attention backward on COLD saved operands (1 GB of other traffic between launches, as inside the step) with and without a
touch of q|k|v + the saved context right before it.  Events bracket the attention launch alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L
ops.set_compute_dtype('fp16')
B, T, H, d = 32, 249, 4, 256
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B, T, 3 * d, generator=g) * 0.7).cuda().half().requires_grad_(True)
km = torch.ones(B, T, dtype=torch.uint8, device='cuda')
gy = torch.randn(B, T, d, generator=g).cuda().half()
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
out = ops.SelfAttentionFn.apply(qkv, km, H, False)
fn = out.grad_fn
def bwd():
    return fn.apply(gy) if hasattr(fn, 'apply') else torch.autograd.grad(out, qkv, gy, retain_graph=True)
def touch(*ts):
    for t in ts:
        t.detach().view(torch.int16).max()
def measure(pre, flush=True):
    evs = []
    for _ in range(12):
        if flush:
            big.add_(1.0)
        gy2 = gy.clone()                      # dO is warm in the step (the launch before produced it)
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.autograd.grad(out, qkv, gy2, retain_graph=True); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[2:])
    return v[len(v) // 2]
print('attention backward: warm %.1f us | cold %.1f | q|k|v + context touched %.1f' % (measure(None, False), measure(None), measure(lambda: touch(qkv, out))))
