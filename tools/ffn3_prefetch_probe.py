#!/usr/bin/env python3
"""Would the split FFN launches gain from having their packed weights touched by the launch before them?  Slab kernels (fp16, 7968
rows, d_ff 2048), 12 distinct weight / activation / save sets cycled so that nothing is found in a cache ("cold", as in the step),
with and without a small kernel that reads the NEXT launch's weights (3 MB) right before it.  Events bracket the FFN launch alone."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L
ops.set_compute_dtype('fp16')
dev = 'cuda'
M, d, F, N = 7968, 256, 2048, 12
hdt = ops.act_dtype()
lib = L.load(); p, st = ops._p, ops._stream
torch.manual_seed(0)
b1 = torch.randn(2 * F, device=dev) * 0.1
Ps = [[t.clone() for t in ops.ffn_packs(torch.randn(2 * F, d, device=dev) / math.sqrt(d), torch.randn(d, F, device=dev) / math.sqrt(F))] for _ in range(N)]
xs = [torch.randn(M, d, device=dev).to(hdt) for _ in range(N)]
das = [(torch.randn(M, d, device=dev) * 0.1).to(hdt) for _ in range(N)]
hs = [torch.zeros(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=hdt, device=dev) for _ in range(N)]
us = [torch.zeros(lib.otr_ffn_split_padded_rows(M), F, dtype=hdt, device=dev) for _ in range(N)]
dhs = [torch.empty(us[0].shape[0], 2 * F, dtype=hdt, device=dev) for _ in range(N)]
slabs = torch.empty(4, M, d, dtype=hdt, device=dev)
def fwd(i):
    L.check(lib.otr_ffn_fwd_split_slab(p(xs[i]), p(Ps[i][0]), p(b1), p(Ps[i][1]), p(hs[i]), p(us[i]), p(slabs), M, F, d, st()), 'fwd')
def bwd(i):
    L.check(lib.otr_ffn_bwd_split_slab(p(das[i]), p(hs[i]), p(Ps[i][2]), p(Ps[i][3]), p(dhs[i]), p(slabs), M, F, d, st()), 'bwd')
def touch(ts):
    for t in ts:
        t.view(torch.int16).max()            # a read-only pass over the pack
for i in range(N):
    fwd(i); bwd(i)
torch.cuda.synchronize()
def measure(kernel, sets, pre):
    evs = []
    for rep in range(4):
        for i in sets:
            if pre is not None:
                pre(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); kernel(i); e1.record()
            evs.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[len(sets):])
    return v[len(v) // 2]
allsets = list(range(N))
print('fwd  one set (warm)            %.1f us' % measure(fwd, [0] * N, None))
print('fwd  12 sets (cold)            %.1f us' % measure(fwd, allsets, None))
print('fwd  12 sets, weights touched  %.1f us' % measure(fwd, allsets, lambda i: touch(Ps[i][:2])))
print('fwd  12 sets, x rows touched   %.1f us' % measure(fwd, allsets, lambda i: touch([xs[i]])))
print('fwd  12 sets, both touched     %.1f us' % measure(fwd, allsets, lambda i: touch(Ps[i][:2] + [xs[i]])))
print('bwd  one set (warm)            %.1f us' % measure(bwd, [0] * N, None))
print('bwd  12 sets (cold)            %.1f us' % measure(bwd, allsets, None))
print('bwd  12 sets, weights touched  %.1f us' % measure(bwd, allsets, lambda i: touch(Ps[i][2:])))
print('bwd  12 sets, dy rows touched  %.1f us' % measure(bwd, allsets, lambda i: touch([das[i]])))
print('bwd  12 sets, weights + dy     %.1f us' % measure(bwd, allsets, lambda i: touch(Ps[i][2:] + [das[i]])))
print('bwd  12 sets, weights + saved tiles touched  %.1f us' % measure(bwd, allsets, lambda i: touch(Ps[i][2:] + [hs[i]])))
print('bwd  12 sets, saved tiles touched  %.1f us' % measure(bwd, allsets, lambda i: touch([hs[i]])))
