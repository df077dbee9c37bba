#!/usr/bin/env python3
"""A few launches of the split FFN kernels (csrc/ffn3.hip) at the benchmark shape, for rocprofv3 --pmc passes (tools/gpu_pmc4.sh)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ops.set_compute_dtype(mode)
dev = 'cuda'
M, d, F = 7968, 256, 2048
hdt = ops.act_dtype()
torch.manual_seed(0)
w1 = torch.randn(2 * F, d, device=dev) / math.sqrt(d)
w2 = torch.randn(d, F, device=dev) / math.sqrt(F)
b1, b2 = torch.randn(2 * F, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
x = torch.randn(M, d, device=dev)
x16 = x.to(hdt)
P = ops.ffn_packs(w1, w2)
y, y16, z = torch.empty_like(x), torch.empty_like(x16), torch.empty_like(x)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
seed = ops.rng_seed_tensor(dev)
lib = L.load()
p, st = ops._p, ops._stream
da = (torch.randn(M, d, device=dev) * 0.01).to(hdt)
nb = lib.otr_ffn_split_scratch_bytes(M)
scratch = torch.empty(nb // 4, device=dev)
sync = ops._ffn_sync(torch.device('cuda', torch.cuda.current_device()))
hsave = torch.zeros(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=hdt, device=dev)
mp = lib.otr_ffn_split_padded_rows(M)
usave = torch.zeros(mp, F, dtype=hdt, device=dev)
dh3 = torch.zeros(mp, 2 * F, dtype=hdt, device=dev)
dx3 = torch.zeros(M, d, device=dev)
for _ in range(n):
    L.check(lib.otr_ffn_ln_fwd_split(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), 0.1, 0, 1e-5, p(y), p(y16),
                                     p(z), p(mean), p(rstd), p(hsave), p(usave), p(scratch), nb, p(sync), sync.numel(), M, F, d, st()), 'fwd3')
    L.check(lib.otr_ffn_bwd_split(p(da), p(hsave), p(P[2]), p(P[3]), p(dh3), p(dx3), p(dx3), p(scratch), nb, p(sync), sync.numel(), M, F, d, st()), 'bwd3')
torch.cuda.synchronize()
print('done')
