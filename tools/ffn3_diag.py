#!/usr/bin/env python3
"""Where do the split FFN kernels (csrc/ffn3.hip) differ from the 32-row kernels?  Error maps of u / dh / dx by (32-row, 32-column) tile."""
import math, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
M = int(sys.argv[2]) if len(sys.argv) > 2 else 7968
ops.set_compute_dtype(mode)
dev = 'cuda'
d, F = 256, 2048
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
dh = torch.empty(M, 2 * F, dtype=hdt, device=dev)
u = torch.empty(M, F, dtype=hdt, device=dev)
dxz = torch.zeros(M, d, device=dev)
bpart = torch.empty((M + 31) // 32, 2 * F, device=dev)
L.check(lib.otr_ffn_bwd(p(x16), p(da), p(P[0]), p(b1), p(P[2]), p(P[3]), p(dh), p(u), p(bpart), None, p(dxz), M, F, d, st()), 'bwd')
nb = lib.otr_ffn_split_scratch_bytes(M)
scratch = torch.empty(nb // 4, device=dev)
sync = ops._ffn_sync(torch.device('cuda', torch.cuda.current_device()))
hsave = torch.zeros(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=hdt, device=dev)
mp = lib.otr_ffn_split_padded_rows(M)
usave = torch.zeros(mp, F, dtype=hdt, device=dev)
dh3 = torch.zeros(mp, 2 * F, dtype=hdt, device=dev)
dx3 = torch.zeros(M, d, device=dev)
L.check(lib.otr_ffn_ln_fwd_split(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0, 1e-5, p(y), p(y16),
                                 p(z), p(mean), p(rstd), p(hsave), p(usave), p(scratch), nb, p(sync), sync.numel(), M, F, d, st()), 'fwd3')
y_save = y.clone()
L.check(lib.otr_ffn_ln_fwd_split(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0, 1e-5, p(y), p(y16),
                                 p(z), p(mean), p(rstd), None, None, p(scratch), nb, p(sync), sync.numel(), M, F, d, st()), 'fwd3')
print('y save vs nosave rel', float((y_save - y).norm() / y.norm()))
L.check(lib.otr_ffn_bwd_split(p(da), p(hsave), p(P[2]), p(P[3]), p(dh3), None, p(dx3), p(scratch), nb, p(sync), sync.numel(), M, F, d, st()), 'bwd3')
torch.cuda.synchronize()
# fp32 reference of u
h = x16.float() @ w1.to(hdt).float().t() + b1
uref = h[:, :F] * torch.sigmoid(h[:, F:])


def tilemap(a, b, name, ct=32):
    e = (a.float() - b.float()).abs()
    R, Cc = e.shape
    Rt = R // 32 * 32
    t = e[:Rt].reshape(Rt // 32, 32, Cc // ct, ct).amax(dim=(1, 3))
    scale = float(b.float().abs().mean())
    bad = (t > 0.05 * scale)
    print(name, 'rel', float((a.float() - b.float()).norm() / b.float().norm()), 'scale', scale, 'bad tiles', int(bad.sum()), 'of', bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print('  first bad (rowtile, coltile):', idx[:12].tolist())
        print('  bad by coltile % 64:', torch.bincount(idx[:, 1] % 64, minlength=64).tolist())
        print('  bad by rowtile % 4:', torch.bincount(idx[:, 0] % 4, minlength=4).tolist())


tilemap(u, uref, 'v1 u vs ref')
tilemap(usave[:M], uref, 'split u vs ref')
tilemap(usave[:M], u, 'split u vs v1 u')
tilemap(dh3[:M], dh, 'split dh vs v1 dh')
tilemap(dx3, dxz, 'split dx vs v1 dx')
