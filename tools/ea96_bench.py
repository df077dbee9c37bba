"""tuning aid: otr_attention_bias_bwd at the Conformer bench shape (B 32, T' 249, 4 heads x 96), timed with events, under otr_debug_set(33, v):
0 = the streamed dQ + dK/dV pair, 1 = csrc/encattn96.hip, 1 | 2a = its ablations (a & 1 no score-term loads, a & 2 no d bias stores, a & 4 no tiles)"""
import ctypes as C
import sys
import torch
from opentransformer_amd import _lib as L, ops

ops.set_compute_dtype('fp16')
lib = L.load()
adt = ops.act_dtype()
B, T, H, dk = 32, 249, 4, 96
d = H * dk
Pp = (2 * T - 1 + 7) // 8 * 8
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B, T, 3 * d, generator=g) * 0.5).to('cuda', adt)
quv = qkv[..., :d].contiguous()
bd = torch.randn(B, T, H, Pp, generator=g).to('cuda')
dout = torch.randn(B, T, d, generator=g).to('cuda', adt)
km = torch.ones(B, T, dtype=torch.uint8, device='cuda')
out = torch.empty(B, T, d, dtype=adt, device='cuda')
lse = torch.empty(B, H, T, dtype=torch.float32, device='cuda')
desc = ops._attn_desc(B, H, T, T, dk, adt, (T * d, d), (T * 3 * d, 3 * d), (T * 3 * d, 3 * d), (T * d, d), False)
L.check(lib.otr_attention_bias_fwd(C.byref(desc), ops._p(quv), ops._p(qkv, d), ops._p(qkv, 2 * d), ops._p(km), ops._p(bd), T * H * Pp, Pp, H * Pp, 1,
                                   ops._p(out), ops._p(lse), ops._stream()), 'fwd')
dbd = torch.zeros(B, T, H, Pp, dtype=adt, device='cuda')
dq = torch.empty(B, T, d, dtype=adt, device='cuda')
dkv = torch.empty(B, T, 3 * d, dtype=adt, device='cuda')
delta = torch.empty_like(lse)


def run():
    return lib.otr_attention_bias_bwd(C.byref(desc), ops._p(quv), ops._p(qkv, d), ops._p(qkv, 2 * d), ops._p(km), ops._p(bd), ops._p(dbd), ops._code(dbd.dtype),
                                      T * H * Pp, Pp, H * Pp, 1, ops._p(out), ops._p(dout), ops._p(lse), ops._p(delta), ops._p(dq), ops._p(dkv, d),
                                      ops._p(dkv, 2 * d), ops._stream())


for v in [int(a) for a in (sys.argv[1:] or ['1', '0'])]:
    lib.otr_debug_set(33, v)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print('debug_set(33, %2d): %.1f us' % (v, e0.elapsed_time(e1) * 50), flush=True)
lib.otr_debug_set(33, 1)
