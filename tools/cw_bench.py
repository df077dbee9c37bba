"""tuning aid: the conv2wide launch alone (conv2 input gradient at the Conformer bench shape), timed with events, under otr_debug_set(32, v)"""
import ctypes as C
import sys
import torch
from opentransformer_amd import _lib as L, ops

ops.set_compute_dtype('fp16')
lib = L.load()
adt = ops.act_dtype()
B, T, Fd, C1, C2 = 32, 1000, 80, 256, 256
T1, F1, T2, F2 = ops.conv_geometry(T, Fd)
g = torch.Generator().manual_seed(1)
act1 = torch.randn(B, T1, F1, C1, generator=g).clamp_min(0).to('cuda', adt)
w2r = (torch.randn(C2, 3, 3, C1, generator=g) / 48).to('cuda', adt)
b2 = torch.zeros(C2, device='cuda')
g2 = torch.randn(B, T2, F2, C2, generator=g).to('cuda', adt)
act2 = torch.empty(B, T2, F2, C2, dtype=adt, device='cuda')
dact1 = torch.empty_like(act1)
desc = L.ConvDesc(B, T, Fd, C1, C2, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
ws = ops._workspace(act1.device)


def run(which):
    return lib.otr_conv2_dgrad_wide(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._p(ws), ops._WS_BYTES, ops._stream())


for abl in [int(a) for a in (sys.argv[1:] or ['0'])]:
    lib.otr_debug_set(32, abl)
    for which in ('dgrad',):
        for _ in range(3):
            assert run(which) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(which)
        e1.record()
        torch.cuda.synchronize()
        print('ablate %2d %-5s %.1f us' % (abl, which, e0.elapsed_time(e1) * 100), flush=True)
lib.otr_debug_set(32, 0)
