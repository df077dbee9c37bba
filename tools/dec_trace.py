#!/usr/bin/env python3
"""Per-workgroup shader-clock timeline of the fused decoder launches (csrc/declayer.hip, otr_debug_trace): phases of every kernel at
the bench shape (B = 32, L = 15, T' = 249, d_ff 2048), median over workgroups, in cycles and us at the measured clock."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from opentransformer_amd import ops, _lib as L
import test_gpu_decoder_fused as T
ops.set_compute_dtype('fp16')
B, Lq, Tk, nl, dff, vocab = 32, 15, 249, 2, 2048, 4234
dec = T.make_decoder(nl, dff, vocab, 0.1, seed=1).train()
tokens, memory, key_mask, gy = T.inputs(B, Lq, Tk, vocab, seed=2)
for _ in range(3):
    T.run_hip(dec, tokens, memory, key_mask, gy, fused=True)
torch.cuda.synchronize()
tr = torch.zeros(16384 + 6 * 256 * 16, dtype=torch.int64, device='cuda')
lib = L.load()
lib.otr_debug_trace(ops._p(tr))
T.run_hip(dec, tokens, memory, key_mask, gy, fused=True)
torch.cuda.synchronize()
lib.otr_debug_trace(None)
t = tr.cpu().numpy()[16384:].reshape(6, 256, 16)
names = {0: ('self_fwd', ['prologue(LN)', 'qkv gemm', 'attention (1 wave)', 'out gemm', 'slab store']),
         1: ('cross_fwd', ['kv issue + prologue(LN)', 'q gemm', 'flash loop', 'partials -> LDS', 'merge', 'out gemm + store']),
         2: ('ffn_fwd', ['prologue(LN)', 'chunk loop', 'tiles -> LDS']),
         3: ('ffn_bwd', ['x stage + LN bwd', 'chunk loop', 'barrier']),
         4: ('cross_bwd', ['kv issue + LN bwd', 'dctx + q load', 'flash-bwd loop', 'dq merge', 'dy gemm + store']),
         5: ('self_bwd', ['LN bwd + qkv load', 'dctx', 'attention bwd (2 waves)', 'dy gemm + store'])}
for k, (nm, ph) in names.items():
    a = t[k]
    live = a[:, 0] > 0
    a = a[live]
    n = len(ph) + 1
    d = np.diff(a[:, :n], axis=1).astype(np.float64)
    tot = a[:, n - 1] - a[:, 0]
    print('%-10s workgroups %3d  total cycles median %7.0f (p90 %7.0f); start spread %d' % (nm, a.shape[0], np.median(tot), np.percentile(tot, 90), a[:, 0].max() - a[:, 0].min()))
    for i, p_ in enumerate(ph):
        print('    %-26s median %7.0f   p90 %7.0f' % (p_, np.median(d[:, i]), np.percentile(d[:, i], 90)))

# tuning build (make DEFS=-DDL_PROBE LIBDIR=../lib_probe, OTR_LIB_DIR=...): the self-attention forward launch's prologue in pieces
a = t[0]
a = a[a[:, 0] > 0]
if a.shape[0] and (a[:, 8] > 0).all():
    for nm, lo, hi in (('entry -> loads issued (LN rows + slabs, q|k|v weight stream)', 0, 8), ('-> everything landed (vmcnt 0)', 8, 9),
                       ('-> LayerNorm arithmetic + LDS image', 9, 10), ('-> barrier', 10, 1)):
        d = (a[:, hi] - a[:, lo]).astype(np.float64)
        print('    probe %-62s median %7.0f   p90 %7.0f' % (nm, np.median(d), np.percentile(d, 90)))
