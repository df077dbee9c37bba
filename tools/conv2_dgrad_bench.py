#!/usr/bin/env python3
"""otr_conv2_dgrad at the AISHELL shape (B=32, 1000 x 80 fbank, C1=64, C2=128): against the
column-matrix pair it replaces (otr_conv2_dgrad_cols + otr_conv2_col2im).  One JSON line."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import _lib as L, ops   # noqa: E402

ops.set_compute_dtype('fp16')
dev = 'cuda:0'
B, T, Fd, C1, C2 = 32, 1000, 80, 64, 128
T1, F1, T2, F2 = ops.conv_geometry(T, Fd)
adt = ops.act_dtype()
g2 = torch.randn(B, T2, F2, C2, device=dev).to(adt)
w2r = (torch.randn(C2, 3, 3, C1, device=dev) / 20).to(adt)
act1 = torch.randn(B, T1, F1, C1, device=dev).clamp_min(0).to(adt)
dact1 = torch.empty_like(act1)
dcol = torch.empty(B * T2 * F2, 9 * C1, device=dev, dtype=adt)
desc = L.ConvDesc(B, T, Fd, C1, C2, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
lib = L.load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def implicit():
    L.check(lib.otr_conv2_dgrad(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._stream()), 'dgrad')


def explicit():
    L.check(lib.otr_conv2_dgrad_cols(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(dcol), ops._stream()), 'cols')
    L.check(lib.otr_conv2_col2im(C.byref(desc), ops._p(dcol), ops._p(act1), ops._p(dact1), ops._stream()), 'col2im')


res = {'shape': [B, T, Fd, C1, C2], 'flops': 2.0 * B * T2 * F2 * C2 * 9 * C1}
res['implicit_us'] = timed(implicit)
a = dact1.clone()
res['column_pair_us'] = timed(explicit)
res['max_abs_diff_vs_column_pair'] = float((a.float() - dact1.float()).abs().max())
res['hbm_bytes_min'] = g2.numel() * 2 + act1.numel() * 2 * 2
res['tflops'] = res['flops'] / res['implicit_us'] / 1e6
res['hbm_gbps_min_traffic'] = res['hbm_bytes_min'] / res['implicit_us'] / 1e3
# ablations (otr_debug_set(10, v)): results are garbage, only the time counts
res['ablations_us'] = {}
for v, name in ((1, 'no mask loads / result stores'), (2, 'operand loads from one line'), (3, 'both')):
    lib.otr_debug_set(10, v)
    res['ablations_us'][name] = timed(implicit, 10)
lib.otr_debug_set(10, 0)
res['implicit_us_again'] = timed(implicit, 10)
# per-workgroup timeline of one launch (otr_debug_trace: 100 MHz real-time stamps at start / fragments built / end, class)
tr = torch.zeros(512 * 4, dtype=torch.int64, device=dev)
lib.otr_debug_trace(C.c_void_p(tr.data_ptr()))
implicit()
torch.cuda.synchronize()
lib.otr_debug_trace(None)
t = tr.view(512, 4).cpu()
t = t[t[:, 0] > 0]
t0 = int(t[:, 0].min())
tl = {}
for c in range(4):
    m = t[t[:, 3] == c]
    if len(m) == 0:
        continue
    us = lambda x: round(float(x) / 100.0, 2)                      # noqa: E731
    tl['class%d' % c] = {'workgroups': int(len(m)),
                         'start_us_min_max': [us(m[:, 0].min() - t0), us(m[:, 0].max() - t0)],
                         'fragments_built_us_median': us((m[:, 1] - m[:, 0]).median()),
                         'tiles_us_min_median_max': [us((m[:, 2] - m[:, 1]).min()), us((m[:, 2] - m[:, 1]).median()), us((m[:, 2] - m[:, 1]).max())],
                         'end_us_min_max': [us(m[:, 2].min() - t0), us(m[:, 2].max() - t0)]}
res['timeline'] = tl
print(json.dumps(res))
