#!/usr/bin/env python3
"""Shader-clock timeline of conv2_fwd_kernel (csrc/conv2fwd.hip, otr_debug_trace) at the bench shape (32 x 1000 x 80)."""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L
ops.set_compute_dtype('fp16')
g = torch.Generator().manual_seed(1)
x = torch.randn(32, 1000, 80, generator=g).cuda()
w1 = (torch.randn(64, 1, 3, 3, generator=g) / 3).cuda(); b1 = torch.zeros(64).cuda()
w2 = (torch.randn(128, 64, 3, 3, generator=g) / math.sqrt(576)).cuda(); b2 = torch.zeros(128).cuda()
lib = L.load()
with torch.no_grad():
    for _ in range(3):
        ops.ConvSubsampleFn.apply(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    tr = torch.zeros(16384 + 10 * 256 * 16, dtype=torch.int64, device='cuda')
    lib.otr_debug_trace(ops._p(tr))
    ops.ConvSubsampleFn.apply(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    lib.otr_debug_trace(None)
t = tr.cpu().numpy()[16384:].reshape(10, 256, 16)[9]
t = t[t[:, 0] > 0]
d = np.diff(t[:, :5], axis=1).astype(np.float64)
print('workgroups %d total cycles median %.0f p90 %.0f' % (t.shape[0], np.median(t[:, 4] - t[:, 0]), np.percentile(t[:, 4] - t[:, 0], 90)))
for i, nm in enumerate(['weights + pads -> first barrier', 'stage item 0', 'tiles item 0', 'items 1..3']):
    print('  %-34s median %7.0f p90 %7.0f' % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
