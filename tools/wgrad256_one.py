"""Launches the 256-wide weight-gradient kernel a few times on the step's 55 wide problems (for rocprofv3 --pmc passes).
usage: python tools/wgrad256_one.py [grid] [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L   # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.set_compute_dtype('fp16')
lib = L.load()
dev, adt, M = 'cuda:0', ops.act_dtype(), 7968
g = torch.Generator(device=dev).manual_seed(1)


def rnd(*s):
    return torch.randn(*s, device=dev, generator=g).to(adt)


items = []
for _ in range(12):
    x, dqkv, ctx, dout = rnd(M, 256), rnd(M, 768), rnd(M, 256), rnd(M, 256)
    x1, dh, u, dy2 = rnd(M, 256), rnd(M, 4096), rnd(M, 2048), rnd(M, 256)
    for dy, xx in ((dqkv, x), (dout, ctx), (dh, x1), (dy2, u)):
        items.append((dy, xx, torch.zeros(dy.shape[1], xx.shape[1], device=dev)))
mem, dkv = rnd(M, 256), rnd(M, 3072)
for i in range(6):
    items.append((dkv[:, 512 * i:512 * (i + 1)], mem, torch.zeros(512, 256, device=dev)))
items.append((rnd(M, 256), rnd(M, 608), torch.zeros(256, 608, device=dev)))
lib.otr_debug_set(6, 1)
lib.otr_debug_set(7, grid)
if len(sys.argv) > 3:
    lib.otr_debug_set(8, int(sys.argv[3]))
for _ in range(n):
    ops._wq['w'], ops._wq['b'] = list(items), []
    ops.flush_weight_grads()
torch.cuda.synchronize()
print('done')
