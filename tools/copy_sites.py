#!/usr/bin/env python3
"""Where do the device-to-device copies of one training step come from?  Wraps Tensor.copy_ / clone / contiguous / to / torch.cat
for one eager step and counts the calling lines inside this package (tuning aid: every copy is a 3-5 us launch)."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota   # noqa: E402
from opentransformer_amd import ops, synthetic as syn   # noqa: E402
from opentransformer_amd.dp import FlatDataParallel, FusedAdam   # noqa: E402

ops.set_compute_dtype('fp16')
dev = torch.device('cuda:0')
cfg = syn.c2_model(residual_dropout=0.1)
model = ota.SpeechToText(cfg)
syn.fill_state_dict_(model.state_dict(), 1234)
model = model.to(dev).train()
dp = FlatDataParallel(model)
opt = FusedAdam(dp)
inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=0)
inputs = {k: v.to(dev) for k, v in inputs.items()}
targets = {k: v.to(dev) for k, v in targets.items()}


def step():
    dp.zero_grad()
    ops.next_dropout_step(dev)
    loss, _ = dp(inputs, targets)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
sites = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'opentransformer_amd' in fr.filename and 'copy_sites' not in fr.filename:
            return '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.line)
    return '?'


def wrap(owner, name, pred):
    orig = getattr(owner, name)

    def f(*a, **k):
        if pred(a, k):
            sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)
    return orig


is_cuda = lambda a: isinstance(a[0], torch.Tensor) and a[0].is_cuda   # noqa: E731
o1 = wrap(torch.Tensor, 'copy_', lambda a, k: is_cuda(a))
o2 = wrap(torch.Tensor, 'clone', lambda a, k: is_cuda(a))
o3 = wrap(torch.Tensor, 'contiguous', lambda a, k: is_cuda(a) and not a[0].is_contiguous())
o4 = wrap(torch.Tensor, 'to', lambda a, k: is_cuda(a))
o5 = wrap(torch.Tensor, 'float', lambda a, k: is_cuda(a) and a[0].dtype != torch.float32)
o6 = wrap(torch, 'cat', lambda a, k: True)
o7 = wrap(torch, 'zeros', lambda a, k: True)
o8 = wrap(torch, 'zeros_like', lambda a, k: True)
o9 = wrap(torch.Tensor, 'fill_', lambda a, k: is_cuda(a))
o10 = wrap(torch.Tensor, 'zero_', lambda a, k: is_cuda(a))
step()
torch.cuda.synchronize()
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
    print('%4d  %-12s %s' % (n, name, where))
