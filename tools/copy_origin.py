#!/usr/bin/env python3
"""Which host-side ops launch the device-to-device copies (and torch's own elementwise kernels) of one eager training step?
torch.profiler with stacks; groups the CPU ops whose device work is a copy / fill / add kernel by (op chain, shapes, first
frame inside this package).  Complements copy_sites.py, which cannot see copies made inside the autograd engine."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota   # noqa: E402
from opentransformer_amd import ops, synthetic as syn   # noqa: E402
from opentransformer_amd.dp import FlatDataParallel, FusedAdam   # noqa: E402

ops.set_compute_dtype('fp16')
dev = torch.device('cuda:0')
model = ota.SpeechToText(syn.c2_model(residual_dropout=0.1))
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
    dp.all_reduce_gradients()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

WANT = ('copyBuffer', 'Memcpy', 'Memset', 'elementwise_kernel', 'FillFunctor', 'CatArray', 'fillBuffer')
groups = collections.Counter()
total = collections.Counter()
for e in prof.events():
    ks = [k for k in getattr(e, 'kernels', []) if any(w in k.name for w in WANT)]
    if not ks:
        continue
    chain, p = [], e
    while p is not None and len(chain) < 5:
        chain.append(p.name[:48])
        p = p.cpu_parent
    frame = next((f for f in (e.stack or []) if 'opentransformer_amd' in f or 'bench' in f), '?')
    key = (' <- '.join(chain), str(e.input_shapes)[:80], frame[-90:], ks[0].name[:40])
    groups[key] += len(ks)
    total[ks[0].name[:40]] += len(ks)
# device-to-device copies are runtime calls (hipMemcpyAsync -> a memcpy node = __amd_rocclr_copyBuffer under hipGraph), not kernels
# of an op: walk up from the runtime event to the aten op / autograd node that made it
mem = collections.Counter()
for e in prof.events():
    if 'hipMemcpy' not in e.name and 'Memcpy' not in e.name:
        continue
    chain, p, frame, shapes = [], e, '?', ''
    while p is not None and len(chain) < 7:
        chain.append(p.name[:48])
        if frame == '?' and p.stack:
            frame = next((f for f in p.stack if 'opentransformer_amd' in f or 'bench' in f), '?')
        if not shapes and getattr(p, 'input_shapes', None):
            shapes = str(p.input_shapes)[:80]
        p = p.cpu_parent
    mem[(' <- '.join(chain), shapes, frame[-100:])] += 1
print('memcpy runtime calls in one step:', sum(mem.values()))
for (chain, shapes, frame), n in mem.most_common(40):
    print('%3d  %s\n       shapes %s\n       at %s' % (n, chain, shapes, frame))
print('device kernels of interest in one step:', dict(total))
for (chain, shapes, frame, kn), n in groups.most_common(60):
    print('%3d  %-38s %s\n       shapes %s\n       at %s' % (n, kn, chain, shapes, frame))
