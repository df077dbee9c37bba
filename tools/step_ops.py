#!/usr/bin/env python3
"""Which Python line launches each small torch (non-library) kernel of the training step: one eager step under torch.profiler with
stacks; prints every ATen kernel of the step with the innermost frame inside this repo."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota
from opentransformer_amd import ops, synthetic as syn
from opentransformer_amd.dp import FlatDataParallel, FusedAdam
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
ops.set_compute_dtype('fp16')
model_name = sys.argv[1] if len(sys.argv) > 1 else 'transformer'
cfg = syn.c2_model(residual_dropout=0.1) if model_name == 'transformer' else syn.conformer_model(False, 0.1)
inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=0)
inputs = {k: v.to(dev) for k, v in inputs.items()}
targets = {k: v.to(dev) for k, v in targets.items()}
model = ota.SpeechToText(cfg)
syn.fill_state_dict_(model.state_dict(), 1234)
model = model.to(dev).train()
dp = FlatDataParallel(model)
opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0, noam=dict(model_size=256, warmup_steps=12000, factor=1.0))
def step():
    dp.zero_grad()
    ops.next_dropout_step(dev)
    loss, _ = dp(inputs, targets)
    loss.backward()
    opt.step(dp.all_reduce_gradients()[0])
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0:
        continue
    if ev.cpu_children and any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        continue                     # count the leaf op only
    site = next((f for f in (ev.stack or []) if root in f and 'tools/step_ops.py' not in f), '(no repo frame: autograd engine / torch internals)')
    key = (ev.name, site.replace(root + '/', ''))
    n, t = rows.get(key, (0, 0.0))
    rows[key] = (n + 1, t + ev.device_time_total)
tot = 0.0
for (name, site), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tot += t
    print('%8.1f us  x%-3d %-28s %s' % (t, n, name, site[:150]))
print('total ATen device time %.1f us' % tot)
