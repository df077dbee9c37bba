#!/usr/bin/env python3
"""Which torch (non-library) kernels does one training step launch, and from where?  (copies, adds, fills: launch overhead)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota
from opentransformer_amd import ops, synthetic as syn
from opentransformer_amd.dp import FlatDataParallel, FusedAdam

ops.set_compute_dtype('fp16')
dev = torch.device('cuda:0')
cfg = syn.c2_model(residual_dropout=0.1)
model = ota.SpeechToText(cfg); syn.fill_state_dict_(model.state_dict(), 1234); model = model.to(dev).train()
dp = FlatDataParallel(model); opt = FusedAdam(dp)
inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=0)
inputs = {k: v.to(dev) for k, v in inputs.items()}; targets = {k: v.to(dev) for k, v in targets.items()}

def step():
    dp.zero_grad(); ops.next_dropout_step(dev)
    loss, _ = dp(inputs, targets); loss.backward()

for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith('aten::'):
        continue
    if e.name in ('aten::empty', 'aten::empty_like', 'aten::view', 'aten::reshape', 'aten::as_strided', 'aten::empty_strided', 'aten::_unsafe_view',
                  'aten::slice', 'aten::select', 'aten::transpose', 'aten::permute', 'aten::t', 'aten::expand', 'aten::unsqueeze', 'aten::squeeze',
                  'aten::detach', 'aten::alias', 'aten::result_type', 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense', 'aten::resize_', 'aten::set_', 'aten::lift_fresh', 'aten::to', 'aten::contiguous', 'aten::view_as'):
        continue
    st = [s for s in (e.stack or []) if 'opentransformer_amd' in s or 'bench' in s or 'torch_ops' in s]
    key = (e.name, st[0].split('/')[-1] if st else '?')
    rows[key] = rows.get(key, 0) + 1
for (name, where), n in sorted(rows.items(), key=lambda kv: -kv[1])[:60]:
    print('%4d  %-28s %s' % (n, name, where))
