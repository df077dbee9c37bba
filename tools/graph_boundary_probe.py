#!/usr/bin/env python3
"""What does a second hipGraph launch per step cost on this stack?  The training step of bench.py (fp16, B = 32) replayed as
 (a) one graph, (b) two separately captured copies of the whole step, alternating, (c) forward | backward as two graphs with a
 shared pool, (d) the same without sharing the pool.  ms per step, median of 5 windows of 20 steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota
from opentransformer_amd import ops, synthetic as syn
from opentransformer_amd.dp import FlatDataParallel, FusedAdam
dev = torch.device('cuda', 0)
ops.set_compute_dtype('fp16')
cfg = syn.c2_model(residual_dropout=0.1)
inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=0)
inputs = {k: v.to(dev) for k, v in inputs.items()}
targets = {k: v.to(dev) for k, v in targets.items()}
model = ota.SpeechToText(cfg)
syn.fill_state_dict_(model.state_dict(), 1234)
model = model.to(dev).train()
dp = FlatDataParallel(model)
opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0, noam=dict(model_size=256, warmup_steps=12000, factor=1.0))
held = []
def fwd():
    dp.zero_grad(); ops.next_dropout_step(dev)
    loss, _ = dp(inputs, targets)
    held[:] = [loss]
def bwd():
    held[0].backward()
def whole():
    fwd(); bwd(); opt.step(1.0)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        whole()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
def cap(fn, **kw):
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g, **kw):
        fn()
    return g
def timeit(step):
    for _ in range(5):
        step()
    res = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e3)
    return sorted(res)[2]
gA = cap(whole); print('(a) one graph                         %.3f ms' % timeit(gA.replay))
gB = cap(whole)
flip = [0]
def ab():
    (gA if flip[0] == 0 else gB).replay(); flip[0] ^= 1
print('(b) two execs of the whole step, a b a b  %.3f ms' % timeit(ab))
g1 = cap(fwd); g2 = cap(lambda: (bwd(), opt.step(1.0)), pool=g1.pool())
print('(c) forward | backward, shared pool       %.3f ms' % timeit(lambda: (g1.replay(), g2.replay())))
del g1, g2
g1 = cap(fwd); g2 = cap(lambda: (bwd(), opt.step(1.0)))
print('(d) forward | backward, separate pools    %.3f ms' % timeit(lambda: (g1.replay(), g2.replay())))
g3 = cap(lambda: opt.step(1.0))
del g1, g2
g1 = cap(lambda: (fwd(), bwd()))
print('(e) forward + backward | optimizer        %.3f ms' % timeit(lambda: (g1.replay(), g3.replay())))
