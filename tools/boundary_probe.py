#!/usr/bin/env python3
"""Where do the ~4.7 us per node of the replayed training step come from?  (review r05, item 1; companion of tools/ubench/boundary.hip)

 A. chains of N dependent launches of LIBRARY kernels through the C ABI -- a trivial one (`otr_touch` of 256 x 256 x 64 B: 256
    workgroups, one 4-byte load per thread), a streaming one (`otr_scale_cast` of [7968, 256] fp32 -> 16-bit: 8.2 MB in, 4.1 MB out) --
    issued three ways: eager from Python, captured with hipStreamBeginCapture / hipGraphLaunch through ctypes (no torch object involved
    in the capture), captured by torch.cuda.CUDAGraph.  us per node, host clock around R replays.
 B. the WHOLE training step of bench.py (fp16, B = 32), captured by torch and captured raw, replayed: ms per step.
 C. the same step with K extra trivial nodes appended inside the graph: (ms(K) - ms(0)) / K = what one more node costs in THIS graph.
 D. the step's encoder forward alone (no autograd) as a chain, both captures.
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota
from opentransformer_amd import _lib as L, ops, synthetic as syn
from opentransformer_amd.dp import FlatDataParallel, FusedAdam

dev = torch.device('cuda', 0)
ops.set_compute_dtype('fp16')
lib = L.load()


def hip_runtime():
    """the libamdhip64 this process already has mapped (torch's), not a second copy"""
    for line in open('/proc/self/maps'):
        if 'libamdhip64' in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError('libamdhip64 is not mapped')


hip = hip_runtime()
for fn in ('hipStreamBeginCapture', 'hipStreamEndCapture', 'hipGraphInstantiate', 'hipGraphLaunch', 'hipGraphExecDestroy', 'hipGraphDestroy',
           'hipStreamSynchronize', 'hipGraphGetNodes'):
    getattr(hip, fn).restype = C.c_int


def hck(e, what):
    if e != 0:
        raise RuntimeError('%s -> hip error %d' % (what, e))


class RawGraph:
    """hipStreamBeginCapture ... hipStreamEndCapture on torch's CURRENT stream, instantiated and launched through ctypes"""

    def __init__(self, fn, stream):
        self.stream = stream
        sp = C.c_void_p(stream.cuda_stream)
        g = C.c_void_p()
        import gc
        gc.collect(); gc.disable()
        try:
            with torch.cuda.stream(stream):
                hck(hip.hipStreamBeginCapture(sp, 2), 'begin capture (relaxed)')       # hipStreamCaptureModeRelaxed
                try:
                    fn()
                finally:
                    hck(hip.hipStreamEndCapture(sp, C.byref(g)), 'end capture')
        finally:
            gc.enable()
        n = C.c_size_t(0)
        hck(hip.hipGraphGetNodes(g, None, C.byref(n)), 'get nodes')
        self.nodes = n.value
        self.ex = C.c_void_p()
        hck(hip.hipGraphInstantiate(C.byref(self.ex), g, None, None, C.c_size_t(0)), 'instantiate')
        hip.hipGraphDestroy(g)

    def replay(self):
        hck(hip.hipGraphLaunch(self.ex, C.c_void_p(self.stream.cuda_stream)), 'graph launch')


def wall(step, reps, inner=1):
    for _ in range(3):
        step()
    res = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / reps / inner * 1e6)
    return sorted(res)[2], res


side = torch.cuda.Stream()


def three_ways(name, launch, N=200, R=20):
    """launch(i) issues node i on torch's current stream"""
    def chain():
        for i in range(N):
            launch(i)
    with torch.cuda.stream(side):
        chain()
        torch.cuda.synchronize()
        eager, _ = wall(chain, 3, N)
    side.synchronize()
    raw = RawGraph(chain, side)
    with torch.cuda.stream(side):
        rawt, _ = wall(raw.replay, R, N)
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        chain()
    tg, _ = wall(g.replay, R, N)
    print('A  %-46s eager %6.2f   raw hipGraph %6.2f (%d nodes)   torch CUDAGraph %6.2f   us per node' % (name, eager, rawt, raw.nodes, tg), flush=True)


buf = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
x32 = torch.randn(7968, 256, device=dev)
y16 = torch.empty(7968, 256, device=dev, dtype=torch.float16)


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


three_ways('otr_touch 256 x 256 x 64 B (256 workgroups)', lambda i: lib.otr_touch(C.c_void_p(buf.data_ptr()), 256 * 256 * 64, st()))
three_ways('otr_touch 64 B (1 workgroup)', lambda i: lib.otr_touch(C.c_void_p(buf.data_ptr()), 64 * 256, st()))
three_ways('otr_scale_cast [7968,256] fp32 -> fp16', lambda i: lib.otr_scale_cast(C.c_void_p(x32.data_ptr()), C.c_void_p(y16.data_ptr()), x32.numel(), 1.0, st()))

# ------------------------------------------------------------------------------------------------ the real step
cfg = syn.c2_model(residual_dropout=0.1)
inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=0)
inputs = {k: v.to(dev) for k, v in inputs.items()}
targets = {k: v.to(dev) for k, v in targets.items()}
model = ota.SpeechToText(cfg)
syn.fill_state_dict_(model.state_dict(), 1234)
model = model.to(dev).train()
dp = FlatDataParallel(model)
opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0, noam=dict(model_size=256, warmup_steps=12000, factor=1.0))
extra = [0]
grid_of_extra = [256]
inter = [0]            # trivial nodes inserted BEHIND every proj_ln_fwd launch of the forward pass (12 sites, each between two heavy kernels)
_pal = ops.proj_add_layernorm


def _pal_probe(*a, **k):
    r = _pal(*a, **k)
    for _ in range(inter[0]):
        lib.otr_touch(C.c_void_p(buf.data_ptr()), 256 * 256 * 64, st())
    return r


ops.proj_add_layernorm = _pal_probe


def whole():
    dp.zero_grad(next_dropout_step=True)
    loss, _ = dp(inputs, targets)
    ops.backward(loss)
    opt.step(1.0)
    for _ in range(extra[0]):
        lib.otr_touch(C.c_void_p(buf.data_ptr()), grid_of_extra[0] * 256 * 64, st())


with torch.cuda.stream(side):
    for _ in range(3):
        whole()
side.synchronize(); torch.cuda.synchronize()


def cap_torch():
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        whole()
    return g


g0 = cap_torch()
base, allb = wall(g0.replay, 20)
print('B  whole step, torch CUDAGraph                  %.1f us per step  (windows %s)' % (base, ' '.join('%.0f' % v for v in allb)), flush=True)
try:
    with torch.cuda.stream(side):
        for _ in range(2):
            whole()
    side.synchronize()
    raw = RawGraph(whole, side)
    with torch.cuda.stream(side):
        rawt, allr = wall(raw.replay, 20)
    print('B  whole step, raw hipGraph through ctypes      %.1f us per step  (%d nodes; windows %s)' % (rawt, raw.nodes, ' '.join('%.0f' % v for v in allr)), flush=True)
except Exception as e:                                             # noqa: BLE001
    print('B  raw capture of the whole step failed: %s: %s' % (type(e).__name__, e), flush=True)
    torch.cuda.synchronize()
for grid in (256, 1):
    grid_of_extra[0] = grid
    for K in (100, 300):
        extra[0] = K
        g = cap_torch()
        t, allt = wall(g.replay, 20)
        t0, _ = wall(g0.replay, 20)
        print('C  + %3d trivial nodes (%3d workgroups each): %.1f us per step, base re-read %.1f -> %.2f us per extra node' % (K, grid, t, t0, (t - t0) / K), flush=True)
        del g
extra[0] = 0
for k in (1, 4, 16):
    inter[0] = k
    g = cap_torch()
    t, _ = wall(g.replay, 20)
    t0, _ = wall(g0.replay, 20)
    print('D  %2d trivial nodes behind each of the 12 proj_ln_fwd launches (between heavy kernels): %.1f us per step, base %.1f -> %.2f us per extra node' % (k, t, t0, (t - t0) / (12 * k)), flush=True)
    del g
inter[0] = 0


# E. the forward pass alone (no autograd graph): ~80 dependent launches of the real kernels, three ways
def fwd_only():
    with torch.no_grad():
        dp(inputs, targets)


with torch.cuda.stream(side):
    for _ in range(2):
        fwd_only()
    e, _ = wall(fwd_only, 5)
side.synchronize()
try:
    raw = RawGraph(fwd_only, side)
    with torch.cuda.stream(side):
        rawt, _ = wall(raw.replay, 20)
    nn_ = raw.nodes
except Exception as ex:                                            # noqa: BLE001
    rawt, nn_ = float('nan'), -1
    print('E  raw capture failed: %s' % ex)
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with ops.graph_capture(g):
    fwd_only()
tg, _ = wall(g.replay, 20)
print('E  forward pass alone (%d nodes: conv, 12 x (rb_linear_ln, attn_fwd, proj_ln_fwd, ffn3_fwd), decoder, loss): eager %.1f   raw hipGraph %.1f   torch CUDAGraph %.1f   us per pass' % (nn_, e, rawt, tg), flush=True)

# eager: what the host can issue
with torch.cuda.stream(side):
    e, _ = wall(whole, 5)
print('B  whole step, eager                            %.1f us per step' % e, flush=True)
