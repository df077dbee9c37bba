"""Times the weight-gradient launch of one backward pass (the 48 wide problems of the 12 encoder layers at B=32: M=7968)
through otr_linear_wgrad_grouped with the 256-wide kernel (csrc/wgrad256.hip) on and off, for several workgroup counts.
Prints one JSON line.  Usage: python tools/wgrad256_bench.py [--mode fp16] [--reps 10] [--grids 0,240,224]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='fp16')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--grids', default='0')
    ap.add_argument('--layers', type=int, default=12)
    ap.add_argument('--rows', type=int, default=7968)
    ap.add_argument('--ablate', default='')
    ap.add_argument('--base', type=int, default=0, help='otr_debug_set(8, v) for the main runs (0 = shipped policy)')
    ap.add_argument('--alias', action='store_true', help='every layer reads the SAME operand tensors (cache-resident working set)')
    a = ap.parse_args()
    ops.set_compute_dtype(a.mode)
    lib = L.load()
    lib.otr_debug_set(8, a.base)
    dev = 'cuda:0'
    adt = ops.act_dtype()
    M = a.rows
    g = torch.Generator(device=dev).manual_seed(1)

    def rnd(*s):
        return torch.randn(*s, device=dev, generator=g).to(adt)
    items = []
    shared = None
    for _ in range(a.layers):
        if shared is None or not a.alias:
            shared = (rnd(M, 256), rnd(M, 768), rnd(M, 256), rnd(M, 256), rnd(M, 256), rnd(M, 4096), rnd(M, 2048), rnd(M, 256))
        x, dqkv, ctx, dout, x1, dh, u, dy2 = shared
        for dy, xx in ((dqkv, x), (dout, ctx), (dh, x1), (dy2, u)):
            items.append((dy, xx, torch.zeros(dy.shape[1], xx.shape[1], device=dev)))
    if a.layers == 12:       # the other wide problems of the step: the decoder's cross-attention key/value slices, the frontend's Linear
        mem, dkv = rnd(M, 256), rnd(M, 3072)
        for i in range(6):
            items.append((dkv[:, 512 * i:512 * (i + 1)], mem, torch.zeros(512, 256, device=dev)))
        items.append((rnd(M, 256), rnd(M, 608), torch.zeros(256, 608, device=dev)))
    flops = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _ in items)
    nbytes = sum(dy.numel() * 2 + x.numel() * 2 + 2 * o.numel() * 4 for dy, x, o in items)

    def run(on, grid):
        lib.otr_debug_set(6, on)
        lib.otr_debug_set(7, grid)
        ops._wq['w'], ops._wq['b'] = list(items), []
        ops.flush_weight_grads()

    def timed(on, grid):
        for _ in range(2):
            run(on, grid)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run(on, grid)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    out = {'problems': len(items), 'gflop': flops / 1e9, 'algorithmic_mb': nbytes / 1e6, 'mode': a.mode}
    run(0, 0)
    torch.cuda.synchronize()
    ref = [o.clone() for _, _, o in items]
    for _, _, o in items:
        o.zero_()
    t = timed(0, 0)
    out['grouped128_ms'] = t
    out['grouped128_tflops'] = flops / t / 1e9
    for gs in a.grids.split(','):
        grid = int(gs)
        for _, _, o in items:
            o.zero_()
        run(1, grid)
        torch.cuda.synchronize()
        err = max(float((o - r).abs().max()) / float(r.abs().max()) for (_, _, o), r in zip(items, ref))
        t = timed(1, grid)
        out['wgrad256_grid%d' % grid] = {'ms': t, 'tflops': flops / t / 1e9, 'tb_per_s': nbytes / t / 1e9, 'max_rel_err_vs_128': err}
    for ab in [int(v) for v in a.ablate.split(',') if v]:
        lib.otr_debug_set(8, ab)
        out['ablate%d_grid-248_ms' % ab] = timed(1, -248)
        if ab >= 8:
            out['ablate%d_grid0_ms' % ab] = timed(1, 0)
    lib.otr_debug_set(8, a.base)
    lib.otr_debug_set(8, 0)
    lib.otr_debug_set(6, -1)
    lib.otr_debug_set(7, 0)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
