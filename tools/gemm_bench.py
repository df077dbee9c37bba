"""Per-shape GEMM timing for the shapes of the C2 train step (B=32, T'=249 -> M=7968)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib

bf, f32 = torch.bfloat16, torch.float32
M = 7968
CASES = [  # name, kind, (M, N, K), x dtype, w dtype, y dtype
    ('qkv fwd', 'fwd', (M, 768, 256), bf, bf, bf), ('out fwd', 'fwd', (M, 256, 256), bf, bf, f32),
    ('w1 fwd', 'fwd', (M, 4096, 256), bf, bf, bf), ('w2 fwd', 'fwd', (M, 256, 2048), bf, bf, f32),
    ('fe fwd', 'fwd', (M, 256, 2560), bf, bf, f32), ('vk fwd', 'fwd', (M, 512, 256), bf, bf, bf),
    ('vocab fwd', 'fwd', (480, 4234, 256), bf, bf, f32),
    ('w1 fwd f32src', 'fwd', (M, 4096, 256), f32, f32, bf),
    ('qkv dgrad', 'dgrad', (M, 768, 256), f32, bf, bf), ('w1 dgrad', 'dgrad', (M, 4096, 256), f32, bf, bf),
    ('w2 dgrad', 'dgrad', (M, 256, 2048), bf, bf, f32), ('out dgrad', 'dgrad', (M, 256, 256), bf, bf, f32),
    ('qkv wgrad', 'wgrad', (M, 768, 256), bf, bf, bf), ('w1 wgrad', 'wgrad', (M, 4096, 256), bf, bf, bf),
    ('w2 wgrad', 'wgrad', (M, 256, 2048), bf, bf, f32), ('out wgrad', 'wgrad', (M, 256, 256), bf, bf, f32),
]


def run(tile, ks):
    lib = _lib.load()
    lib.otr_debug_set(0, tile)
    lib.otr_debug_set(1, ks)
    print('--- tile %s ksplit %s' % (tile or 'auto', ks or 'auto'))
    for name, kind, (m, n, k), xdt, wdt, ydt in CASES:
        x = torch.randn(m, k, device='cuda').to(xdt)
        w = (torch.randn(n, k, device='cuda') / 16).to(wdt)
        dy = torch.randn(m, n, device='cuda').to(ydt)
        b = torch.randn(n, device='cuda')
        if kind == 'fwd':
            fn = lambda: ops.linear_fwd_raw(x, w, b, ydt)
        elif kind == 'dgrad':
            fn = lambda: ops.linear_dgrad_raw(dy, w, xdt)
        else:
            fn = lambda: ops.linear_wgrad_raw(dy, x, w)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%-14s %-6s M=%5d N=%5d K=%5d  %8.1f us  %7.1f TF/s' % (name, kind, m, n, k, us, 2.0 * m * n * k / us / 1e6))


if __name__ == '__main__':
    ops.set_compute_dtype('bf16')
    for tile, ks in [(0, 0), (64, 0), (128, 0)]:
        run(tile, ks)
