"""Tile x split-K sweep over the C2 train-step GEMM shapes (tuning input for gemm_launch_tiles)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib
from tools.gemm_bench import CASES

bf, f32 = torch.bfloat16, torch.float32
EXTRA = [('q dec fwd', 'fwd', (480, 256, 256), bf, bf, bf), ('w1 dec fwd', 'fwd', (480, 4096, 256), bf, bf, bf),
         ('w2 dec fwd', 'fwd', (480, 256, 2048), bf, bf, f32), ('w1 dec wgrad', 'wgrad', (480, 4096, 256), bf, bf, bf),
         ('w2 dec wgrad', 'wgrad', (480, 256, 2048), bf, bf, f32), ('vk wgrad', 'wgrad', (7968, 512, 256), bf, bf, bf),
         ('vocab wgrad', 'wgrad', (480, 4234, 256), bf, bf, f32), ('vocab dgrad', 'dgrad', (480, 4234, 256), f32, bf, f32)]


def main():
    ops.set_compute_dtype('bf16')
    lib = _lib.load()
    for name, kind, (m, n, k), xdt, wdt, ydt in CASES + EXTRA:
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
        res = {}
        for tile in (0, 64, 128):
            for ks in ((0,) if tile == 0 else (1, 2, 4, 8, 16, 32, 64)):
                lib.otr_debug_set(0, tile)
                lib.otr_debug_set(1, ks)
                for _ in range(2):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[(tile, ks)] = e0.elapsed_time(e1) / 10 * 1e3
        best = min(res, key=res.get)
        row = ' '.join('%d/%d:%.0f' % (t, s, v) for (t, s), v in res.items() if t)
        print('%-13s %-5s %5dx%5dx%5d auto %6.1f best %s %6.1f | %s' % (name, kind, m, n, k, res[(0, 0)], best, res[best], row))
    lib.otr_debug_set(0, 0)
    lib.otr_debug_set(1, 0)


if __name__ == '__main__':
    main()
