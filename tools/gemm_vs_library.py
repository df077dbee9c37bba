"""How far are the tile GEMMs from the vendor library on the Conformer's shapes?  Times otr_linear_fwd (ops.linear_fwd_raw) against
torch.nn.functional.linear (hipBLASLt / rocBLAS underneath) for y[M,N] = x[M,K] w[N,K]^T in fp16, 100 back-to-back launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L          # noqa: E402


def bench(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ops.set_compute_dtype('fp16')
    M = 7968
    lib = L.load()
    print('%6s %6s %6s | %9s %9s %9s | %9s %9s | TF/s ours, library' % ('M', 'N', 'K', 'ours f16', 'ours f32o', 'library', '128t f16', '128t f32o'))
    for N, K in [(384, 384), (384, 768), (384, 1152), (384, 1536), (768, 384), (1152, 384), (1536, 384), (256, 256), (768, 256), (256, 2048), (4096, 256)]:
        x = torch.randn(M, K, device='cuda', dtype=torch.float16)
        w = torch.randn(N, K, device='cuda', dtype=torch.float16)
        t_h = bench(lambda: ops.linear_fwd_raw(x, w, None, torch.float16))
        t_f = bench(lambda: ops.linear_fwd_raw(x, w, None, torch.float32))
        t_l = bench(lambda: torch.nn.functional.linear(x, w))
        lib.otr_debug_set(0, 128)                      # force 128 x 128 tiles
        t_h8 = bench(lambda: ops.linear_fwd_raw(x, w, None, torch.float16))
        t_f8 = bench(lambda: ops.linear_fwd_raw(x, w, None, torch.float32))
        lib.otr_debug_set(0, 0)
        fl = 2.0 * M * N * K
        print('%6d %6d %6d | %7.1f us %7.1f us %7.1f us | %7.1f us %7.1f us | %6.0f %6.0f' % (M, N, K, t_h, t_f, t_l, t_h8, t_f8, fl / t_h / 1e6, fl / t_l / 1e6))


if __name__ == '__main__':
    main()
