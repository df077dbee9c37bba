#!/usr/bin/env python3
"""Time the row-block fused FFN kernels (csrc/ffn_fused.hip) at the benchmark shape: M = 32 x 249 rows, d 256, d_ff 2048."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops, _lib as L     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=7968)
    ap.add_argument('--dff', type=int, default=2048)
    ap.add_argument('--mode', default='bf16')
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    ops.set_compute_dtype(a.mode)
    dev = 'cuda'
    M, d, F = a.rows, 256, a.dff
    hdt = ops.act_dtype()
    w1 = torch.randn(2 * F, d, device=dev) / math.sqrt(d)
    w2 = torch.randn(d, F, device=dev) / math.sqrt(F)
    b1, b2 = torch.randn(2 * F, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    x = torch.randn(M, d, device=dev)
    x16 = x.to(hdt)
    P = ops.ffn_packs(w1, w2)
    y, y16, z = torch.empty_like(x), torch.empty_like(x16), torch.empty_like(x)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    seed = ops.rng_seed_tensor(dev)
    lib = L.load()
    p, st = ops._p, ops._stream

    def fwd(pd):
        L.check(lib.otr_ffn_ln_fwd(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), pd, 0, 1e-5,
                                   p(y), p(y16), p(z), p(mean), p(rstd), M, F, d, st()), 'fwd')
    da = (torch.randn(M, d, device=dev) * 0.01).to(hdt)
    dh = torch.empty(M, 2 * F, dtype=hdt, device=dev)
    u = torch.empty(M, F, dtype=hdt, device=dev)
    dx = torch.zeros(M, d, device=dev)
    bpart = torch.empty((M + 31) // 32, 2 * F, device=dev)

    def bwd():
        L.check(lib.otr_ffn_bwd(p(x16), p(da), p(P[0]), p(b1), p(P[2]), p(P[3]), p(dh), p(u), p(bpart), p(dx), p(dx), M, F, d, st()), 'bwd')

    nb = lib.otr_ffn_split_scratch_bytes(M)
    scratch = torch.empty(nb // 4, device=dev)
    sync = ops._ffn_sync(torch.device('cuda', torch.cuda.current_device()))

    hsave = torch.empty(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=hdt, device=dev)
    mp = lib.otr_ffn_split_padded_rows(M)
    usave = torch.empty(mp, F, dtype=hdt, device=dev)
    dh3 = torch.empty(mp, 2 * F, dtype=hdt, device=dev)
    dx3 = torch.zeros(M, d, device=dev)

    def fwd3(pd, save=False):
        L.check(lib.otr_ffn_ln_fwd_split(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), pd, 0, 1e-5, p(y), p(y16),
                                         p(z), p(mean), p(rstd), p(hsave) if save else None, p(usave) if save else None, p(scratch), nb,
                                         p(sync), sync.numel(), M, F, d, st()), 'fwd3')

    def bwd3():
        L.check(lib.otr_ffn_bwd_split(p(da), p(hsave), p(P[2]), p(P[3]), p(dh3), None, p(dx3), p(scratch), nb, p(sync), sync.numel(),
                                      M, F, d, st()), 'bwd3')

    def wgrad():
        ops.linear_wgrad_raw(dh, x16, None)
        ops.linear_wgrad_raw(da, u, None)

    def old_fwd():
        h = torch.empty(M, 2 * F, dtype=hdt, device=dev)
        uu = torch.empty(M, F, dtype=hdt, device=dev)
        L.check(lib.otr_ffn_glu_fwd(p(x16), d, p(ops.weight_lp(w1)), d, p(b1), p(h), p(uu), M, F, d, st()), 'glu')
        a_ = ops.linear_fwd_raw(uu, ops.weight_lp(w2), b2, hdt)
        return a_

    def timeit(fn, n):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    res = {'rows': M, 'dff': F, 'mode': a.mode}
    res['v1_fwd_us'] = timeit(lambda: fwd(0.0), a.iters)
    res['v1_fwd_drop_us'] = timeit(lambda: fwd(0.1), a.iters)
    res['v1_bwd_us'] = timeit(bwd, a.iters)
    fwd(0.0)
    y1 = y.clone()
    res['split_fwd_us'] = timeit(lambda: fwd3(0.0), a.iters)
    res['split_fwd_drop_us'] = timeit(lambda: fwd3(0.1), a.iters)
    fwd3(0.0)
    res['split_vs_v1_y_rel'] = float((y - y1).norm() / y1.norm())
    res['split_arrivals_mod4'] = int((sync.view(-1, 8)[:, 0] % 4).abs().sum().item())
    res['split_fwd_save_us'] = timeit(lambda: fwd3(0.0, True), a.iters)
    res['split_bwd_us'] = timeit(bwd3, a.iters)
    # parity of the split backward (saved tiles) against the 32-row kernel (recompute): dx without skip, dh, u
    dxz = torch.zeros(M, d, device=dev)
    L.check(lib.otr_ffn_bwd(p(x16), p(da), p(P[0]), p(b1), p(P[2]), p(P[3]), p(dh), p(u), p(bpart), None, p(dxz), M, F, d, st()), 'bwd')
    fwd3(0.0, True)
    bwd3()
    res['split_vs_v1_dx_rel'] = float((dx3 - dxz).norm() / dxz.norm())
    res['split_vs_v1_dh_rel'] = float((dh3[:M].float() - dh.float()).norm() / dh.float().norm())
    res['split_vs_v1_u_rel'] = float((usave[:M].float() - u.float()).norm() / u.float().norm())
    for ab in (1, 2, 3):
        lib.otr_debug_set(4, ab)
        res['split_fwd_ablate%d_us' % ab] = timeit(lambda: fwd3(0.0), a.iters)
        res['split_bwd_ablate%d_us' % ab] = timeit(bwd3, a.iters)
    for ab in (4, 8, 12):     # 4 = no global stores of the saved tiles / dh, 8 = no tile loads (backward), 12 = neither
        lib.otr_debug_set(4, ab)
        if ab == 4:
            res['split_fwd_save_ablate4_us'] = timeit(lambda: fwd3(0.0, True), a.iters)
        res['split_bwd_ablate%d_us' % ab] = timeit(bwd3, a.iters)
    lib.otr_debug_set(4, 0)
    # in the step every layer saves into its own buffers (12 x 98 MB) and the backward kernels read them back much later: the
    # same launches cycling through N distinct (hsave, usave / dh) sets
    for ncyc in (2, 12):
        hs = [torch.empty_like(hsave) for _ in range(ncyc)]
        us = [torch.empty_like(usave) for _ in range(ncyc)]
        ds = [torch.empty_like(dh3) for _ in range(ncyc)]
        cnt = [0]

        def fwd_c():
            i = cnt[0] % ncyc
            cnt[0] += 1
            L.check(lib.otr_ffn_ln_fwd_split(p(x), p(x16), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0, 1e-5, p(y),
                                             p(y16), p(z), p(mean), p(rstd), p(hs[i]), p(us[i]), p(scratch), nb, p(sync), sync.numel(),
                                             M, F, d, st()), 'fwd3')

        def bwd_c():
            i = cnt[0] % ncyc
            cnt[0] += 1
            L.check(lib.otr_ffn_bwd_split(p(da), p(hs[i]), p(P[2]), p(P[3]), p(ds[i]), None, p(dx3), p(scratch), nb, p(sync),
                                          sync.numel(), M, F, d, st()), 'bwd3')
        res['split_fwd_save_cycle%d_us' % ncyc] = timeit(fwd_c, 48)
        res['split_bwd_cycle%d_us' % ncyc] = timeit(bwd_c, 48)
        del hs, us, ds
    # ... and through 12 distinct weight sets as well (in the step no layer finds its packed weights in the L2)
    Ps = [ops.ffn_packs(torch.randn(2 * F, d, device=dev) / math.sqrt(d), torch.randn(d, F, device=dev) / math.sqrt(F)) for _ in range(12)]
    Ps = [[t.clone() for t in q] for q in Ps]
    hs = [torch.empty_like(hsave) for _ in range(12)]
    us = [torch.empty_like(usave) for _ in range(12)]
    ds = [torch.empty_like(dh3) for _ in range(12)]
    xs = [(torch.randn(M, d, device=dev), ) for _ in range(12)]
    xs = [(t[0], t[0].to(hdt)) for t in xs]
    cnt = [0]

    def fwd_w():
        i = cnt[0] % 12
        cnt[0] += 1
        L.check(lib.otr_ffn_ln_fwd_split(p(xs[i][0]), p(xs[i][1]), p(Ps[i][0]), p(b1), p(Ps[i][1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0,
                                         1e-5, p(y), p(y16), p(z), p(mean), p(rstd), p(hs[i]), p(us[i]), p(scratch), nb, p(sync),
                                         sync.numel(), M, F, d, st()), 'fwd3')

    def bwd_w():
        i = cnt[0] % 12
        cnt[0] += 1
        L.check(lib.otr_ffn_bwd_split(p(xs[i][1]), p(hs[i]), p(Ps[i][2]), p(Ps[i][3]), p(ds[i]), None, p(dx3), p(scratch), nb, p(sync),
                                      sync.numel(), M, F, d, st()), 'bwd3')
    res['split_fwd_save_cold_us'] = timeit(fwd_w, 48)
    res['split_bwd_cold_us'] = timeit(bwd_w, 48)

    def fwd_wn():           # cold weights, nothing saved: is it the saves' store stream that pushes the weights out of the L2?
        i = cnt[0] % 12
        cnt[0] += 1
        L.check(lib.otr_ffn_ln_fwd_split(p(xs[i][0]), p(xs[i][1]), p(Ps[i][0]), p(b1), p(Ps[i][1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0,
                                         1e-5, p(y), p(y16), p(z), p(mean), p(rstd), None, None, p(scratch), nb, p(sync),
                                         sync.numel(), M, F, d, st()), 'fwd3')

    def fwd_xn():           # warm weights (one set), cold x, nothing saved
        i = cnt[0] % 12
        cnt[0] += 1
        L.check(lib.otr_ffn_ln_fwd_split(p(xs[i][0]), p(xs[i][1]), p(P[0]), p(b1), p(P[1]), p(b2), p(gamma), p(beta), p(seed), 0.0, 0,
                                         1e-5, p(y), p(y16), p(z), p(mean), p(rstd), None, None, p(scratch), nb, p(sync),
                                         sync.numel(), M, F, d, st()), 'fwd3')
    res['split_fwd_nosave_cold_us'] = timeit(fwd_wn, 48)
    res['split_fwd_nosave_coldx_warmw_us'] = timeit(fwd_xn, 48)
    lib.otr_debug_set(4, 1)
    res['split_fwd_save_cold_nodma_us'] = timeit(fwd_w, 48)
    res['split_bwd_cold_nodma_us'] = timeit(bwd_w, 48)
    lib.otr_debug_set(4, 0)
    del Ps, hs, us, ds, xs
    res['split_fwd_tflops'] = 2.0 * M * 3 * F * d / res['split_fwd_us'] / 1e6
    res['split_bwd_tflops'] = 2.0 * M * 3 * F * d / res['split_bwd_us'] / 1e6
    res['wgrad_pair_us'] = timeit(wgrad, 10)
    try:
        res['old_fwd_us'] = timeit(old_fwd, 20)
    except Exception as e:                                # noqa: BLE001
        res['old_fwd_us'] = str(e)
    fl_f = 2.0 * M * (2 * F * d + F * d)
    fl_b = 2.0 * M * (2 * F * d + F * d + 2 * F * d)
    res['v1_fwd_tflops'] = fl_f / res['v1_fwd_us'] / 1e6
    res['v1_bwd_tflops'] = fl_b / res['v1_bwd_us'] / 1e6
    print(json.dumps(res))


if __name__ == '__main__':
    main()
