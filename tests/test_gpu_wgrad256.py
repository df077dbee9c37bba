"""256 x 256-tile weight-gradient launch (csrc/wgrad256.hip) behind otr_linear_wgrad_grouped: parity with a plain fp32 torch
reference of dw += dy^T x (the aten mm_backward the reference reaches through train/trainer.py:208), on the headline
shapes (M = 7968 rows; module/ffn.py:38-41 and module/attention.py:62-75 weights), on ragged row counts, with the
(tile, slab) space cut into few and into many chunks (pieces of one tile meeting at the turnstile), run twice for bitwise
determinism -- plus the hardware probe that pins the lane mapping of ds_read_b64_tr_b16 the kernel is built on."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _lib():
    from opentransformer_amd import _lib as L
    return L, L.load()


def test_trread_lane_mapping():
    """Within each 16-lane group, lane p receives element p % 4 of the 8-byte pieces addressed by lanes 4j + p // 4
    (j = 0..3): checked with a scattered address pattern (tools/emu/wgrad256_emu.py uses the same model)."""
    from opentransformer_amd import ops
    L, lib = _lib()
    img = torch.arange(2048, dtype=torch.int16)
    rng = np.random.default_rng(3)
    for trial in range(3):
        if trial == 0:
            addr = np.arange(64, dtype=np.int32) * 8                      # linear: a [16][4]-piece block per group
        else:
            addr = (rng.permutation(512)[:64] * 8).astype(np.int32)       # any 8-byte aligned pieces
        a = torch.from_numpy(addr).to(DEV)
        out = torch.zeros(256, dtype=torch.int16, device=DEV)
        L.check(lib.otr_debug_trread(ops._p(img.to(DEV)), ops._p(a), ops._p(out), ops._stream()), 'otr_debug_trread')
        got = out.cpu().numpy().reshape(64, 4)
        want = np.zeros((64, 4), np.int64)
        for lane in range(64):
            g, p = lane >> 4, lane & 15
            for j in range(4):
                want[lane, j] = addr[16 * g + 4 * j + (p >> 2)] // 2 + (p & 3)
        assert np.array_equal(got, want), (trial, got[:20], want[:20])


def _run(items, mode, grid, on=1, all_taken=True):
    """items: list of (dy, x, out).  Runs otr_linear_wgrad_grouped with the 256-wide launch forced on / off."""
    from opentransformer_amd import ops
    L, lib = _lib()
    L.check(lib.otr_debug_set(6, on), 'debug_set')
    L.check(lib.otr_debug_set(7, grid), 'debug_set')
    try:
        ops._wq['w'], ops._wq['b'] = list(items), []
        ops.flush_weight_grads()
        torch.cuda.synchronize()
        if on and all_taken:   # no piece gave up at a turnstile (the grouped kernel's table would overwrite the counter)
            assert lib.otr_debug_wgrad256_errors(ops._p(ops._workspace(items[0][0].device))) == 0
    finally:
        lib.otr_debug_set(6, -1)
        lib.otr_debug_set(7, 0)


SHAPES_SMALL = [(1032, 256, 256), (2048, 512, 256), (1544, 256, 768)]
SHAPES_RAGGED = [(2048, 256, 608), (1032, 384, 136), (1544, 136, 392), (2048, 256, 256)]
SHAPES_HEAD = [(7968, 256, 256), (7968, 768, 256), (7968, 4096, 256), (7968, 256, 2048)]


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('shapes,grid', [(SHAPES_SMALL, 0), (SHAPES_SMALL, 3), (SHAPES_SMALL, 7), (SHAPES_SMALL, -16), (SHAPES_RAGGED, 0),
                                         (SHAPES_RAGGED, 5), (SHAPES_HEAD, 0), (SHAPES_HEAD, 96), (SHAPES_HEAD, -250)])
def test_wgrad256_matches_reference(mode, shapes, grid):
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        gen = torch.Generator().manual_seed(5)
        items, refs = [], []
        for (m, n, k) in shapes:
            # an operand embedded in a wider matrix (ld > columns), like the packed q|k|v gradient
            dyw = torch.randn(m, n + 64, generator=gen).to(DEV, adt)
            dy = dyw[:, 32 * 0:n] if n % 512 else dyw[:, :n]
            x = torch.randn(m, k, generator=gen).to(DEV, adt)
            out = torch.full((n, k), 0.5, device=DEV)
            items.append((dy, x, out))
            refs.append(0.5 + dy.float().t() @ x.float())
        _run(items, mode, grid)
        first = [o.clone() for _, _, o in items]
        for (dy, x, out), ref in zip(items, refs):
            err = float((out - ref).abs().max()) / float(ref.abs().max())
            assert err < 2e-5, (tuple(dy.shape), tuple(x.shape), grid, err)     # fp32 accumulation of exact 16-bit products
        # second run on fresh buffers: bitwise identical (fixed accumulation order through the turnstile)
        for _, _, o in items:
            o.fill_(0.5)
        _run(items, mode, grid)
        for a, (_, _, o) in zip(first, items):
            assert torch.equal(a, o)
        # and equal (to rounding) to the 128-wide grouped kernel it replaces
        for _, _, o in items:
            o.fill_(0.5)
        _run(items, mode, grid, on=0)
        for a, (_, _, o) in zip(first, items):
            assert float((a - o).abs().max()) / float(a.abs().max()) < 2e-5
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('shapes,grid', [(SHAPES_SMALL, 0), (SHAPES_SMALL, 7), (SHAPES_RAGGED, 5), (SHAPES_HEAD, 0), (SHAPES_HEAD, -250)])
def test_wgrad256_stores_into_cleared_single_writer_buffers(shapes, grid):
    """otr_wgrad_item_t.overwrite (ops.register_single_writer_grads / gradients_cleared): a registered buffer that was just cleared
    receives the first partial sum of every tile as a STORE (through the turnstile too: piece 0 stores, the others accumulate);
    the result is bit-identical to the accumulating launch on zeros.  A second launch before the next clear accumulates.  That the
    store path really runs is shown on a buffer that is NOT zero when it is declared cleared: the old content is gone."""
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp16')
    try:
        adt = ops.act_dtype()
        gen = torch.Generator().manual_seed(11)
        items, refs = [], []
        for (m, n, k) in shapes:
            dy = torch.randn(m, n, generator=gen).to(DEV, adt)
            x = torch.randn(m, k, generator=gen).to(DEV, adt)
            items.append((dy, x, torch.zeros(n, k, device=DEV)))
            refs.append(dy.float().t() @ x.float())
        ptrs = {o.data_ptr() for _, _, o in items}
        _run(items, 'fp16', grid)                                   # unregistered buffers: read-modify-write on zeros
        base = [o.clone() for _, _, o in items]
        ops.register_single_writer_grads(ptrs)
        try:
            for _, _, o in items:
                o.zero_()
            ops.gradients_cleared(ptrs)
            _run(items, 'fp16', grid)                               # cleared + registered: the store path
            for a, (_, _, o), ref in zip(base, items, refs):
                assert torch.equal(a, o)
                assert float((o - ref).abs().max()) / float(ref.abs().max()) < 2e-5
            _run(items, 'fp16', grid)                               # not cleared in between: accumulates
            for a, (_, _, o) in zip(base, items):
                assert float((o - 2 * a).abs().max()) / float(a.abs().max()) < 1e-6
            for _, _, o in items:
                o.fill_(7.0)
            ops.gradients_cleared(ptrs)                             # (a lie, to see the store: nothing of the 7.0 survives)
            _run(items, 'fp16', grid)
            for a, (_, _, o) in zip(base, items):
                assert torch.equal(a, o)
            # the same buffer twice in one flush (a Linear applied twice): never stored, and the second product runs in a launch of
            # its own behind the first (two problems of ONE launch read-modify-write the same tiles unordered)
            for _, _, o in items:
                o.zero_()
            ops.gradients_cleared(ptrs)
            _run(items + [items[0]], 'fp16', grid)
            assert float((items[0][2] - 2 * base[0]).abs().max()) / float(base[0].abs().max()) < 1e-6
        finally:
            ops.unregister_single_writer_grads(ptrs)
    finally:
        ops.set_compute_dtype('bf16')


def test_wgrad256_mixed_with_unqualified_items():
    """Items the 256-wide launch cannot take (short contraction, odd sizes, fp32 operand) stay on the grouped kernel in
    the same call; results of both groups are right."""
    from opentransformer_amd import ops
    ops.set_compute_dtype('bf16')
    adt = ops.act_dtype()
    gen = torch.Generator().manual_seed(6)
    shapes = [(4000, 256, 256, adt), (480, 4096, 256, adt), (4000, 256, 256, torch.float32), (1000, 130, 68, adt),
              (4000, 512, 512, adt)]
    items, refs = [], []
    for (m, n, k, dyt) in shapes:
        dy = torch.randn(m, n, generator=gen).to(DEV, dyt)
        x = torch.randn(m, k, generator=gen).to(DEV, adt)
        out = torch.zeros(n, k, device=DEV)
        items.append((dy, x, out))
        refs.append(dy.float().t() @ x.float())
    _run(items, 'bf16', 0, all_taken=False)
    for (dy, x, out), ref in zip(items, refs):
        tol = 2e-5 if dy.dtype == adt else 1e-2           # an fp32 dy is rounded to the 16-bit type by the grouped kernel
        assert float((out - ref).abs().max()) / float(ref.abs().max()) < tol, tuple(dy.shape)


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('grid', [0, 5, -7])
def test_wgrad256_bias_gradient_rides_along(mode, grid):
    """a bias gradient (column sums of dy) queued next to the weight gradient of the same Linear is folded into the 256-wide
    launch (otr_wgrad_item_t.dbias); one whose matrix is not a weight-gradient operand stays on the column-sum kernel"""
    from opentransformer_amd import ops
    L, lib = _lib()
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        gen = torch.Generator().manual_seed(8)
        items, biases, refs = [], [], []
        for (m, n, k) in [(2048, 768, 256), (2048, 256, 512), (1544, 136, 392), (2048, 512, 256)]:
            wide = torch.randn(m, n + 8, generator=gen).to(DEV, adt)
            dy = wide[:, :n]
            x = torch.randn(m, k, generator=gen).to(DEV, adt)
            out, bias = torch.zeros(n, k, device=DEV), torch.full((n,), 2.0, device=DEV)
            items.append((dy, x, out))
            biases.append((dy, bias))
            refs.append((dy.float().t() @ x.float(), 2.0 + dy.float().sum(0)))
        lone = torch.randn(1000, 96, generator=gen).to(DEV, adt)
        lone_b = torch.zeros(96, device=DEV)
        L.check(lib.otr_debug_set(6, 1), 'debug_set')
        L.check(lib.otr_debug_set(7, grid), 'debug_set')
        try:
            ops._wq['w'], ops._wq['b'] = list(items), list(biases) + [(lone, lone_b)]
            ops.flush_weight_grads()
            torch.cuda.synchronize()
        finally:
            lib.otr_debug_set(6, -1)
            lib.otr_debug_set(7, 0)
        for (dy, x, out), (_, bias), (rw, rb) in zip(items, biases, refs):
            assert float((out - rw).abs().max()) / float(rw.abs().max()) < 2e-5
            assert float((bias - rb).abs().max()) / float(rb.abs().max()) < 2e-5, tuple(dy.shape)
        assert float((lone_b - lone.float().sum(0)).abs().max()) < 1e-3
    finally:
        ops.set_compute_dtype('bf16')


def test_turnstile_give_up_reaches_the_optimizer():
    """a turnstile wait that gives up leaves a finite but WRONG sum, which passes the NaN guard.  With the
    spin bound forced to 1 (otr_debug_set(11, 1)) pieces of split tiles give up; the sticky fault word
    (otr_set_fault_counter) must then make FusedAdam skip the update and count the event, and a healthy step afterwards must
    go through."""
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    L, lib = _lib()
    ops.set_compute_dtype('bf16')
    adt = ops.act_dtype()
    lin = torch.nn.Linear(256, 256).to(DEV)
    dp = FlatDataParallel(lin)
    opt = FusedAdam(dp, lr=1e-3, clip_grad=5.0)
    fault = ops.fault_counter(torch.device(DEV))
    gen = torch.Generator().manual_seed(8)
    items = []
    for (m, n, k) in SHAPES_HEAD:                       # 28 tiles of equal row count on 96 workgroups: 3 concurrent row ranges per tile
        dy = torch.randn(m, n, generator=gen).to(DEV, adt)
        x = torch.randn(m, k, generator=gen).to(DEV, adt)
        items.append((dy, x, torch.zeros(n, k, device=DEV)))
    w0 = dp.flat_param.clone()
    try:
        L.check(lib.otr_debug_set(11, 1), 'debug_set')
        dp.zero_grad()
        dp.flat_grad.fill_(1e-3)
        _run(items, 'bf16', 96, all_taken=False)
        torch.cuda.synchronize()
        assert int(fault.item()) > 0, 'no piece gave up with a spin bound of 1: the test does not exercise the give-up path'
        opt.step(1.0)
        st = opt.stats()
        assert st['skipped'] == 1 and st['faults'] > 0 and st['step'] == 0, st
        assert int(fault.item()) == 0                   # read and cleared by the update
        assert torch.equal(dp.flat_param, w0)           # nothing was applied
    finally:
        lib.otr_debug_set(11, 0)
    for _, _, o in items:
        o.zero_()
    _run(items, 'bf16', 96)
    assert int(fault.item()) == 0
    opt.step(1.0)
    st = opt.stats()
    assert st['skipped'] == 1 and st['step'] == 1 and not torch.equal(dp.flat_param, w0), st
