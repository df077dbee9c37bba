"""Row-block projections of the attention sub-layers (csrc/rowblock.hip): otr_rb_linear, otr_proj_ln_fwd, otr_ln_bwd_proj
against plain fp32 torch references of the same maths (module/attention.py:62-75,120-140; the residual + LayerNorm of
encoder/transformer.py:54-56), and the fused sub-layer against the unfused kernels it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = {'bf16': 2e-2, 'fp16': 3e-3}


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('M', [7968, 1000, 33])
@pytest.mark.parametrize('N,K', [(256, 256), (768, 256)])
def test_rb_linear_forward_and_input_gradient(mode, M, N, K):
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        gen = torch.Generator().manual_seed(3)
        w = torch.nn.Parameter((torch.randn(N, K, generator=gen) / K ** 0.5).to(DEV))
        b = torch.randn(N, generator=gen).to(DEV)
        packs = ops.lin_packs(w)
        assert packs is not None
        wl = ops.weight_lp(w).float()
        x = torch.randn(M, K, generator=gen).to(DEV, adt)
        for out_dtype in (torch.float32, adt):
            y = ops.rb_linear_raw(x, packs[0], N, b, out_dtype)
            ref = x.float() @ wl.t() + b
            assert _rel(y, ref) < (1e-5 if out_dtype == torch.float32 else TOL[mode] / 4), (out_dtype, _rel(y, ref))
        # input gradient with the skip-connection gradient accumulated in place: dx = skip + dy . W
        dy = torch.randn(M, N, generator=gen).to(DEV, adt)
        skip = torch.randn(M, K, generator=gen).to(DEV)
        ref = skip + dy.float() @ wl
        dx = ops.rb_linear_raw(dy, packs[1], K, None, torch.float32, skip=skip)
        assert dx.data_ptr() == skip.data_ptr()
        assert _rel(dx, ref) < 1e-5, _rel(dx, ref)
        # embedded operand (row stride > columns), like a slice of a packed projection
        wide = torch.randn(M, K + 64, generator=gen).to(DEV, adt)
        y = ops.rb_linear_raw(wide[:, :K], packs[0], N, None, torch.float32)
        assert _rel(y, wide[:, :K].float() @ wl.t()) < 1e-5
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('M', [7968, 480, 45])
def test_proj_ln_matches_reference_and_unfused_path(mode, M):
    """forward and every gradient of LN(x + c W^T + b) (dropout 0) against torch autograd in fp32 on the same 16-bit operands"""
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        d = 256
        gen = torch.Generator().manual_seed(4)
        w = torch.nn.Parameter((torch.randn(d, d, generator=gen) / d ** 0.5).to(DEV))
        b = torch.nn.Parameter(torch.randn(d, generator=gen).to(DEV) * 0.1)
        gamma = torch.nn.Parameter((1 + 0.1 * torch.randn(d, generator=gen)).to(DEV))
        beta = torch.nn.Parameter((0.1 * torch.randn(d, generator=gen)).to(DEV))
        x = torch.randn(M, d, generator=gen).to(DEV).requires_grad_(True)
        c = torch.randn(M, d, generator=gen).to(DEV, adt).requires_grad_(True)
        gy = torch.randn(M, d, generator=gen).to(DEV)
        packs = ops.proj_ln_packs(x, c, w, gamma)
        assert packs is not None
        y = ops.proj_add_layernorm(x, c, w, b, gamma, beta, 0.0, 1e-5, packs)
        assert ops.lp_of(y) is not None and _rel(ops.lp_of(y), y) < TOL[mode] / 4
        y.backward(gy)
        got = [y.detach(), x.grad, c.grad, w.grad, b.grad, gamma.grad, beta.grad]
        # reference: same 16-bit operands, fp32 maths
        wl = ops.weight_lp(w).float().detach().requires_grad_(True)
        x2 = x.detach().clone().requires_grad_(True)
        c2 = c.detach().float().requires_grad_(True)
        b2, g2, be2 = (t.detach().clone().requires_grad_(True) for t in (b, gamma, beta))
        ref = F.layer_norm(x2 + c2 @ wl.t() + b2, (d,), g2, be2, 1e-5)
        ref.backward(gy)
        want = [ref.detach(), x2.grad, c2.grad, wl.grad, b2.grad, g2.grad, be2.grad]
        names = ['y', 'dx', 'dc', 'dw', 'db', 'dgamma', 'dbeta']
        for n, a, r in zip(names, got, want):
            tol = 2e-5 if n in ('y', 'dx') else TOL[mode]      # dc is stored 16-bit; dw / db / affine sums see the 16-bit da
            assert _rel(a, r) < tol, (n, _rel(a, r))
    finally:
        ops.set_compute_dtype('bf16')


def test_proj_ln_dropout_masks_agree_between_forward_and_backward():
    """with p > 0 the backward kernel regenerates the forward mask: the branch gradient is exactly zero where the forward
    pass dropped the branch, and the keep rate is 1 - p"""
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp16')
    try:
        adt, d, M, p = ops.act_dtype(), 256, 2000, 0.3
        gen = torch.Generator().manual_seed(5)
        w = torch.nn.Parameter(torch.eye(d).to(DEV))                   # branch = c (exactly representable)
        gamma = torch.nn.Parameter(torch.ones(d, device=DEV))
        beta = torch.nn.Parameter(torch.zeros(d, device=DEV))
        x = torch.zeros(M, d, device=DEV, requires_grad=True)
        c = (torch.randn(M, d, generator=gen).abs() + 1.0).to(DEV, adt).requires_grad_(True)
        packs = ops.proj_ln_packs(x, c, w, gamma)
        y = ops.proj_add_layernorm(x, c, w, None, gamma, beta, p, 1e-5, packs)
        y.backward(torch.randn(M, d, generator=gen).to(DEV))
        # identity weight: dc = da; dropped positions have da == 0 exactly
        dropped = (c.grad == 0)
        rate = float(dropped.float().mean())
        assert abs(rate - p) < 0.01, rate
        # forward: z = x + mask * c / (1-p): y is the LayerNorm of z; rows where everything was kept or dropped aside,
        # the smallest entries of each row of y are the dropped ones (z = 0 there, kept entries are >= 1/(1-p) > 0)
        z_like = y.detach()
        row_min = z_like.min(dim=1, keepdim=True).values
        fwd_dropped = (z_like - row_min).abs() < 1e-6
        rows = dropped.any(dim=1)
        assert torch.equal(fwd_dropped[rows], dropped[rows])
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
def test_encoder_layer_rowblock_equals_generic_path(mode, monkeypatch):
    """a post-norm encoder layer (encoder/transformer.py:16-90) forward + backward with the row-block kernels on and off"""
    from opentransformer_amd import nn as onn, ops
    ops.set_compute_dtype(mode)
    try:
        torch.manual_seed(11)
        layer = onn.TransformerEncoderLayer(4, 256, 2048, 0.0, 0.0, 0.0, activation='glu').to(DEV).train()
        x0 = torch.randn(4, 250, 256, device=DEV)
        mask = torch.ones(4, 250, dtype=torch.bool, device=DEV)
        mask[1, 200:] = False
        res = []
        for rb in (True, False):
            monkeypatch.setattr(ops, '_RB', rb)
            for prm in layer.parameters():
                prm.grad = None
            x = ops.attach_lp(x0.clone().requires_grad_(True), None)
            xin = ops.add_layernorm(x, None, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), 0.0, 1e-5)   # gives x a 16-bit twin
            y, _ = layer(xin, mask)
            y.square().mean().backward()
            res.append([y.detach().clone(), x.grad.clone()] + [prm.grad.clone() for prm in layer.parameters()])
        scale = max(float(t.abs().max()) for t in res[1][2:])
        for a, b in zip(*res):
            # (the key bias gradient is analytically zero: both paths leave rounding noise there, compared on the absolute scale)
            err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3 * scale)
            assert err < 4 * TOL[mode], err
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
def test_packs_follow_the_optimizer_for_every_row_block_linear(mode):
    """a (256,256) Linear that FlatDataParallel's NAME list did not cover -- vk_proj with
    share_vk_proj=True (module/attention.py:128-132) -- reached the row-block path in ops.LinearFn through a cache keyed by
    (_version, data_ptr), which FusedAdam's raw-pointer update never changes: forward and input gradient kept using the
    INITIAL weights.  Packs are now registered by shape; two optimizer steps on the row-block path must track the same two
    steps on the generic GEMM path (OTR_NO_ROWBLOCK)."""
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    from opentransformer_amd.nn import MultiHeadedCrossAttention
    ops.set_compute_dtype(mode)
    was = ops._RB
    try:
        gen = torch.Generator().manual_seed(12)
        B, T, Lq, d = 8, 160, 12, 256                       # 1280 memory rows: above the row-block threshold
        memory = torch.randn(B, T, d, generator=gen).to(DEV)
        query = torch.randn(B, Lq, d, generator=gen).to(DEV)
        mask = torch.ones(B, T, dtype=torch.bool, device=DEV)
        outs = {}
        for rb in (True, False):
            ops._RB = rb
            torch.manual_seed(5)
            att = MultiHeadedCrossAttention(4, d, d, 0.0, share_vk_proj=True).to(DEV)
            dp = FlatDataParallel(att)
            opt = FusedAdam(dp, lr=1e-2, clip_grad=0.0, weight_decay=0.0, loss_scale=0.0)     # 15 % of a weight per step: a stale operand shows
            if rb:
                assert getattr(att.vk_proj.weight, '_otr_lin_packs', None) is not None, 'vk_proj packs not registered by shape'
            trace = []
            for _ in range(3):
                dp.zero_grad()
                y, _ = dp(ops.attach_lp(query, query.to(ops.act_dtype())), ops.attach_lp(memory, memory.to(ops.act_dtype())), mask)
                trace.append(y.detach().float().clone())
                (y.float() ** 2).mean().backward()
                opt.step(1.0)
            outs[rb] = trace
        assert _rel(outs[True][0], outs[False][0]) < TOL[mode]
        moved = _rel(outs[False][2], outs[False][0])
        assert moved > 8 * TOL[mode], ('the optimizer did not move the output enough for this test to see a stale pack', moved)
        for i in (1, 2):                                      # after the updates the two paths still agree: a stale pack would
            e = _rel(outs[True][i], outs[False][i])           # leave the row-block output where it started, i.e. `moved` away
            assert e < max(4 * TOL[mode], 0.1 * moved), (i, e, moved)
    finally:
        ops._RB = was
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode,p_drop', [('fp16', 0.0), ('fp16', 0.1), ('bf16', 0.1)])
def test_ffn_layernorm_backward_in_the_next_projections_launch(mode, p_drop, monkeypatch):
    """Two post-norm encoder layers under FlatDataParallel (in-place flat gradients, deferred weight gradients): layer 2's q|k|v
    input-gradient launch runs the backward of layer 1's closing LayerNorm in its epilogue (ops.LnOutLink, otr_rb_linear_ln_bwd)
    -- same gradients as with the separate LayerNorm-backward launch (OTR_LNOUT_LINK=0), dropout masks included, and the fused
    launch is really taken (once: layer 2 has no successor)."""
    from opentransformer_amd import nn as onn, ops
    from opentransformer_amd.dp import FlatDataParallel

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1 = onn.TransformerEncoderLayer(4, 256, 2048, 0.0, 0.0, p_drop, activation='glu')
            self.l2 = onn.TransformerEncoderLayer(4, 256, 2048, 0.0, 0.0, p_drop, activation='glu')
            self.g = torch.nn.Parameter(torch.ones(256))
            self.b = torch.nn.Parameter(torch.zeros(256))

        def forward(self, x, mask):
            h = ops.add_layernorm(ops.attach_lp(x, None), None, self.g, self.b, 0.0, 1e-5)     # gives the stream its 16-bit twin
            h, _ = self.l1(h, mask)
            h, _ = self.l2(h, mask)
            return h

    ops.set_compute_dtype(mode)
    try:
        torch.manual_seed(5)
        model = Two().to(DEV).train()
        dp = FlatDataParallel(model)
        x0 = torch.randn(9, 250, 256, device=DEV)
        mask = torch.ones(9, 250, dtype=torch.bool, device=DEV)
        mask[2, 180:] = False
        gy = torch.randn(9, 250, 256, device=DEV)
        calls = []
        raw = ops.rb_linear_ln_bwd_raw
        monkeypatch.setattr(ops, 'rb_linear_ln_bwd_raw', lambda *a: (calls.append(1), raw(*a))[1])
        res = []
        for fused in (True, False):
            monkeypatch.setattr(ops, '_LNOUT', fused)
            dp.zero_grad()
            ops._state['rng_offset'] = 0                     # same dropout masks in both passes
            x = x0.clone().requires_grad_(True)
            y = dp(x, mask)
            y.backward(gy)
            res.append((y.detach().clone(), x.grad.clone(), dp.flat_grad.clone()))
        assert len(calls) == 1
        for a, b, name in zip(res[0], res[1], ('y', 'dx', 'flat gradient')):
            assert _rel(a, b) < 3e-5, (name, _rel(a, b))      # the affine / bias sums are grouped by 32 instead of 16 rows (unseeded input: 1.1e-5 seen)
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('M,K,with_skip', [(7968, 768, True), (1000, 768, False), (45, 256, True)])
def test_rb_linear_ln_bwd_matches_reference(mode, M, K, with_skip):
    """otr_rb_linear_ln_bwd against torch autograd in fp32 on the same 16-bit operands: for y = LN(z) (affine gamma, beta) and
    dy = skip + g . W, the launch returns dz, the 16-bit branch gradient (= dz, dropout 0) and per-workgroup sums whose totals
    are dgamma, dbeta and the column sums of the branch gradient."""
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        adt, d = ops.act_dtype(), 256
        gen = torch.Generator().manual_seed(6)
        w = torch.nn.Parameter((torch.randn(K, d, generator=gen) / d ** 0.5).to(DEV))       # y_next = x W^T: W is [K outputs, 256 inputs]
        packs = ops.lin_packs(w)
        assert packs is not None
        wl = ops.weight_lp(w).float()
        g = torch.randn(M, K, generator=gen).to(DEV, adt)
        skip = torch.randn(M, d, generator=gen).to(DEV) if with_skip else None
        z = torch.randn(M, d, generator=gen).to(DEV) * 1.5 + 0.3
        gamma = (1 + 0.1 * torch.randn(d, generator=gen)).to(DEV)
        beta = (0.1 * torch.randn(d, generator=gen)).to(DEV)
        mean = z.mean(1)
        rstd = (z.var(1, unbiased=False) + 1e-5).rsqrt()
        dx, da, part = ops.rb_linear_ln_bwd_raw(g, packs[1], skip, (z, mean, rstd, gamma, None, 0.0, 0))
        # reference
        dy = g.float() @ wl + (skip if skip is not None else 0.0)
        z2, g2, b2 = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        F.layer_norm(z2, (d,), g2, b2, 1e-5).backward(dy)
        assert _rel(dx, z2.grad) < 2e-5, _rel(dx, z2.grad)
        assert _rel(da, z2.grad) < TOL[mode] / 4
        sums = part.sum(0)
        assert _rel(sums[:d], g2.grad) < 2e-5 and _rel(sums[d:2 * d], b2.grad) < 2e-5
        assert _rel(sums[2 * d:], z2.grad.sum(0)) < 1e-4
    finally:
        ops.set_compute_dtype('bf16')
