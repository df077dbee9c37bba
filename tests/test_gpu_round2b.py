"""Second batch of round-2 kernels: ReLU backward with the bias-gradient sums in the same pass (frontend/conv.py:63-66),
the re-tiled GLU backward (module/ffn.py:40, module/conformer.py:44-46), row mask with a type change, the pre-norm residual
link (encoder/conformer.py:50-73) and the in-place BatchNorm / depthwise-conv parameter gradients of the Conformer
convolution module (module/conformer.py:36-57) -- each against plain fp32 torch of the same maths or against the unfused path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('rows,cols', [(159360, 128), (1000, 64), (33, 256), (5, 8), (4097, 1024)])
def test_relu_bwd_colsum(mode, rows, cols):
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        gen = torch.Generator().manual_seed(rows + cols)
        y = torch.randn(rows, cols, generator=gen).clamp_min(0).to(DEV, adt)
        g = torch.randn(rows, cols, generator=gen).to(DEV, adt)
        res = ops.relu_bwd_colsum_raw(y, g)
        if res is None:
            assert cols // (4 if adt == torch.float32 else 8) > 256 or cols % 8
            return
        out, part = res
        want = torch.where(y > 0, g, torch.zeros_like(g))
        assert torch.equal(out, want)
        assert part.shape[1] == cols and part.dtype == torch.float32
        assert _rel(part.sum(0), want.float().sum(0)) < 1e-5
        assert torch.equal(out, ops.relu_bwd_raw(y, g))
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('M,F,masked,sig', [(7968, 1536, False, True), (240, 384, True, False), (1000, 2048, False, False),
                                            (33, 24, True, False), (65, 40, False, True)])
def test_glu_bwd_tilings(mode, M, F, masked, sig):
    """dh = GLU'(h) * du and the per-strip bias partials for wide, narrow (16 / 32 / 64 column threads) and odd shapes"""
    import ctypes as C
    from opentransformer_amd import _lib as L, ops
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        gen = torch.Generator().manual_seed(M + F)
        h = torch.randn(M, 2 * F, generator=gen)
        if sig:
            h[:, F:] = torch.sigmoid(h[:, F:])
        h = h.to(DEV, adt)
        du = torch.randn(M, F, generator=gen).to(DEV, adt)
        mask = (torch.rand(M, generator=gen) > 0.2).to(DEV, torch.uint8) if masked else None
        dh = torch.empty_like(h)
        nblk = (M + ops.GLU_RPB - 1) // ops.GLU_RPB
        part = torch.full((nblk, 2 * F), float('nan'), device=DEV)
        L.check(L.load().otr_glu_bwd(ops._p(h), ops._p(du), ops._p(dh), ops._p(part), ops._code(adt), M, F, ops._p(mask), int(sig),
                                     ops._stream()), 'otr_glu_bwd')
        a, gt = h[:, :F].float(), h[:, F:].float()
        s = gt if sig else torch.sigmoid(gt)
        d = du.float() * (mask.float().unsqueeze(1) if masked else 1.0)
        want = torch.cat([d * s, d * a * s * (1 - s)], dim=1)
        tol = 1e-6 if mode == 'fp32' else (6e-3 if mode == 'bf16' else 8e-4)
        assert _rel(dh, want) < tol, _rel(dh, want)
        assert _rel(part.sum(0), want.sum(0)) < 1e-4          # partials are summed before the 16-bit rounding of dh
    finally:
        ops.set_compute_dtype('bf16')


def test_row_mask_cast():
    from opentransformer_amd import _lib as L, ops
    for mode in ('bf16', 'fp16'):
        ops.set_compute_dtype(mode)
        try:
            adt = ops.act_dtype()
            gen = torch.Generator().manual_seed(7)
            x = torch.randn(777, 384, generator=gen).to(DEV)
            m = (torch.rand(777, generator=gen) > 0.3).to(DEV, torch.uint8)
            for src in (x, x.to(adt)):
                for odt in (torch.float32, adt):
                    out = torch.empty(777, 384, dtype=odt, device=DEV)
                    L.check(L.load().otr_row_mask_cast(ops._p(src), ops._code(src.dtype), ops._p(m), ops._p(out), ops._code(odt), 777, 384,
                                                       ops._stream()), 'otr_row_mask_cast')
                    want = (src.float() * m.float().unsqueeze(1)).to(odt)
                    assert torch.equal(out, want), (src.dtype, odt)
        finally:
            ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp32', 'fp16'])
def test_conformer_block_prenorm_link_equals_autograd_sum(mode, monkeypatch):
    """x + f(LN(x)): with the link the LayerNorm backward adds the skip-connection gradient itself; without it autograd sums the
    two -- same numbers (one fp32 add either way), every parameter gradient included"""
    from opentransformer_amd import nn as onn, ops
    ops.set_compute_dtype(mode)
    try:
        torch.manual_seed(3)
        blk = onn.ConformerEncoderBlock(64, 128, 5, 4, residual_dropout=0.0).to(DEV).train()
        x0 = torch.randn(3, 50, 64, device=DEV)
        mask = torch.ones(3, 50, dtype=torch.bool, device=DEV)
        mask[1, 40:] = False
        pos = onn.relative_sinusoid(50, 64, DEV)
        res = []
        for linked in (True, False):
            if not linked:
                monkeypatch.setattr(ops, 'new_prenorm_link', lambda: None)
            for prm in blk.parameters():
                prm.grad = None
            x = x0.clone().requires_grad_(True)
            y, _ = blk(x, mask, pos)
            y.square().mean().backward()
            res.append([y.detach().clone(), x.grad.clone()] + [prm.grad.clone() for prm in blk.parameters() if prm.grad is not None])   # (post_ffn is never applied, as shipped)
        scale = max(float(t.abs().max()) for t in res[1][2:])
        for a, b in zip(*res):
            # (the rel-pos attention backward accumulates with atomics: runs differ at 1e-5; gradients that are analytically zero --
            #  a bias in front of BatchNorm -- are rounding noise on both sides and are compared on the absolute scale)
            assert float((a - b).abs().max()) < 5e-4 * max(float(b.abs().max()), 1e-3 * scale), _rel(a, b)
    finally:
        ops.set_compute_dtype('bf16')


def test_conformer_conv_module_inplace_parameter_gradients():
    """under FlatDataParallel the BatchNorm and depthwise-conv parameter gradients are accumulated where they live (the
    reduction launch of otr_bn_swish_bwd, the sums of otr_dwconv_bwd) -- equal to the gradients autograd returns without it"""
    import copy
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel
    ops.set_compute_dtype('fp16')
    try:
        mod = ota.ConformerConvolutionModule(128, 7).to(DEV).train()
        syn.fill_state_dict_(mod.state_dict(), 9)
        ref = copy.deepcopy(mod)
        dp = FlatDataParallel(mod)
        dp.zero_grad()
        gen = torch.Generator().manual_seed(1)
        x = torch.randn(4, 90, 128, generator=gen).to(DEV)
        g = torch.randn(4, 90, 128, generator=gen).to(DEV, ops.act_dtype())
        mask = torch.ones(4, 90, dtype=torch.bool, device=DEV)
        mask[2, 70:] = False
        for rep in range(2):                               # second pass: accumulation on top of the first
            mod(x.clone().requires_grad_(True), mask).backward(g)
        want = torch.autograd.grad(ref(x.clone().requires_grad_(True), mask), list(ref.parameters()), g)
        for (name, p), w in zip(mod.named_parameters(), want):
            if name == 'depthwise_conv.bias':              # zero gradient in front of BatchNorm: roundoff on both sides
                continue
            assert _rel(p.grad, 2 * w) < 2e-3, (name, _rel(p.grad, 2 * w))
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('B,T,Fdim,C1,C2', [(32, 1000, 80, 64, 128), (3, 97, 40, 64, 128), (2, 200, 80, 32, 64), (1, 7, 3, 64, 128),
                                            (5, 331, 83, 64, 128)])
def test_conv2_implicit_input_gradient_equals_column_path(mode, B, T, Fdim, C1, C2, monkeypatch):
    """otr_conv2_dgrad (four parity-class GEMMs, no column matrix) against otr_conv2_dgrad_cols + otr_conv2_col2im on the
    same operands: same products, fp32 accumulation in a different order, one 16-bit rounding at the end on both sides --
    and against torch's conv2d input gradient in fp32 on the small cases (frontend/conv.py:63-66 backward)."""
    import math
    import torch.nn.functional as F
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        gen = torch.Generator().manual_seed(B * T + Fdim)
        x = torch.randn(B, T, Fdim, generator=gen).to(DEV)
        w1 = (torch.randn(C1, 1, 3, 3, generator=gen) / 3).to(DEV).requires_grad_(True)
        b1 = (0.1 * torch.randn(C1, generator=gen)).to(DEV).requires_grad_(True)
        w2 = (torch.randn(C2, C1, 3, 3, generator=gen) / math.sqrt(9 * C1)).to(DEV).requires_grad_(True)
        b2 = (0.1 * torch.randn(C2, generator=gen)).to(DEV).requires_grad_(True)
        res = []
        for implicit in (True, False):
            monkeypatch.setattr(ops, '_CONV2_IMPLICIT_DGRAD', implicit)
            act2 = ops.ConvSubsampleFn.apply(x, w1, b1, w2, b2)
            g = torch.randn(act2.shape, generator=torch.Generator().manual_seed(5)).to(DEV, act2.dtype)
            res.append(torch.autograd.grad(act2, (w1, b1, w2, b2), g))
        # dw1 / db1 are sums over dact1 (the tensor the two paths produce differently); dw2 / db2 do not depend on it
        for name, a, b in zip(('dw1', 'db1', 'dw2', 'db2'), *res):
            # (the column path rounds every tap's contribution to 16 bits before col2im sums them; the implicit kernel rounds once)
            assert _rel(a, b) < (6e-3 if mode == 'bf16' else 8e-4), (name, _rel(a, b))
            if name == 'dw2':                                  # (db2 goes through the atomics of the stand-alone column sum here)
                assert torch.equal(a, b), name
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
def test_conv2_implicit_input_gradient_elementwise(mode):
    """the dact1 tensor itself, through the C ABI, against an fp32 torch conv_transpose of the same 16-bit operands"""
    import ctypes as C
    import torch.nn.functional as F
    from opentransformer_amd import _lib as L, ops
    ops.set_compute_dtype(mode)
    try:
        adt = ops.act_dtype()
        for (B, T, Fd, C1, C2) in [(2, 61, 30, 64, 128), (3, 40, 17, 32, 64), (1, 250, 80, 64, 128), (9, 333, 80, 64, 128), (1, 7, 3, 32, 64)]:
            T1, F1, T2, F2 = ops.conv_geometry(T, Fd)
            gen = torch.Generator().manual_seed(T)
            g2 = torch.randn(B, T2, F2, C2, generator=gen).to(DEV, adt)
            w2r = (torch.randn(C2, 3, 3, C1, generator=gen) / 20).to(DEV, adt)
            act1 = torch.randn(B, T1, F1, C1, generator=gen).clamp_min(0).to(DEV, adt)
            dact1 = torch.full_like(act1, float('nan'))
            desc = L.ConvDesc(B, T, Fd, C1, C2, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
            rc = L.load().otr_conv2_dgrad(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._stream())
            assert rc == 0, rc
            # reference: conv2d(act1 [B,C1,T1,F1], w [C2,C1,3,3], stride 2, pad (0,1)) input gradient
            w = w2r.float().permute(0, 3, 1, 2).contiguous()
            a1 = act1.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
            out = F.conv2d(a1, w, None, stride=2, padding=(0, 1))
            assert out.shape == (B, C2, T2, F2)
            (gi,) = torch.autograd.grad(out, a1, g2.float().permute(0, 3, 1, 2).contiguous())
            want = (gi * (a1 > 0)).permute(0, 2, 3, 1)
            assert not torch.isnan(dact1.float()).any()
            assert _rel(dact1, want) < (4e-3 if mode == 'bf16' else 5e-4), _rel(dact1, want)
            assert bool((dact1[act1 <= 0] == 0).all())
    finally:
        ops.set_compute_dtype('bf16')
