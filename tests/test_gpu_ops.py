"""GPU (-m gpu): every C-ABI kernel family against a plain fp32 torch reference of the same op.

fp32 mode (v_mfma_f32_16x16x4_f32) must agree to roundoff; bf16 mode to bf16 resolution.  Each test
runs both modes.  Shapes include ragged / unaligned cases that force the scalar (non-vector) paths."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'
MODES = ['fp32', 'bf16', 'fp16']
TOL = {'fp32': 2e-5, 'bf16': 1.5e-2, 'fp16': 2e-3}
HALF = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.fixture(params=MODES)
def mode(request):
    from opentransformer_amd import ops
    ops.set_compute_dtype(request.param)
    yield request.param
    ops.set_compute_dtype('bf16')


def adt(mode):
    return torch.float32 if mode == 'fp32' else HALF[mode]


# ------------------------------------------------------------------------------------------ linear
@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (300, 200, 96), (7, 5, 3), (257, 131, 72), (1, 768, 256),
                                   (480, 4234, 256), (1000, 64, 2560)])
def test_linear_fwd_bwd(mode, M, N, K):
    from opentransformer_amd import ops
    for xdt in ([torch.float32] if mode == 'fp32' else [torch.float32, HALF[mode]]):
        x = rnd(M, K, seed=1).to(xdt).requires_grad_(True)
        w = (rnd(N, K, seed=2) / math.sqrt(K)).requires_grad_(True)
        b = rnd(N, seed=3).requires_grad_(True)
        for odt in ([torch.float32] if mode == 'fp32' else [torch.float32, HALF[mode]]):
            y = ops.linear(x, w, b, out_dtype=odt)
            yr = F.linear(x.float(), w, b)
            assert rel(y.float(), yr) < TOL[mode], ('fwd', xdt, odt)
            g = rnd(M, N, seed=4).to(odt)
            dx, dw, db = torch.autograd.grad(y, (x, w, b), g)
            dxr, dwr, dbr = torch.autograd.grad(yr, (x, w, b), g.float())
            assert rel(dx.float(), dxr.float()) < TOL[mode], ('dgrad', xdt, odt)
            assert rel(dw, dwr) < TOL[mode], ('wgrad', xdt, odt)
            assert rel(db, dbr) < TOL[mode], ('dbias', xdt, odt)


def test_linear_identity_asymmetric(mode):
    """A = I with an asymmetric B catches operand / output transposes (guide rule G9)."""
    from opentransformer_amd import ops
    n = 64
    x = torch.eye(n, device=DEV)
    w = (torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) % 97) / 16.0
    y = ops.linear(x, w, None)
    assert rel(y, w.t()) < TOL[mode]


def test_linear_relu_and_strided_input(mode):
    from opentransformer_amd import ops
    big = rnd(50, 200, seed=5)
    x = big[:, 8:104]                       # row stride 200, 96 columns
    w = rnd(40, 96, seed=6) / 10
    b = rnd(40, seed=7)
    y = ops.linear(x, w, b, relu=True)
    assert rel(y, F.relu(F.linear(x, w, b))) < TOL[mode]


# ------------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, key_mask, causal, H):
    B, Tq, d = q.shape
    Tk = k.shape[1]
    dk = d // H
    qh = q.float().view(B, Tq, H, dk).transpose(1, 2)
    kh = k.float().view(B, Tk, H, dk).transpose(1, 2)
    vh = v.float().view(B, Tk, H, dk).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) / math.sqrt(dk)
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], float('-inf'))
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(Tq, Tk, device=q.device)).bool(), float('-inf'))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Tq, d)


@pytest.mark.parametrize('B,T,H,dk,ragged,causal', [(3, 249, 4, 64, True, False), (2, 37, 4, 16, True, False),
                                                     (2, 15, 4, 64, False, True), (1, 130, 2, 32, False, False),
                                                     (2, 70, 4, 16, False, True)])
def test_self_attention(mode, B, T, H, dk, ragged, causal):
    from opentransformer_amd import ops
    d = H * dk
    qkv = (rnd(B, T, 3 * d, seed=11) * 0.7).to(adt(mode)).requires_grad_(True)
    km = None
    if ragged:
        lens = [T - 13 * i for i in range(B)]
        km = torch.zeros(B, T, dtype=torch.bool, device=DEV)
        for i, n in enumerate(lens):
            km[i, :n] = True
    out = ops.SelfAttentionFn.apply(qkv, km.to(torch.uint8) if km is not None else None, H, causal)
    qf = qkv.detach().float().requires_grad_(True)
    ref = ref_attention(qf[..., :d], qf[..., d:2 * d], qf[..., 2 * d:], km, causal, H)
    assert rel(out.float(), ref) < TOL[mode]
    g = rnd(B, T, d, seed=12).to(adt(mode))
    (dqkv,) = torch.autograd.grad(out, qkv, g)
    (dref,) = torch.autograd.grad(ref, qf, g.float())
    for nm, sl in (('dq', slice(0, d)), ('dk', slice(d, 2 * d)), ('dv', slice(2 * d, 3 * d))):
        assert rel(dqkv[..., sl].float(), dref[..., sl]) < 2 * TOL[mode], nm


@pytest.mark.parametrize('B,T,H,dk,causal', [(3, 249, 4, 64, False), (2, 15, 4, 64, True), (2, 130, 2, 32, False), (2, 70, 4, 16, True)])
def test_attention_bwd_one_launch_equals_two(mode, B, T, H, dk, causal):
    """The backward pass is ONE launch (dQ workgroups + dK/dV workgroups, the dK/dV half forming its own delta = rowsum(dO * O));
    otr_debug_set(13, 1) brings back the two-launch form (dQ writes delta, dK/dV reads it).  Same results: dQ bit for bit, dK / dV
    up to the summation order of delta."""
    from opentransformer_amd import ops, _lib as L
    d = H * dk
    qkv = (rnd(B, T, 3 * d, seed=31) * 0.7).to(adt(mode)).requires_grad_(True)
    km = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    for i in range(B):
        km[i, :T - 11 * i] = True
    g = rnd(B, T, d, seed=32).to(adt(mode))
    lib = L.load()
    res = []
    try:
        L.check(lib.otr_debug_set(21, 0), 'debug_set')        # the generic kernels (the encoder-shape kernel has its own tests)
        for two in (0, 1):
            L.check(lib.otr_debug_set(13, two), 'debug_set')
            out = ops.SelfAttentionFn.apply(qkv, km.to(torch.uint8), H, causal)
            (dqkv,) = torch.autograd.grad(out, qkv, g)
            res.append(dqkv.float())
    finally:
        lib.otr_debug_set(13, 0)
        lib.otr_debug_set(21, 1)
    assert torch.equal(res[0][..., :d], res[1][..., :d])
    assert rel(res[0][..., d:], res[1][..., d:]) < (1e-5 if mode == 'fp32' else 2e-3)


@pytest.mark.parametrize('B,T,H,dk,causal', [(3, 249, 4, 64, False), (5, 15, 4, 64, True), (2, 130, 2, 32, False)])
def test_attention_workgroup_mapping_does_not_change_results(mode, B, T, H, dk, causal):
    """The attention launches place the blocks of one (head, utterance) on one XCD (a 1-D grid decoded in the kernel; H * B not a
    multiple of 8 leaves padding workgroups that exit); otr_debug_set(16, 0) brings back the 3-D grid.  Same blocks, same
    arithmetic: bit-identical outputs and gradients."""
    from opentransformer_amd import ops, _lib as L
    d = H * dk
    qkv = (rnd(B, T, 3 * d, seed=41) * 0.7).to(adt(mode)).requires_grad_(True)
    km = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    for i in range(B):
        km[i, :T - 3 * i] = True
    g = rnd(B, T, d, seed=42).to(adt(mode))
    lib = L.load()
    res = []
    try:
        L.check(lib.otr_debug_set(21, 0), 'debug_set')        # the generic kernels (the encoder-shape kernel's grid is fixed)
        for xmap in (1, 0):
            L.check(lib.otr_debug_set(16, xmap), 'debug_set')
            out = ops.SelfAttentionFn.apply(qkv, km.to(torch.uint8), H, causal)
            (dqkv,) = torch.autograd.grad(out, qkv, g)
            res.append((out.detach().clone(), dqkv.clone()))
    finally:
        lib.otr_debug_set(16, 1)
        lib.otr_debug_set(21, 1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize('B,L,T,H,dk', [(3, 15, 249, 4, 64), (2, 1, 49, 4, 16), (2, 70, 100, 4, 16)])
def test_cross_attention(mode, B, L, T, H, dk):
    from opentransformer_amd import ops
    d = H * dk
    q = (rnd(B, L, d, seed=21) * 0.7).to(adt(mode)).requires_grad_(True)
    kv = (rnd(B, T, 2 * d, seed=22) * 0.7).to(adt(mode)).requires_grad_(True)
    km = torch.ones(B, T, dtype=torch.bool, device=DEV)
    km[0, T - 9:] = False
    out = ops.CrossAttentionFn.apply(q, kv, km.to(torch.uint8), H)
    qf = q.detach().float().requires_grad_(True)
    kvf = kv.detach().float().requires_grad_(True)
    ref = ref_attention(qf, kvf[..., :d], kvf[..., d:], km, False, H)
    assert rel(out.float(), ref) < TOL[mode]
    g = rnd(B, L, d, seed=23).to(adt(mode))
    dq, dkv = torch.autograd.grad(out, (q, kv), g)
    dqr, dkvr = torch.autograd.grad(ref, (qf, kvf), g.float())
    assert rel(dq.float(), dqr) < 2 * TOL[mode]
    assert rel(dkv.float(), dkvr) < 2 * TOL[mode]


# ------------------------------------------------------------------------------------------ add + layernorm
@pytest.mark.parametrize('M,d', [(100, 256), (33, 64), (7, 384), (1000, 256)])
def test_add_layernorm(mode, M, d):
    from opentransformer_amd import ops
    x = rnd(M, d, seed=31).requires_grad_(True)
    a = rnd(M, d, seed=32).to(adt(mode)).requires_grad_(True)
    gam = (1 + 0.1 * rnd(d, seed=33)).requires_grad_(True)
    bet = (0.1 * rnd(d, seed=34)).requires_grad_(True)
    y = ops.add_layernorm(x, a, gam, bet, 0.0)
    yr = F.layer_norm(x + a.float(), (d,), gam, bet, 1e-5)
    assert rel(y, yr) < 1e-5
    g = rnd(M, d, seed=35)
    grads = torch.autograd.grad(y, (x, a, gam, bet), g)
    gref = torch.autograd.grad(yr, (x, a, gam, bet), g)
    for nm, u, v in zip(('dx', 'da', 'dgamma', 'dbeta'), grads, gref):
        assert rel(u.float(), v.float()) < (1e-4 if nm != 'da' or mode == 'fp32' else 1e-2), nm
    # plain LayerNorm (a = None)
    y2 = ops.add_layernorm(x, None, gam, bet, 0.0)
    assert rel(y2, F.layer_norm(x, (d,), gam, bet, 1e-5)) < 1e-5


def test_add_layernorm_dropout_statistics():
    """Dropout parity is statistical: keep-rate, determinism per (seed, offset),
    fresh masks per step, and the backward regenerating exactly the forward's mask."""
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp32')
    try:
        M, d, p = 512, 256, 0.1
        dev = torch.device(DEV)
        x = torch.zeros(M, d, device=DEV, requires_grad=True)
        a = torch.ones(M, d, device=DEV, requires_grad=True)
        gam = torch.ones(d, device=DEV)
        bet = torch.zeros(d, device=DEV)
        ops.next_dropout_step(dev)
        y = ops.add_layernorm(x, a, gam, bet, p)
        ops._state['rng_offset'] = 0
        assert torch.equal(y, ops.add_layernorm(x, a, gam, bet, p))      # same seed+offset -> same mask
        ops.next_dropout_step(dev)
        y2 = ops.add_layernorm(x, a, gam, bet, p)
        assert not torch.equal(y, y2)                                     # new step -> new mask
        dropped = y2 < 0            # z in {0, 1/(1-p)}: dropped elements sit below the row mean
        assert abs(dropped.float().mean().item() - p) < 0.01
        g = rnd(M, d, seed=36)
        (da,) = torch.autograd.grad(y2, a, g)
        assert bool(((da == 0) == dropped).all())                         # bwd mask == fwd mask
    finally:
        ops.set_compute_dtype('bf16')


# ------------------------------------------------------------------------------------------ FFN / GLU / misc
@pytest.mark.parametrize('M,d,dff', [(200, 64, 256), (130, 256, 2048)])
def test_ffn_glu(mode, M, d, dff):
    from opentransformer_amd import ops
    x = rnd(M, d, seed=41).requires_grad_(True)
    w1 = (rnd(2 * dff, d, seed=42) / math.sqrt(d)).requires_grad_(True)
    b1 = (0.1 * rnd(2 * dff, seed=43)).requires_grad_(True)
    w2 = (rnd(d, dff, seed=44) / math.sqrt(dff)).requires_grad_(True)
    b2 = (0.1 * rnd(d, seed=45)).requires_grad_(True)
    y = ops.FeedForwardGLUFn.apply(x, w1, b1, w2, b2)
    yr = F.linear(F.glu(F.linear(x, w1, b1), -1), w2, b2)
    assert rel(y, yr) < TOL[mode]
    g = rnd(M, d, seed=46)
    grads = torch.autograd.grad(y, (x, w1, b1, w2, b2), g)
    gref = torch.autograd.grad(yr, (x, w1, b1, w2, b2), g)
    for nm, u, v in zip(('dx', 'dw1', 'db1', 'dw2', 'db2'), grads, gref):
        assert rel(u, v) < 2 * TOL[mode], nm


@pytest.mark.parametrize('M,d,dff', [(200, 64, 256), (130, 256, 2048), (1030, 256, 2048), (77, 64, 40)])
def test_ffn_glu_fused_bf16_path(M, d, dff):
    """The production bf16 data path: x carries its bf16 twin and the output gradient arrives in bf16, so the fused
    kernels run (otr_ffn_glu_fwd: w_1 GEMM + GLU epilogue keeping (a | sigmoid(b)); otr_ffn_glu_bwd: dy.W_2 GEMM + GLU'
    epilogue) on 64- and 128-wide tiles; dff = 40 does not qualify and must fall back with the same result."""
    from opentransformer_amd import ops
    ops.set_compute_dtype('bf16')
    try:
        x = rnd(M, d, seed=41).requires_grad_(True)
        ops.attach_lp(x, ops.cast_bf16(x.detach()))
        w1 = (rnd(2 * dff, d, seed=42) / math.sqrt(d)).requires_grad_(True)
        b1 = (0.1 * rnd(2 * dff, seed=43)).requires_grad_(True)
        w2 = (rnd(d, dff, seed=44) / math.sqrt(dff)).requires_grad_(True)
        b2 = (0.1 * rnd(d, seed=45)).requires_grad_(True)
        y = ops.FeedForwardGLUFn.apply(x, w1, b1, w2, b2, False, torch.bfloat16)
        assert y.dtype == torch.bfloat16
        yr = F.linear(F.glu(F.linear(x, w1, b1), -1), w2, b2)
        assert rel(y.float(), yr) < TOL['bf16']
        g = rnd(M, d, seed=46)
        grads = torch.autograd.grad(y, (x, w1, b1, w2, b2), g.to(torch.bfloat16))
        gref = torch.autograd.grad(yr, (x, w1, b1, w2, b2), g)
        for nm, u, v in zip(('dx', 'dw1', 'db1', 'dw2', 'db2'), grads, gref):
            assert rel(u, v) < 2 * TOL['bf16'], nm
    finally:
        ops.set_compute_dtype('bf16')


def test_posenc_and_embedding(mode):
    from opentransformer_amd import ops
    from oracle import otrans_oracle as orc
    x = rnd(3, 49, 64, seed=51).requires_grad_(True)
    y = ops.posenc(x)
    yr = orc.add_posenc(x.detach().cpu())
    assert rel(y.cpu(), yr) < 1e-5
    (dx,) = torch.autograd.grad(y, x, torch.ones_like(y))
    assert rel(dx, torch.full_like(dx, 8.0)) < 1e-6
    E = rnd(100, 64, seed=52).requires_grad_(True)
    tok = torch.randint(0, 100, (4, 11), generator=torch.Generator().manual_seed(1)).to(DEV)
    tok[0, :3] = 7                                    # repeated ids exercise the atomic scatter
    e = ops.embed_posenc(tok, E)
    er = orc.add_posenc(F.embedding(tok.cpu(), E.detach().cpu()))
    assert rel(e.cpu(), er) < 1e-5
    g = rnd(4, 11, 64, seed=53)
    (dE,) = torch.autograd.grad(e, E, g)
    Er = E.detach().clone().requires_grad_(True)
    (dEr,) = torch.autograd.grad(F.embedding(tok, Er) * 8.0, Er, g)
    assert rel(dE, dEr) < 1e-5


def test_colsum_unaligned(mode):
    from opentransformer_amd import ops
    for M, N in [(1000, 256), (77, 4234), (5, 3)]:
        a = rnd(M, N, seed=61).to(adt(mode))
        assert rel(ops.colsum_raw(a), a.float().sum(0)) < (1e-5 if mode == 'fp32' else 1e-5)


# ------------------------------------------------------------------------------------------ conv frontend
@pytest.mark.parametrize('B,T,Fdim,C1,C2', [(2, 200, 80, 32, 64), (3, 97, 40, 64, 128), (1, 1000, 80, 64, 128),
                                            (2, 120, 80, 256, 256), (1, 77, 80, 128, 64)])      # r06: conv1 on the matrix pipe for any multiple of 64 channels
def test_conv_subsample(mode, B, T, Fdim, C1, C2):
    from opentransformer_amd import ops
    x = rnd(B, T, Fdim, seed=71)
    w1 = (rnd(C1, 1, 3, 3, seed=72) / 3).requires_grad_(True)
    b1 = (0.1 * rnd(C1, seed=73)).requires_grad_(True)
    w2 = (rnd(C2, C1, 3, 3, seed=74) / math.sqrt(9 * C1)).requires_grad_(True)
    b2 = (0.1 * rnd(C2, seed=75)).requires_grad_(True)
    act2 = ops.ConvSubsampleFn.apply(x, w1, b1, w2, b2)
    h1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=(0, 1)))
    h2 = F.relu(F.conv2d(h1, w2, b2, stride=2, padding=(0, 1)))          # [B,C2,T2,F2]
    ref = h2.permute(0, 2, 3, 1).reshape(B, h2.size(2), -1)               # [B,T2,F2*C2] channel-last
    assert act2.shape == ref.shape
    assert rel(act2.float(), ref) < TOL[mode]
    g = rnd(*ref.shape, seed=76).to(adt(mode))
    grads = torch.autograd.grad(act2, (w1, b1, w2, b2), g)
    gref = torch.autograd.grad(ref, (w1, b1, w2, b2), g.float())
    for nm, u, v in zip(('dw1', 'db1', 'dw2', 'db2'), grads, gref):
        # every gradient here passes through the ReLU mask of act2, which the reference takes from UNROUNDED pre-activations:
        # operand rounding eps flips a fraction ~eps of the mask bits and each flip moves an O(1) random upstream element,
        # so the error goes like sqrt(eps) (measured 1.3e-2 in fp16 here; 1.5e-3 inside the real model, whose upstream
        # gradient is smooth)
        assert rel(u, v) < (10 if mode == 'fp16' else 3) * TOL[mode], nm


# ------------------------------------------------------------------------------------------ losses
def test_label_smoothing_loss():
    from opentransformer_amd import ops
    from oracle import otrans_oracle as orc
    for R, V in [(44, 100), (480, 4234)]:
        logits = rnd(4, R // 4, V, seed=81).requires_grad_(True)
        tgt = torch.randint(1, V, (4, R // 4), generator=torch.Generator().manual_seed(2))
        tgt[1, -3:] = 0
        tgt[3, -1:] = 0
        loss = ops.LabelSmoothingLossFn.apply(logits, tgt.to(DEV), 0.1, 0)
        lc = logits.detach().cpu().requires_grad_(True)
        ref = orc.label_smoothing_loss(lc, tgt, 0.1)
        assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
        (dl,) = torch.autograd.grad(loss * 3.0, logits)
        (dr,) = torch.autograd.grad(ref * 3.0, lc)
        assert rel(dl.cpu(), dr) < 1e-5


def test_log_softmax_and_ctc():
    from opentransformer_amd import ops
    B, T, V, Lm = 4, 49, 100, 11
    logits = rnd(B, T, V, seed=91).requires_grad_(True)
    assert rel(ops.log_softmax(logits.detach()), F.log_softmax(logits.detach(), -1)) < 1e-6
    gen = torch.Generator().manual_seed(3)
    tgt = torch.randint(1, V, (B, Lm), generator=gen)
    tgt[0, 2] = tgt[0, 1]                               # repeated label needs the blank path
    in_len = torch.tensor([49, 45, 30, 5])
    tgt_len = torch.tensor([11, 8, 11, 6])              # last utterance is infeasible (T=5 < L=6)
    loss = ops.CTCLossFn.apply(logits, tgt.to(DEV), in_len.to(DEV), tgt_len.to(DEV), 0)
    lc = logits.detach().cpu().requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(lc, -1).transpose(0, 1), tgt, in_len, tgt_len, blank=0, zero_infinity=True)
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    (dl,) = torch.autograd.grad(loss, logits)
    (dr,) = torch.autograd.grad(ref, lc)
    assert rel(dl.cpu(), dr) < 1e-4


def test_cpu_tensor_is_refused():
    from opentransformer_amd import ops, _lib
    with pytest.raises(_lib.OtransHipError):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8), None)


# ------------------------------------------------------------------------------------------ conformer pieces
@pytest.mark.parametrize('B,T,H,dk', [(2, 49, 4, 16), (2, 70, 4, 96)])
def test_relpos_attention(mode, B, T, H, dk):
    import opentransformer_amd as ota
    from opentransformer_amd.nn import relative_sinusoid
    from opentransformer_amd import synthetic as syn
    from oracle import otrans_oracle as orc
    d = H * dk
    torch.manual_seed(0)
    mod = ota.MultiHeadedSelfAttentionWithRelPos(H, d).to(DEV)
    syn.fill_state_dict_(mod.state_dict(), 5)
    x = rnd(B, T, d, seed=101).requires_grad_(True)
    mask = torch.ones(B, 1, T, dtype=torch.bool, device=DEV)
    mask[1, 0, T - 7:] = False
    pos = relative_sinusoid(T, d, DEV)
    out, _ = mod(x, mask, pos)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    xc = x.detach().cpu().requires_grad_(True)
    ref = orc.relpos_self_attention(sd, xc, mask.cpu(), pos.cpu(), H)
    assert rel(out.float().cpu(), ref) < TOL[mode]
    g = rnd(B, T, d, seed=102).to(out.dtype)
    names = ['qvk_proj.weight', 'pos_proj.weight', 'posu', 'posv']
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(out, [x] + [params[n] for n in names], g)
    gref = torch.autograd.grad(ref, [xc] + [sd[n] for n in names], g.float().cpu())
    for nm, u, v in zip(['dx'] + names, grads, gref):
        assert rel(u.float().cpu(), v) < 3 * TOL[mode], nm


@pytest.mark.parametrize('B,T,C', [(3, 49, 64), (2, 120, 384)])
def test_conformer_conv_module(mode, B, T, C):
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn
    from oracle import otrans_oracle as orc
    torch.manual_seed(0)
    mod = ota.ConformerConvolutionModule(C, 5).to(DEV).train()
    syn.fill_state_dict_(mod.state_dict(), 6)
    sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
    x = rnd(B, T, C, seed=111).requires_grad_(True)
    mask = torch.ones(B, T, dtype=torch.bool, device=DEV)
    mask[0, T - 11:] = False
    out = mod(x, mask)
    for k in sd:
        if sd[k].is_floating_point() and 'running' not in k:
            sd[k].requires_grad_(True)
    xc = x.detach().cpu().requires_grad_(True)
    run_m, run_v = sd['batch_norm.running_mean'].clone(), sd['batch_norm.running_var'].clone()
    ref = orc.conformer_conv_module(sd, xc, mask.cpu(), training=True)
    assert out.dtype == adt(mode)           # the branch leaves in the activation type (the residual add takes it as such)
    assert rel(out.float().cpu(), ref) < TOL[mode]
    # running statistics were updated like torch's BatchNorm1d (momentum 0.1, unbiased variance)
    y = torch.nn.functional.batch_norm(torch.zeros(1, C, 2), run_m, run_v, training=False)   # noqa: F841 (buffers untouched)
    g = rnd(B, T, C, seed=112).to(out.dtype)
    names = [k for k in sd if sd[k].requires_grad]
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(out, [x] + [params[n] for n in names], g)
    gref = torch.autograd.grad(ref, [xc] + [sd[n] for n in names], g.float().cpu())
    for nm, u, v in zip(['dx'] + names, grads, gref):
        if nm == 'depthwise_conv.bias':     # a bias in front of BatchNorm has zero gradient: both sides are roundoff
            assert float(u.abs().max()) < 1e-3 * float(gref[1].abs().max() + 1e-6) + 1e-4, nm
            continue
        assert rel(u.float().cpu(), v) < 4 * TOL[mode], nm
    # eval mode uses the running statistics
    mod.eval()
    with torch.no_grad():
        oe = mod(x, mask)
        sde = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        re = orc.conformer_conv_module(sde, xc.detach(), mask.cpu(), training=False)
    assert rel(oe.float().cpu(), re) < TOL[mode]


def test_residual_add_dropout():
    from opentransformer_amd import ops
    x = rnd(64, 256, seed=121).requires_grad_(True)
    a = rnd(64, 256, seed=122).requires_grad_(True)
    y = ops.residual_add(x, a, 0.5, 0.0)
    assert rel(y, x + 0.5 * a) < 1e-6
    ops.next_dropout_step(torch.device(DEV))
    yd = ops.residual_add(x, a, 1.0, 0.25)
    kept = ((yd - x).abs() > 0)
    assert abs(kept.float().mean().item() - 0.75) < 0.02
    (da,) = torch.autograd.grad(yd, a, torch.ones_like(yd))
    assert bool(((da != 0) == kept).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('kind', ['gelu', 'tanh', 'swish'])
def test_ffn_activation_kernels(kind, dtype):
    """otr_act_fwd / otr_act_bwd vs torch (erf gelu, tanh, x*sigmoid(x)); n = 8k + 3 exercises the scalar tail"""
    from opentransformer_amd import ops
    ref = {'gelu': torch.nn.functional.gelu, 'tanh': torch.tanh, 'swish': lambda t: t * torch.sigmoid(t)}[kind]
    torch.manual_seed(0)
    x = (3 * torch.randn(37, 1003, device=DEV)).to(dtype).requires_grad_(True)
    dy = torch.randn(37, 1003, device=DEV).to(dtype)
    y = ops.activation(x, kind)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.float())
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and x.grad.dtype == dtype
    assert (y.float() - yr).abs().max().item() <= tol * max(1.0, yr.abs().max().item())
    assert (x.grad.float() - xr.grad).abs().max().item() <= tol * max(1.0, xr.grad.abs().max().item())


def shared_projection_inputs():
    """seeded CPU inputs shared with oracle/make_golden.py:golden_shared_projections"""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 21, 64, generator=g)
    mem = torch.randn(3, 50, 48, generator=g)
    xmask = torch.arange(21).unsqueeze(0) < torch.tensor([21, 17, 9]).unsqueeze(1)
    mmask = torch.arange(50).unsqueeze(0) < torch.tensor([50, 33, 41]).unsqueeze(1)
    dy = torch.randn(3, 21, 64, generator=g)
    return x, mem, xmask, mmask, dy


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'fp16'])
def test_shared_qvk_and_vk_projections_match_reference(golden, mode):
    """share_qvk_proj / share_vk_proj (module/attention.py:71-72,131-132): query = key = value = one projection"""
    from opentransformer_amd import ops, nn as onn
    from opentransformer_amd import synthetic as syn
    gz = golden('modules_shared.npz')
    g = {k: torch.from_numpy(gz[k]) for k in gz.files}
    x, mem, xmask, mmask, dy = shared_projection_inputs()
    tol = 2e-4 if mode == 'fp32' else 4e-2
    ops.set_compute_dtype(mode)
    try:
        sa = onn.MultiHeadedSelfAttention(4, 64, 0.0, share_qvk_proj=True)
        ca = onn.MultiHeadedCrossAttention(4, 64, 48, 0.0, share_vk_proj=True)
        syn.fill_state_dict_(sa.state_dict(), 31)
        syn.fill_state_dict_(ca.state_dict(), 32)
        sa, ca = sa.to(DEV), ca.to(DEV)
        valid = xmask
        xs = x.to(DEV).requires_grad_(True)
        y, _ = sa(xs, xmask.to(DEV).unsqueeze(1))
        y.float().backward(dy.to(DEV))
        assert rel(y.detach().float().cpu()[valid], g['sa_y'][valid]) < tol
        # padded query rows see real keys: their dy flows into dk/dv in the reference as well
        assert rel(xs.grad.cpu(), g['sa_dx']) < tol
        for k, p in sa.named_parameters():
            assert rel(p.grad.cpu(), g['sa_grad:' + k]) < tol, k
        xq, ms = x.to(DEV).requires_grad_(True), mem.to(DEV).requires_grad_(True)
        y, _ = ca(xq, ms, mmask.to(DEV).unsqueeze(1))
        y.float().backward(dy.to(DEV))
        assert rel(y.detach().float().cpu(), g['ca_y']) < tol
        assert rel(xq.grad.cpu(), g['ca_dq']) < tol and rel(ms.grad.cpu(), g['ca_dmem']) < tol
        for k, p in ca.named_parameters():
            assert rel(p.grad.cpu(), g['ca_grad:' + k]) < tol, k
    finally:
        ops.set_compute_dtype('bf16')


def loss_option_inputs():
    """seeded CPU inputs shared with oracle/make_golden.py:golden_loss_options"""
    g = torch.Generator().manual_seed(91)
    logits = 2 * torch.randn(3, 9, 57, generator=g)
    target = torch.randint(1, 57, (3, 9), generator=g)
    target[0, 7:] = 0
    target[2, 4:] = 0
    mask = torch.rand(3, 9, generator=g) < 0.3
    return logits, target, mask


def test_label_smoothing_mask_and_sum_normalisation(golden):
    """LabelSmoothingLoss(mask=..., normalize_length=False): module/loss.py:31-35,43-46"""
    from opentransformer_amd import nn as onn
    g = golden('module_loss.npz')
    logits, target, mask = loss_option_inputs()
    for name, (use_mask, norm) in {'mask': (True, True), 'sum': (False, False), 'mask_sum': (True, False)}.items():
        lg = logits.to(DEV).requires_grad_(True)
        loss = onn.LabelSmoothingLoss(57, 0.1, normalize_length=norm)(lg, target.to(DEV), mask.to(DEV) if use_mask else None)
        loss.backward()
        assert abs(loss.item() - float(g[name + '_loss'])) < 1e-5 * abs(float(g[name + '_loss'])), name
        assert rel(lg.grad.cpu(), torch.from_numpy(g[name + '_grad'])) < 1e-5, name


def optimizer_inputs():
    """seeded parameters / per-step gradients shared with oracle/make_golden.py:golden_optimizer.  Step 2 carries a huge
    gradient (clipped to norm 5), step 4 a NaN (the update is skipped and the schedule does not advance)."""
    g = torch.Generator().manual_seed(123)
    shapes = [(37, 64), (64,), (5, 3, 3), (130,)]
    params = [0.3 * torch.randn(*sh, generator=g) for sh in shapes]
    grads = []
    for step in range(7):
        scale = 400.0 if step == 2 else 0.05
        gs = [scale * torch.randn(*sh, generator=g) for sh in shapes]
        if step == 4:
            gs[1][7] = float('nan')
        grads.append(gs)
    hp = dict(lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip=5.0, model_size=256, warmup_steps=4, factor=1.0)
    return shapes, params, grads, hp


@pytest.mark.parametrize('world', [1, 4])
def test_fused_optimizer_step_matches_reference(golden, world):
    """otr_optimizer_step (clip + NaN guard + Noam + Adam with L2 decay over the flat buffers) against the reference's
    TransformerScheduler + torch.optim.Adam loop (train/trainer.py:221-234); world > 1: gradients arrive as a sum over
    ranks and 1/world is folded into the update (grad_scale)."""
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    g = golden('optimizer_steps.npz')
    shapes, params, grads, hp = optimizer_inputs()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(p.clone().to(DEV)) for p in params])

    deferral = ops._wq['on']
    try:
        dp = FlatDataParallel(Holder())
        opt = FusedAdam(dp, lr=hp['lr'], betas=hp['betas'], eps=hp['eps'], weight_decay=hp['weight_decay'], clip_grad=hp['clip'],
                        noam=dict(model_size=hp['model_size'], warmup_steps=hp['warmup_steps'], factor=hp['factor']))
        for step, gs in enumerate(grads):
            for p, gr in zip(dp.params, gs):
                p.grad.copy_(gr.to(DEV) * world)
            opt.step(grad_scale=1.0 / world)
            st = opt.stats()
            got = torch.cat([p.detach().reshape(-1) for p in dp.params]).cpu()
            want = torch.from_numpy(g['params_%d' % step])
            assert rel(got, want) < 5e-6, (step, rel(got, want))
            assert abs(st['lr'] - float(g['lr'][step])) <= 1e-6 * float(g['lr'][step]), (step, st['lr'], g['lr'][step])
            assert int(st['skipped']) == int(g['skipped'][:step + 1].sum()), (step, st)
            if not g['skipped'][step]:
                assert abs(st['grad_sqnorm'] ** 0.5 / world - float(g['grad_norm'][step])) <= 2e-5 * float(g['grad_norm'][step])
            if dp.flat_param_lp is not None:      # the bf16 shadow is refreshed in the same pass
                assert torch.equal(dp.flat_param_lp.float().cpu(), dp.flat_param.to(torch.bfloat16).float().cpu())
    finally:
        ops.defer_weight_grads(deferral)


def test_transpose_batched():
    """one launch transposes a list of ragged 2-D matrices packed in a flat buffer (bf16 weight shadows)"""
    import ctypes as C
    from opentransformer_amd import _lib as L
    gen = torch.Generator().manual_seed(3)
    shapes = [(768, 256), (100, 64), (1, 7), (65, 129), (4234, 256), (3, 3)]
    total = sum(a * b for a, b in shapes) + 5
    src = torch.randn(total, generator=gen).to(DEV, torch.bfloat16)
    dst = torch.zeros_like(src)
    table, tiles, off = [], 0, 2          # leading gap: offsets need not start at 0
    for a, b in shapes:
        table.append([off, a, b, tiles])
        tiles += ((a + 63) // 64) * ((b + 63) // 64)
        off += a * b
    tab = torch.tensor(table, dtype=torch.int64, device=DEV)
    L.check(L.load().otr_transpose_batched(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(tab.data_ptr()),
                                           len(shapes), tiles, 2, C.c_void_p(torch.cuda.current_stream().cuda_stream)), 't')
    for (o, a, b, _) in table:
        assert torch.equal(dst[o:o + a * b].view(b, a), src[o:o + a * b].view(a, b).t()), (a, b)
    assert float(dst[:2].abs().sum()) == 0.0 and float(dst[off:].abs().sum()) == 0.0


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'fp16'])
def test_deferred_bias_gradient_through_layernorm(mode):
    """linear(defer_bias=True) + add_layernorm(a_bias=b): the bias gradient comes out of the LayerNorm backward
    (with dropout) and equals the plain path's, both as a returned tensor and accumulated in place."""
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        gen = torch.Generator().manual_seed(11)
        M, K, d = 300, 96, 128
        x0 = torch.randn(M, d, generator=gen).to(DEV)
        h = torch.randn(M, K, generator=gen).to(DEV)
        w = (torch.randn(d, K, generator=gen) / 8).to(DEV).requires_grad_()
        gam = torch.randn(d, generator=gen).to(DEV).requires_grad_()
        bet = torch.randn(d, generator=gen).to(DEV).requires_grad_()
        g = torch.randn(M, d, generator=gen).to(DEV)
        res = []
        for defer, inplace in ((False, False), (True, False), (True, True)):
            b = torch.randn(d, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_()
            if inplace:
                b.grad = torch.full((d,), 0.5, device=DEV)
                b._otr_grad_inplace = True
            ops._state['rng_offset'] = 0
            a = ops.linear(h, w, b, defer_bias=defer)
            y = ops.add_layernorm(x0, a, gam, bet, 0.1, a_bias=b if defer else None)
            for t in (w, gam, bet):
                t.grad = None
            y.backward(g)
            res.append((b.grad - 0.5) if inplace else b.grad.clone())
        tol = 1e-4
        torch.testing.assert_close(res[1], res[0], rtol=tol, atol=tol * float(res[0].abs().max()))
        torch.testing.assert_close(res[2], res[0], rtol=tol, atol=tol * float(res[0].abs().max()))
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'fp16'])
def test_grouped_weight_and_bias_gradients(mode):
    """otr_linear_wgrad_grouped / otr_colsum_grouped == the per-layer launches they replace (accumulating), over
    ragged shapes: long and short contractions, small and large outputs, both dy dtypes, one misaligned item."""
    import ctypes as C
    from opentransformer_amd import _lib as L
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        gen = torch.Generator().manual_seed(21)
        adt = ops.act_dtype()
        shapes = [(1000, 768, 256), (1000, 256, 256), (480, 4096, 256), (1000, 256, 2048), (60, 64, 64), (7, 128, 200),
                  (333, 100, 36), (1000, 130, 68)]
        flat = torch.zeros(sum(n * k for _, n, k in shapes) + 64, device=DEV)
        items_w, items_b, refs, off = [], [], [], 0
        for idx, (m, n, k) in enumerate(shapes):
            dyt = adt if idx % 2 == 0 else torch.float32
            dy = torch.randn(m, n, generator=gen).to(DEV, dyt)
            x = torch.randn(m, k, generator=gen).to(DEV, adt)
            if idx == 6:
                off += 1                                  # misaligned gradient view -> the stand-alone fallback
            out = flat[off:off + n * k].view(n, k)
            out.fill_(0.25)
            off += n * k
            bias = torch.full((n,), -1.0, device=DEV)
            items_w.append((dy, x, out))
            items_b.append((dy, bias))
            refs.append((0.25 + dy.float().t() @ x.float(), -1.0 + dy.float().sum(0)))
        ops._wq['w'], ops._wq['b'] = list(items_w), list(items_b)
        ops.flush_weight_grads()
        tol = 2e-5 if mode == 'fp32' else 2e-2
        for (dy, x, out), (dyb, bias), (rw, rb) in zip(items_w, items_b, refs):
            scale = float(rw.abs().max())
            torch.testing.assert_close(out, rw, rtol=tol, atol=tol * scale)
            torch.testing.assert_close(bias, rb, rtol=1e-4, atol=1e-3 * float(rb.abs().max()))
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('M,d,F_', [(1030, 256, 2048), (130, 64, 96)])
def test_ffn_glu_fused_entries_direct(M, d, F_):
    """otr_ffn_glu_fwd / otr_ffn_glu_bwd through the C ABI: h = (value | sigmoid(gate)), u, dh and the bias partials;
    an unqualified shape returns 1 and launches nothing."""
    import ctypes as C
    from opentransformer_amd import _lib as L
    from opentransformer_amd import ops
    lib = L.load()
    bf = torch.bfloat16
    x = rnd(M, d, seed=61).to(bf)
    w1 = (rnd(2 * F_, d, seed=62) / math.sqrt(d)).to(bf)
    b1 = 0.1 * rnd(2 * F_, seed=63)
    h = torch.zeros(M, 2 * F_, device=DEV, dtype=bf)
    u = torch.zeros(M, F_, device=DEV, dtype=bf)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    assert lib.otr_ffn_glu_fwd(p(x), d, p(w1), d, p(b1), p(h), p(u), M, F_, d, st) == 0
    t = x.float() @ w1.float().t() + b1
    a, sg = t[:, :F_], torch.sigmoid(t[:, F_:])
    assert rel(h[:, :F_].float(), a) < 1e-2 and rel(h[:, F_:].float(), sg) < 1e-2 and rel(u.float(), a * sg) < 1e-2
    # backward on the saved (a | sigmoid(b))
    dy = rnd(M, d, seed=64).to(bf)
    w2t = (rnd(F_, d, seed=65) / math.sqrt(F_)).to(bf)            # [F, d] = w_2^T
    dh = torch.zeros_like(h)
    cap = (M + 63) // 64
    part = torch.zeros(cap, 2 * F_, device=DEV)
    rows = C.c_int32(0)
    assert lib.otr_ffn_glu_bwd(p(dy), 1, d, p(w2t), d, p(h), 1, p(dh), p(part), cap, C.byref(rows), M, F_, d, st) == 0
    du = (dy.float() @ w2t.float().t()).to(bf).float()
    hs, hg = h[:, :F_].float(), h[:, F_:].float()
    ra, rb = du * hg, du * hs * hg * (1 - hg)
    assert rel(dh[:, :F_].float(), ra) < 1.5e-2 and rel(dh[:, F_:].float(), rb) < 1.5e-2
    assert 0 < rows.value <= cap
    got = part[:rows.value].sum(0)
    assert rel(got[:F_], dh[:, :F_].float().sum(0)) < 1e-2 and rel(got[F_:], dh[:, F_:].float().sum(0)) < 1e-2
    # F not a multiple of the tile's value width: refused, nothing written
    h2 = torch.full((M, 80), 7.0, device=DEV, dtype=bf)
    assert lib.otr_ffn_glu_fwd(p(x), d, p(w1), d, p(b1), p(h2), p(u), M, 40, d, st) == 1
    torch.cuda.synchronize()
    assert bool((h2 == 7.0).all())


def test_optimizer_gradient_noise_and_loss_scale():
    """otr_optimizer_step extras: Gaussian gradient noise added after clipping (train/trainer.py:223-227) and the device-side
    dynamic loss scale (divide out, halve + skip on a non-finite norm, grow after `growth_interval` finite updates)"""
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam

    class Holder(torch.nn.Module):
        def __init__(self, value=0.0):
            super().__init__()
            self.w = torch.nn.Parameter(torch.full((1 << 18,), value, device=DEV))

    deferral = ops._wq['on']
    try:
        # noise: zero gradients, so exp_avg = (1 - beta1) * noise
        dp = FlatDataParallel(Holder(1.0))
        opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, clip_grad=5.0, grad_noise=0.05, loss_scale=0.0)
        dp.zero_grad()
        opt.step()
        m = opt.exp_avg[:1 << 18] / 0.1
        assert abs(float(m.mean())) < 1e-3 and abs(float(m.std()) - 0.05) < 1e-3
        # cells whose parameter AND gradient are exactly zero are padding of the flat buffers (alignment gaps, the extra rows of a
        # row-padded Linear): no noise there, they stay zero (include/otrans_hip.h)
        dp = FlatDataParallel(Holder(0.0))
        opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, clip_grad=5.0, grad_noise=0.05, loss_scale=0.0)
        dp.zero_grad()
        dp.params[0].grad[:1024].fill_(1e-3)            # the first 1024 cells are live (non-zero gradient): they do get noise
        opt.step()
        assert float(dp.flat_param[1024:].abs().max()) == 0.0 and float(opt.exp_avg[1024:].abs().max()) == 0.0
        assert float(opt.exp_avg[:1024].std()) > 1e-3
        # loss scale: gradients arrive 1024x too large
        dp = FlatDataParallel(Holder())
        opt = FusedAdam(dp, lr=1e-3, weight_decay=0.0, clip_grad=0.0, loss_scale=1024.0, loss_scale_growth=2)
        g = torch.randn(1 << 18, device=DEV) * 0.01
        dp.params[0].grad.copy_(g * 1024.0)
        opt.step()
        st = opt.stats()
        assert st['skipped'] == 0 and abs(st['grad_sqnorm'] - float((g * g).sum())) < 1e-3 * float((g * g).sum())
        assert rel(opt.exp_avg[:1 << 18], 0.1 * g) < 1e-5                       # scale divided out
        dp.params[0].grad.copy_(g * 1024.0)
        opt.step()                                                               # second finite update: growth_interval reached
        assert opt.stats()['loss_scale'] == 2048.0
        before = dp.flat_param.clone()
        dp.params[0].grad.fill_(float('inf'))                                    # overflow: skip, halve
        opt.step()
        st = opt.stats()
        assert st['skipped'] == 1 and st['loss_scale'] == 1024.0 and torch.equal(dp.flat_param, before)
    finally:
        ops.defer_weight_grads(deferral)
