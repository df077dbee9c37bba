"""GPU (-m gpu): the encoder's split FFN in SLAB mode (csrc/ffn3.hip SLAB kernels, otr_rb_linear_ln, otr_ln_bwd_proj_slabs): the four
hidden slices of a 128-row block leave 16-bit partial sums, the NEXT layer's q|k|v projection finishes dropout + residual + LayerNorm
in its prologue (encoder/transformer.py:58-63 then :47-49 of the next layer), and on the way back the attention sub-layer's closing
launch sums the slices' input-gradient shares while it loads its rows.

Compared against
 * the exchange form of the same kernels (one launch, partial sums exchanged inside it), dropout ON -- both draw the masks from the same
   counter RNG in the same order, so they agree to rounding;
 * a plain fp32 torch restatement of the post-norm encoder layers, dropout off;
and the paths around it: no-grad, a last layer whose LayerNorm nobody projects (otr_dec_ln), a consumer that is not the q|k|v
projection (materialize), the result read twice (gradient handed over AND a real gradient)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_encoder(n_blocks, d_ff, p_drop, seed):
    import opentransformer_amd.nn as onn
    torch.manual_seed(seed)
    enc = onn.TransformerEncoder(d_model=256, n_heads=4, d_ff=d_ff, n_blocks=n_blocks, residual_dropout=p_drop, activation='glu').to(DEV)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return enc


def inputs(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, 256, generator=g).to(DEV)
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    lens[0] = T
    mask = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).to(DEV)
    gy = torch.randn(B, T, 256, generator=g).to(DEV)
    return x, mask, gy


def run_hip(enc, x, mask, gy, slab, count=None):
    from opentransformer_amd import ops
    was = ops._FFN_SLAB
    ops._FFN_SLAB = slab
    recs = []
    try:
        ops.next_dropout_step(DEV)
        ops.rng_seed_tensor(DEV).fill_(4321)
        xin = x.clone().requires_grad_(True)
        if count is not None:
            ops.set_kernel_timer(recs)
        out, _, _ = enc(xin, mask)
        grads = torch.autograd.grad(out, [xin] + list(enc.parameters()), gy)
        torch.cuda.synchronize()
        if count is not None:
            count.extend(r[0] if isinstance(r, (tuple, list)) else r for r in recs)
        return out.detach(), [g.detach() for g in grads]
    finally:
        ops.set_kernel_timer(None)
        ops._FFN_SLAB = was


def torch_encoder(enc, x, mask, round_qkv=None):
    """fp32 torch restatement of the encoder stack; round_qkv = a 16-bit dtype: the FIRST layer's q | k | v are formed as the HIP path forms them
    (16-bit rows and weights, fp32 sums, rounded; straight-through gradient), every other operation stays fp32 -- the reference that shares the HIP path's attention OPERANDS"""
    d, H = 256, 4
    B, T, _ = x.shape
    pos = torch.arange(T, device=x.device, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, device=x.device, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(T, d, device=x.device)
    pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    x = x * math.sqrt(d) + pe
    km = mask.view(B, 1, 1, T)

    def heads(t):
        return t.view(B, T, H, d // H).transpose(1, 2)
    for li, b in enumerate(enc.blocks):
        sa, ff = b.slf_attn, b.feed_forward
        if round_qkv is not None and li == 0:
            # the HIP path's operands: 16-bit input rows and weights, fp32 accumulation, 16-bit q | k | v (straight-through gradients)
            def st(t):
                return t + (t.to(round_qkv).float() - t).detach()
            qkv = st(F.linear(st(x), st(sa.qvk_proj.weight), sa.qvk_proj.bias))
        else:
            qkv = F.linear(x, sa.qvk_proj.weight, sa.qvk_proj.bias)
        q, k, v = qkv.split(d, dim=-1)
        s = (heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(d // H)).masked_fill(~km, float('-inf'))
        c = (torch.softmax(s, dim=-1).masked_fill(~km, 0.0) @ heads(v)).transpose(1, 2).reshape(B, T, d)
        x = F.layer_norm(x + F.linear(c, sa.output_proj.weight, sa.output_proj.bias), (d,), b.norm1.weight, b.norm1.bias, b.norm1.eps)
        h = F.linear(x, ff.w_1.weight, ff.w_1.bias)
        x = F.layer_norm(x + F.linear(F.glu(h, dim=-1), ff.w_2.weight, ff.w_2.bias), (d,), b.norm2.weight, b.norm2.bias, b.norm2.eps)
    return x


@pytest.mark.parametrize('B,T,nl,dff,p_drop', [(10, 249, 3, 2048, 0.1), (17, 130, 2, 1024, 0.25), (32, 249, 2, 2048, 0.0)])
def test_slab_mode_matches_exchange_mode(B, T, nl, dff, p_drop):
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp16')
    try:
        enc = make_encoder(nl, dff, p_drop, seed=B)
        enc.train()
        x, mask, gy = inputs(B, T, seed=T)
        names_a, names_b = [], []
        a, ga = run_hip(enc, x, mask, gy, slab=True, count=names_a)
        b, gb = run_hip(enc, x, mask, gy, slab=False, count=names_b)
        sa, sb = ' '.join(map(str, names_a)), ' '.join(map(str, names_b))
        assert sa.count('ffn_fwd_slab') == nl and sa.count('ffn_bwd_slab') == nl and 'ffn_ln_fwd_split' not in sa, sa
        assert sa.count('rb_linear_ln ') == nl - 1 and sa.count('dec_ln') == 1, sa              # the last LayerNorm stands alone
        assert sb.count('ffn_ln_fwd_split') == nl and 'slab' not in sb, sb
        assert rel(a, b) < 2e-3, rel(a, b)
        names = ['x'] + [n for n, _ in enc.named_parameters()]
        worst = max((rel(u, v), n) for n, u, v in zip(names, ga, gb) if float(v.norm()) > 1e-6)
        assert worst[0] < 1e-2, worst
        if p_drop > 0:
            enc.eval()
            c, _ = run_hip(enc, x, mask, gy, slab=True)
            assert rel(a, c) > 5e-2
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
def test_slab_mode_matches_fp32_torch(mode):
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        enc = make_encoder(2, 2048, 0.0, seed=1)
        enc.train()
        x, mask, gy = inputs(12, 200, seed=5)
        got, ggot = run_hip(enc, x, mask, gy, slab=True)
        xr = x.clone().requires_grad_(True)
        ref = torch_encoder(enc, xr, mask)
        gref = torch.autograd.grad(ref, [xr] + list(enc.parameters()), gy)
        ty, tg = (2e-3, 2e-2) if mode == 'fp16' else (1.5e-2, 8e-2)
        assert rel(got, ref) < ty, rel(got, ref)
        names = ['x'] + [n for n, _ in enc.named_parameters()]
        # the key third of the q|k|v bias gradient is zero in exact arithmetic and is measured apart
        # (tests/helpers.py: key_aware_grad_errors); every other gradient meets the flat bound -- no comparison against the HIP
        # path's own exchange form any more
        errs, key_errs = H.key_aware_grad_errors(names, ggot, gref)
        worst = max((e, n) for n, e in errs.items())
        over = {n: e for n, e in errs.items() if e >= tg}
        H.log_tolerance_cases('ffn_slab', {'mode': mode, 'tg': tg, 'worst': worst, 'over_tg': over, 'key_bias_residue': key_errs})
        # The named tensors (measured 0.026 / 0.20 against the fp32 reference): the q|k|v gradient of the FIRST layer's self-attention and
        # the input gradient it feeds.  Layer 0 attends over sqrt(d)-scaled inputs: its scores have a standard deviation of ~85, the
        # softmax saturates, and the 16-bit rounding of the projected q and k (2^-11 / 2^-8 of values ~ 9) moves the few unsaturated
        # probabilities by percents -- BEFORE any backward arithmetic runs.  r06: this is now MEASURED instead of asserted by name
        # (tools/delta_study.py: with exact arithmetic on the rounded operands the error is the same 1.6e-2; delta from P . dP instead
        # of dO . O changes the third digit): the same gradients are compared with an fp32 reference whose layer-0 q|k|v are rounded to
        # the compute type and whose every other operation is fp32, and against THAT reference they must meet the flat bound.
        named = ('x', 'blocks.0.slf_attn.qvk_proj.weight', 'blocks.0.slf_attn.qvk_proj.bias')
        if over:
            assert all(n in named for n in over), ('gradient over the flat bound and not a named exception', over, tg)
            xr2 = x.clone().requires_grad_(True)
            ref2 = torch_encoder(enc, xr2, mask, round_qkv=ops.act_dtype())
            gref2 = torch.autograd.grad(ref2, [xr2] + list(enc.parameters()), gy)
            errs2, _ = H.key_aware_grad_errors(names, ggot, gref2)
            same_operands = {n: errs2[n] for n in over}
            H.log_tolerance_cases('ffn_slab_same_operands', {'mode': mode, 'tg': tg, 'vs_fp32': over, 'vs_fp32_on_rounded_qkv': same_operands})
            print('encoder slab parity', mode, 'named tensors against the reference on rounded q|k|v:', same_operands)
            assert all(e < tg for e in same_operands.values()), (same_operands, tg)
        assert all(e < (8e-3 if mode == 'fp16' else 4.5e-2) for e in key_errs.values()), key_errs      # measured 1.0e-3 / 7.5e-3
        print('encoder slab parity', mode, 'out %.2e worst grad %.2e %s' % (rel(got, ref), worst[0], worst[1]))
    finally:
        ops.set_compute_dtype('bf16')


def test_slab_mode_no_grad_and_other_consumers():
    from opentransformer_amd import ops
    import opentransformer_amd.nn as onn
    ops.set_compute_dtype('fp16')
    try:
        enc = make_encoder(2, 1024, 0.0, seed=2).eval()
        x, mask, gy = inputs(9, 260, seed=3)
        with torch.no_grad():
            a = enc(x, mask)[0]
            ops._FFN_SLAB = False
            try:
                b = enc(x, mask)[0]
            finally:
                ops._FFN_SLAB = True
        assert rel(a, b) < 2e-3
        # a pending LayerNorm read by something that is not the q|k|v projection, and read twice
        blk = enc.blocks[0]
        enc.train()
        km = mask.to(torch.uint8).unsqueeze(1)
        w = torch.randn(256, 256, device=DEV) * 0.05

        def chain(defer):
            xin = x.clone().requires_grad_(True)
            h, _ = enc.pos_emb(xin)
            blk._defer_ln = defer
            try:
                y, _ = blk(h, km, None)
            finally:
                blk._defer_ln = False
            assert (getattr(y, '_otr_pending', None) is not None) == defer
            z = ops.linear(y, w)                   # a 256 -> 256 Linear: not fused, materializes
            out = z + ops.materialize(y)           # second reader
            return out, torch.autograd.grad(out, [xin] + list(blk.parameters()), gy)
        o1, g1 = chain(True)
        o2, g2 = chain(False)
        assert rel(o1, o2) < 2e-3
        for u, v in zip(g1, g2):
            if float(v.norm()) > 1e-6:
                assert rel(u, v) < 1e-2, rel(u, v)
    finally:
        ops.set_compute_dtype('bf16')
