"""GPU (-m gpu): the fused decoder stack (csrc/declayer.hip: otr_dec_self / cross / ffn _fwd / _bwd, otr_dec_ln, otr_dec_sum) --
three launches per layer and direction, cut along (utterance group, head) / (32-row block, hidden slice) -- against

 * a plain fp32 torch restatement of decoder/transformer.py:47-90,161-183 (post-norm layers: causal self-attention,
   cross-attention over the masked encoder memory, GLU feed-forward; module/attention.py:23-46,60-84,120-145, module/ffn.py:38-41)
   evaluated on the same parameters, forward and every gradient;
 * the unfused HIP path (one launch per operator), with dropout ON: both draw their masks from the same counter RNG in the same
   order, so they must agree to rounding -- which pins the mask regeneration of the fused backward prologues.

Shapes cover: two utterances per 32-row group (L = 15, the AISHELL batch), four (L = 7), one (L = 20, L = 32), a last group that is
not full, ragged key masks, key counts that are not a multiple of the 32-key tile, and the bench shape itself."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# Gradients allowed over the flat bound tg, by name pattern, each with its own bound per mode (<= 1.5 x the worst value measured over
# SHAPES on MI355X, profiles/r05_tolerance_cases.jsonl: fp16 0.027, bf16 0.22).  They are ONE error, seen three times: the q|k|v
# gradient of the FIRST layer's self-attention and what it feeds (the embedding).  Layer 0 attends over sqrt(d) x embedding + PE
# rows (|x| ~ 16 x a later layer's LayerNorm output): its softmax saturates, P is close to one-hot, and dS = P o (dP - rowsum(dO o O))
# cancels to the rounding of the 16-bit O / dO / P operands.  Every other tensor of every shape meets tg; the per-operator HIP path
# shows the same three values (what the old "as far off as the unfused path" branch compared against).
OVER_TG = {'blocks.0.slf_attn.qvk_proj.': {'fp16': 4e-2, 'bf16': 3e-1}, 'embedding.weight': {'fp16': 4e-2, 'bf16': 3e-1}}
# residue the 16-bit paths leave in the analytically-zero key-bias gradient, relative to the live slices of the same bias
KEY_RESIDUE = {'fp16': 8e-3, 'bf16': 4.5e-2}        # <= 2.2 x measured (profiles/r05_tolerance_cases.jsonl: 3.7e-3 / 2.1e-2)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def row_rel(a, b):
    a, b = a.double().reshape(-1, a.shape[-1]), b.double().reshape(-1, b.shape[-1])
    return float(((a - b).norm(dim=1) / (b.norm(dim=1) + 1e-6)).max())


def make_decoder(n_blocks, d_ff, vocab, p_drop, seed):
    import opentransformer_amd.nn as onn
    torch.manual_seed(seed)
    dec = onn.TransformerDecoder(vocab, d_model=256, n_heads=4, d_ff=d_ff, memory_dim=256, n_blocks=n_blocks, residual_dropout=p_drop,
                                 activation='glu', normalize_before=False).to(DEV)
    with torch.no_grad():                       # default init leaves biases / norms trivial: make every term matter
        for n, p in dec.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return dec


def torch_decoder(dec, tokens, memory, key_mask, round_qkv=None):
    """fp32 restatement with torch ops only (dropout off); round_qkv = a 16-bit dtype: the FIRST layer's self-attention
    q | k | v are formed as the HIP path forms them (16-bit rows and weights, fp32 sums, rounded; straight-through gradient) -- the reference that shares the HIP path's attention operands"""
    d, H = 256, 4
    B, Lq = tokens.shape
    x = dec.embedding.weight[tokens] * math.sqrt(d)
    pos = torch.arange(Lq, device=x.device, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, device=x.device, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(Lq, d, device=x.device)
    pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    x = x + pe
    causal = torch.tril(torch.ones(Lq, Lq, dtype=torch.bool, device=x.device))

    def heads(t):
        return t.view(t.shape[0], t.shape[1], H, d // H).transpose(1, 2)

    def attend(q, k, v, mask):
        s = heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(d // H)
        s = s.masked_fill(~mask, float('-inf'))
        p = torch.softmax(s, dim=-1).masked_fill(~mask, 0.0)
        return (p @ heads(v)).transpose(1, 2).reshape(q.shape[0], q.shape[1], d)
    for li, b in enumerate(dec.blocks):
        sa, ca, ff = b.slf_attn, b.src_attn, b.feed_forward
        if round_qkv is not None and li == 0:
            # the HIP path's operands: 16-bit input rows and weights, fp32 accumulation, 16-bit q | k | v (straight-through gradients)
            def st(t):
                return t + (t.to(round_qkv).float() - t).detach()
            qkv = st(F.linear(st(x), st(sa.qvk_proj.weight), sa.qvk_proj.bias))
        else:
            qkv = F.linear(x, sa.qvk_proj.weight, sa.qvk_proj.bias)
        q, k, v = qkv.split(d, dim=-1)
        x = F.layer_norm(x + F.linear(attend(q, k, v, causal.view(1, 1, Lq, Lq)), sa.output_proj.weight, sa.output_proj.bias), (d,),
                         b.norm1.weight, b.norm1.bias, b.norm1.eps)
        q = F.linear(x, ca.q_proj.weight, ca.q_proj.bias)
        k, v = F.linear(memory, ca.vk_proj.weight, ca.vk_proj.bias).split(d, dim=-1)
        x = F.layer_norm(x + F.linear(attend(q, k, v, key_mask.view(B, 1, 1, -1)), ca.output_proj.weight, ca.output_proj.bias), (d,),
                         b.norm2.weight, b.norm2.bias, b.norm2.eps)
        h = F.linear(x, ff.w_1.weight, ff.w_1.bias)
        x = F.layer_norm(x + F.linear(F.glu(h, dim=-1), ff.w_2.weight, ff.w_2.bias), (d,), b.norm3.weight, b.norm3.bias, b.norm3.eps)
    return F.linear(x, dec.output_layer.weight, dec.output_layer.bias)


def inputs(B, Lq, T, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(1, vocab, (B, Lq), generator=g).to(DEV)
    memory = torch.randn(B, T, 256, generator=g).to(DEV)
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    lens[0] = T
    key_mask = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).to(DEV)
    gy = torch.randn(B, Lq, vocab, generator=g).to(DEV)
    return tokens, memory, key_mask, gy


def run_hip(dec, tokens, memory, key_mask, gy, fused):
    from opentransformer_amd import ops
    was = ops._DEC_FUSED
    ops._DEC_FUSED = fused
    try:
        ops.next_dropout_step(DEV)
        ops.rng_seed_tensor(DEV).fill_(1234)
        mem = ops.attach_lp(memory.clone().requires_grad_(True), memory.to(ops.act_dtype()))
        logits, _ = dec(tokens, mem, key_mask)
        ps = list(dec.parameters())
        grads = torch.autograd.grad(logits, [mem] + ps, gy)
        return logits.detach(), [g.detach() for g in grads]
    finally:
        ops._DEC_FUSED = was


SHAPES = [  # B, L, T', layers, d_ff
    (5, 15, 70, 2, 1024),      # two utterances per group, the last group half full, ragged keys, T' % 32 != 0
    (6, 7, 33, 1, 1024),       # four per group (6 = 4 + 2)
    (3, 20, 64, 1, 2048),      # one per group, padding rows inside the tile
    (2, 32, 40, 1, 1024),      # a full tile per utterance
    (32, 15, 249, 2, 2048),    # the AISHELL bench shape (two of its six layers)
]


@pytest.mark.parametrize('group', [0, 32])          # utterances per attention workgroup: 0 = the library's choice (round 5: smaller groups
@pytest.mark.parametrize('mode', ['fp16', 'bf16'])   # when the grid would be thin), 32 = full 32-row tiles (the round-4 geometry)
@pytest.mark.parametrize('B,Lq,T,nl,dff', SHAPES)
def test_fused_decoder_matches_fp32_torch(mode, B, Lq, T, nl, dff, group):
    from opentransformer_amd import ops, _lib
    ops.set_compute_dtype(mode)
    _lib.check(_lib.load().otr_debug_set(23, group), 'otr_debug_set')
    try:
        vocab = 200
        dec = make_decoder(nl, dff, vocab, 0.0, seed=B + Lq)
        dec.train()
        tokens, memory, key_mask, gy = inputs(B, Lq, T, vocab, seed=7 * B + Lq)
        assert ops.decoder_stack_applies(ops.attach_lp(torch.zeros(B, Lq, 256, device=DEV), torch.zeros(B, Lq, 256, device=DEV, dtype=ops.act_dtype())),
                                         memory, dec.blocks, False) > 0
        got, ggot = run_hip(dec, tokens, memory, key_mask, gy, fused=True)
        memr = memory.clone().requires_grad_(True)
        ref = torch_decoder(dec, tokens, memr, key_mask)
        gref = torch.autograd.grad(ref, [memr] + list(dec.parameters()), gy)
        ty, tg = (2e-3, 2e-2) if mode == 'fp16' else (1.5e-2, 8e-2)
        assert rel(got, ref) < ty, rel(got, ref)
        assert row_rel(got, ref) < 8 * ty, row_rel(got, ref)             # no single (utterance, position) is off
        names = ['memory'] + [n for n, _ in dec.named_parameters()]
        # no tensor is judged against the HIP path itself any more.  The key-bias slices (zero in exact arithmetic)
        # are measured apart (tests/helpers.py: key_aware_grad_errors); every other gradient meets the flat bound tg, except the
        # tensors NAMED in OVER_TG with their own measured bound
        errs, key_errs = H.key_aware_grad_errors(names, ggot, gref)
        worst = max((e, n) for n, e in errs.items())
        over = {n: e for n, e in errs.items() if e >= tg}
        H.log_tolerance_cases('decoder_fused', {'mode': mode, 'shape': [B, Lq, T, nl, dff], 'group': group, 'tg': tg, 'worst': worst, 'over_tg': over,
                                                'key_bias_residue': key_errs})
        for n, e in over.items():
            bound = next((b for pat, b in OVER_TG.items() if pat in n), None)
            assert bound is not None and e < bound[mode], ('gradient over the flat bound and not a named exception', n, e, tg)
        if over:
            # r06: the cause is MEASURED, not only named: against an fp32 reference whose layer-0 q|k|v are rounded to the compute type
            # (everything else fp32) the same gradients meet the flat bound -- the gap to the plain fp32 reference is the rounding of the
            # attention OPERANDS in front of a saturated softmax, not the backward arithmetic (tools/delta_study.py)
            memr2 = memory.clone().requires_grad_(True)
            ref2 = torch_decoder(dec, tokens, memr2, key_mask, round_qkv=ops.act_dtype())
            gref2 = torch.autograd.grad(ref2, [memr2] + list(dec.parameters()), gy)
            errs2, _ = H.key_aware_grad_errors(names, ggot, gref2)
            same_operands = {n: errs2[n] for n in over}
            H.log_tolerance_cases('decoder_fused_same_operands', {'mode': mode, 'shape': [B, Lq, T, nl, dff], 'group': group, 'tg': tg,
                                                                  'vs_fp32': over, 'vs_fp32_on_rounded_qkv': same_operands})
            assert all(e < tg for e in same_operands.values()), (same_operands, tg)
        assert all(e < KEY_RESIDUE[mode] for e in key_errs.values()), key_errs
        print('decoder parity', mode, (B, Lq, T), 'logits %.2e rows %.2e worst grad %.2e %s' % (rel(got, ref), row_rel(got, ref), worst[0], worst[1]))
        # keys beyond an utterance's length receive no gradient
        assert float(ggot[0][~key_mask].abs().max()) < 1e-3 * float(ggot[0].abs().max())
    finally:
        _lib.load().otr_debug_set(23, 0)
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('B,Lq,T,nl,dff,p_drop', [(5, 15, 70, 2, 1024, 0.25), (32, 15, 249, 1, 2048, 0.1), (6, 7, 33, 2, 1024, 0.0)])
def test_fused_decoder_matches_unfused_hip_path_with_dropout(B, Lq, T, nl, dff, p_drop):
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp16')
    try:
        vocab = 200
        dec = make_decoder(nl, dff, vocab, p_drop, seed=3)
        dec.train()
        tokens, memory, key_mask, gy = inputs(B, Lq, T, vocab, seed=11)
        a, ga = run_hip(dec, tokens, memory, key_mask, gy, fused=True)
        b, gb = run_hip(dec, tokens, memory, key_mask, gy, fused=False)
        assert rel(a, b) < 3e-3, rel(a, b)
        assert row_rel(a, b) < 2e-2, row_rel(a, b)
        names = ['memory'] + [n for n, _ in dec.named_parameters()]
        worst = max((rel(x, y), n) for n, x, y in zip(names, ga, gb) if float(y.norm()) > 1e-6)
        assert worst[0] < 2e-2, worst
        if p_drop > 0:      # the mask matters: without it the outputs differ by far more than rounding
            dec.eval()
            c, _ = run_hip(dec, tokens, memory, key_mask, gy, fused=True)
            assert rel(a, c) > 5e-2
    finally:
        ops.set_compute_dtype('bf16')


def test_fused_decoder_no_grad_and_fallbacks():
    """inference (no autograd) takes the fused path too; shapes it does not serve fall back to the per-operator path"""
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp16')
    try:
        vocab = 120
        dec = make_decoder(2, 1024, vocab, 0.1, seed=5).eval()
        tokens, memory, key_mask, _ = inputs(4, 9, 50, vocab, seed=2)
        with torch.no_grad():
            mem = ops.attach_lp(memory, memory.to(ops.act_dtype()))
            a, _ = dec(tokens, mem, key_mask)
            ops._DEC_FUSED = False
            try:
                b, _ = dec(tokens, mem, key_mask)
            finally:
                ops._DEC_FUSED = True
            assert rel(a, b) < 3e-3
            tokens2 = torch.randint(1, vocab, (2, 40), device=DEV)          # L > 32: not served
            x = ops.embed_posenc(tokens2, dec.embedding.weight)
            assert ops.decoder_stack_applies(x, memory[:2], dec.blocks, False) == 0
            c, _ = dec(tokens2, ops.attach_lp(memory[:2], memory[:2].to(ops.act_dtype())), key_mask[:2])
            assert torch.isfinite(c).all()
    finally:
        ops.set_compute_dtype('bf16')
