"""GPU (-m gpu): the encoder-shape self-attention backward kernel (csrc/encattn.hip: 16-bit, head dim 64, no causal mask, T <= 256; one
(utterance, head) staged whole in LDS, one workgroup per orientation) against

 * a plain fp32 torch restatement of module/attention.py:23-46 on the same 16-bit inputs, and
 * the generic streamed kernels of attention.hip (otr_debug_set(21, 0)), which round at the same places.

Shapes: the AISHELL one (249 frames), exact tile multiples, one frame, a partial last tile, head counts that leave padding workgroups in
the XCD-aware grid, ragged key masks (an utterance with ONE live key included), and T = 256 (every wave busy).  r06: 257 <= T <= 512 run
on the same kernel (own blocks of 256 rows x streamed super-chunks of 256 rows: 257 = a second block of ONE row, 349 / 363 = AISHELL's
longest utterances, 384, 512 = every wave of both blocks busy); T = 513 falls back to the generic kernels."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
HALF = {'bf16': torch.bfloat16, 'fp16': torch.float16}
TOL = {'bf16': 1.5e-2, 'fp16': 2e-3}


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ref_attention(q, k, v, key_mask, H):
    B, T, d = q.shape
    dk = d // H
    qh, kh, vh = (t.float().view(B, T, H, dk).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(2, 3) / math.sqrt(dk)
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], float('-inf'))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, d)


def run(qkv, km, H, g, enc):
    from opentransformer_amd import ops, _lib as L
    lib = L.load()
    L.check(lib.otr_debug_set(21, 1 if enc else 0), 'debug_set')
    try:
        x = qkv.detach().clone().requires_grad_(True)
        out = ops.SelfAttentionFn.apply(x, km.to(torch.uint8) if km is not None else None, H, False)
        (dqkv,) = torch.autograd.grad(out, x, g)
        return out.detach(), dqkv
    finally:
        lib.otr_debug_set(21, 1)


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
@pytest.mark.parametrize('B,T,H,ragged', [(3, 249, 4, True), (32, 249, 4, True), (5, 32, 4, False), (2, 33, 3, True), (9, 256, 1, True),
                                          (2, 1, 4, False), (4, 100, 2, True), (1, 64, 4, False), (3, 257, 4, True),
                                          # r06: beyond 256 frames -- own blocks x streamed super-chunks (AISHELL's longest utterances: T' ~ 349-363)
                                          (4, 349, 4, True), (32, 349, 4, True), (3, 363, 4, True), (2, 384, 2, True), (2, 512, 1, True), (5, 289, 3, True),
                                          (1, 513, 2, False)])
def test_encoder_attention_backward(mode, B, T, H, ragged):
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        d = 64 * H
        gen = torch.Generator().manual_seed(100 * B + T)
        qkv = (torch.randn(B, T, 3 * d, generator=gen) * 0.7).to(DEV).to(HALF[mode])
        g = torch.randn(B, T, d, generator=gen).to(DEV).to(HALF[mode])
        km = None
        if ragged:
            lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=gen)
            lens[0] = T
            if B > 1:
                lens[-1] = 1                                  # one live key: softmax over a single element
            km = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).to(DEV)
        out, dq = run(qkv, km, H, g, enc=True)
        out0, dq0 = run(qkv, km, H, g, enc=False)
        assert torch.equal(out, out0)                          # the forward kernel is shared
        qf = qkv.float().requires_grad_(True)
        ref = ref_attention(qf[..., :d], qf[..., d:2 * d], qf[..., 2 * d:], km, H)
        (dref,) = torch.autograd.grad(ref, qf, g.float())
        for nm, sl in (('dq', slice(0, d)), ('dk', slice(d, 2 * d)), ('dv', slice(2 * d, 3 * d))):
            if float(dref[..., sl].norm()) < 1e-6:            # one frame: the softmax is constant, dq = dk = 0 exactly
                assert float(dq[..., sl].float().abs().max()) < 1e-3, nm
                continue
            e_new, e_old = rel(dq[..., sl].float(), dref[..., sl]), rel(dq0[..., sl].float(), dref[..., sl])
            assert e_new < 2 * TOL[mode], (nm, e_new, e_old)
            assert e_new < 1.5 * e_old + 1e-4, (nm, e_new, e_old)     # as close to fp32 as the generic kernels
        assert torch.isfinite(dq.float()).all()
        if km is not None:             # masked keys receive no gradient (their dK / dV rows are written, as zeros)
            dead = ~km
            assert float(dq[..., d:][dead].float().abs().max()) == 0.0
    finally:
        ops.set_compute_dtype('bf16')
