"""GPU (-m gpu): round-5 launches.

* otr_label_smoothing_loss_fused (module/loss.py:21-48 in ONE launch: rows read once, the last block sums the row losses in the
  three-kernel form's order, the gradient pre-multiplied by a device scalar, targets as a strided view) against the three-kernel
  form and against plain fp32 torch; deterministic; replayable from a hipGraph (the arrival ticket returns to zero);
* the embedding on token VIEWS (truth[:, :-1]) and the fused decoder stack's input gradient summed by the embedding's backward
  (otr_embed_bwd_ld) against the otr_dec_sum form;
* the whole model: views + fused loss + folded loss scale give the loss / gradients of the round-4 forms."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def torch_ls_loss(logits, target, smoothing, pad):
    """module/loss.py:21-48 restated with torch ops (fp64)"""
    V = logits.shape[-1]
    x = logits.reshape(-1, V).double()
    t = target.reshape(-1)
    keep = t != pad
    true = torch.full_like(x, smoothing / (V - 1))
    true.scatter_(1, t.clamp(min=0).unsqueeze(1), 1.0 - smoothing)
    kl = F.kl_div(torch.log_softmax(x, dim=1), true, reduction='none')
    return kl.masked_fill(~keep.unsqueeze(1), 0.0).sum() / keep.sum()


CASES = [  # B, L, V, padded rows (ld), pad positions, strided target view
    (32, 15, 4234, True, True, True),       # the AISHELL output layer: head of a [480, 4240] product, truth[:, 1:] view
    (32, 15, 4234, True, False, False),
    (4, 7, 100, False, True, True),         # V % 4 == 0: contiguous rows are aligned
    (3, 5, 2048, False, True, False),       # the 2-quad instantiation's upper edge
    (2, 9, 5120, False, False, True),       # the 5-quad instantiation's upper edge
    (2, 3, 8192, False, True, False),       # 8 quads
    (1, 1, 36, False, False, False),        # one row
]


@pytest.mark.parametrize('B,L,V,padded,with_pad,view', CASES)
def test_fused_loss_matches_three_kernel_form_and_torch(B, L, V, padded, with_pad, view):
    from opentransformer_amd import ops
    g = torch.Generator().manual_seed(B * 131 + L * 7 + V)
    R = B * L
    ld = (V + 7) // 8 * 8 if padded else V
    buf = (3.0 * torch.randn(R, ld, generator=g)).to(DEV)
    truth = torch.randint(1, V, (B, L + 1), generator=g).to(DEV)
    if with_pad:
        truth[0, L // 2 + 1:] = 0
        truth[-1, -1] = 0
    target = truth[:, 1:] if view else truth[:, 1:].contiguous()
    scale = torch.tensor([1024.0], device=DEV)

    def run(fused, gs):
        was = ops._LS_FUSED
        ops._LS_FUSED = fused
        try:
            base = buf.clone().requires_grad_(True)
            logits = base[:, :V].view(B, L, V)
            loss = ops.label_smoothing_loss(logits, target, 0.1, 0, grad_scale=gs)
            ops.backward(loss)
            return loss.detach().clone(), base.grad.clone()
        finally:
            ops._LS_FUSED = was
    l1, g1 = run(True, scale)
    l0, g0 = run(False, scale)
    assert float(g1[:, V:].abs().max()) == 0.0 if ld > V else True            # the zero tail of a padded row
    assert abs(float(l1) - float(l0)) <= 2e-6 * abs(float(l0)), (float(l1), float(l0))
    assert rel(g1, g0) < 2e-6, rel(g1, g0)
    ref_in = buf[:, :V].double().clone().requires_grad_(True)
    ref = torch_ls_loss(ref_in.view(B, L, V), target, 0.1, 0)
    ref.backward()
    assert abs(float(l1) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel(g1[:, :V], 1024.0 * ref_in.grad) < 5e-6
    l2, g2 = run(True, scale)                                                     # deterministic: bit-equal on a second run
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    l3, g3 = run(True, None)                                                      # no scale: the plain gradient
    assert rel(g3 * 1024.0, g1) < 1e-6 and torch.equal(l3, l1)
    # a gradient that is NOT the unit seed of ops.backward takes the general path
    base = buf.clone().requires_grad_(True)
    loss = ops.label_smoothing_loss(base[:, :V].view(B, L, V), target, 0.1, 0, grad_scale=scale)
    (2.5 * loss).backward()
    assert rel(base.grad, 2.5 * g1) < 1e-6


def test_fused_loss_replays_from_a_hipgraph():
    """the arrival ticket is left at zero by every launch: a captured loss launch replays any number of times"""
    from opentransformer_amd import ops
    B, L, V = 32, 15, 4234
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(B * L, 4240, generator=g).to(DEV)
    target = torch.randint(1, V, (B, L), generator=g).to(DEV)
    logits = buf[:, :V].view(B, L, V)
    want = ops.label_smoothing_loss(logits, target, 0.1, 0).clone()            # eager first: allocates the ticket outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.label_smoothing_loss(logits, target, 0.1, 0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with ops.graph_capture(graph):
        out = ops.label_smoothing_loss(logits, target, 0.1, 0)
    for _ in range(4):
        out.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), (float(out), float(want))
    assert int(ops._ls_ticket(buf.device)[0]) == 0


def test_embedding_reads_token_views_and_takes_the_decoder_slabs():
    from opentransformer_amd import ops
    from tests.test_gpu_decoder_fused import make_decoder, inputs, run_hip
    ops.set_compute_dtype('fp16')
    try:
        B, Lq, T, vocab = 5, 15, 70, 200
        dec = make_decoder(2, 1024, vocab, 0.0, seed=3)
        dec.train()
        tokens, memory, key_mask, gy = inputs(B, Lq, T, vocab, seed=11)
        truth = torch.cat([tokens, tokens[:, :1]], dim=1)                         # [B, L + 1]: tokens = truth[:, :-1] as a VIEW
        view = truth[:, :-1]
        assert not view.is_contiguous()
        names = ['memory'] + [n for n, _ in dec.named_parameters()]
        res = {}
        for label, sink, tok in (('sink+view', True, view), ('sum+contiguous', False, tokens)):
            was = ops._EMBED_SINK
            ops._EMBED_SINK = sink
            try:
                recs = []
                ops.set_kernel_timer(recs)
                try:
                    res[label] = run_hip(dec, tok, memory, key_mask, gy, fused=True)
                finally:
                    ops.set_kernel_timer(None)
            finally:
                ops._EMBED_SINK = was
        (ya, ga), (yb, gb) = res['sink+view'], res['sum+contiguous']
        assert torch.equal(ya, yb)
        for n, a, b in zip(names, ga, gb):
            assert rel(a, b) < 1e-5, (n, rel(a, b))       # two runs of one backward pass differ in the last bits (float atomics), nothing more
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'fp32'])
def test_model_step_with_views_and_fused_loss_matches_the_round4_forms(mode):
    """SpeechToText.forward (model/speech2text.py:44-62): the shifted target VIEWS + the one-launch loss with the loss scale folded in
    against torch.stack + three-kernel loss + ScaleGradFn, same parameters, same batch: same loss, same gradients"""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype(mode)
    try:
        cfg = syn.c1_model(0.0)
        inputs, targets = syn.synthetic_batch(batch=4, frames=200, feat_dim=80, vocab=100, tgt_len=10, seed=0, lengths=[200, 180, 150, 97],
                                              tgt_lengths=[10, 8, 10, 5])
        inputs, targets = {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in targets.items()}
        out = {}
        for label, fused in (('new', True), ('old', False)):
            was = (ops._LS_FUSED, ops._EMBED_SINK)
            ops._LS_FUSED = ops._EMBED_SINK = fused
            try:
                model = ota.SpeechToText(cfg)
                syn.fill_state_dict_(model.state_dict(), 21)
                model = model.to(DEV).train()
                dp = FlatDataParallel(model)
                opt = FusedAdam(dp, lr=1e-3, loss_scale=(512.0 if mode == 'fp16' else None))
                dp.zero_grad()
                loss, _ = dp(inputs, targets)
                ops.backward(loss)
                torch.cuda.synchronize()
                out[label] = (loss.detach().clone(), dp.flat_grad.clone())
            finally:
                ops._LS_FUSED, ops._EMBED_SINK = was
        assert abs(float(out['new'][0]) - float(out['old'][0])) < 2e-6 * abs(float(out['old'][0]))
        assert rel(out['new'][1], out['old'][1]) < (1e-5 if mode == 'fp32' else 1e-3), rel(out['new'][1], out['old'][1])
    finally:
        ops.set_compute_dtype('bf16')


def test_zero_tick_clears_and_advances_the_dropout_seed():
    from opentransformer_amd import ops
    buf = torch.randn(1000003, device=DEV)[:1000000]               # 16-byte aligned head, n % 4 == 0 and a ragged case below
    seed = ops.rng_seed_tensor(buf.device)
    before = int(seed[0])
    ops.zero_and_next_dropout_step(buf)
    torch.cuda.synchronize()
    assert float(buf.abs().max()) == 0.0
    after = int(seed[0])
    ops.rng_seed_tensor(buf.device).fill_(before)
    ops.next_dropout_step(buf.device)
    assert int(seed[0]) == after                                       # the same step as ops.next_dropout_step
    odd = torch.randn(4099, device=DEV)
    guard = odd[4097:].clone()
    ops.zero_and_next_dropout_step(odd[:4097])
    torch.cuda.synchronize()
    assert float(odd[:4097].abs().max()) == 0.0 and torch.equal(odd[4097:], guard)


def test_posenc_launch_also_casts_the_key_mask():
    """the frame mask behind the two stride-2 convolutions is a strided bool view (frontend/conv.py:78-83): its uint8 form leaves the
    positional-encoding launch, bit-identical outputs, and ops._mask_u8 finds it without a cast"""
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp16')
    try:
        B, T0, d = 5, 1000, 256
        g = torch.Generator().manual_seed(3)
        lens = torch.tensor([1000, 873, 640, 999, 512])
        mask0 = (torch.arange(T0).unsqueeze(0) < lens.unsqueeze(1)).to(DEV)
        t1 = (T0 - 3) // 2 + 1
        m1 = mask0[:, 1::2][:, :t1]
        T = (t1 - 3) // 2 + 1
        mask = m1[:, 1::2][:, :T]
        assert not mask.is_contiguous()
        x = torch.randn(B, T, d, generator=g).to(DEV)
        y0 = ops.posenc(x)
        y1 = ops.posenc(x, mask)
        assert torch.equal(y0, y1) and torch.equal(ops.lp_of(y0), ops.lp_of(y1))
        u8 = ops._mask_u8(mask, B, T)
        assert u8.dtype == torch.uint8 and u8.is_contiguous() and torch.equal(u8.bool(), mask)
        assert getattr(mask, '_otr_u8')[1] is u8
        ref = x * 16.0
        pos = torch.arange(T, device=DEV, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2, device=DEV, dtype=torch.float32) * -(torch.log(torch.tensor(10000.0)) / d))
        pe = torch.zeros(T, d, device=DEV)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        assert rel(y1, ref + pe) < 1e-6
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('accumulate', [False, True])
def test_frontend_linear_takes_its_gradient_as_a_16bit_operand(accumulate):
    """ops._Grad16Link: the positional encoding's backward hands sqrt(d) dy to the frontend's output Linear as a 16-bit operand (its
    weight gradient then joins the grouped 256-wide launch through a kernel-order staging image + one regrouping add).  Against the
    fp32-gradient path on the same model and batch: every gradient behind the hand-over (frontend) agrees to 16-bit rounding, every
    other gradient exactly; with two backward passes per step (accumulation) the staging image is not counted twice."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype('fp16')
    try:
        cfg = syn.c2_model(0.0)
        cfg['encoder']['n_blocks'] = 2
        cfg['decoder']['n_blocks'] = 1
        inputs, targets = syn.synthetic_batch(batch=4, frames=400, feat_dim=80, vocab=4234, tgt_len=8, seed=2, lengths=[400, 333, 280, 395])
        inputs, targets = {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in targets.items()}
        out = {}
        for label, on in (('g16', True), ('fp32', False)):
            was = ops._G16
            ops._G16 = on
            try:
                model = ota.SpeechToText(cfg)
                syn.fill_state_dict_(model.state_dict(), 5)
                model = model.to(DEV).train()
                dp = FlatDataParallel(model)
                FusedAdam(dp, lr=1e-3, loss_scale=256.0)
                names = recs = []
                ops.set_kernel_timer(recs)
                try:
                    dp.zero_grad()
                    for _ in range(2 if accumulate else 1):
                        loss, _ = dp(inputs, targets)
                        ops.backward(loss)
                finally:
                    ops.set_kernel_timer(None)
                torch.cuda.synchronize()
                out[label] = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
                out[label + '_staged'] = getattr(model.frontend.output_layer.weight, '_otr_regroup_state', {'dirty': None})['dirty']
            finally:
                ops._G16 = was
        assert out['g16_staged'] is True and out['fp32_staged'] is False          # the hand-over really ran / really did not
        worst = max((rel(g, out['fp32'][n]), n) for n, g in out['g16'].items())
        # two hand-overs are on: the loss launch's gradient (16-bit operand of the output layer: one fp16 rounding, eps 4.9e-4, at the
        # top of the backward pass -- every gradient moves by about that much) and the positional encoding's (the frontend's Linear)
        assert worst[0] < 1e-2, worst
        assert float(out['g16']['frontend.output_layer.weight'].abs().sum()) > 0
        assert float(out['g16'].get('decoder.output_layer.weight', out['g16']['decoder.embedding.weight']).abs().sum()) > 0
    finally:
        ops.set_compute_dtype('bf16')


def test_gradient_accumulation_with_store_first_weight_gradients():
    """two backward passes into one cleared gradient buffer (dp.no_sync-style accumulation at world size 1) give twice the gradient of
    one: the first pass's weight-gradient launch STORES into the cleared buffers (ops.gradients_cleared), the second accumulates"""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype('fp16')
    try:
        cfg = syn.c2_model(0.0)
        cfg['encoder']['n_blocks'] = 1
        cfg['decoder']['n_blocks'] = 1
        inputs, targets = syn.synthetic_batch(batch=8, frames=1000, feat_dim=80, vocab=4234, tgt_len=15, seed=4)
        inputs, targets = {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in targets.items()}
        model = ota.SpeechToText(cfg)
        syn.fill_state_dict_(model.state_dict(), 9)
        model = model.to(DEV).train()
        dp = FlatDataParallel(model)
        FusedAdam(dp, lr=1e-3, loss_scale=256.0)
        res = []
        for passes in (1, 2):
            dp.zero_grad()
            for _ in range(passes):
                loss, _ = dp(inputs, targets)
                ops.backward(loss)
            torch.cuda.synchronize()
            res.append(dp.flat_grad.clone())
        assert rel(res[1], 2 * res[0]) < 1e-5, rel(res[1], 2 * res[0])
        was = ops._WG_OVERWRITE
        ops._WG_OVERWRITE = False                                      # and the same gradient with the switch off
        try:
            dp.zero_grad()
            loss, _ = dp(inputs, targets)
            ops.backward(loss)
            torch.cuda.synchronize()
            assert rel(dp.flat_grad, res[0]) < 1e-5
        finally:
            ops._WG_OVERWRITE = was
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
@pytest.mark.parametrize('R,pos,maxlen', [(80, 0, 61), (80, 1, 61), (80, 29, 61), (80, 60, 61), (20, 63, 130), (20, 64, 130), (20, 127, 130), (7, 5, 13)])
def test_cached_self_attention_vector_form_matches_plain_torch_and_the_serial_kernel(mode, R, pos, maxlen):
    """otr_decode_self_attention on 16-bit operands with head dim 64 (csrc/decode.hip: 16-byte loads, a chunk's value rows all in
    flight) against plain fp32 torch on the same caches and ancestor table, and against the serial kernel it replaces
    (otr_debug_set(24, 0)); the new key / value land in the caches at [r, pos]; positions past 64 take the second chunk."""
    from opentransformer_amd import ops, _lib
    ops.set_compute_dtype(mode)
    try:
        H, dk = 4, 64
        d = H * dk
        adt = ops.act_dtype()
        g = torch.Generator().manual_seed(R * 1000 + pos)
        qkv = torch.randn(R, 3 * d, generator=g).to(DEV, adt)
        kc0 = torch.randn(R, maxlen, d, generator=g).to(DEV, adt)
        vc0 = torch.randn(R, maxlen, d, generator=g).to(DEV, adt)
        anc = torch.randint(0, R, (R, maxlen), generator=g).to(torch.int32).to(DEV)
        p = torch.tensor([pos], dtype=torch.int32, device=DEV)
        outs = {}
        for form in (1, 0):
            _lib.check(_lib.load().otr_debug_set(24, form), 'otr_debug_set')
            kc, vc = kc0.clone(), vc0.clone()
            outs[form] = (ops.decode_self_attention(qkv, kc, vc, anc, p, H), kc, vc)
        _lib.load().otr_debug_set(24, 1)
        (o1, k1, v1), (o0, k0, v0) = outs[1], outs[0]
        assert torch.equal(k1, k0) and torch.equal(v1, v0)
        assert torch.equal(k1[:, pos], qkv[:, d:2 * d]) and torch.equal(v1[:, pos], qkv[:, 2 * d:])
        # plain torch on the same numbers
        q = qkv[:, :d].float().view(R, H, dk)
        rows = torch.arange(R, device=DEV)
        keys = torch.stack([kc0[anc[:, j].long(), j] for j in range(pos)] + [qkv[:, d:2 * d]], dim=1).float().view(R, pos + 1, H, dk)
        vals = torch.stack([vc0[anc[:, j].long(), j] for j in range(pos)] + [qkv[:, 2 * d:]], dim=1).float().view(R, pos + 1, H, dk)
        sc = torch.einsum('rhd,rjhd->rhj', q, keys) / 8.0
        ref = torch.einsum('rhj,rjhd->rhd', torch.softmax(sc, dim=-1), vals).reshape(R, d)
        tol = 4e-3 if mode == 'fp16' else 2e-2              # the 16-bit rounding of the output
        assert rel(o1.float(), ref) < tol, rel(o1.float(), ref)
        assert rel(o1.float(), o0.float()) < tol
    finally:
        ops.set_compute_dtype('bf16')


# ----------------------------------------------------------------------------------- cached decode step: fused self-attention launch
@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
@pytest.mark.parametrize('R,pos,nslab', [(13, 0, 0), (13, 5, 8), (80, 37, 16), (21, 70, 4), (8, 63, 0), (30, 64, 16)])
def test_dec_self_step_against_torch(mode, R, pos, nslab):
    """otr_dec_self_step (csrc/declayer.hip) = LayerNorm of the layer below + q|k|v + attention over the ancestors' cached positions
    + the heads' shares of the output projection, checked against the same arithmetic in torch fp32 on the 16-bit rounded operands
    (module/attention.py:60-84 for one new position; the cache addressing of otr_decode_self_attention).  Ragged last block
    (R % 8 != 0), first step (pos 0), more than one 64-position chunk (pos 70), chunk boundary (pos 63)."""
    import ctypes as C
    from opentransformer_amd import _lib as L, ops
    ops.set_compute_dtype(mode)
    try:
        lib, hdt, d, H = L.load(), ops.half_dtype(), 256, 4
        maxlen = pos + 3
        gen = torch.Generator().manual_seed(100 * R + pos)
        rnd = lambda *sh, s=1.0: (torch.randn(*sh, generator=gen) * s).to(DEV)      # noqa: E731
        wqkv, bqkv, wo = torch.nn.Parameter(rnd(768, d, s=0.06)), rnd(768, s=0.1), torch.nn.Parameter(rnd(d, d, s=0.06))
        kc, vc = rnd(R, maxlen, d, s=0.7).to(hdt), rnd(R, maxlen, d, s=0.7).to(hdt)
        kc0, vc0 = kc.clone(), vc.clone()
        anc = torch.randint(0, R, (R, maxlen), generator=gen).to(DEV, torch.int32)
        post = torch.tensor([pos], dtype=torch.int32, device=DEV)
        xres = rnd(R, d)
        y, y16 = torch.zeros(R, d, device=DEV), torch.zeros(R, d, dtype=hdt, device=DEV)
        if nslab:
            sl_in = rnd(nslab, R, d, s=0.3).to(hdt)
            bias, gamma, beta = rnd(d, s=0.1), 1 + rnd(d, s=0.1), rnd(d, s=0.1)
            ln = ops._dec_ln(xres, None, sl_in, nslab, bias, gamma, beta, None, 0.0, 1e-5, 0, y, y16)
            want_y = torch.nn.functional.layer_norm(xres + sl_in.float().sum(0) + bias, (d,), gamma, beta, 1e-5)
            xin = want_y.to(hdt).float()
        else:
            x16 = xres.to(hdt)
            ln = ops._dec_ln(None, x16, None, 0)
            xin = x16.float()
        slabs = torch.full((H, R, d), float('nan'), dtype=hdt, device=DEV)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pq, po = ops.lin_packs(wqkv)[0], ops.lin_packs(wo)[0]
        L.check(lib.otr_dec_self_step(C.byref(ln), R, ops._p(pq), ops._p(bqkv), ops._p(po), ops._p(kc), ops._p(vc), ops._p(anc), ops._p(post),
                                      maxlen, ops._p(slabs), st), 'otr_dec_self_step')
        torch.cuda.synchronize()
        tol = dict(rtol=2e-2, atol=2e-2) if mode == 'bf16' else dict(rtol=3e-3, atol=3e-3)
        if nslab:
            torch.testing.assert_close(y, want_y, rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(y16.float(), want_y, **tol)
        w16 = wqkv.detach().to(hdt).float()
        qkv = (xin @ w16.T + bqkv).to(hdt).float()
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        # the caches: position `pos` of every row appended, everything else untouched
        torch.testing.assert_close(kc[:, pos].float(), k, **tol)
        torch.testing.assert_close(vc[:, pos].float(), v, **tol)
        keep = torch.ones(maxlen, dtype=torch.bool, device=DEV)
        keep[pos] = False
        assert torch.equal(kc[:, keep], kc0[:, keep]) and torch.equal(vc[:, keep], vc0[:, keep])
        rows = anc[:, :pos].long()                                          # [R, pos]
        jj = torch.arange(pos, device=DEV)
        K = torch.cat((kc0[rows, jj].float(), k[:, None]), 1).view(R, pos + 1, H, 64)
        V = torch.cat((vc0[rows, jj].float(), v[:, None]), 1).view(R, pos + 1, H, 64)
        s = torch.einsum('rhd,rjhd->rhj', q.view(R, H, 64), K) * 0.125
        ctx = torch.einsum('rhj,rjhd->rhd', torch.softmax(s, -1), V).to(hdt).float()        # [R, H, 64]
        wo16 = wo.detach().to(hdt).float()
        for h in range(H):
            want = ctx[:, h] @ wo16[:, 64 * h:64 * (h + 1)].T
            torch.testing.assert_close(slabs[h].float(), want, **tol)
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
def test_cached_search_fused_step_matches_unfused_step(mode):
    """the cached beam search at the C5 shapes on otr_dec_self_step + the fused tail vs the same search on the per-operator launches
    (qkv GEMM, otr_decode_self_attention, output projection + LayerNorm): same hypotheses wherever the margin is clear, close scores"""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, recognize, synthetic as syn
    ops.set_compute_dtype(mode)
    try:
        model = ota.SpeechToText(syn.c2_model(0.0, n_enc=2))
        syn.fill_state_dict_(model.state_dict(), 7)
        lm = recognize.TransformerLanguageModel(syn.lm_config(4234, num_blocks=2))
        syn.fill_state_dict_(lm.state_dict(), 8)
        with torch.no_grad():
            model.decoder.output_layer.bias[1] = -30.0          # EOS never wins: every hypothesis runs max_len steps
        model, lm = model.to(DEV).eval(), lm.to(DEV).eval()
        inputs, _ = syn.synthetic_batch(batch=3, frames=400, feat_dim=80, vocab=4234, tgt_len=5, seed=3, lengths=[400, 333, 250])
        x, m = inputs['inputs'].to(DEV), inputs['mask'].to(DEV)
        kw = dict(beam_width=10, nbest=10, max_len=20, penalty=0.6, lamda=5, lm=lm, lm_weight=0.1, idx2unit={i: str(i) for i in range(4234)})
        res = {}
        for fused in (True, False):
            recognize._DECODE_STEP_FUSED = fused
            rec = recognize.SpeechToTextRecognizer(model, apply_cache=True, **kw)
            res[fused] = rec.recognize(x, m)
            stt = next(iter(rec._cached_states.values()))
            assert stt.fused_dec == fused and stt.fused_lm == fused
        (h1, s1), (h0, s0) = res[True], res[False]
        s1, s0 = s1.numpy(), s0.numpy()
        tol = 0.15 if mode == 'bf16' else 0.03
        assert abs(s1[:, 0] - s0[:, 0]).max() < tol, (s1[:, 0], s0[:, 0])
        for i in range(len(h0)):
            if s0[i, 0] - s0[i, 1] > 2 * tol:
                assert h1[i][0] == h0[i][0], i
    finally:
        recognize._DECODE_STEP_FUSED = True
        ops.set_compute_dtype('bf16')


# ----------------------------------------------------------------------------------- Conformer: per-head GEMMs in one launch, dp products deferred
@pytest.mark.parametrize('mode', ['fp16', 'fp32'])
def test_conformer_through_the_engine_matches_bare_model(mode):
    """MultiHeadedSelfAttentionWithRelPos (module/attention.py:217-253) on the batched per-head GEMM launch, and -- through
    FlatDataParallel's in-place gradients -- its dp_h = dbd_h^T (q+v)_h products in the grouped weight-gradient launch with dW_pos behind it
    (ops.RelPosAttentionFn.backward): every parameter gradient of the small Conformer equals the bare model's (per-operator launches,
    OTR_GEMM_BATCHED / OTR_POS_DEFER off) up to summation order."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn, nn as nn_mod
    from opentransformer_amd.dp import FlatDataParallel
    ops.set_compute_dtype(mode)
    try:
        cfg = syn.conformer_model(small=True)
        inputs, targets = syn.synthetic_batch(batch=4, frames=120, feat_dim=80, vocab=100, tgt_len=6, seed=11, lengths=[120, 97, 80, 111])
        di = {k: v.to(DEV) for k, v in inputs.items()}
        dt = {k: v.to(DEV) for k, v in targets.items()}

        def grads(engine, batched):
            ops._GEMM_BATCHED, ops._POS_DEFER, ops._DW_PART, ops._BN_PART, ops._DBD_PERSIST = batched, batched, batched, batched, batched
            nn_mod._RES_LN = nn_mod._LN2 = nn_mod._MASK_FOLD = batched
            model = ota.SpeechToText(cfg)
            syn.fill_state_dict_(model.state_dict(), 77)
            model = model.to(DEV).train()
            if engine:
                dp = FlatDataParallel(model)
                dp.zero_grad()
                loss, _ = dp(di, dt)
                ops.backward(loss)
            else:
                loss, _ = model(di, dt)
                loss.backward()
            torch.cuda.synchronize()
            return float(loss), {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}

        l0, g0 = grads(False, False)
        l1, g1 = grads(True, True)
        assert abs(l0 - l1) <= (2e-3 if mode == 'fp16' else 1e-5) * abs(l0), (l0, l1)
        tol = 2e-2 if mode == 'fp16' else 2e-4
        # (a bias in front of a BatchNorm on batch statistics has an analytically zero gradient, module/conformer.py:103-110: what the two
        #  runs hold there is rounding noise of different summation orders -- measured against the largest gradient instead of itself)
        gmax = max(float(v.norm()) for v in g0.values())
        worst = max((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-3 * gmax)), k) for k in g0)
        assert worst[0] < tol, worst
        assert any('pos_proj' in k for k in g0)
    finally:
        ops._GEMM_BATCHED, ops._POS_DEFER, ops._DW_PART, ops._BN_PART, ops._DBD_PERSIST = True, True, True, True, True
        nn_mod._RES_LN = nn_mod._LN2 = nn_mod._MASK_FOLD = True
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'fp32'])
@pytest.mark.parametrize('p_drop', [0.0, 0.3])
def test_residual_layernorm_fn_against_torch(mode, p_drop):
    """ops.ResidualLnFn: (z, y) = (x + scale dropout(a), LN(z)) and its backward with gradients arriving on BOTH outputs, against torch
    autograd on the same maths; with dropout the mask is read back from z and must be the one the backward pass regenerates."""
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        M, d, scale = 333, 384, 0.5
        gen = torch.Generator().manual_seed(5)
        rnd = lambda *sh: torch.randn(*sh, generator=gen).to(DEV)      # noqa: E731
        adt = ops.act_dtype()
        x = rnd(M, d).requires_grad_(True)
        a = rnd(M, d).to(adt).requires_grad_(True)
        gamma, beta = (1 + 0.1 * rnd(d)).requires_grad_(True), (0.1 * rnd(d)).requires_grad_(True)
        gz, gy = rnd(M, d), rnd(M, d)
        z, y = ops.residual_layernorm(x, a, scale, p_drop, gamma, beta, 1e-5)
        if mode == 'fp16':
            assert ops.lp_of(y) is not None and torch.equal(ops.lp_of(y).float(), y.to(adt).float())
        ((z * gz).sum() + (y * gy).sum()).backward()
        af = a.detach().float()
        if p_drop > 0:
            m = ((z.detach() - x.detach()).abs() > 0).float()
            keep = float(m[af.abs() > 1e-3].mean())
            assert abs(keep - (1 - p_drop)) < 0.01, keep
            m = m / (1 - p_drop)
        else:
            m = torch.ones_like(af)
        xr, ar = x.detach().clone().requires_grad_(True), af.clone().requires_grad_(True)
        gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
        zr = xr + scale * m * ar
        yr = F.layer_norm(zr, (d,), gr, br, 1e-5)
        ((zr * gz).sum() + (yr * gy).sum()).backward()
        tol = dict(rtol=2e-3, atol=2e-3) if mode == 'fp16' else dict(rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(z.detach(), zr.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(y.detach(), yr.detach(), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(a.grad.float(), ar.grad, **tol)
        torch.testing.assert_close(gamma.grad, gr.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(beta.grad, br.grad, rtol=1e-3, atol=1e-3)
        # only one of the two outputs used
        x.grad = None
        z2, y2 = ops.residual_layernorm(x, a, scale, 0.0, gamma, beta, 1e-5)
        (y2 * gy).sum().backward()
        xr.grad = None
        F.layer_norm(xr + scale * ar, (d,), gr, br, 1e-5).mul(gy).sum().backward()
        torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-4)
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'fp32'])
@pytest.mark.parametrize('d,M', [(384, 333), (256, 50), (640, 77)])
def test_residual_double_layernorm_against_torch(mode, d, M):
    """ops.ResidualLnFn with a second LayerNorm, y = LN2(LN1(x + scale a)) (encoder/conformer.py:87-89 post_ffn_norm + final_norm):
    otr_add_layernorm2_fwd / _bwd against torch autograd, gradients on both outputs, all four affine gradients."""
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        scale = 1.0
        gen = torch.Generator().manual_seed(d + M)
        rnd = lambda *sh: torch.randn(*sh, generator=gen).to(DEV)      # noqa: E731
        adt = ops.act_dtype()
        x = rnd(M, d).requires_grad_(True)
        a = rnd(M, d).to(adt).requires_grad_(True)
        g1, b1 = (1 + 0.2 * rnd(d)).requires_grad_(True), (0.3 * rnd(d)).requires_grad_(True)
        g2, b2 = (1 + 0.2 * rnd(d)).requires_grad_(True), (0.3 * rnd(d)).requires_grad_(True)
        gz, gy = rnd(M, d), rnd(M, d)
        am = (torch.rand(M, generator=gen) > 0.3).to(DEV, torch.uint8)      # rows that take no branch (a_mask)
        z, y = ops.residual_layernorm(x, a, scale, 0.0, g1, b1, 1e-5, None, g2, b2, a_mask=am)
        ((z * gz).sum() + (y * gy).sum()).backward()
        ref = [t.detach().float().clone().requires_grad_(True) for t in (x, a, g1, b1, g2, b2)]
        zr = ref[0] + scale * ref[1] * am.float()[:, None]
        yr = F.layer_norm(F.layer_norm(zr, (d,), ref[2], ref[3], 1e-5), (d,), ref[4], ref[5], 1e-5)
        ((zr * gz).sum() + (yr * gy).sum()).backward()
        torch.testing.assert_close(z.detach(), zr.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(y.detach(), yr.detach(), rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(x.grad, ref[0].grad, rtol=5e-4, atol=5e-4)
        tol = dict(rtol=3e-3, atol=3e-3) if mode == 'fp16' else dict(rtol=5e-4, atol=5e-4)
        torch.testing.assert_close(a.grad.float(), ref[1].grad, **tol)
        for got, want, nm in zip((g1, b1, g2, b2), ref[2:], ('gamma', 'beta', 'gamma2', 'beta2')):
            torch.testing.assert_close(got.grad, want.grad, rtol=2e-3, atol=2e-3, msg=lambda m, nm=nm: nm + ': ' + m)
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'fp32'])
def test_relpos_attention_gradient_tensor_kept_across_steps(mode):
    """ops.RelPosAttentionFn keeps the gradient tensor of the score term across steps (zeroed once: `_persistent_dbd`); the attention
    backward rewrites every in-range band entry, so a second pass with a DIFFERENT key mask (and different data) on the same module
    gives the gradients a fresh tensor gives (ops._DBD_PERSIST off)."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.nn import relative_sinusoid
    ops.set_compute_dtype(mode)
    try:
        B, T, H, dk = 3, 70, 4, 96
        d = H * dk
        mod = ota.MultiHeadedSelfAttentionWithRelPos(H, d).to(DEV)
        syn.fill_state_dict_(mod.state_dict(), 5)
        pos = relative_sinusoid(T, d, DEV)
        gen = torch.Generator().manual_seed(3)

        def run(seed, valid, persist):
            ops._DBD_PERSIST = persist
            x = (torch.randn(B, T, d, generator=torch.Generator().manual_seed(seed)) * 0.5).to(DEV).requires_grad_(True)
            mask = torch.zeros(B, 1, T, dtype=torch.bool, device=DEV)
            for b, n in enumerate(valid):
                mask[b, 0, :n] = True
            out, _ = mod(x, mask, pos)
            g = torch.randn(B, T, d, generator=torch.Generator().manual_seed(seed + 1)).to(DEV).to(out.dtype)
            names = ['qvk_proj.weight', 'pos_proj.weight', 'posu', 'posv']
            params = dict(mod.named_parameters())
            return [t.float().clone() for t in torch.autograd.grad(out, [x] + [params[n] for n in names], g)]

        pw = mod.pos_proj.weight
        if hasattr(pw, '_otr_dbd'):
            del pw._otr_dbd
        run(11, [70, 70, 70], True)                     # fills the whole band of the kept tensor
        kept = pw._otr_dbd['buf']
        second = run(12, [70, 41, 9], True)             # most of it masked now: stale entries would show up in every gradient
        assert pw._otr_dbd['buf'] is kept               # the SAME tensor served the second pass (it lives on the layer: dies with the model)
        fresh = run(12, [70, 41, 9], False)
        for a, b in zip(second, fresh):
            assert torch.equal(a, b)
    finally:
        ops._DBD_PERSIST = True
        ops.set_compute_dtype('bf16')
