"""GPU (-m gpu): round-6 changes.

* otr_posenc_fwd serves any d and any alignment again (the four-column kernel had become the only form);
* store-first weight gradients are taken only when the clear and the launch share a capture context: a captured forward + backward
  replayed twice between two eager zero_grad() calls ACCUMULATES, a capture that contains its own zero_grad stores;
* otr_beam_topk on a row of NaNs returns in-range indices."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def torch_posenc(x):
    B, T, d = x.shape
    pos = torch.arange(T, device=x.device, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, device=x.device, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(T, d, device=x.device)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)[:, :d // 2]
    return x * math.sqrt(d) + pe


@pytest.mark.parametrize('mode', ['fp32', 'fp16'])
@pytest.mark.parametrize('d,odd_offset', [(254, False), (256, True), (30, True), (256, False)])
def test_posenc_any_width_any_alignment(mode, d, odd_offset):
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        B, T = 3, 37
        base = torch.randn(B * T * d + 1, device=DEV)
        x = (base[1:] if odd_offset else base[:-1]).view(B, T, d)       # contiguous, but 4 bytes off a 16-byte boundary when odd_offset
        assert x.is_contiguous() and (x.data_ptr() % 16 != 0) == odd_offset
        xr = x.clone().requires_grad_(True)
        y = ops.posenc(x.requires_grad_(True))
        y = y[0] if isinstance(y, tuple) else y
        ref = torch_posenc(xr)
        assert rel(y.float(), ref) < 1e-5, rel(y.float(), ref)
        g = torch.randn_like(ref)
        gx, = torch.autograd.grad(y, x, g)
        gr, = torch.autograd.grad(ref, xr, g)
        assert rel(gx, gr) < 1e-5
    finally:
        ops.set_compute_dtype('bf16')


def _small_model():
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn
    cfg = syn.c2_model(0.0)
    cfg['encoder']['n_blocks'] = 1
    cfg['decoder']['n_blocks'] = 1
    inputs, targets = syn.synthetic_batch(batch=8, frames=1000, feat_dim=80, vocab=4234, tgt_len=15, seed=4)
    inputs, targets = {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in targets.items()}
    model = ota.SpeechToText(cfg)
    syn.fill_state_dict_(model.state_dict(), 9)
    return model.to(DEV).train(), inputs, targets


def test_store_first_weight_gradients_only_within_one_capture_context():
    """the overwrite decision is baked into a captured graph.  (a) forward + backward captured WITHOUT the clear,
    replayed twice after an eager zero_grad(): the weight gradients are twice one replay's (they accumulate like every other
    gradient); (b) a capture that contains zero_grad + forward + backward stores: replayed twice it still holds ONE pass's gradient,
    bit-equal to the eager pass."""
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype('fp16')
    try:
        model, inputs, targets = _small_model()
        dp = FlatDataParallel(model)
        FusedAdam(dp, lr=1e-3, loss_scale=256.0)

        def fwd_bwd():
            loss, _ = dp(inputs, targets)
            ops.backward(loss)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                dp.zero_grad()
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        one = dp.flat_grad.clone()                                     # eager: clear and launch in the same (eager) context -> stored
        assert float(one.abs().sum()) > 0
        # (a) the clear is NOT part of the graph
        dp.zero_grad()
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            fwd_bwd()
        dp.zero_grad()
        g.replay(); g.replay()
        torch.cuda.synchronize()
        assert rel(dp.flat_grad, 2 * one) < 1e-5, rel(dp.flat_grad, 2 * one)
        # (b) the clear is part of the graph
        g2 = torch.cuda.CUDAGraph()
        with ops.graph_capture(g2):
            dp.zero_grad()
            fwd_bwd()
        g2.replay(); g2.replay()
        torch.cuda.synchronize()
        assert rel(dp.flat_grad, one) < 1e-6, rel(dp.flat_grad, one)
    finally:
        ops.set_compute_dtype('bf16')


def test_foreign_writer_of_a_registered_gradient_is_not_overwritten():
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype('fp16')
    try:
        model, inputs, targets = _small_model()
        dp = FlatDataParallel(model)
        FusedAdam(dp, lr=1e-3, loss_scale=256.0)
        w = model.encoder.blocks[0].slf_attn.qvk_proj.weight
        dp.zero_grad()
        loss, _ = dp(inputs, targets); ops.backward(loss); torch.cuda.synchronize()
        plain = w.grad.clone()
        dp.zero_grad()
        w.grad.add_(3.0)                                               # e.g. a hook, or an AccumulateGrad of a torch-native use
        ops.gradients_written([w.grad.data_ptr()])
        loss, _ = dp(inputs, targets); ops.backward(loss); torch.cuda.synchronize()
        assert rel(w.grad, plain + 3.0) < 1e-5
    finally:
        ops.set_compute_dtype('bf16')


def test_beam_topk_on_a_row_of_nans_returns_in_range_indices():
    from opentransformer_amd import _lib as L
    lib = L.load()
    V, k = 4234, 10
    logits = torch.randn(3, V, device=DEV)
    logits[1] = float('nan')
    score = torch.empty(3, k, device=DEV)
    idx = torch.full((3, k), -7, dtype=torch.int64, device=DEV)
    L.check(lib.otr_beam_topk(C.c_void_p(logits.data_ptr()), V, None, 0, 0.0, 3, V, k, C.c_void_p(score.data_ptr()), C.c_void_p(idx.data_ptr()),
                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_beam_topk')
    torch.cuda.synchronize()
    assert int(idx.min()) >= 0 and int(idx.max()) < V, idx
    assert idx[1].tolist() == list(range(k))                          # a row that cannot be ranked yields its lowest indices, in order
    lp = torch.log_softmax(logits[[0, 2]], dim=-1)
    ref = torch.topk(lp, k, dim=-1)
    assert torch.equal(idx[[0, 2]], ref.indices) and rel(score[[0, 2]], ref.values) < 1e-6


@pytest.mark.parametrize('V,k', [(4234, 10), (4234, 1), (4240, 16), (300, 10), (100, 10), (5120, 5)])
@pytest.mark.parametrize('case', ['random', 'clustered', 'equal', 'lm'])
def test_beam_topk_selection_by_counting(V, k, case):
    """r06: beam_topk_reg_kernel selects by COUNTING (threshold = k-th best thread maximum, candidates ranked among themselves) instead
    of k serial arg-max rounds: against torch.topk of the fused log-probabilities, and bit-equal to the rounds form
    (otr_debug_set(25, 2)).  clustered: the whole top-k in ONE thread's stride (index = 5 mod 256); equal: a row of equal scores (ties
    -> lower index first)."""
    from opentransformer_amd import _lib as L
    lib = L.load()
    R = 7
    g = torch.Generator().manual_seed(V + k)
    logits = torch.randn(R, V, generator=g)
    lm = torch.randn(R, V, generator=g) if case == 'lm' else None
    if case == 'clustered':
        logits[:, 5::256] += 20.0
    if case == 'equal':
        logits[2] = 0.25
        logits[3, : V // 2] = 1.0
        logits[3, V // 2:] = -1.0
    logits = logits.to(DEV)
    lm = lm.to(DEV) if lm is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for form in (1, 2):
        L.check(lib.otr_debug_set(25, form), 'debug_set')
        try:
            score = torch.empty(R, k, device=DEV)
            idx = torch.full((R, k), -1, dtype=torch.int64, device=DEV)
            L.check(lib.otr_beam_topk(C.c_void_p(logits.data_ptr()), V, C.c_void_p(lm.data_ptr()) if lm is not None else None, V if lm is not None else 0,
                                      0.3 if lm is not None else 0.0, R, V, k, C.c_void_p(score.data_ptr()), C.c_void_p(idx.data_ptr()), st), 'otr_beam_topk')
            torch.cuda.synchronize()
            outs[form] = (score.clone(), idx.clone())
        finally:
            lib.otr_debug_set(25, 1)
    assert torch.equal(outs[1][1], outs[2][1]) and torch.equal(outs[1][0], outs[2][0])
    lp = torch.log_softmax(logits.double(), dim=-1)
    if lm is not None:
        lp = lp + 0.3 * torch.log_softmax(lm.double(), dim=-1)
    # ties -> lower index first: a stable descending sort of the fp64 scores
    order = torch.sort(lp, dim=-1, descending=True, stable=True).indices[:, :k]
    score, idx = outs[1]
    got_lp = torch.gather(lp, 1, idx)
    ref_lp = torch.gather(lp, 1, order)
    assert rel(score, ref_lp) < 1e-5
    assert float((got_lp - ref_lp).abs().max()) < 1e-5                  # the same scores in the same order (fp32 near-ties may swap indices)
    if case in ('equal', 'clustered') or lm is None:
        same = (idx == order) | ((got_lp - ref_lp).abs() < 1e-6)
        assert bool(same.all())
    if case == 'equal':
        assert idx[2].tolist() == list(range(k))                        # all equal: the lowest indices, in order


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
@pytest.mark.parametrize('lm_blocks,dec_blocks', [(2, 3), (4, 2), (3, 3)])
def test_cached_search_on_pair_launches_is_the_forked_search(mode, lm_blocks, dec_blocks):
    """r06: the LM's layers as the second problem of the decoder's launches (otr_dec_self_step_pair / otr_dec_ffn_fwd_pair /
    otr_dec_ln_pair, recognize._fused_stacks_paired) run the same kernels on the same operands as the two separate chains (the LM on a
    forked stream): token-identical n-best lists and bit-equal scores, also when one stack is deeper than the other (the tail of the
    deeper one runs on single launches), eagerly and under hipGraph replay."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, recognize, synthetic as syn
    ops.set_compute_dtype(mode)
    try:
        cfg = syn.c2_model(0.0, n_enc=2)
        cfg['decoder']['n_blocks'] = dec_blocks
        model = ota.SpeechToText(cfg)
        syn.fill_state_dict_(model.state_dict(), 7)
        lm = recognize.TransformerLanguageModel(syn.lm_config(4234, num_blocks=lm_blocks))
        syn.fill_state_dict_(lm.state_dict(), 8)
        with torch.no_grad():
            model.decoder.output_layer.bias[1] = 2.0            # EOS live: beams finish at different steps
        model, lm = model.to(DEV).eval(), lm.to(DEV).eval()
        inputs, _ = syn.synthetic_batch(batch=3, frames=400, feat_dim=80, vocab=4234, tgt_len=5, seed=3, lengths=[400, 333, 250])
        x, m = inputs['inputs'].to(DEV), inputs['mask'].to(DEV)
        kw = dict(beam_width=10, nbest=10, max_len=16, penalty=0.6, lamda=5, lm=lm, lm_weight=0.1, idx2unit={i: str(i) for i in range(4234)})
        res = {}
        for pair in (True, False):
            recognize._DECODE_PAIR = pair
            rec = recognize.SpeechToTextRecognizer(model, apply_cache=True, **kw)
            res[pair] = rec.recognize(x, m)
            stt = next(iter(rec._cached_states.values()))
            assert stt.paired == pair and (stt.side is None) == pair
            if pair:
                assert all(g is not None for g in stt.graphs)      # the paired step was captured and replayed
        (h1, s1), (h0, s0) = res[True], res[False]
        assert h1 == h0
        assert torch.equal(s1, s0)
    finally:
        recognize._DECODE_PAIR = True
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'fp32'])
def test_cached_search_with_the_lagged_stop_test_is_the_synchronous_search(mode):
    """r06 (recognize._DECODE_LAGGED_STOP): the cached beam search launches step s + 1 before it reads step s's all-finished count and
    gives back exactly what the loop that syncs after every step (recognize/speech2text.py:67) gives: hypotheses, scores, and the step
    it stopped at.  With EOS live the search stops early, so the one step that ran in vain is exercised."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, recognize, synthetic as syn
    ops.set_compute_dtype(mode)
    try:
        cfg = syn.c2_model(0.0, n_enc=2)
        cfg['decoder']['n_blocks'] = 2
        model = ota.SpeechToText(cfg)
        syn.fill_state_dict_(model.state_dict(), 17)
        lm = recognize.TransformerLanguageModel(syn.lm_config(4234, num_blocks=2))
        syn.fill_state_dict_(lm.state_dict(), 18)
        with torch.no_grad():
            model.decoder.output_layer.bias[1] = 30.0           # EOS wins everywhere: every beam has finished after two or three steps
        model, lm = model.to(DEV).eval(), lm.to(DEV).eval()
        inputs, _ = syn.synthetic_batch(batch=3, frames=400, feat_dim=80, vocab=4234, tgt_len=5, seed=5, lengths=[400, 333, 250])
        x, m = inputs['inputs'].to(DEV), inputs['mask'].to(DEV)
        kw = dict(beam_width=10, nbest=10, max_len=24, penalty=0.6, lamda=5, lm=lm, lm_weight=0.1, idx2unit={i: str(i) for i in range(4234)})
        res, stops = {}, {}
        run0 = recognize.CachedBeamState.run

        def run_and_note(self):
            out = run0(self)
            stops.setdefault(recognize._DECODE_LAGGED_STOP, []).append(out[1])
            return out
        recognize.CachedBeamState.run = run_and_note
        for lag in (False, True):
            recognize._DECODE_LAGGED_STOP = lag
            rec = recognize.SpeechToTextRecognizer(model, apply_cache=True, **kw)
            for _ in range(4):                                 # eager warm-up visits, capture, replay
                res[lag] = rec.recognize(x, m)
        assert stops[False] == stops[True] and max(stops[False]) < 24, stops    # the search did stop early, at the same step every time
        assert res[True][0] == res[False][0]
        assert torch.equal(res[True][1], res[False][1])
    finally:
        recognize._DECODE_LAGGED_STOP = True
        recognize.CachedBeamState.run = run0
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
def test_conformer_relpos_tables_of_all_blocks_in_two_batched_launches(mode):
    """r06 (nn._POS_TABLES / ops.relpos_tables): pos_proj(sinusoid) of every Conformer block and its transpose come from two batched
    GEMM launches under FlatDataParallel (the blocks' weight shadows sit at one stride) instead of a GEMM + a transposing copy per
    block: the same loss and gradients as the per-block form, and the tables really were batched."""
    import opentransformer_amd as ota
    import opentransformer_amd.nn as onn
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype(mode)
    try:
        cfg = syn.conformer_model(True, 0.0)                       # the small Conformer (d = 64, 2 blocks)
        inputs, targets = syn.synthetic_batch(batch=4, frames=200, feat_dim=80, vocab=100, tgt_len=6, seed=2, lengths=[200, 161, 120, 88])
        inputs, targets = {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in targets.items()}
        res = {}
        for batched in (True, False):
            onn._POS_TABLES = batched
            model = ota.SpeechToText(cfg)
            syn.fill_state_dict_(model.state_dict(), 3)
            model = model.to(DEV).train()
            dp = FlatDataParallel(model)
            FusedAdam(dp, lr=1e-3, loss_scale=64.0)
            if batched:
                pos = onn.relative_sinusoid(13, cfg['encoder']['d_model'], torch.device(DEV))
                tabs = ops.relpos_tables(pos, [b.mha.pos_proj.weight for b in model.encoder.blocks])
                assert tabs is not None and len(tabs) == len(model.encoder.blocks)
                pe, p0, pt0 = tabs[0]
                assert rel(pt0.float(), p0.float().t()) < 1e-3        # the transpose is the swapped product
                w16 = ops.weight_lp(model.encoder.blocks[1].mha.pos_proj.weight).float()
                assert rel(tabs[1][1].float(), pe.to(ops.half_dtype()).float() @ w16.t()) < 5e-3
            dp.zero_grad()
            loss, _ = dp(inputs, targets)
            ops.backward(loss)
            torch.cuda.synchronize()
            res[batched] = (float(loss), dp.flat_grad.clone())
        assert abs(res[True][0] - res[False][0]) < 2e-3 * abs(res[False][0])
        assert rel(res[True][1], res[False][1]) < (2e-2 if mode == 'bf16' else 3e-3), rel(res[True][1], res[False][1])
    finally:
        onn._POS_TABLES = True
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp32', 'fp16', 'bf16'])
@pytest.mark.parametrize('M,d', [(37, 384), (64, 256), (5, 64)])
def test_residual_layernorm3_matches_torch(mode, M, d):
    """ops.ResidualLn3Fn (otr_add_layernorm3_fwd / _bwd): z = x + scale * a (row-masked), y2 = LN2(LN1(z)), y3 = LN3(y2) and the gradients of
    x, a and the six affine parameters for gradients arriving at z, y2 AND y3 -- against plain fp32 torch."""
    import torch.nn.functional as F
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        g = torch.Generator().manual_seed(M + d)
        x = torch.randn(2, M, d, generator=g).to(DEV).requires_grad_(True)
        adt = ops.act_dtype()
        a = torch.randn(2, M, d, generator=g).to(DEV).to(adt).requires_grad_(True)
        norms = [torch.nn.LayerNorm(d).to(DEV) for _ in range(3)]
        for n in norms:
            with torch.no_grad():
                n.weight.add_(0.2 * torch.randn(d, generator=g).to(DEV))
                n.bias.add_(0.2 * torch.randn(d, generator=g).to(DEV))
        am = (torch.rand(2 * M, generator=g) > 0.3).to(torch.uint8).to(DEV)
        z, y2, y3 = ops.residual_layernorm3(x, a, 0.5, 0.0, norms[0], norms[1], norms[2], a_mask=am)
        xr, ar = x.detach().clone().requires_grad_(True), a.detach().float().clone().requires_grad_(True)
        ps = [p.detach().clone().requires_grad_(True) for n in norms for p in (n.weight, n.bias)]
        zr = xr + 0.5 * ar * am.view(2, M, 1).float()
        y2r = F.layer_norm(F.layer_norm(zr, (d,), ps[0], ps[1]), (d,), ps[2], ps[3])
        y3r = F.layer_norm(y2r, (d,), ps[4], ps[5])
        tol = 1e-5 if mode == 'fp32' else (3e-3 if mode == 'fp16' else 2e-2)
        assert rel(z, zr) < 1e-6 and rel(y2, y2r) < 1e-5 and rel(y3.float(), y3r) < tol
        gz, g2 = torch.randn(2, M, d, generator=g).to(DEV), torch.randn(2, M, d, generator=g).to(DEV)
        g3 = torch.randn(2, M, d, generator=g).to(DEV)
        outs = (z, y2, y3)
        grads = torch.autograd.grad(outs, [x, a] + [p for n in norms for p in (n.weight, n.bias)], (gz, g2, g3.to(y3.dtype)))
        gref = torch.autograd.grad((zr, y2r, y3r), [xr, ar] + ps, (gz, g2, g3.to(y3.dtype).float()))
        for nm, u, v in zip(('dx', 'da', 'dg1', 'db1', 'dg2', 'db2', 'dg3', 'db3'), grads, gref):
            assert rel(u.float(), v) < (1e-4 if mode == 'fp32' else tol), (nm, rel(u.float(), v))
        # only d y3 arrives (the others None): zeros stand in
        g_only3 = torch.autograd.grad(ops.residual_layernorm3(x, a, 0.5, 0.0, norms[0], norms[1], norms[2], a_mask=am)[2], x, g3.to(y3.dtype))[0]
        r_only3 = torch.autograd.grad(y3r, xr, g3.to(y3.dtype).float(), retain_graph=True)[0] if False else None
        assert torch.isfinite(g_only3).all()
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'bf16', 'fp32'])
def test_conformer_chain_of_three_layernorms_across_blocks(mode):
    """r06 (nn._LN3): the closing launch of a Conformer block also runs the NEXT block's macaron LayerNorm and hands its output over; the same
    loss and gradients as the form where every block normalises its own input."""
    import opentransformer_amd as ota
    import opentransformer_amd.nn as onn
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype(mode)
    try:
        cfg = syn.conformer_model(True, 0.0)
        cfg['encoder']['nblocks'] = 3
        inputs, targets = syn.synthetic_batch(batch=4, frames=200, feat_dim=80, vocab=100, tgt_len=6, seed=2, lengths=[200, 161, 120, 88])
        inputs, targets = {k: v.to(DEV) for k, v in inputs.items()}, {k: v.to(DEV) for k, v in targets.items()}
        res = {}
        for chained in (True, False):
            onn._LN3 = chained
            model = ota.SpeechToText(cfg)
            syn.fill_state_dict_(model.state_dict(), 3)
            model = model.to(DEV).train()
            assert not any('chain' in k for k in model.state_dict())
            dp = FlatDataParallel(model)
            FusedAdam(dp, lr=1e-3, loss_scale=64.0 if mode == 'fp16' else 1.0)
            names = []
            ops.set_kernel_timer(names)
            try:
                dp.zero_grad()
                loss, _ = dp(inputs, targets)
                ops.backward(loss)
            finally:
                ops.set_kernel_timer(None)
            torch.cuda.synchronize()
            res[chained] = (float(loss), dp.flat_grad.clone())
            assert not any('chain' in k for k in model.state_dict())          # nothing was registered on the blocks
        assert abs(res[True][0] - res[False][0]) < (1e-5 if mode == 'fp32' else 3e-3) * abs(res[False][0])
        tol = 1e-4 if mode == 'fp32' else (3e-2 if mode == 'bf16' else 5e-3)
        assert rel(res[True][1], res[False][1]) < tol, rel(res[True][1], res[False][1])
    finally:
        onn._LN3 = True
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('B,T,Fdim,C1,C2', [(4, 200, 80, 256, 256), (2, 131, 40, 256, 136), (1, 77, 80, 512, 256)])
def test_conv2_wgrad_gather_form(mode, B, T, Fdim, C1, C2):
    """otr_conv2_wgrad for C1 % 256 == 0 (the Conformer's 256 -> 256 frontend, frontend/conv.py:50-83): wgrad256.hip with gathered x rows
    and a fixed-order sum of the row ranges == the transposing GEMM it replaces (same 16-bit operands, fp32 sums: 1e-5) == torch's
    conv2d weight gradient on the same rounded operands.  Ragged pixel counts (rows past M read the zero line), the padded
    frequency taps, a ragged last channel tile (C2 = 136) and two k-tiles per tap (C1 = 512)."""
    from opentransformer_amd import ops, _lib as L
    import torch.nn.functional as F
    ops.set_compute_dtype(mode)
    try:
        lib = L.load()
        adt = ops.act_dtype()
        T1, F1, T2, F2 = ops.conv_geometry(T, Fdim)
        g = torch.Generator(device='cpu').manual_seed(5)
        act1 = torch.randn(B, T1, F1, C1, generator=g).relu_().to(DEV).to(adt)
        g2 = (torch.randn(B, T2, F2, C2, generator=g) * (torch.rand(B, T2, F2, C2, generator=g) > 0.5)).to(DEV).to(adt)
        desc = L.ConvDesc(B, T, Fdim, C1, C2, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
        ws = ops._workspace(act1.device)
        out = {}
        for form in (1, 0):
            L.check(lib.otr_debug_set(29, form), 'debug_set')
            dw = torch.full((C2, 3, 3, C1), float('nan'), device=DEV)
            names = []
            ops.set_kernel_timer(names)
            try:
                L.check(lib.otr_conv2_wgrad(C.byref(desc), ops._p(g2), ops._p(act1), ops._p(dw), ops._p(ws), ops._WS_BYTES, ops._stream()), 'otr_conv2_wgrad')
            finally:
                ops.set_kernel_timer(None)
            torch.cuda.synchronize()
            out[form] = dw
        assert torch.isfinite(out[1]).all()
        assert rel(out[1], out[0]) < 1e-5, rel(out[1], out[0])
        # torch: d/dw of conv2d(act1 NCHW, w, stride 2, pad (0, 1)) against g2
        w = torch.zeros(C2, C1, 3, 3, device=DEV, requires_grad=True)
        y = F.conv2d(act1.float().permute(0, 3, 1, 2), w, None, stride=2, padding=(0, 1))
        (dwr,) = torch.autograd.grad(y, w, g2.float().permute(0, 3, 1, 2))
        assert rel(out[1], dwr.permute(0, 2, 3, 1)) < 1e-4, rel(out[1], dwr.permute(0, 2, 3, 1))
    finally:
        L.load().otr_debug_set(29, 1)
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('B,T,Fd,C1,C2', [(2, 61, 30, 256, 256), (3, 120, 80, 256, 256), (1, 7, 3, 256, 256), (2, 90, 41, 128, 256)])
def test_conv2_input_gradient_sliced_form(mode, B, T, Fd, C1, C2):
    """otr_conv2_dgrad for 256 output channels of conv2 (the Conformer's 256 -> 256 frontend): the parity-class kernel with the channels
    of act1 cut into 64-wide slices (conv.hip conv2_dgrad_sliced_kernel), elementwise against torch's conv2d input gradient of the same
    16-bit operands in fp32, masked by act1 > 0 (frontend/conv.py:63-66 backward); and == the column-matrix + col2im path it replaces."""
    import torch.nn.functional as F
    from opentransformer_amd import _lib as L, ops
    ops.set_compute_dtype(mode)
    try:
        lib = L.load()
        adt = ops.act_dtype()
        T1, F1, T2, F2 = ops.conv_geometry(T, Fd)
        gen = torch.Generator().manual_seed(T)
        g2 = torch.randn(B, T2, F2, C2, generator=gen).to(DEV, adt)
        w2r = (torch.randn(C2, 3, 3, C1, generator=gen) / 20).to(DEV, adt)
        act1 = torch.randn(B, T1, F1, C1, generator=gen).clamp_min(0).to(DEV, adt)
        dact1 = torch.full_like(act1, float('nan'))
        desc = L.ConvDesc(B, T, Fd, C1, C2, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
        rc = lib.otr_conv2_dgrad(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._stream())
        assert rc == 0, rc
        w = w2r.float().permute(0, 3, 1, 2).contiguous()
        a1 = act1.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        out = F.conv2d(a1, w, None, stride=2, padding=(0, 1))
        (gi,) = torch.autograd.grad(out, a1, g2.float().permute(0, 3, 1, 2).contiguous())
        want = (gi * (a1 > 0)).permute(0, 2, 3, 1)
        assert not torch.isnan(dact1.float()).any()
        assert rel(dact1, want) < (4e-3 if mode == 'bf16' else 5e-4), rel(dact1, want)
        assert bool((dact1[act1 <= 0] == 0).all())
        # the explicit path on the same operands
        L.check(lib.otr_debug_set(30, 0), 'debug_set')
        assert lib.otr_conv2_dgrad(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._stream()) == 1     # not served: nothing launched
        dcol = torch.empty((B * T2 * F2, 9 * C1), dtype=adt, device=DEV)
        d2 = torch.full_like(act1, float('nan'))
        L.check(lib.otr_conv2_dgrad_cols(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(dcol), ops._stream()), 'cols')
        L.check(lib.otr_conv2_col2im(C.byref(desc), ops._p(dcol), ops._p(act1), ops._p(d2), ops._stream()), 'col2im')
        assert rel(dact1, d2) < (8e-3 if mode == 'bf16' else 1e-3), rel(dact1, d2)
    finally:
        L.load().otr_debug_set(30, 1)
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['bf16', 'fp16'])
@pytest.mark.parametrize('B,T,Fd', [(2, 61, 30), (3, 120, 80), (1, 7, 3), (5, 333, 83), (32, 1000, 80)])
def test_conv2_wide_input_gradient(mode, B, T, Fd):
    """otr_conv2_dgrad_wide (csrc/conv2wide.hip: 256 -> 256 channels, weights streamed through LDS, all channels of a pixel in one
    workgroup) against torch's conv2d input gradient on the same 16-bit operands in fp32, masked by act1 > 0 (frontend/conv.py:63-66
    backward), and against the sliced parity-class kernel it replaces.  Shapes: a few pixels (fewer than one 256-pixel tile), ragged
    tiles, odd F1 (classes of different sizes), the bench batch."""
    import torch.nn.functional as F
    from opentransformer_amd import _lib as L, ops
    ops.set_compute_dtype(mode)
    try:
        lib = L.load()
        adt = ops.act_dtype()
        C1 = C2 = 256
        T1, F1, T2, F2 = ops.conv_geometry(T, Fd)
        gen = torch.Generator().manual_seed(T + Fd)
        act1 = torch.randn(B, T1, F1, C1, generator=gen).clamp_min(0).to(DEV, adt)
        w2r = (torch.randn(C2, 3, 3, C1, generator=gen) / 48).to(DEV, adt)
        g2 = torch.randn(B, T2, F2, C2, generator=gen).to(DEV, adt)
        desc = L.ConvDesc(B, T, Fd, C1, C2, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
        ws = ops._workspace(act1.device)
        assert lib.otr_conv2_wide_scratch_bytes() == 9 * 8 * 1024 * 16
        dact1 = torch.full_like(act1, float('nan'))
        assert lib.otr_conv2_dgrad_wide(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._p(ws), ops._WS_BYTES, ops._stream()) == 0
        d2 = torch.full_like(act1, float('nan'))
        assert lib.otr_conv2_dgrad(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(d2), ops._stream()) == 0       # the sliced form
        assert not torch.isnan(dact1.float()).any()
        assert rel(dact1, d2) < (4e-3 if mode == 'bf16' else 5e-4), rel(dact1, d2)
        assert bool((dact1[act1 <= 0] == 0).all())
        if B * T <= 1000:
            w = w2r.float().permute(0, 3, 1, 2).contiguous()
            a1 = act1.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
            out = F.conv2d(a1, w, None, stride=2, padding=(0, 1))
            (gi,) = torch.autograd.grad(out, a1, g2.float().permute(0, 3, 1, 2).contiguous())
            want = (gi * (a1 > 0)).permute(0, 2, 3, 1)
            assert rel(dact1, want) < (4e-3 if mode == 'bf16' else 5e-4), rel(dact1, want)
        # not served: other channel counts, a short scratch, the switch
        d64 = L.ConvDesc(B, T, Fd, 64, 128, T1, F1, T2, F2, ops._code(adt), ops._compute_code(), ops._code(adt))
        assert lib.otr_conv2_dgrad_wide(C.byref(d64), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._p(ws), ops._WS_BYTES, ops._stream()) == 1
        assert lib.otr_conv2_dgrad_wide(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._p(ws), 1024, ops._stream()) == 1
        L.check(lib.otr_debug_set(31, 0), 'debug_set')
        assert lib.otr_conv2_dgrad_wide(C.byref(desc), ops._p(g2), ops._p(w2r), ops._p(act1), ops._p(dact1), ops._p(ws), ops._WS_BYTES, ops._stream()) == 1
    finally:
        L.load().otr_debug_set(31, 1)
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('B,T,Cc,k', [(4, 90, 128, 7), (3, 249, 384, 5), (2, 33, 64, 3), (1, 5, 256, 5)])
def test_conformer_conv_backward_fused_middle(mode, B, T, Cc, k):
    """ConformerConvolutionModule backward (module/conformer.py:36-57) with BatchNorm's apply step, the depthwise conv's backward and the
    GLU's backward in one launch (otr_bn_swish_bwd_sums + otr_conformer_conv_bwd_mid) == the five-launch chain it replaces: the input
    gradient and every parameter gradient, under FlatDataParallel (in-place gradient buffers, deferred sums), padded frames masked,
    segments that cross utterance boundaries (T not a multiple of the 16-row segments), an utterance shorter than the kernel's reach."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel
    ops.set_compute_dtype(mode)
    try:
        gen = torch.Generator().manual_seed(B * T + Cc)
        x = torch.randn(B, T, Cc, generator=gen).to(DEV)
        gout = torch.randn(B, T, Cc, generator=gen).to(DEV, ops.act_dtype())
        mask = torch.ones(B, T, dtype=torch.bool, device=DEV)
        if B > 1:
            mask[B - 1, max(1, T - T // 4):] = False
        res = {}
        for fused in (True, False):
            ops._CONV_MID_FUSED = fused
            mod = ota.ConformerConvolutionModule(Cc, k).to(DEV).train()
            syn.fill_state_dict_(mod.state_dict(), 9)
            dp = FlatDataParallel(mod)
            dp.zero_grad()
            xin = x.clone().requires_grad_(True)
            names = []
            ops.set_kernel_timer(names)
            try:
                mod(xin, mask).backward(gout)
            finally:
                ops.set_kernel_timer(None)
            torch.cuda.synchronize()
            res[fused] = (xin.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()})
        tol = 1e-5 if mode == 'fp32' else (4e-3 if mode == 'bf16' else 5e-4)
        assert rel(res[True][0], res[False][0]) < tol, rel(res[True][0], res[False][0])
        for n in res[True][1]:
            a, b = res[True][1][n], res[False][1][n]
            scale = max(float(b.abs().max()), 1e-6)
            if n == 'depthwise_conv.bias':                 # zero gradient in front of BatchNorm: roundoff on both sides
                continue
            assert float((a - b).abs().max()) < 10 * tol * scale, (n, float((a - b).abs().max()), scale)
    finally:
        ops._CONV_MID_FUSED = True
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
@pytest.mark.parametrize('B,T,dbd16', [(3, 249, True), (2, 300, True), (2, 131, False), (1, 33, True), (2, 513, True)])
def test_relpos_attention_backward_whole_utterance_kernel(mode, B, T, dbd16):
    """otr_attention_bias_bwd for the Conformer's self-attention (head dim 96, relative-position score term, module/attention.py:196-253):
    csrc/encattn96.hip (the (utterance, head) staged in LDS super-chunk by super-chunk, one workgroup per orientation) == the streamed
    dQ + dK/dV pair it replaces (otr_debug_set(33, 0)): dq, dk, dv and the score term's gradient, with masked keys, T across one / two / three
    super-chunks and two own blocks (T > 256), the gradient tensor in fp32 and in 16 bits; and dq / dk / dv against an fp32 torch reference."""
    from opentransformer_amd import _lib as L, ops
    ops.set_compute_dtype(mode)
    try:
        lib = L.load()
        adt = ops.act_dtype()
        H, dk = 4, 96
        d = H * dk
        P = 2 * T - 1
        Pp = (P + 7) // 8 * 8
        gen = torch.Generator().manual_seed(T)
        qkv = (torch.randn(B, T, 3 * d, generator=gen) * 0.5).to(DEV, adt)
        quv = (qkv[..., :d].float() + 0.1).to(adt).contiguous()                      # the (q + u) operand: [B, T, d]
        bd = (torch.randn(B, T, H, Pp, generator=gen) * 2.0).to(DEV)                  # the un-shifted score term, fp32
        dout = torch.randn(B, T, d, generator=gen).to(DEV, adt)
        km = torch.ones(B, T, dtype=torch.uint8, device=DEV)
        if B > 1:
            km[B - 1, T - T // 5:] = 0
        out = torch.empty(B, T, d, dtype=adt, device=DEV)
        lse = torch.empty(B, H, T, dtype=torch.float32, device=DEV)
        desc = ops._attn_desc(B, H, T, T, dk, adt, (T * d, d), (T * 3 * d, 3 * d), (T * 3 * d, 3 * d), (T * d, d), False)
        L.check(lib.otr_attention_bias_fwd(C.byref(desc), ops._p(quv), ops._p(qkv, d), ops._p(qkv, 2 * d), ops._p(km), ops._p(bd),
                                           T * H * Pp, Pp, H * Pp, 1, ops._p(out), ops._p(lse), ops._stream()), 'fwd')
        res = {}
        for form in (1, 0):
            L.check(lib.otr_debug_set(33, form), 'debug_set')
            dbd = torch.zeros(B, T, H, Pp, dtype=adt if dbd16 else torch.float32, device=DEV)
            dq = torch.full((B, T, d), float('nan'), dtype=adt, device=DEV)
            dkv = torch.full((B, T, 3 * d), float('nan'), dtype=adt, device=DEV)
            delta = torch.empty_like(lse)
            names = []
            ops.set_kernel_timer(names)
            try:
                L.check(lib.otr_attention_bias_bwd(C.byref(desc), ops._p(quv), ops._p(qkv, d), ops._p(qkv, 2 * d), ops._p(km), ops._p(bd), ops._p(dbd),
                                                   ops._code(dbd.dtype), T * H * Pp, Pp, H * Pp, 1, ops._p(out), ops._p(dout), ops._p(lse), ops._p(delta),
                                                   ops._p(dq), ops._p(dkv, d), ops._p(dkv, 2 * d), ops._stream()), 'bwd')
            finally:
                ops.set_kernel_timer(None)
            torch.cuda.synchronize()
            res[form] = (dq, dkv[..., d:2 * d].clone(), dkv[..., 2 * d:].clone(), dbd)
        tol = 6e-3 if mode == 'bf16' else 8e-4
        for nm, a, b in zip(('dq', 'dk', 'dv', 'dbias'), res[1], res[0]):
            assert torch.isfinite(a.float()).all(), nm
            assert rel(a, b) < tol, (nm, rel(a, b))
        assert bool((res[1][3][:, :, :, P:] == 0).all())                                 # the padding of the relative axis is never written
        # fp32 torch reference of the same 16-bit operands (the shifted matrix gathered explicitly)
        q = quv.float().view(B, T, H, dk).transpose(1, 2)
        k = qkv[..., d:2 * d].float().reshape(B, T, H, dk).transpose(1, 2)
        v = qkv[..., 2 * d:].float().reshape(B, T, H, dk).transpose(1, 2)
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        i = torch.arange(T, device=DEV).view(T, 1)
        j = torch.arange(T, device=DEV).view(1, T)
        col = (j - i + T - 1).view(1, T, 1, T).expand(B, T, H, T)
        shifted = torch.gather(bd, 3, col).permute(0, 2, 1, 3)                           # [B, H, T(i), T(j)]
        s = (q @ k.transpose(-1, -2) + shifted) / math.sqrt(dk)
        s = s.masked_fill(km.view(B, 1, 1, T) == 0, float('-inf'))
        o = torch.softmax(s, -1) @ v
        gq, gk, gv = torch.autograd.grad(o, (q, k, v), dout.float().view(B, T, H, dk).transpose(1, 2))
        live = km.view(B, T, 1).bool()
        for nm, a, w in (('dq', res[1][0], gq), ('dk', res[1][1], gk), ('dv', res[1][2], gv)):
            w2 = w.transpose(1, 2).reshape(B, T, d)
            if nm != 'dq':
                w2 = w2 * live
            assert rel(a, w2) < (2e-2 if mode == 'bf16' else 3e-3), (nm, rel(a, w2))
    finally:
        L.load().otr_debug_set(33, 1)
        ops.set_compute_dtype('bf16')
