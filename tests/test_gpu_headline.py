"""GPU (-m gpu): parity at the QUOTED configurations.

 * BASELINE.json configs[1] exactly as bench.py times it -- transformer_baseline.yaml (+input_size 80), B = 32 utterances
   x 1000 frames, 15 decoder rows, V = 4234, dropout off -- loss, logits, encoder memory and EVERY parameter gradient
   against the CPU oracle (oracle/otrans_oracle.py, pinned to the reference by tests/test_oracle_golden.py) in all three
   compute modes.  North-star bar: 1e-3 relative on loss and logits; bench.py's mode (fp16) must meet it.
 * configs[4] (C5) at full size -- 12+6 layers, beam 10, 4-block TransformerLM shallow fusion, V = 4234 -- hypotheses and
   scores of the KV-cached hipGraph decoder against the ORACLE's beam search (recognize/speech2text.py:39-192 restated),
   not against the product's own re-forward loop.  In the 16-bit modes rounding may flip a near-tie inside the search, so
   the product's per-step beams (SpeechToTextRecognizer.trace) are checked to be a VALID beam search under the oracle's
   own scores: at every step the kept candidates are a top-`beam` set of the ORACLE's candidate scores up to twice the
   measured score drift, and the product's cumulative scores sit within the drift of the oracle's.
 * (r03) configs[3] (C4, conformer_baseline.yaml) at the bench batch -- B = 32 x 1000 frames, ragged lengths -- and
   configs[1] with the joint CTC term (ctc_weight 0.3) at the same size: loss (+ the CTC term), logits, encoder memory and
   every parameter gradient against the oracle in all three modes."""
import json
import os

import numpy as np
import pytest
import torch

from opentransformer_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def c2_oracle():
    """oracle forward + backward on the bench batch (a few seconds of CPU)"""
    from oracle import otrans_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = syn.c2_model(0.0)
    inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=0)
    parts = H.require_grad(H.filled_state(cfg))
    loss, aux = orc.speech2text_forward(parts, cfg, inputs, targets)
    loss.backward()
    return cfg, inputs, targets, loss.detach(), {k: v.detach() for k, v in aux.items()}, H.flat_named(parts)


# (loss, logits / memory, worst parameter gradient): <= 2x the drift measured on MI355X (profiles/r02_parity_headline.json)
# measured r02: fp32 (0, 4.7e-7, 2.7e-5); fp16 (1.2e-6, 5.4e-4, 3.9e-3); bf16 (1.1e-4, 4.2e-3, 2.7e-2, worst = a decoder q_proj weight)
HEADLINE_TOL = {'fp32': (1e-5, 5e-6, 1e-4), 'fp16': (1e-4, 1e-3, 8e-3), 'bf16': (1e-3, 8e-3, 5e-2)}


def _oracle_run(cfg, inputs, targets, seed=1234):
    from oracle import otrans_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    parts = H.require_grad(H.filled_state(cfg, seed=seed))
    loss, aux = orc.speech2text_forward(parts, cfg, inputs, targets)
    loss.backward()
    return loss.detach(), {k: v.detach() for k, v in aux.items()}, H.flat_named(parts)


def _compare_with_oracle(cfg, inputs, targets, ref_loss, ref_aux, ref_flat, mode, tag, label, tol, seed=1234):
    """product forward + backward on the GPU in `mode`; relative errors of loss (+ CTC term), logits, encoder memory and every
    parameter gradient against the oracle's; written to gpurun_out/parity_<tag>_<mode>.json and asserted against tol =
    (loss, logits / memory, worst gradient)"""
    import opentransformer_amd as ota
    from opentransformer_amd import ops
    ops.set_compute_dtype(mode)
    try:
        model = ota.SpeechToText(cfg)
        syn.fill_state_dict_(model.state_dict(), seed)
        model = model.to(DEV).train()
        di = {k: v.to(DEV) for k, v in inputs.items()}
        dt = {k: v.to(DEV) for k, v in targets.items()}
        with H.loss_scaled(mode) as ls:
            fe, fm = model.frontend(di['inputs'], di['mask'])
            memory, mm, _ = model.encoder(fe, fm)
            logits, _ = model.decoder(dt['targets'][:, :-1].contiguous(), memory, mm)
            loss, aux = model(di, dt)
            loss.backward()
            ls.unscale(model)
        r = {'config': label, 'mode': mode,
             'loss_rel': abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()),
             'logits_rel': rel(logits.detach(), ref_aux['logits']), 'memory_rel': rel(memory.detach(), ref_aux['memory'])}
        if 'ctc_loss' in ref_aux:
            r['ctc_loss_rel'] = abs(float(aux['CTCLoss']) - float(ref_aux['ctc_loss'])) / abs(float(ref_aux['ctc_loss']))
        worst, wkey, rels = 0.0, None, {}
        n_el = sum(v.grad.numel() for v in ref_flat.values() if v.grad is not None)
        g_rms = (sum(float(v.grad.double().pow(2).sum()) for v in ref_flat.values() if v.grad is not None) / n_el) ** 0.5
        zero_worst = 0.0
        for k, p in model.named_parameters():
            if p.grad is None:
                assert ref_flat[k].grad is None or float(ref_flat[k].grad.abs().max()) == 0.0, k
                continue
            g_ref = ref_flat[k].grad
            if k.endswith('conv.depthwise_conv.bias'):
                # ANALYTICALLY ZERO: a bias in front of a BatchNorm on batch statistics (module/conformer.py:103-110) shifts the
                # mean the norm subtracts.  The oracle itself holds rounding noise there (4e-5 of the model's RMS gradient element
                # in fp32), so there is nothing to be relative to: both sides must be ~0 on the scale of a typical gradient element
                zero_worst = max(zero_worst, float(p.grad.double().pow(2).mean().sqrt()) / g_rms)
                assert float(g_ref.double().pow(2).mean().sqrt()) < 1e-3 * g_rms, k
                continue
            e = rel(p.grad, g_ref)
            rels[k] = e
            if e > worst:
                worst, wkey = e, k
        r['zero_grad_rms_over_model_rms'] = zero_worst
        r['grad_worst'], r['grad_worst_key'] = worst, wkey
        r['grad_median'] = float(np.median(list(rels.values())))
        r['grad_conv1_weight'] = rels.get('frontend.conv1.conv_layer.weight')
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_%s_%s.json' % (tag, mode)), 'w') as f:
            json.dump(r, f, indent=1)
        print(json.dumps(r))
        tl, ta, tm, tg = tol if len(tol) == 4 else (tol[0], tol[1], tol[1], tol[2])
        assert r['loss_rel'] < tl and r['logits_rel'] < ta and r['memory_rel'] < tm, r
        assert r.get('ctc_loss_rel', 0.0) < tl, r
        assert worst < tg, r
        assert zero_worst < (1e-3 if mode == 'fp32' else 5e-2), r
    finally:
        ops.set_compute_dtype('bf16')


@pytest.mark.parametrize('mode', ['fp32', 'fp16', 'bf16'])
def test_c2_batch32_matches_oracle(c2_oracle, mode):
    cfg, inputs, targets, ref_loss, ref_aux, ref_flat = c2_oracle
    _compare_with_oracle(cfg, inputs, targets, ref_loss, ref_aux, ref_flat, mode, 'headline', 'C2 B=32 x 1000 frames (bench batch)',
                         HEADLINE_TOL[mode])


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
def test_c2_batch32_through_the_bench_engine_matches_oracle(c2_oracle, mode):
    """The TIMED configuration, not only the bare model: bench.py's construction -- FlatDataParallel (flat in-place gradients, deferred
    grouped weight gradients in the 256-wide launch, the row-padded output layer, the 16-bit gradient hand-overs, the staged frontend
    weight gradient) + FusedAdam's device-side loss scale, `dp.zero_grad(next_dropout_step=True)`, `ops.backward(loss)` -- at the bench
    batch against the oracle: loss and every parameter gradient (read from the flat buffer, loss scale divided out)."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    cfg, inputs, targets, ref_loss, ref_aux, ref_flat = c2_oracle
    ops.set_compute_dtype(mode)
    try:
        model = ota.SpeechToText(cfg)
        syn.fill_state_dict_(model.state_dict(), 1234)
        model = model.to(DEV).train()
        dp = FlatDataParallel(model)
        opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0,
                        noam=dict(model_size=256, warmup_steps=12000, factor=1.0))
        di = {k: v.to(DEV) for k, v in inputs.items()}
        dt = {k: v.to(DEV) for k, v in targets.items()}
        dp.zero_grad(next_dropout_step=True)
        loss, _ = dp(di, dt)
        ops.backward(loss)
        torch.cuda.synchronize()
        ls = float(opt.state[6]) or 1.0
        worst, wkey, rels = 0.0, None, {}
        for k, p in model.named_parameters():
            g_ref = ref_flat[k].grad
            if g_ref is None:
                continue
            e = rel(p.grad.detach() / ls, g_ref)
            rels[k] = e
            if e > worst:
                worst, wkey = e, k
        staged = getattr(model.frontend.output_layer.weight, '_otr_regroup_state', {'dirty': None})['dirty']
        r = {'config': 'C2 B=32 x 1000 frames through bench.py\'s engine (FlatDataParallel + FusedAdam loss scale)', 'mode': mode,
             'loss_rel': abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()), 'loss_scale': ls, 'grad_worst': worst,
             'grad_worst_key': wkey, 'grad_median': float(np.median(list(rels.values()))),
             'grad_frontend_output_layer': rels.get('frontend.output_layer.weight'),
             'grad_conv1_weight': rels.get('frontend.conv1.conv_layer.weight'),
             'grad_output_layer': rels.get('decoder.output_layer.weight', rels.get('decoder.embedding.weight')),
             'frontend_weight_gradient_staged': staged, 'faults': opt.stats()['faults']}
        with open(os.path.join(ROOT, 'gpurun_out', 'parity_headline_engine_%s.json' % mode), 'w') as f:
            json.dump(r, f, indent=1)
        print(json.dumps(r))
        tl, _, tg = HEADLINE_TOL[mode]
        assert r['loss_rel'] < tl and worst < tg, r
        assert staged is True and r['faults'] == 0, r
    finally:
        ops.set_compute_dtype('bf16')


def _ragged(batch, lo, hi, seed):
    """utterance lengths in frames: the longest fills the batch, the rest are drawn from [lo, hi]"""
    rng = np.random.default_rng(seed)
    n = [int(v) for v in rng.integers(lo, hi + 1, batch)]
    n[int(rng.integers(0, batch))] = hi
    return n


@pytest.fixture(scope='module')
def c2_ctc_oracle():
    """configs[1] with the joint CTC term of model/speech2text.py:59-64 (ctc_weight 0.3) on a RAGGED bench-size batch: CTC log-probs
    [249, 32, 4234], up to 16 labels per utterance"""
    cfg = syn.c2_model(0.0, ctc_weight=0.3)
    rng = np.random.default_rng(11)
    inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=5, lengths=_ragged(32, 520, 1000, 6),
                                          tgt_lengths=[int(v) for v in rng.integers(4, 16, 32)])
    return (cfg, inputs, targets) + _oracle_run(cfg, inputs, targets)


# (loss and CTC term, logits / memory, worst gradient): <= 2x measured on MI355X (profiles/r03_parity_c2ctc_*.json): fp32 (1.4e-7,
# 7.1e-7, 2.8e-4), fp16 (4.8e-7, 5.1e-4, 5.3e-3), bf16 (2.0e-5, 3.9e-3, 3.1e-2).  The fp32 gradient figure is the noise floor of
# ANY fp32 CTC: the term is 0.3 x 166 of a loss of 55 here, its logit gradient softmax - occupancy cancels, and the ORACLE'S
# OWN fp32 gradients sit 7.5e-5 (median) / 1.2e-4 (worst) from the same oracle evaluated in float64 (measured, r03)
C2CTC_TOL = {'fp32': (1e-5, 5e-6, 6e-4), 'fp16': (1e-4, 1e-3, 1.1e-2), 'bf16': (1e-3, 8e-3, 6e-2)}


@pytest.mark.parametrize('mode', ['fp32', 'fp16', 'bf16'])
def test_c2_ctc_batch32_matches_oracle(c2_ctc_oracle, mode):
    cfg, inputs, targets, ref_loss, ref_aux, ref_flat = c2_ctc_oracle
    _compare_with_oracle(cfg, inputs, targets, ref_loss, ref_aux, ref_flat, mode, 'c2ctc',
                         'C2 + CTC 0.3, B=32 x 1000 frames, ragged', C2CTC_TOL[mode])


@pytest.fixture(scope='module')
def c4_oracle():
    """configs[3]: conformer_baseline.yaml (d = 384, 12 blocks, BatchNorm on batch statistics) at the bench batch, ragged"""
    cfg = syn.conformer_model(False, 0.0)
    rng = np.random.default_rng(12)
    inputs, targets = syn.synthetic_batch(32, 1000, 80, 4234, 15, seed=7, lengths=_ragged(32, 520, 1000, 8),
                                          tgt_lengths=[int(v) for v in rng.integers(4, 16, 32)])
    return (cfg, inputs, targets) + _oracle_run(cfg, inputs, targets, seed=31)


# (loss, logits, encoder memory, worst gradient): <= 2x measured on MI355X (profiles/r03_parity_c4_*.json): fp32 (1.1e-7, 8.1e-7,
# 2.0e-6, .), fp16 (1.4e-5, 6.2e-4, 1.5e-3, .), bf16 (1.9e-5, 4.9e-3, 1.2e-2, .)
C4_TOL = {'fp32': (1e-5, 5e-6, 5e-6, 5e-4), 'fp16': (1e-4, 1.3e-3, 3e-3, 3e-2), 'bf16': (1e-3, 1e-2, 2.4e-2, 1e-1)}


@pytest.mark.parametrize('mode', ['fp32', 'fp16', 'bf16'])
def test_c4_batch32_matches_oracle(c4_oracle, mode):
    cfg, inputs, targets, ref_loss, ref_aux, ref_flat = c4_oracle
    _compare_with_oracle(cfg, inputs, targets, ref_loss, ref_aux, ref_flat, mode, 'c4', 'C4 conformer_baseline B=32 x 1000 frames, ragged',
                         C4_TOL[mode], seed=31)


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
def test_c4_batch32_through_the_bench_engine_matches_oracle(c4_oracle, mode):
    """VERDICT r05 "missing 4": the Conformer through bench.py's construction -- FlatDataParallel (flat in-place gradients, grouped weight
    gradients incl. the DEFERRED dp products of the relative-position attention and dW_pos behind them, the persistent score-gradient
    tensor, BatchNorm partial sums, the residual + double-LayerNorm launches) + FusedAdam's device-side loss scale -- at the bench batch
    against the ORACLE (not against the bare HIP model): loss and every parameter gradient, read from the flat buffer."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    cfg, inputs, targets, ref_loss, ref_aux, ref_flat = c4_oracle
    ops.set_compute_dtype(mode)
    try:
        model = ota.SpeechToText(cfg)
        syn.fill_state_dict_(model.state_dict(), 31)
        model = model.to(DEV).train()
        dp = FlatDataParallel(model)
        opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0,
                        noam=dict(model_size=cfg['encoder']['d_model'], warmup_steps=12000, factor=1.0))
        di = {k: v.to(DEV) for k, v in inputs.items()}
        dt = {k: v.to(DEV) for k, v in targets.items()}
        g_rms = None
        for rep_ in range(2):                        # twice: the second pass runs on the persistent score-gradient tensors of the first
            dp.zero_grad(next_dropout_step=True)
            loss, _ = dp(di, dt)
            ops.backward(loss)
        torch.cuda.synchronize()
        ls = float(opt.state[6]) or 1.0
        n_el = sum(v.grad.numel() for v in ref_flat.values() if v.grad is not None)
        g_rms = (sum(float(v.grad.double().pow(2).sum()) for v in ref_flat.values() if v.grad is not None) / n_el) ** 0.5
        worst, wkey, rels, zero_worst = 0.0, None, {}, 0.0
        for k, p in model.named_parameters():
            g_ref = ref_flat[k].grad
            if g_ref is None:
                continue
            if k.endswith('conv.depthwise_conv.bias'):           # analytically zero (see _compare_with_oracle)
                zero_worst = max(zero_worst, float((p.grad.detach() / ls).double().pow(2).mean().sqrt()) / g_rms)
                continue
            e = rel(p.grad.detach() / ls, g_ref)
            rels[k] = e
            if e > worst:
                worst, wkey = e, k
        r = {'config': 'C4 conformer_baseline B=32 x 1000 frames, ragged, through bench.py\'s engine (FlatDataParallel + FusedAdam loss scale), second of two passes',
             'mode': mode, 'loss_rel': abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()), 'loss_scale': ls, 'grad_worst': worst,
             'grad_worst_key': wkey, 'grad_median': float(np.median(list(rels.values()))),
             'grad_pos_proj_worst': max(v for k, v in rels.items() if 'pos_proj' in k),
             'grad_posu_posv_worst': max(v for k, v in rels.items() if k.endswith('posu') or k.endswith('posv')),
             'zero_grad_rms_over_model_rms': zero_worst, 'faults': opt.stats()['faults']}
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'parity_c4_engine_%s.json' % mode), 'w') as f:
            json.dump(r, f, indent=1)
        print(json.dumps(r))
        tl, _, _, tg = C4_TOL[mode]
        assert r['loss_rel'] < tl and worst < tg, r
        assert zero_worst < 5e-2 and r['faults'] == 0, r
    finally:
        ops.set_compute_dtype('bf16')


def _beam_search_validity(orc, parts, cfg, inputs, lm, lm_weight, beam, trace, tol):
    """Replay the product's per-step beams (trace: [(prefixes [B*beam, t+1], scores [B*beam])] after step t = 1, 2, ...) on the
    ORACLE's log-probabilities (recognize/speech2text.py:95-146 restated: decoder.inference + lm_weight * lm.predict on the
    product's own prefixes).  Returns
      worst_kept_below_cut  max over steps / utterances of (oracle's beam-th best candidate score - oracle score of a candidate
                            the product kept): <= 0 when the product kept a true top-beam set, positive by the margin it
                            mis-ranked otherwise
      worst_score_drift     max |product cumulative score - oracle cumulative score of the same prefix|
      min_cut_margin[b]     min over steps of (oracle beam-th best - (beam+1)-th best candidate score): how close the ORACLE's
                            own search came to a tie at its cut"""
    with torch.no_grad():
        x, mask = orc.conv_frontend(parts['frontend'], inputs['inputs'], inputs['mask'])
        memory, mmask = orc.transformer_encoder(parts['encoder'], x, mask, cfg['encoder'])
        B, T, D = memory.shape
        R = B * beam
        bm = memory.unsqueeze(1).repeat(1, beam, 1, 1).view(R, T, D)
        bmask = mmask.unsqueeze(1).repeat(1, beam, 1).view(R, T)
        prev = torch.full((R, 1), 1, dtype=torch.long)                        # BOS
        cum = torch.tensor([0.0] + [-float('inf')] * (beam - 1), dtype=torch.float64).repeat(B)
        worst_cut, worst_drift, min_margin = -float('inf'), 0.0, [float('inf')] * B
        for step, (pref, sc) in enumerate(trace):
            pref, sc = pref.cpu(), sc.double().cpu().view(-1)
            lp = orc.decoder_inference(parts['decoder'], prev, bm, bmask, cfg['decoder'])
            if lm is not None:
                lp = lp + lm_weight * orc.transformer_lm_predict(lm[0], lm[1], prev)
            lp = lp.double()
            if step > 0:
                # finished beams (last token EOS; BOS shares the id, hence not at step 0) have ONE live branch: EOS at score 0
                # (mask_finished_scores / mask_finished_preds, recognize/speech2text.py:156-192)
                fin = prev[:, -1] == orc.EOS
                if bool(fin.any()):
                    lp = lp.clone()
                    lp[fin] = -float('inf')
                    lp[fin, orc.EOS] = 0.0
            cand = (cum.view(R, 1) + lp).view(B, -1)                           # every (beam, token) expansion of the product's beams
            V = lp.size(1)
            top = torch.topk(cand, beam + 1, dim=-1).values
            new_cum = torch.empty(R, dtype=torch.float64)
            for r in range(R):
                b = r // beam
                # the parent of the kept prefix: the beam of this utterance whose prefix it extends (first match; duplicates
                # carry identical scores)
                par = [q for q in range(b * beam, (b + 1) * beam) if torch.equal(prev[q], pref[r, :-1]) and cum[q] > -float('inf')]
                assert par, ('kept prefix does not extend a beam of the previous step', r)
                new_cum[r] = max(float(cum[q] + lp[q, int(pref[r, -1])]) for q in par)
                worst_cut = max(worst_cut, float(top[b, beam - 1] - new_cum[r]))
                worst_drift = max(worst_drift, abs(float(sc[r]) - float(new_cum[r])))
            for b in range(B):
                min_margin[b] = min(min_margin[b], float(top[b, beam - 1] - top[b, beam]))
            prev, cum = pref, new_cum
    return {'worst_kept_below_cut': worst_cut, 'worst_score_drift': worst_drift, 'min_cut_margin': min_margin, 'steps': len(trace)}


@pytest.mark.parametrize('mode', ['fp32', 'fp16', 'bf16'])
def test_c5_full_size_decode_matches_oracle_beam_search(mode):
    import opentransformer_amd as ota
    from opentransformer_amd import ops
    from opentransformer_amd.recognize import SpeechToTextRecognizer, TransformerLanguageModel
    from oracle import otrans_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = syn.c2_model(0.0)                                   # full size: 12 encoder / 6 decoder layers
    lm_cfg = syn.lm_config(4234, num_blocks=4)                # transformer_lm.yaml with vocab_size forced to 4234
    beam, max_len = 10, 10
    parts = H.filled_state(cfg, seed=7)
    parts['decoder']['output_layer.bias'][1] = -30.0          # EOS never wins: every hypothesis runs max_len steps
    inputs, _ = syn.synthetic_batch(batch=2, frames=1000, feat_dim=80, vocab=4234, tgt_len=5, seed=3, lengths=[1000, 873])
    ops.set_compute_dtype(mode)
    try:
        model = ota.SpeechToText(cfg)
        for name, sd in (('frontend', model.frontend), ('encoder', model.encoder), ('decoder', model.decoder)):
            sd.load_state_dict(parts[name], strict=True)
        lm = TransformerLanguageModel(lm_cfg)
        syn.fill_state_dict_(lm.state_dict(), 8)
        lm_sd = {k: v.detach().clone() for k, v in lm.state_dict().items()}
        model, lm = model.to(DEV).eval(), lm.to(DEV).eval()
        ref_h, ref_s = orc.beam_search(parts, cfg, inputs['inputs'], inputs['mask'], beam=beam, max_len=max_len, penalty=0.6,
                                       lamda=5, nbest=beam, lm=(lm_sd, lm_cfg), lm_weight=0.1)
        rec = SpeechToTextRecognizer(model, apply_cache=True, beam_width=beam, nbest=beam, max_len=max_len, penalty=0.6, lamda=5,
                                     lm=lm, lm_weight=0.1, idx2unit={i: str(i) for i in range(4234)})
        rec.trace = rec_trace = []
        got_h, got_s = rec.recognize(inputs['inputs'].to(DEV), inputs['mask'].to(DEV))
        got_tok = [[[int(t) for t in s.split()] for s in utt] for utt in got_h]
        ref_s, got_s = ref_s.numpy(), got_s.numpy()
        if mode == 'fp32':
            assert got_tok == ref_h
            np.testing.assert_allclose(got_s, ref_s, rtol=1e-4, atol=1e-4)
            return
        # 16-bit operands: rounding can flip a near-tie INSIDE the search (a candidate pruned at some step in one run survives
        # in the other), which inserts / drops whole hypotheses further down the n-best list.  "Identical hypotheses" is
        # therefore asserted in the only form rounding leaves meaningful: the product's search, replayed step by step on
        # the ORACLE's scores, never keeps a candidate the oracle scores more than 2 x drift below its own cut, never
        # reports a cumulative score further than the drift from the oracle's, and ends in the oracle's ranking wherever
        # the oracle's final scores are further apart than 2 x drift.  drift = measured worst |score - oracle score| x 2
        # (profiles/r03_decode_validity_*.json: fp16 0.008, bf16 0.035 nat over 10 steps)
        tol = 0.02 if mode == 'fp16' else 0.08
        rep = _beam_search_validity(orc, parts, cfg, inputs, (lm_sd, lm_cfg), 0.1, beam, rec_trace, tol)
        rep.update(mode=mode, tol=tol, nbest_identical=[got_tok[b] == ref_h[b] for b in range(len(ref_h))],
                   nbest_shared=[len(set(map(tuple, got_tok[b])) & set(map(tuple, ref_h[b]))) for b in range(len(ref_h))])
        with open(os.path.join(ROOT, 'gpurun_out', 'decode_validity_%s.json' % mode), 'w') as f:
            json.dump(rep, f, indent=1)
        print(json.dumps(rep))
        assert rep['worst_kept_below_cut'] < 2 * tol, rep
        assert rep['worst_score_drift'] < tol, rep
        for b in range(len(ref_h)):
            # final ranking (length penalty is a constant here: EOS never wins): same order as the oracle's own scores of the
            # product's hypotheses wherever those are separated by more than 2 x drift
            assert abs(got_s[b, 0] - ref_s[b, 0]) < tol, (mode, b, got_s[b, 0], ref_s[b, 0])
            assert got_tok[b][0] == ref_h[b][0] or ref_s[b, 0] - ref_s[b, 1] < 2 * tol, (mode, b)
            ref_map = {tuple(h): float(ref_s[b, n]) for n, h in enumerate(ref_h[b])}
            if rep['min_cut_margin'][b] > 2 * tol:         # no near-tie at any cut of this utterance's search: the same n-best set
                assert set(map(tuple, got_tok[b])) == set(ref_map), (mode, b)
            for n, h in enumerate(got_tok[b]):             # shared hypotheses: same score, and in the oracle's order up to the drift
                if tuple(h) in ref_map:
                    assert abs(got_s[b, n] - ref_map[tuple(h)]) < tol, (mode, b, n)
                    later = [ref_map[tuple(g)] for g in got_tok[b][n + 1:] if tuple(g) in ref_map]
                    assert all(ref_map[tuple(h)] > v - 2 * tol for v in later), (mode, b, n)
    finally:
        ops.set_compute_dtype('bf16')


# worst cumulative-score drift of the 16-bit modes over the 60-step search, x 2 (measured: profiles/r05_decode_eos_live_*.json)
LIVE_EOS_TOL = {'fp16': 0.03, 'bf16': 0.25}


@pytest.mark.parametrize('mode', ['fp32', 'fp16', 'bf16'])
def test_c5_full_size_decode_with_live_eos(mode):
    """the full-size C5 search with EOS LIVE -- beams finish at different steps (hypothesis lengths 0 .. 60 in one
    n-best list), so mask_finished_scores / mask_finished_preds (recognize/speech2text.py:156-192), the all-finished early exit
    (:117-118) and the length penalty over ragged lengths (:128-131) run at V = 4234 with LM fusion, B = 8 ragged utterances,
    beam 10, max_len 60, against orc.beam_search.  The EOS logit gets +4 (random weights otherwise never emit it).
    16-bit modes: no count of shared hypotheses is asserted -- a rounding flip of a near-tie deep in the search
    reshuffles the tail of an n-best list without being wrong.  The product's per-step beams are REPLAYED on the oracle's scores
    (_beam_search_validity, with the finished-beam masking): every kept candidate lies within 2 x tol of the oracle's cut at
    that step, every cumulative score within tol of the oracle's, and the final (length-normalised) ranking agrees wherever the
    oracle's own search never came closer than 2 x tol to a tie at a cut."""
    import opentransformer_amd as ota
    from opentransformer_amd import ops
    from opentransformer_amd.recognize import SpeechToTextRecognizer, TransformerLanguageModel
    from oracle import otrans_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = syn.c2_model(0.0)
    lm_cfg = syn.lm_config(4234, num_blocks=4)
    beam, max_len, B = 10, 60, 8
    parts = H.filled_state(cfg, seed=7)
    parts['decoder']['output_layer.bias'][1] = 4.0
    inputs, _ = syn.synthetic_batch(batch=B, frames=1000, feat_dim=80, vocab=4234, tgt_len=5, seed=3,
                                    lengths=[1000, 873, 640, 999, 512, 931, 777, 404])
    ops.set_compute_dtype(mode)
    try:
        model = ota.SpeechToText(cfg)
        for name, sd in (('frontend', model.frontend), ('encoder', model.encoder), ('decoder', model.decoder)):
            sd.load_state_dict(parts[name], strict=True)
        lm = TransformerLanguageModel(lm_cfg)
        syn.fill_state_dict_(lm.state_dict(), 8)
        lm_sd = {k: v.detach().clone() for k, v in lm.state_dict().items()}
        model, lm = model.to(DEV).eval(), lm.to(DEV).eval()
        ref_h, ref_s = orc.beam_search(parts, cfg, inputs['inputs'], inputs['mask'], beam=beam, max_len=max_len, penalty=0.6,
                                       lamda=5, nbest=beam, lm=(lm_sd, lm_cfg), lm_weight=0.1)
        lens = sorted({len(h) for utt in ref_h for h in utt})
        assert len(lens) >= 5 and lens[0] < 3 and lens[-1] >= 20, lens       # the case does what it is for: ragged finishing steps
        ref_s = ref_s.numpy()
        report = {'mode': mode, 'lengths': lens}
        for cache in (True, False) if mode == 'fp32' else (True,):
            rec = SpeechToTextRecognizer(model, apply_cache=cache, beam_width=beam, nbest=beam, max_len=max_len, penalty=0.6, lamda=5,
                                         lm=lm, lm_weight=0.1, idx2unit={i: str(i) for i in range(4234)})
            rec.trace = rec_trace = []
            got_h, got_s = rec.recognize(inputs['inputs'].to(DEV), inputs['mask'].to(DEV))
            got_tok = [[[int(t) for t in s.split()] for s in utt] for utt in got_h]
            got_s = got_s.numpy()
            if mode == 'fp32':
                assert got_tok == ref_h, cache
                np.testing.assert_allclose(got_s, ref_s, rtol=2e-4, atol=2e-4)
                if cache:      # the replay criterion itself, where the answer is known: an exact search is a valid search
                    rep = _beam_search_validity(orc, parts, cfg, inputs, (lm_sd, lm_cfg), 0.1, beam, rec_trace, 1e-3)
                    report.update(rep)
                    assert rep['worst_kept_below_cut'] < 1e-3 and rep['worst_score_drift'] < 1e-3, rep
                continue
            # tol: <= 2 x the measured worst cumulative-score drift over up to 60 steps (profiles/r05_decode_eos_live_*.json)
            tol = LIVE_EOS_TOL[mode]
            rep = _beam_search_validity(orc, parts, cfg, inputs, (lm_sd, lm_cfg), 0.1, beam, rec_trace, tol)
            drift_final, shared = 0.0, []
            for b in range(B):
                ref_map = {tuple(h): float(ref_s[b, n]) for n, h in enumerate(ref_h[b])}
                if rep['min_cut_margin'][b] > 2 * tol:     # no near-tie at any cut of this utterance's search: the same n-best set
                    assert set(map(tuple, got_tok[b])) == set(ref_map), (mode, b)
                n_sh = 0
                for n, h in enumerate(got_tok[b]):
                    if tuple(h) in ref_map:
                        n_sh += 1
                        drift_final = max(drift_final, abs(float(got_s[b, n]) - ref_map[tuple(h)]))
                shared.append(n_sh)
            # The 1-best, per utterance: identical to the oracle's OUTRIGHT wherever the oracle's own final 1-best / 2-best
            # gap exceeds twice the final score drift MEASURED in this run (each of the two scores may move by the drift) -- no 2 x tol
            # escape, no appeal to near-ties at cuts.  (r06 on MI355X: 8 / 8 identical in fp16 and bf16; the gaps are 9.3-9.8 nat
            # against a drift of 3.3e-3 / 3.4e-2.)  one_best_flip_possible records the other MEASURED way a 1-best could differ
            # legitimately -- the oracle's search came closer to a tie at some cut than the depth below the cut at which this run is
            # measured to keep candidates -- for the report only.
            one_best = [got_tok[b][0] == ref_h[b][0] for b in range(B)]
            gap12 = [float(ref_s[b, 0] - ref_s[b, 1]) for b in range(B)]
            near_cut = [rep['min_cut_margin'][b] <= 2 * rep['worst_kept_below_cut'] for b in range(B)]
            report.update(rep, tol=tol, worst_final_score_drift=drift_final, nbest_shared=shared,
                          nbest_identical=[got_tok[b] == ref_h[b] for b in range(B)], one_best_identical=one_best,
                          oracle_gap_1best_2best=gap12, one_best_flip_possible=near_cut)
            assert all(one_best[b] or gap12[b] <= 2 * drift_final for b in range(B)), report
            assert rep['worst_kept_below_cut'] < 2 * tol, report
            assert rep['worst_score_drift'] < tol, report
            assert drift_final < tol, report
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'decode_eos_live_%s.json' % mode), 'w') as f:
            json.dump(report, f, indent=1)
        print(json.dumps(report))
    finally:
        ops.set_compute_dtype('bf16')
