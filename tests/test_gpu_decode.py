"""GPU (-m gpu): decode parity -- token-identical greedy / beam / LM-fusion hypotheses and scores
against the fixtures the REAL reference recognizers produced on a briefly trained C1 model
(tests/golden/c1_decode.npz, oracle/make_golden.py:golden_decode)."""
import numpy as np
import pytest
import torch

from opentransformer_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def load_models(g, mode):
    import opentransformer_amd as ota
    from opentransformer_amd import ops
    from opentransformer_amd.recognize import TransformerLanguageModel
    ops.set_compute_dtype(mode)
    model = ota.SpeechToText(syn.c1_model(0.0, ctc_weight=0.3))
    model.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w:')}, strict=True)
    lm = TransformerLanguageModel(syn.lm_config(100, d_model=64, d_ff=128, num_blocks=2))
    syn.fill_state_dict_(lm.state_dict(), 4321)
    return model.to(DEV).eval(), lm.to(DEV).eval()


def hyp_arr(nbest, like):
    a = -np.ones_like(like)
    for i, utt in enumerate(nbest):
        for j, s in enumerate(utt):
            toks = [int(t) for t in s.split()]
            a[i, j, :len(toks)] = toks
    return a


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_beam_search_matches_reference(golden, mode):
    from opentransformer_amd import ops
    from opentransformer_amd.recognize import SpeechToTextRecognizer
    g = golden('c1_decode.npz')
    try:
        model, lm = load_models(g, mode)
        x, m = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['mask']).to(DEV)
        idx2unit = {i: str(i) for i in range(100)}
        for tag, kw in [('greedy', dict(beam_width=1, nbest=1, max_len=12, penalty=0.0)),
                        ('beam5', dict(beam_width=5, nbest=5, max_len=12, penalty=0.6, lamda=5)),
                        ('beam5_lm', dict(beam_width=5, nbest=3, max_len=12, penalty=0.6, lamda=5, lm=lm, lm_weight=0.3))]:
            rec = SpeechToTextRecognizer(model, idx2unit=idx2unit, ngpu=1, **kw)
            nbest, scores = rec.recognize(x, m)
            want = g[tag + '_hyp']
            got = hyp_arr(nbest, want)
            if mode == 'fp32':
                assert np.array_equal(got, want), tag
                np.testing.assert_allclose(scores.numpy(), g[tag + '_score'], rtol=2e-4, atol=2e-4)
            else:
                # bf16: the 1-best must be token-identical wherever the REFERENCE's own 1-best/2-best
                # score margin exceeds 0.1 nat; below that bf16 rounding may legitimately swap beams
                # (one such utterance exists: LM-fused scores -12.427 / -12.477 / -12.551).
                ref_s = g[tag + '_score']
                clear = np.ones(len(want), bool) if ref_s.shape[1] < 2 else (ref_s[:, 0] - ref_s[:, 1]) > 0.1
                assert clear.sum() >= 3, tag
                assert np.array_equal(got[clear, 0], want[clear, 0]), tag
                np.testing.assert_allclose(scores.numpy()[clear, 0], ref_s[clear, 0], rtol=5e-2, atol=5e-2)
    finally:
        ops.set_compute_dtype('bf16')


def test_decoder_inference_lm_and_ctc_greedy_match_reference(golden):
    from opentransformer_amd import ops
    from opentransformer_amd.recognize import CTCRecognizer
    g = golden('c1_decode.npz')
    try:
        model, lm = load_models(g, 'fp32')
        x, m = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['mask']).to(DEV)
        with torch.no_grad():
            fe, fm, _ = model.frontend.inference(x, m, None)
            mem, mm, _ = model.encoder(fe, fm)
            preds = torch.from_numpy(g['inference_preds']).to(DEV)
            lp, _, _ = model.decoder.inference(preds, mem, mm)
            np.testing.assert_allclose(lp.cpu().numpy(), g['inference_logp'], rtol=1e-4, atol=1e-4)
            llp = lm.predict(preds, last_frame=True).squeeze(1)
            np.testing.assert_allclose(llp.cpu().numpy(), g['lm_logp'], rtol=1e-4, atol=1e-4)
            clp, cln = model.assistor.inference(mem, mm)
            np.testing.assert_allclose(clp.cpu().numpy(), g['ctc_head_logp'], rtol=1e-4, atol=2e-4)
        greedy = CTCRecognizer(model, idx2unit={i: str(i) for i in range(100)}, mode='greedy').recognize_greedy(x, m)
        want = [[t for t in row if t >= 0] for row in g['ctc_greedy'].tolist()]
        assert greedy == want
    finally:
        ops.set_compute_dtype('bf16')


def test_beam_kernels_against_torch():
    """otr_beam_topk / otr_beam_prune vs the reference's own torch formulation on random scores."""
    import ctypes as C
    from opentransformer_amd import _lib as L
    from oracle import otrans_oracle as orc     # noqa: F401  (semantics documented there)
    lib = L.load()
    gen = torch.Generator().manual_seed(0)
    R, V, k = 12, 4234, 5
    logits = torch.randn(R, 3, V, generator=gen).to(DEV)
    lm = torch.randn(R, 3, V, generator=gen).to(DEV)
    ks = torch.empty(R, k, device=DEV)
    ki = torch.empty(R, k, dtype=torch.long, device=DEV)
    p = lambda t, off=0: C.c_void_p(t.data_ptr() + off * t.element_size())      # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.otr_beam_topk(p(logits, 2 * V), 3 * V, p(lm, 2 * V), 3 * V, 0.3, R, V, k, p(ks), p(ki), st), 'topk')
    ref = torch.log_softmax(logits[:, -1], -1) + 0.3 * torch.log_softmax(lm[:, -1], -1)
    rs, ri = ref.topk(k)
    assert torch.equal(ki, ri)
    torch.testing.assert_close(ks, rs, rtol=1e-5, atol=1e-5)
