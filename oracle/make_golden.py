#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

TEST INFRASTRUCTURE.  Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py
The reference ships no golden vectors, so these fixtures -- outputs of the
reference's own modules on seeded inputs/weights -- are what pins both the oracle restatement
(oracle/otrans_oracle.py) and the HIP path.  Weights are NOT stored: they are regenerated on
any machine by opentransformer_amd.synthetic.fill_state_dict_ (numpy Generator, crc32-keyed).
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
# second entry only satisfies the bare `from activation import Swish` at otrans/module/ffn.py:9
sys.path[:0] = ['/root/reference', '/root/reference/otrans/module']

from otrans.model import End2EndModel, LanguageModel            # noqa: E402
from otrans.model.ctc import CTCAssistor                        # noqa: E402
from otrans.recognize.speech2text import SpeechToTextRecognizer  # noqa: E402
from otrans.recognize.ctc import CTCRecognizer                  # noqa: E402

from opentransformer_amd import synthetic as syn                # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def probe_vector(key, numel):
    rng = np.random.default_rng(zlib.crc32(('probe:' + key).encode()))
    return rng.standard_normal(numel).astype(np.float32)


def grad_summary(named_params):
    """Per-parameter [L2 norm, dot with a crc-seeded probe vector] (2 scalars pin a tensor)."""
    keys, vals = [], []
    for k, p in named_params:
        if p.grad is None:
            continue
        g = p.grad.detach().double().reshape(-1).numpy()
        keys.append(k)
        vals.append([np.sqrt((g * g).sum()), float((g * probe_vector(k, g.size)).sum())])
    return np.array(keys), np.array(vals, dtype=np.float64)


def build(model_cfg, seed=1234):
    torch.manual_seed(seed)
    model = End2EndModel[model_cfg['type']](model_cfg)
    syn.fill_state_dict_(model.state_dict(), seed)
    return model


def parts_state(model):
    return {'frontend': model.frontend.state_dict(), 'encoder': model.encoder.state_dict(),
            'decoder': model.decoder.state_dict()}


def golden_train(name, cfg, batch_kw, store_full_grads=()):
    model = build(cfg)
    model.train()       # dropout rates are 0 in cfg
    inputs, targets = syn.synthetic_batch(**batch_kw)
    fe_out, fe_mask = model.frontend(inputs['inputs'], inputs['mask'])
    memory, mem_mask, _ = model.encoder(fe_out, fe_mask)
    logits, _ = model.decoder(targets['targets'][:, :-1].clone(), memory, mem_mask)
    ce = model.crit(logits, targets['targets'][:, 1:].clone())
    loss, aux = model(inputs, targets)
    loss.backward()
    keys, gsum = grad_summary(list(model.named_parameters()))
    out = dict(loss=loss.item(), ce=ce.item(), ctc=(aux or {}).get('CTCLoss', np.nan),
               fe_out=fe_out.detach().numpy(), fe_mask=fe_mask.numpy(), memory=memory.detach().numpy(),
               logits=logits.detach().numpy(), grad_keys=keys, grad_summary=gsum)
    named = dict(model.named_parameters())
    for k in store_full_grads:
        out['grad:' + k] = named[k].grad.numpy()
    if cfg['ctc_weight'] > 0:
        lp, ln = model.assistor.inference(memory.detach(), mem_mask)
        out['ctc_log_probs'] = lp.detach().numpy()
        out['ctc_len'] = ln.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, 'loss', out['loss'], 'ce', out['ce'], 'ctc', out['ctc'], 'nparams',
          sum(p.numel() for p in model.parameters()))
    return model


TEMPLATE_SEED = 5


def template_batch(B, rng, V=100, F=80):
    """Learnable synthetic ASR task: every token id has a fixed random F-dim template that is
    held for 12-19 frames, plus noise.  Used to train C1 briefly so hypotheses are not
    degenerate (a random-init model emits EOS at once: SURVEY.md 8c)."""
    templ = np.random.default_rng(TEMPLATE_SEED).standard_normal((V, F)).astype(np.float32)
    feats, toks = [], []
    for _ in range(B):
        n = rng.integers(3, 10)
        tk = rng.integers(3, V, n)
        seg = rng.integers(12, 20, n)
        f = np.concatenate([np.repeat(templ[t][None], s, 0) for t, s in zip(tk, seg)])
        feats.append((f + 0.3 * rng.standard_normal(f.shape)).astype(np.float32))
        toks.append(tk)
    T = max(len(f) for f in feats)
    L = max(len(t) for t in toks)
    x = np.zeros((B, T, F), np.float32)
    m = np.zeros((B, T), bool)
    tg = np.zeros((B, L + 2), np.int64)
    tl = []
    for b in range(B):
        x[b, :len(feats[b])] = feats[b]
        m[b, :len(feats[b])] = True
        tg[b, 0] = 1
        tg[b, 1:1 + len(toks[b])] = toks[b]
        tg[b, 1 + len(toks[b])] = 1
        tl.append(len(toks[b]) + 1)
    return ({'inputs': torch.from_numpy(x), 'mask': torch.from_numpy(m)},
            {'targets': torch.from_numpy(tg), 'targets_length': torch.tensor(tl, dtype=torch.int32)})


def train_c1(cfg, steps=1500):
    """Brief CPU training of the reference C1 model (reference modules, torch Adam)."""
    torch.manual_seed(1234)
    model = End2EndModel[cfg['type']](cfg)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(6)
    model.train()
    for step in range(steps):
        for g in opt.param_groups:
            g['lr'] = 1e-3 * min(1.0, (step + 1) / 300.0)
        i, t = template_batch(16, rng)
        loss, _ = model(i, t)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        if step % 250 == 0:
            print('train_c1', step, loss.item(), flush=True)
    return model


def golden_decode(name, cfg):
    """Beam / greedy hypotheses from the reference recognizers on a briefly trained C1 model
    (weights stored in the fixture as 'w:<state_dict key>')."""
    model = train_c1(cfg)
    model.eval()
    inputs, tgt = template_batch(4, np.random.default_rng(77))
    V = cfg['decoder']['vocab_size']
    idx2unit = {i: str(i) for i in range(V)}
    out = {'inputs': inputs['inputs'].numpy(), 'mask': inputs['mask'].numpy(), 'truth': tgt['targets'].numpy()}
    for k, v in model.state_dict().items():
        out['w:' + k] = v.detach().numpy()

    def to_arr(nbest):   # [[str,...],...] -> int array padded with -1
        rows = [[[int(t) for t in s.split()] for s in utt] for utt in nbest]
        L = max(1, max(len(h) for u in rows for h in u))
        a = -np.ones((len(rows), len(rows[0]), L), dtype=np.int64)
        for i, u in enumerate(rows):
            for j, h in enumerate(u):
                a[i, j, :len(h)] = h
        return a

    lm_cfg = syn.lm_config(V, d_model=cfg['decoder']['d_model'], d_ff=128, num_blocks=2)
    torch.manual_seed(7)
    lm = LanguageModel['transformer_lm'](lm_cfg)
    syn.fill_state_dict_(lm.state_dict(), 4321)
    lm.eval()
    for tag, kw in [('greedy', dict(beam_width=1, nbest=1, max_len=12, penalty=0.0)),
                    ('beam5', dict(beam_width=5, nbest=5, max_len=12, penalty=0.6, lamda=5)),
                    ('beam5_lm', dict(beam_width=5, nbest=3, max_len=12, penalty=0.6, lamda=5, lm=lm, lm_weight=0.3))]:
        rec = SpeechToTextRecognizer(model, idx2unit=idx2unit, ngpu=0, **kw)
        nbest, scores = rec.recognize(inputs['inputs'], inputs['mask'])
        out[tag + '_hyp'] = to_arr(nbest)
        out[tag + '_score'] = scores.numpy()
        print(name, tag, [u[0] for u in nbest], scores[:, 0].tolist())
    # one decoder.inference call pinned exactly
    with torch.no_grad():
        fe, fm, _ = model.frontend.inference(inputs['inputs'], inputs['mask'], None)
        mem, mm, _ = model.encoder(fe, fm)
        preds = torch.tensor([[1, 5, 9], [1, 7, 3], [1, 4, 4], [1, 50, 60]])[:mem.size(0)]
        lp, _, _ = model.decoder.inference(preds, mem, mm)
        out['inference_preds'] = preds.numpy()
        out['inference_logp'] = lp.numpy()
        out['lm_logp'] = lm.predict(preds, last_frame=True).squeeze(1).numpy()
    # CTC greedy via the assistor head (CTCModel.inference is broken in the reference: SURVEY 3.4)
    with torch.no_grad():
        lp, ln = model.assistor.inference(mem, mm)

    class _M:                                   # minimal stand-in for the model CTCRecognizer expects
        def eval(self):
            return self

        def inference(self, a, b):
            return lp, ln
    greedy = CTCRecognizer(_M(), idx2unit=idx2unit, ngpu=0, mode='greedy', beam_width=1).recognize_greedy(None, None)
    L = max(1, max(len(g) for g in greedy))
    ga = -np.ones((len(greedy), L), dtype=np.int64)
    for i, g in enumerate(greedy):
        ga[i, :len(g)] = g
    out['ctc_greedy'] = ga
    out['ctc_head_logp'] = lp.numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)


def golden_decode_rnnlm(name='c1_decode_rnnlm.npz', base='c1_decode.npz'):
    """Shallow fusion with the RECURRENT language model (model/lm.py:33-91 through recognize/base.py:26-37): the trained C1 weights
    of c1_decode.npz (not re-trained), a seeded 2-layer LSTM LM, the reference's beam search; plus RecurrentLanguageModel.predict on
    a token batch with and without a carried state."""
    g = np.load(os.path.join(OUT, base))
    cfg = syn.c1_model(residual_dropout=0.0, ctc_weight=0.3)
    torch.manual_seed(1234)
    model = End2EndModel[cfg['type']](cfg)
    model.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w:')}, strict=True)
    model.eval()
    V = cfg['decoder']['vocab_size']
    idx2unit = {i: str(i) for i in range(V)}
    inputs = {'inputs': torch.from_numpy(g['inputs']), 'mask': torch.from_numpy(g['mask'])}
    lm_cfg = syn.rnn_lm_config(V, hidden_size=cfg['decoder']['d_model'], num_layers=2)
    torch.manual_seed(7)
    lm = LanguageModel['rnn_lm'](lm_cfg)
    syn.fill_state_dict_(lm.state_dict(), 4321)
    lm.eval()
    out = {}

    def to_arr(nbest):
        rows = [[[int(t) for t in s.split()] for s in utt] for utt in nbest]
        L = max(1, max(len(h) for u in rows for h in u))
        a = -np.ones((len(rows), len(rows[0]), L), dtype=np.int64)
        for i, u in enumerate(rows):
            for j, h in enumerate(u):
                a[i, j, :len(h)] = h
        return a
    rec = SpeechToTextRecognizer(model, idx2unit=idx2unit, ngpu=0, beam_width=5, nbest=3, max_len=12, penalty=0.6, lamda=5, lm=lm,
                                 lm_weight=0.3)
    with torch.no_grad():
        nbest, scores = rec.recognize(inputs['inputs'], inputs['mask'])
    out['beam5_rnnlm_hyp'] = to_arr(nbest)
    out['beam5_rnnlm_score'] = scores.numpy()
    print(name, [u[0] for u in nbest], scores[:, 0].tolist())
    with torch.no_grad():
        toks = torch.tensor([[1, 5, 9, 33, 2], [1, 7, 3, 98, 64], [1, 4, 4, 4, 4]])
        lp, hid = lm.predict(toks)                                  # whole sequences from the zero state
        out['predict_tokens'] = toks.numpy()
        out['predict_logp'] = lp.numpy()
        out['predict_h'], out['predict_c'] = hid[0].numpy(), hid[1].numpy()
        lp2, hid2 = lm.predict(toks[:, :2], hid)                    # ... and two more steps carried on from that state
        out['predict2_logp'] = lp2.numpy()
        out['predict2_h'], out['predict2_c'] = hid2[0].numpy(), hid2[1].numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)


def golden_tools():
    """tests/golden/tools_average.npz: the reference's own average_parameters (otrans/utils.py:46-102) on the toy
    checkpoints of tests/test_tools.py:make_checkpoints."""
    import tempfile
    from otrans.utils import average_parameters
    from tests.test_tools import make_checkpoints
    with tempfile.TemporaryDirectory() as d:
        make_checkpoints(d, 5)
        out = average_parameters(d, N=3)
        state = torch.load(out)
    arrs = {}
    for part in ('frontend', 'encoder', 'decoder'):
        for k, v in state[part].items():
            arrs['%s/%s' % (part, k)] = v.numpy()
    np.savez(os.path.join(OUT, 'tools_average.npz'), **arrs)
    print('tools_average.npz', sorted(arrs))


def golden_data():
    """tests/golden/data_collate.npz: the reference's collate_fn_with_eos_bos (data/loader.py:66-108) and spec_augment
    (data/augment.py:9-41) on seeded toy utterances.  otrans.data.loader imports torchaudio / kaldiio /
    python_speech_features (absent here): stub modules stand in for them, the two functions do not use them."""
    import random
    import types
    for name in ('torchaudio', 'kaldiio', 'python_speech_features', 'prefetch_generator'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.BackgroundGenerator = object
            sys.modules[name] = m
    from otrans.data.loader import collate_fn_with_eos_bos
    from otrans.data.augment import spec_augment
    from tests.test_data import toy_batch
    batch = toy_batch()
    ids, inputs, targets = collate_fn_with_eos_bos(batch)
    arrs = {'inputs': inputs['inputs'].numpy(), 'inputs_length': inputs['inputs_length'].numpy(), 'mask': inputs['mask'].numpy(),
            'targets': targets['targets'].numpy(), 'targets_length': targets['targets_length'].numpy(),
            'targets_mask': targets['mask'].numpy()}
    np.random.seed(11)
    random.seed(12)
    aug = []
    for _, feat, flen, _, _ in toy_batch():
        aug.append(spec_augment(feat[:flen].numpy().copy(), freq_mask_num=2, time_mask_num=2, freq_mask_rate=0.3,
                                time_mask_rate=0.2, max_mask_time_len=100))
    for i, a in enumerate(aug):
        arrs['aug%d' % i] = a
    np.savez(os.path.join(OUT, 'data_collate.npz'), **arrs)
    print('data_collate.npz', ids, [a.shape for a in aug])


def golden_bucket():
    """tests/golden/data_bucket.npz: the reference's BySequenceLengthSampler (data/bucket.py:14-170) on a seeded toy
    length table, in its four modes; per case the batches of the constructor split, of two epochs of __iter__, and of a
    shuffle_batch_in_bucket() re-split, flattened as (values, offsets)."""
    import random
    import types
    for name in ('torchaudio', 'kaldiio', 'python_speech_features', 'prefetch_generator'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.BackgroundGenerator = object
            sys.modules[name] = m
    from otrans.data.bucket import BySequenceLengthSampler
    from tests.test_data import ToyLengths, BUCKET_CASES, flatten_batches
    arrs = {}
    for name, kw in BUCKET_CASES.items():
        random.seed(21)
        s = BySequenceLengthSampler(ToyLengths(), **kw)
        seqs = [[b for _, b in s.batch_list], list(s), list(s)]
        s.shuffle_batch_in_bucket()
        seqs.append(list(s))
        for i, seq in enumerate(seqs):
            arrs['%s_%d_v' % (name, i)], arrs['%s_%d_o' % (name, i)] = flatten_batches(seq)
        print('data_bucket', name, [len(q) for q in seqs])
    np.savez(os.path.join(OUT, 'data_bucket.npz'), **arrs)


VARIANTS = {'prenorm': (True, False, False), 'concat': (False, True, False), 'prenorm_concat': (True, True, False),
            'relpos': (False, False, True), 'relpos_prenorm_concat': (True, True, True)}


def golden_shared_projections():
    """tests/golden/modules_shared.npz: MultiHeadedSelfAttention(share_qvk_proj=True) and
    MultiHeadedCrossAttention(share_vk_proj=True) (module/attention.py:49-145) -- options no encoder/decoder constructor passes
    on, so they are pinned at module level: seeded weights and inputs, outputs and all gradients."""
    from otrans.module.attention import MultiHeadedSelfAttention, MultiHeadedCrossAttention
    from tests.test_gpu_ops import shared_projection_inputs
    x, mem, xmask, mmask, dy = shared_projection_inputs()
    out = {}
    sa = MultiHeadedSelfAttention(4, 64, 0.0, share_qvk_proj=True)
    ca = MultiHeadedCrossAttention(4, 64, 48, 0.0, share_vk_proj=True)
    syn.fill_state_dict_(sa.state_dict(), 31)
    syn.fill_state_dict_(ca.state_dict(), 32)
    xs = x.clone().requires_grad_(True)
    y, _ = sa(xs, xmask.unsqueeze(1))
    y.backward(dy)
    out['sa_y'], out['sa_dx'] = y.detach().numpy(), xs.grad.numpy()
    for k, p in sa.named_parameters():
        out['sa_grad:' + k] = p.grad.numpy()
    xq, ms = x.clone().requires_grad_(True), mem.clone().requires_grad_(True)
    y, _ = ca(xq, ms, mmask.unsqueeze(1))
    y.backward(dy)
    out['ca_y'], out['ca_dq'], out['ca_dmem'] = y.detach().numpy(), xq.grad.numpy(), ms.grad.numpy()
    for k, p in ca.named_parameters():
        out['ca_grad:' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'modules_shared.npz'), **out)
    print('modules_shared.npz', {k: v.shape for k, v in out.items() if not k.startswith(('sa_grad', 'ca_grad'))})


def golden_optimizer():
    """tests/golden/optimizer_steps.npz: the trainer's update (train/trainer.py:221-234: clip_grad_norm_(5.0), NaN guard,
    scheduler.step(), optimizer.step(), zero_grad) with the reference's TransformerScheduler (train/scheduler.py:16-53,
    129-138) driving torch.optim.Adam(betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6) (transformer_baseline.yaml:82-88),
    on seeded parameters and gradients.  Step 2 has a huge gradient (clipped), step 4 a NaN gradient (skipped)."""
    import importlib.util
    import math
    spec = importlib.util.spec_from_file_location('ref_scheduler', '/root/reference/otrans/train/scheduler.py')
    sch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sch)
    from tests.test_gpu_ops import optimizer_inputs
    shapes, params, grads, hp = optimizer_inputs()
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = sch.BuildOptimizer['adam'](ps, lr=hp['lr'], betas=hp['betas'], eps=hp['eps'], weight_decay=hp['weight_decay'])
    scheduler = sch.TransformerScheduler(opt, hp['model_size'], hp['warmup_steps'], hp['factor'])
    out = {'lr0': np.float64(scheduler.lr if scheduler.lr is not None else np.nan), 'global_step0': scheduler.global_step}
    lrs, norms, skipped = [], [], []
    for step, gs in enumerate(grads):
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        grad_norm = torch.nn.utils.clip_grad_norm_(ps, hp['clip'])
        if math.isnan(grad_norm):
            skipped.append(1)
        else:
            skipped.append(0)
            scheduler.step()
            opt.step()
        opt.zero_grad()
        lrs.append(opt.param_groups[0]['lr'])
        norms.append(float(grad_norm))
        out['params_%d' % step] = np.concatenate([p.detach().reshape(-1).numpy() for p in ps])
    out.update(lr=np.asarray(lrs, np.float64), grad_norm=np.asarray(norms, np.float64), skipped=np.asarray(skipped))
    np.savez_compressed(os.path.join(OUT, 'optimizer_steps.npz'), **out)
    print('optimizer_steps.npz lr', lrs, 'norm', norms, 'skipped', skipped, 'init', out['lr0'], out['global_step0'])


def golden_loss_options():
    """tests/golden/module_loss.npz: LabelSmoothingLoss (module/loss.py:12-48) with its `mask` argument and
    normalize_length=False -- options the model never uses, pinned at module level."""
    from otrans.module.loss import LabelSmoothingLoss
    from tests.test_gpu_ops import loss_option_inputs
    logits, target, mask = loss_option_inputs()
    out = {}
    for name, (use_mask, norm) in LOSS_CASES.items():
        lg = logits.clone().requires_grad_(True)
        loss = LabelSmoothingLoss(logits.size(-1), 0.1, normalize_length=norm)(lg, target, mask if use_mask else None)
        loss.backward()
        out[name + '_loss'], out[name + '_grad'] = loss.detach().numpy(), lg.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'module_loss.npz'), **out)
    print('module_loss.npz', {k: float(v) for k, v in out.items() if k.endswith('_loss')})


LOSS_CASES = {'mask': (True, True), 'sum': (False, False), 'mask_sum': (True, False)}
ACTIVATION_CASES = [('gelu', 'swish'), ('tanh', 'relu')]


def golden_variants(c1_batch=None):
    """tests/golden/c1_<variant>.npz: the reference's pre-norm / concat_after layer variants (encoder/transformer.py:41-65,
    decoder/transformer.py:47-90) at plumbing size -- no shipped yaml turns them on, the constructors accept them."""
    c1_batch = c1_batch or dict(batch=4, frames=200, feat_dim=80, vocab=100, tgt_len=10, seed=0,
                                lengths=[200, 180, 150, 97], tgt_lengths=[10, 8, 10, 5])
    for name, (pre, cat, rel) in VARIANTS.items():
        full = ['encoder.blocks.1.norm1.weight', 'decoder.blocks.0.norm3.bias']
        if pre:
            full += ['encoder.norm.weight', 'decoder.after_norm.bias']
        if cat:
            full += ['encoder.blocks.0.concat_linear.weight', 'decoder.blocks.1.concat_linear2.bias']
        if rel:
            full += ['encoder.blocks.0.slf_attn.posu', 'encoder.blocks.1.slf_attn.pos_proj.weight']
        golden_train('c1_%s.npz' % name, syn.c1_variant(pre, cat, relative_positional=rel), c1_batch, store_full_grads=full)
    golden_train('c1_frontend_ln.npz', syn.c1_frontend_ln(), c1_batch,
                 store_full_grads=['frontend.layer_norm.weight', 'frontend.output_layer.bias'])
    for steps in (2, 5):
        golden_train('c1_lookahead%d.npz' % steps, syn.c1_lookahead(steps), c1_batch,
                     store_full_grads=['assistor.lookahead_conv.weight', 'assistor.output_layer.bias'])
    golden_shared_projections()
    golden_loss_options()
    golden_optimizer()
    for enc_act, dec_act in ACTIVATION_CASES:
        golden_train('c1_act_%s_%s.npz' % (enc_act, dec_act), syn.c1_activations(enc_act, dec_act), c1_batch,
                     store_full_grads=['encoder.blocks.0.feed_forward.w_1.weight', 'decoder.blocks.1.feed_forward.w_1.bias'])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    c1 = syn.c1_model(residual_dropout=0.0, ctc_weight=0.3)
    c1_batch = dict(batch=4, frames=200, feat_dim=80, vocab=100, tgt_len=10, seed=0,
                    lengths=[200, 180, 150, 97], tgt_lengths=[10, 8, 10, 5])
    golden_train('c1_train.npz', c1, c1_batch,
                 store_full_grads=('frontend.conv1.conv_layer.weight', 'frontend.conv1.conv_layer.bias',
                                   'encoder.blocks.0.slf_attn.qvk_proj.bias', 'decoder.embedding.weight',
                                   'encoder.blocks.1.norm2.weight', 'assistor.output_layer.bias'))
    c1d = syn.c1_model(residual_dropout=0.0, ctc_weight=0.3)
    golden_decode('c1_decode.npz', c1d)
    golden_variants(c1_batch)
    # full-size transformer_baseline (+80-d), tiny batch, ragged, dropout 0: pins C2 numerics
    c2 = syn.c2_model(residual_dropout=0.0)
    c2_batch = dict(batch=2, frames=1000, feat_dim=80, vocab=4234, tgt_len=15, seed=0,
                    lengths=[1000, 873], tgt_lengths=[15, 11])
    golden_train('c2_train_b2.npz', c2, c2_batch)
    # Conformer (BASELINE configs[3]): plumbing size, ragged, dropout 0 (the reference applies F.dropout even
    # in eval: SURVEY.md section 7), BatchNorm in training mode (batch statistics)
    c4s = syn.conformer_model(small=True)
    golden_train('c4_conformer_small.npz', c4s, c1_batch,
                 store_full_grads=('encoder.blocks.0.mha.posu', 'encoder.blocks.0.mha.posv',
                                   'encoder.blocks.0.mha.pos_proj.weight', 'encoder.blocks.1.conv.depthwise_conv.weight',
                                   'encoder.blocks.1.conv.batch_norm.weight', 'encoder.blocks.0.conv.pointwise_conv1.bias'))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'tools':
        golden_tools()
    elif len(sys.argv) > 1 and sys.argv[1] == 'data':
        golden_data()
    elif len(sys.argv) > 1 and sys.argv[1] == 'bucket':
        golden_bucket()
    elif len(sys.argv) > 1 and sys.argv[1] == 'variants':
        golden_variants()
    elif len(sys.argv) > 1 and sys.argv[1] == 'optimizer':
        golden_optimizer()
    elif len(sys.argv) > 1 and sys.argv[1] == 'rnnlm':
        golden_decode_rnnlm()
    else:
        main()
        golden_tools()
        golden_data()
        golden_bucket()
