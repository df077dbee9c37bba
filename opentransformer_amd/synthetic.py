"""Workload definitions shared by bench.py, the tests and oracle/make_golden.py.

Host-side, numpy only (stable across machines by numpy's Generator spec): the named
configurations of BASELINE.json, a deterministic weight fill keyed by state_dict key, and
the seeded synthetic fbank batch of SURVEY.md section 8(d).
"""
import copy
import math
import zlib

import numpy as np
import torch

# egs/aishell/conf/transformer_baseline.yaml:33-76 with frontend.input_size forced to 80
# (the 80-d benchmark uses this yaml + input_size 80).
C2_MODEL = {
    'type': 'speech2text',
    'frontend_type': 'conv',
    'frontend': dict(input_size=80, output_size=256, in_channel=1, mid_channel=64, out_channel=128,
                     kernel_size=[[3, 3], [3, 3]], stride=[2, 2], dropout=0.0, act_func_type='relu',
                     front_end_layer_norm=False),
    'encoder_type': 'transformer',
    'encoder': dict(d_model=256, n_heads=4, d_ff=2048, n_blocks=12, pos_dropout=0.0, slf_attn_dropout=0.0,
                    ffn_dropout=0.0, residual_dropout=0.1, normalize_before=False, concat_after=False,
                    activation='glu', relative_positional=False),
    'decoder_type': 'transformer',
    'decoder': dict(vocab_size=4234, d_model=256, n_heads=4, d_ff=2048, memory_dim=256, n_blocks=6,
                    pos_dropout=0.0, slf_attn_dropout=0.0, src_attn_dropout=0.0, ffn_dropout=0.0,
                    residual_dropout=0.1, activation='glu', normalize_before=False, concat_after=False,
                    share_embedding=True),
    'ctc_weight': 0.0,
    'smoothing': 0.1,
    'encoder_output_size': 256,
}


def c2_model(residual_dropout=0.1, ctc_weight=0.0, n_enc=None, n_dec=None):
    m = copy.deepcopy(C2_MODEL)
    m['encoder']['residual_dropout'] = residual_dropout
    m['decoder']['residual_dropout'] = residual_dropout
    m['ctc_weight'] = ctc_weight
    if n_enc is not None:
        m['encoder']['n_blocks'] = n_enc
    if n_dec is not None:
        m['decoder']['n_blocks'] = n_dec
    return m


def c1_model(residual_dropout=0.0, ctc_weight=0.0):
    """BASELINE.json configs[0]: 2-layer enc/dec, d_model=64, 80x200 fbank, batch 4 (plumbing)."""
    m = copy.deepcopy(C2_MODEL)
    m['frontend'].update(output_size=64, mid_channel=32, out_channel=64)
    m['encoder'].update(d_model=64, n_heads=4, d_ff=256, n_blocks=2, residual_dropout=residual_dropout)
    m['decoder'].update(vocab_size=100, d_model=64, n_heads=4, d_ff=256, memory_dim=64, n_blocks=2,
                        residual_dropout=residual_dropout)
    m['ctc_weight'] = ctc_weight
    m['encoder_output_size'] = 64
    return m


def c1_variant(normalize_before, concat_after, ctc_weight=0.3, relative_positional=False):
    """C1 with the layer variants the shipped yamls leave off (encoder/transformer.py:16-65, decoder/transformer.py:18-90)"""
    m = c1_model(0.0, ctc_weight)
    for part in ('encoder', 'decoder'):
        m[part].update(normalize_before=normalize_before, concat_after=concat_after)
    m['encoder']['relative_positional'] = relative_positional      # the decoder hard-wires False (decoder/transformer.py:144)
    return m


def c1_lookahead(steps, ctc_weight=0.3):
    """C1 with the CTC head's look-ahead convolution (model/ctc.py:17-24; yaml key `lookahead_steps`)"""
    m = c1_model(0.0, ctc_weight)
    m['lookahead_steps'] = steps
    return m


def c1_frontend_ln(ctc_weight=0.3):
    """C1 with front_end_layer_norm=True (frontend/conv.py:128-129)"""
    m = c1_model(0.0, ctc_weight)
    m['frontend']['front_end_layer_norm'] = True
    return m


def c1_activations(enc_act, dec_act, ctc_weight=0.3):
    """C1 with the FFN activations of module/ffn.py:15-21 other than the yamls' glu"""
    m = c1_model(0.0, ctc_weight)
    m['encoder']['activation'], m['decoder']['activation'] = enc_act, dec_act
    return m


def conformer_model(small=False, residual_dropout=0.0):
    """egs/aishell/conf/conformer_baseline.yaml model section (80-d); small=True is a plumbing-size variant."""
    m = copy.deepcopy(C2_MODEL)
    if small:
        m['frontend'].update(output_size=64, mid_channel=32, out_channel=64)
        enc = dict(d_model=64, d_ff=128, cov_kernel_size=5, n_heads=4, nblocks=2)
        m['decoder'].update(vocab_size=100, d_model=64, n_heads=4, d_ff=256, memory_dim=64, n_blocks=2)
        m['encoder_output_size'] = 64
    else:
        m['frontend'].update(output_size=384, mid_channel=256, out_channel=256)
        enc = dict(d_model=384, d_ff=768, cov_kernel_size=5, n_heads=4, nblocks=12)
        m['decoder'].update(d_model=384, memory_dim=384, d_ff=768)
        m['encoder_output_size'] = 384
    enc.update(pos_dropout=0.0, slf_attn_dropout=0.0, ffn_dropout=0.0, residual_dropout=residual_dropout,
               conv_dropout=0.0, macaron_style=True, ffn_scale=0.5, conv_bias=True, activation='glu',
               positional_encoding=True, relative_positional=True)
    m['encoder_type'] = 'conformer'
    m['encoder'] = enc
    m['decoder']['residual_dropout'] = residual_dropout
    return m


def lm_config(vocab_size, d_model=256, n_heads=4, d_ff=2048, num_blocks=4):
    """egs/aishell/conf/transformer_lm.yaml model section with vocab forced equal to the ASR
    vocab (the shipped yaml's 4233 would fail the add)."""
    return dict(type='transformer_lm', vocab_size=vocab_size, d_model=d_model, n_heads=n_heads,
                d_ff=d_ff, num_blocks=num_blocks, residual_dropout=0.0, share_embedding=True,
                smoothing=0.1)


def rnn_lm_config(vocab_size, hidden_size=256, num_layers=2):
    """a `rnn_lm` model section (model/lm.py:33-60: vocab_size, hidden_size, num_layers, dropout, share_embedding, smoothing); the
    reference ships no yaml for it -- sizes follow transformer_lm.yaml's width"""
    return dict(type='recurrent_lm', vocab_size=vocab_size, hidden_size=hidden_size, num_layers=num_layers, dropout=0.0,
                share_embedding=True, smoothing=0.1)


def fill_state_dict_(sd, seed=1234):
    """Deterministic, machine-independent weight fill.  Each tensor gets its own numpy
    Generator seeded by crc32(key)^seed.  Matrices ~ U(-1/sqrt(fan_in), +); biases small;
    LayerNorm affine perturbed away from (1,0) so its gradients are exercised."""
    done = {}
    for k in sorted(sd.keys()):
        t = sd[k]
        if t.data_ptr() in done:          # tied weights (decoder embedding/output_layer)
            continue
        rng = np.random.default_rng((zlib.crc32(k.encode()) ^ seed) & 0xFFFFFFFF)
        shape = tuple(t.shape)
        if k.endswith('num_batches_tracked'):
            continue
        if k.endswith('running_var'):
            a = 1.0 + 0.2 * rng.uniform(0.0, 1.0, shape)
        elif k.endswith('running_mean'):
            a = 0.05 * rng.standard_normal(shape)
        elif k.endswith('posu') or k.endswith('posv'):
            a = 0.2 * rng.standard_normal(shape)
        elif 'norm' in k and k.endswith('weight'):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif 'norm' in k and k.endswith('bias'):
            a = 0.05 * rng.standard_normal(shape)
        elif k.endswith('bias'):
            a = 0.02 * rng.standard_normal(shape)
        elif 'embedding' in k:
            a = rng.standard_normal(shape) / math.sqrt(shape[-1])
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            a = rng.uniform(-1.0, 1.0, shape) / math.sqrt(fan_in)
        with torch.no_grad():
            t.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)))
        done[t.data_ptr()] = k
    return sd


def synthetic_batch(batch, frames, feat_dim, vocab, tgt_len, seed=0, lengths=None, tgt_lengths=None):
    """randn fbank [B,T,F], bool mask, targets [BOS] tokens [EOS] PAD*.

    lengths / tgt_lengths (lists) give the ragged variant: padded frames are zero, padded
    target positions are PAD(0); targets_length counts the EOS (data/loader.py:85,93)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, frames, feat_dim)).astype(np.float32)
    lengths = [frames] * batch if lengths is None else list(lengths)
    tgt_lengths = [tgt_len] * batch if tgt_lengths is None else list(tgt_lengths)
    mask = np.zeros((batch, frames), dtype=bool)
    tg = np.zeros((batch, tgt_len + 2), dtype=np.int64)
    for b in range(batch):
        mask[b, :lengths[b]] = True
        x[b, lengths[b]:] = 0.0
        n = tgt_lengths[b]
        tg[b, 0] = 1
        tg[b, 1:1 + n] = rng.integers(3, vocab, n)
        tg[b, 1 + n] = 1
    inputs = {'inputs': torch.from_numpy(x), 'mask': torch.from_numpy(mask),
              'inputs_length': torch.tensor(lengths, dtype=torch.int32)}
    targets = {'targets': torch.from_numpy(tg),
               'targets_length': torch.tensor([n + 1 for n in tgt_lengths], dtype=torch.int32),
               'mask': torch.from_numpy(tg != 0)}
    return inputs, targets


def flops_per_utt(model, frames, dec_rows, fwd_only=False):
    """SURVEY.md Appendix B closed-form FLOP model (MAC = 2 FLOP), transformer configs."""
    fe, en, de = model['frontend'], model['encoder'], model['decoder']
    T1 = (frames - 3) // 2 + 1
    T2 = (T1 - 3) // 2 + 1
    F1 = (fe['input_size'] + 2 - 3) // 2 + 1
    F2 = (F1 + 2 - 3) // 2 + 1
    m, c, d = fe['mid_channel'], fe['out_channel'], en['d_model']
    f = en['d_ff']
    g = 2 if en['activation'] == 'glu' else 1
    L, V = dec_rows, de['vocab_size']
    conv1 = 2 * m * T1 * F1 * 9
    conv2 = 2 * c * T2 * F2 * 9 * m
    fe_lin = 2 * T2 * (c * F2) * d
    enc = 2 * T2 * d * 3 * d + 2 * (2 * T2 * T2 * d) + 2 * T2 * d * d + 2 * T2 * d * g * f + 2 * T2 * f * d
    fd = de['d_ff']
    gd = 2 if de['activation'] == 'glu' else 1
    dec = (2 * L * d * 3 * d + 2 * (2 * L * L * d) + 2 * L * d * d
           + 2 * L * d * d + 2 * T2 * d * 2 * d + 2 * (2 * L * T2 * d) + 2 * L * d * d
           + 2 * L * d * gd * fd + 2 * L * fd * d)
    out = 2 * L * d * V
    fwd = conv1 + conv2 + fe_lin + en['n_blocks'] * enc + de['n_blocks'] * dec + out
    if model.get('ctc_weight', 0.0) > 0:
        fwd += 2 * T2 * d * V
    return fwd if fwd_only else 3 * fwd - conv1
