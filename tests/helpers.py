"""Shared test helpers: build oracle state dicts / product models from a config, probes."""
import zlib

import numpy as np
import torch

from opentransformer_amd import synthetic as syn


def probe_vector(key, numel):
    """Same crc32-seeded probe as oracle/make_golden.py:probe_vector."""
    rng = np.random.default_rng(zlib.crc32(('probe:' + key).encode()))
    return rng.standard_normal(numel).astype(np.float32)


def _linear(sd, name, out_f, in_f, bias=True):
    sd[name + '.weight'] = torch.empty(out_f, in_f)
    if bias:
        sd[name + '.bias'] = torch.empty(out_f)


def _ln(sd, name, d):
    sd[name + '.weight'] = torch.empty(d)
    sd[name + '.bias'] = torch.empty(d)


def empty_state(cfg, with_ctc=None):
    """Allocate (uninitialised) tensors with the reference's state_dict keys/shapes
     without importing the reference or the product."""
    fe, en, de = cfg['frontend'], cfg['encoder'], cfg['decoder']
    F1 = (fe['input_size'] + 2 - 3) // 2 + 1
    F2 = (F1 + 2 - 3) // 2 + 1
    f = {'conv1.conv_layer.weight': torch.empty(fe['mid_channel'], fe['in_channel'], 3, 3),
         'conv1.conv_layer.bias': torch.empty(fe['mid_channel']),
         'conv2.conv_layer.weight': torch.empty(fe['out_channel'], fe['mid_channel'], 3, 3),
         'conv2.conv_layer.bias': torch.empty(fe['out_channel'])}
    _linear(f, 'output_layer', fe['output_size'], fe['out_channel'] * F2)
    if fe.get('front_end_layer_norm', False):
        _ln(f, 'layer_norm', fe['output_size'])

    def ffn(sd, p, d, dff, act):
        _linear(sd, p + 'feed_forward.w_1', dff * 2 if act == 'glu' else dff, d)
        _linear(sd, p + 'feed_forward.w_2', d, dff)

    e = {}
    d = en['d_model']
    if cfg.get('encoder_type', 'transformer') == 'conformer':
        h = en['n_heads']
        for i in range(en['nblocks']):
            p = 'blocks.%d.' % i
            for nm in ('pre_ffn', 'post_ffn'):
                _linear(e, p + nm + '.w_1', en['d_ff'] * 2, d)
                _linear(e, p + nm + '.w_2', d, en['d_ff'])
            for nm in ('macaron_ffn_norm', 'mha_norm', 'conv_norm', 'post_ffn_norm', 'final_norm'):
                _ln(e, p + nm, d)
            e[p + 'mha.posu'] = torch.empty(1, 1, h, d // h)
            e[p + 'mha.posv'] = torch.empty(1, 1, h, d // h)
            _linear(e, p + 'mha.qvk_proj', 3 * d, d)
            _linear(e, p + 'mha.pos_proj', d, d, bias=False)
            _linear(e, p + 'conv.pointwise_conv1', 2 * d, d)
            e[p + 'conv.depthwise_conv.weight'] = torch.empty(d, 1, en['cov_kernel_size'])
            e[p + 'conv.depthwise_conv.bias'] = torch.empty(d)
            _ln(e, p + 'conv.batch_norm', d)
            e[p + 'conv.batch_norm.running_mean'] = torch.empty(d)
            e[p + 'conv.batch_norm.running_var'] = torch.empty(d)
            e[p + 'conv.batch_norm.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
            _linear(e, p + 'conv.pointwise_conv2', d, d)
    for i in range(en.get('n_blocks', 0) if cfg.get('encoder_type', 'transformer') != 'conformer' else 0):
        p = 'blocks.%d.' % i
        if en.get('relative_positional', False):      # no output_proj: the shipped constructor bug
            hh = en['n_heads']
            e[p + 'slf_attn.posu'] = torch.empty(1, 1, hh, d // hh)
            e[p + 'slf_attn.posv'] = torch.empty(1, 1, hh, d // hh)
            _linear(e, p + 'slf_attn.pos_proj', d, d, bias=False)
        else:
            _linear(e, p + 'slf_attn.output_proj', d, d)
        _linear(e, p + 'slf_attn.qvk_proj', 3 * d, d)
        ffn(e, p, d, en['d_ff'], en['activation'])
        _ln(e, p + 'norm1', d)
        _ln(e, p + 'norm2', d)
        if en.get('concat_after', False):
            _linear(e, p + 'concat_linear', d, 2 * d)
    if en.get('normalize_before', False):
        _ln(e, 'norm', d)
    dd = {}
    d = de['d_model']
    dd['embedding.weight'] = torch.empty(de['vocab_size'], d)
    for i in range(de['n_blocks']):
        p = 'blocks.%d.' % i
        _linear(dd, p + 'slf_attn.output_proj', d, d)
        _linear(dd, p + 'slf_attn.qvk_proj', 3 * d, d)
        _linear(dd, p + 'src_attn.output_proj', d, d)
        _linear(dd, p + 'src_attn.q_proj', d, d)
        _linear(dd, p + 'src_attn.vk_proj', 2 * d, de['memory_dim'])
        ffn(dd, p, d, de['d_ff'], de['activation'])
        for n in ('norm1', 'norm2', 'norm3'):
            _ln(dd, p + n, d)
        if de.get('concat_after', False):
            _linear(dd, p + 'concat_linear1', d, 2 * d)
            _linear(dd, p + 'concat_linear2', d, 2 * d)
    if de.get('normalize_before', True):
        _ln(dd, 'after_norm', d)
    if de.get('share_embedding', False):
        dd['output_layer.weight'] = dd['embedding.weight']
    else:
        dd['output_layer.weight'] = torch.empty(de['vocab_size'], d)
    dd['output_layer.bias'] = torch.empty(de['vocab_size'])
    out = {'frontend': f, 'encoder': e, 'decoder': dd}
    if with_ctc or (with_ctc is None and cfg.get('ctc_weight', 0.0) > 0):
        c = {}
        _linear(c, 'output_layer', de['vocab_size'], cfg['encoder_output_size'])
        if cfg.get('lookahead_steps', 0) > 0:
            c['lookahead_conv.weight'] = torch.empty(cfg['encoder_output_size'], 1, cfg['lookahead_steps'] + 1)
        out['ctc'] = c
    return out


def flat_named(parts):
    """{'frontend': sd, ...} -> {'frontend.key': tensor} using the reference model's attribute
    names (the CTC head is `assistor`: otrans/model/speech2text.py:33)."""
    ren = {'ctc': 'assistor'}
    return {ren.get(p, p) + '.' + k: v for p, sd in parts.items() for k, v in sd.items()}


def filled_state(cfg, seed=1234, with_ctc=None):
    """Oracle-side weights identical to what make_golden.py put into the reference model."""
    parts = empty_state(cfg, with_ctc)
    flat = flat_named(parts)
    if cfg['decoder'].get('share_embedding', False):
        # the reference's named state_dict lists both tied keys; fill_state_dict_ fills the
        # first in sorted order ('decoder.embedding.weight') and skips the alias.
        pass
    syn.fill_state_dict_(flat, seed)
    return parts


def lm_state(cfg, seed=4321):
    d, V = cfg['d_model'], cfg['vocab_size']
    sd = {'embedding.weight': torch.empty(V, d)}
    for i in range(cfg['num_blocks']):
        p = 'blocks.%d.' % i
        _linear(sd, p + 'slf_attn.output_proj', d, d)
        _linear(sd, p + 'slf_attn.qvk_proj', 3 * d, d)
        _linear(sd, p + 'feed_forward.w_1', 2 * cfg['d_ff'], d)
        _linear(sd, p + 'feed_forward.w_2', d, cfg['d_ff'])
        _ln(sd, p + 'norm1', d)
        _ln(sd, p + 'norm2', d)
    sd['output_project.weight'] = sd['embedding.weight']
    sd['output_project.bias'] = torch.empty(V)
    syn.fill_state_dict_(sd, seed)
    return sd


def rnn_lm_state(cfg, seed=4321):
    """state_dict of the reference's RecurrentLanguageModel (model/lm.py:46-60: embedding, nn.LSTM `rnn`, tied output_project),
    filled like oracle/make_golden.py fills the reference module"""
    H_, V, nl = cfg['hidden_size'], cfg['vocab_size'], cfg['num_layers']
    sd = {'embedding.weight': torch.empty(V, H_)}
    for k in range(nl):
        sd['rnn.weight_ih_l%d' % k] = torch.empty(4 * H_, H_)
        sd['rnn.weight_hh_l%d' % k] = torch.empty(4 * H_, H_)
        sd['rnn.bias_ih_l%d' % k] = torch.empty(4 * H_)
        sd['rnn.bias_hh_l%d' % k] = torch.empty(4 * H_)
    sd['output_project.weight'] = sd['embedding.weight'] if cfg.get('share_embedding', True) else torch.empty(V, H_)
    sd['output_project.bias'] = torch.empty(V)
    syn.fill_state_dict_(sd, seed)
    return sd


def require_grad(parts):
    for sd in parts.values():
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                v.requires_grad_(True)
    return parts


# ---------------------------------------------------------------------------------------------- fp16 parity runs
LOSS_SCALE = 1024.0     # static loss scale for fp16 parity runs without an optimizer (FusedAdam scales dynamically)


class loss_scaled:
    """with loss_scaled(mode, device): ... registers a static device-side loss scale in fp16 mode (ops.ScaleGradFn seeds
    the backward pass with it); unscale(model_or_tensor) divides it back out."""

    def __init__(self, mode, device='cuda'):
        self.on, self.device = mode == 'fp16', device

    def __enter__(self):
        from opentransformer_amd import ops
        if self.on:
            ops.set_loss_scale_tensor(torch.full((1,), LOSS_SCALE, device=self.device))
        return self

    def __exit__(self, *a):
        from opentransformer_amd import ops
        ops.set_loss_scale_tensor(None)

    def unscale(self, obj):
        if not self.on:
            return obj
        if isinstance(obj, torch.Tensor):
            return obj.div_(LOSS_SCALE)
        for p in obj.parameters():
            if p.grad is not None:
                p.grad.div_(LOSS_SCALE)
        return obj


def key_aware_grad_errors(names, got, ref, d_model=256):
    """Relative errors of parameter gradients, with the KEY-bias slices taken out of the norm they cannot be judged by.

    The gradient of a key bias is zero in exact arithmetic: d b_k = sum_j dk_j = sum_i q_i (sum_j dS_ij) and every row of dS sums
    to zero (softmax; module/attention.py:23-46), masked rows included.  The fp32 reference leaves round-off there, a 16-bit path
    leaves the rounding of P / dS -- a ratio of two noises says nothing.  So for `*qvk_proj.bias` (split order q, k, v:
    module/attention.py:73) and `*vk_proj.bias` (k, v: :134) the LIVE slices are compared like every other tensor, and the key slice
    is reported apart, as |got_k - ref_k| relative to the norm of the live slices of the same tensor.

    Returns (errs {name: rel error over the live part}, key_errs {name: key-slice residue / live norm})."""
    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / (b.norm() + 1e-30))
    errs, key_errs = {}, {}
    for n, a, b in zip(names, got, ref):
        if float(b.norm()) <= 1e-6:
            continue
        if n.endswith('qvk_proj.bias') and a.numel() == 3 * d_model:
            live = torch.cat([torch.arange(0, d_model), torch.arange(2 * d_model, 3 * d_model)]).to(a.device)
            key = torch.arange(d_model, 2 * d_model, device=a.device)
        elif n.endswith('vk_proj.bias') and a.numel() == 2 * d_model:
            live = torch.arange(d_model, 2 * d_model, device=a.device)
            key = torch.arange(0, d_model, device=a.device)
        else:
            errs[n] = rel(a, b)
            continue
        errs[n] = rel(a[live], b[live])
        key_errs[n] = float((a[key].double() - b[key].double()).norm() / (b[live].double().norm() + 1e-30))
    return errs, key_errs


def log_tolerance_cases(tag, record):
    """append one JSON line to gpurun_out/tolerance_cases.jsonl: which tensors needed more than the flat gradient bound, and why"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(root, 'gpurun_out', 'tolerance_cases.jsonl'), 'a') as f:
        f.write(json.dumps(dict(record, test=tag)) + '\n')
