"""CPU oracle for the otrans speech-transformer hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state_dict-in, tensors-out) fp32 CPU
restatement of the reference algorithm for the path SURVEY.md section 8 names.
It is NOT the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product path (opentransformer_amd) never
touches it and fails loudly when the HIP library is missing.

Parity pinning: the reference ships no golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the build
container by oracle/make_golden.py (which imports /root/reference) and committed
under tests/golden/.  tests/test_oracle_golden.py checks every function here
against those fixtures.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  Arithmetic is third-party torch CPU fp32 kernels, exactly as
in the reference (SURVEY.md 8c "third-party arithmetic").
"""
import math

import torch
import torch.nn.functional as F

PAD = 0   # otrans/data/__init__.py:7-12
BLK = 0
BOS = 1
EOS = 1


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------- frontend
def conv_out_len(t):
    """3x3 stride-2 conv, no time padding: otrans/frontend/conv.py:11-12,28 (time pad 0)."""
    return (t - 3) // 2 + 1


def conv2d_layer(w, b, x, mask):
    """Conv2dLayer.forward + return_output_mask: otrans/frontend/conv.py:50-83.

    relu(conv2d(x, 3x3, stride 2, pad (0,1))); mask[:, 1::2][:, :t]."""
    out = F.relu(F.conv2d(x, w, b, stride=2, padding=(0, 1)))
    mask = mask[:, 1::2][:, :out.size(2)]
    return out, mask


def conv_frontend(sd, x, mask):
    """ConvFrontEnd.forward: otrans/frontend/conv.py:131-153.

    x [B,T,F] -> [B,T2,d]; flatten order c*F2+f (conv.py:145)."""
    x = x.unsqueeze(1)
    x, mask = conv2d_layer(sd['conv1.conv_layer.weight'], sd['conv1.conv_layer.bias'], x, mask)
    x, mask = conv2d_layer(sd['conv2.conv_layer.weight'], sd['conv2.conv_layer.bias'], x, mask)
    b, c, t, f = x.shape
    x = x.permute(0, 2, 1, 3).reshape(b, t, c * f)
    x = F.linear(x, sd['output_layer.weight'], sd['output_layer.bias'])
    if 'layer_norm.weight' in sd:            # front_end_layer_norm=True (conv.py:128-129,150-151)
        x = F.layer_norm(x, (x.size(-1),), sd['layer_norm.weight'], sd['layer_norm.bias'], 1e-5)
    return x, mask


# --------------------------------------------------------------------------- primitives
def sinusoid(positions, d):
    """PositionalEncoding._embedding_from_positions: otrans/module/pos.py:30-42."""
    pos = positions.float().unsqueeze(-1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(positions.shape + (d,), dtype=torch.float32)
    pe[..., 0::2] = torch.sin(pos * div)
    pe[..., 1::2] = torch.cos(pos * div)
    return pe


def add_posenc(x):
    """PositionalEncoding.forward, scale_learnable False: otrans/module/pos.py:44-57."""
    d = x.size(-1)
    pe = sinusoid(torch.arange(x.size(1)).reshape(1, -1), d)
    return x * math.sqrt(d) + pe


def _heads(x, h):
    b, t, d = x.shape
    return x.reshape(b, t, h, d // h).transpose(1, 2)


def _context(sd, v, scores, mask):
    """BasedAttention.compute_context: otrans/module/attention.py:23-46."""
    if mask is not None:
        scores = scores.masked_fill(~mask, -float('inf'))
    w = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(w, v)
    b, n, t, dv = ctx.shape
    ctx = ctx.transpose(1, 2).reshape(b, t, n * dv)
    return F.linear(ctx, sd['output_proj.weight'], sd['output_proj.bias'])


def self_attention(sd, x, mask, h):
    """MultiHeadedSelfAttention.forward: otrans/module/attention.py:60-84.

    qvk_proj rows are ordered q,k,v (attention.py:73); mask [B,1|T,T]."""
    d = x.size(-1)
    y = F.linear(x, sd['qvk_proj.weight'], sd['qvk_proj.bias'])
    q, k, v = (y, y, y) if y.size(-1) == d else torch.split(y, d, dim=-1)      # share_qvk_proj (attention.py:71-72)
    q, k, v = _heads(q, h), _heads(k, h), _heads(v, h)
    scores = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(d // h)
    return _context(sd, v, scores, mask.unsqueeze(1) if mask is not None else None)


def cross_attention(sd, x, memory, memory_mask, h):
    """MultiHeadedCrossAttention.forward: otrans/module/attention.py:119-145.

    vk_proj rows are ordered k,v (attention.py:134)."""
    d = x.size(-1)
    q = F.linear(x, sd['q_proj.weight'], sd['q_proj.bias'])
    m = F.linear(memory, sd['vk_proj.weight'], sd['vk_proj.bias'])
    k, v = (m, m) if m.size(-1) == d else torch.split(m, d, dim=-1)              # share_vk_proj (attention.py:131-132)
    q, k, v = _heads(q, h), _heads(k, h), _heads(v, h)
    scores = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(d // h)
    return _context(sd, v, scores, memory_mask.unsqueeze(1))


def feed_forward(sd, x, activation):
    """PositionwiseFeedForward.forward: otrans/module/ffn.py:38-41 (+ _ACTIVATION :15-21)."""
    hdn = F.linear(x, sd['w_1.weight'], sd['w_1.bias'])
    if activation == 'glu':
        hdn = F.glu(hdn, dim=-1)
    elif activation == 'relu':
        hdn = F.relu(hdn)
    elif activation == 'gelu':
        hdn = F.gelu(hdn)
    elif activation == 'tanh':
        hdn = torch.tanh(hdn)
    elif activation == 'swish':
        hdn = hdn * torch.sigmoid(hdn)
    else:
        raise ValueError(activation)
    return F.linear(hdn, sd['w_2.weight'], sd['w_2.bias'])


def _ln(sd, name, x):
    return F.layer_norm(x, (x.size(-1),), sd[name + '.weight'], sd[name + '.bias'], 1e-5)


# --------------------------------------------------------------------------- encoder
def _attn_residual(sd, concat_name, x, att):
    """x + att, or with concat_after x + concat_linear(cat(x, att)) (encoder/transformer.py:51-54; no dropout there)"""
    if concat_name + '.weight' in sd:
        return x + F.linear(torch.cat((x, att), dim=-1), sd[concat_name + '.weight'], sd[concat_name + '.bias'])
    return x + att


def encoder_layer(sd, x, mask, h, activation, normalize_before=False, pos=None):
    """TransformerEncoderLayer.forward: otrans/encoder/transformer.py:41-65 (dropout off).

    The pre-norm variant takes the residual AFTER the norm (transformer.py:42-44); concat_after is on when the state
    holds a `concat_linear`."""
    if normalize_before:
        x = _ln(sd, 'norm1', x)
    if 'slf_attn.posu' in sd:       # relative_positional=True (transformer.py:23-24,47-48)
        att = relpos_self_attention(_sub(sd, 'slf_attn.'), x, mask, pos, h)
    else:
        att = self_attention(_sub(sd, 'slf_attn.'), x, mask, h)
    x = _attn_residual(sd, 'concat_linear', x, att)
    if not normalize_before:
        x = _ln(sd, 'norm1', x)
    if normalize_before:
        x = _ln(sd, 'norm2', x)
    x = x + feed_forward(_sub(sd, 'feed_forward.'), x, activation)
    if not normalize_before:
        x = _ln(sd, 'norm2', x)
    return x


def transformer_encoder(sd, x, mask, cfg):
    """TransformerEncoder.forward: otrans/encoder/transformer.py:114-134."""
    pos = None
    if cfg.get('relative_positional', False):       # transformer.py:116-120: raw inputs + relative sinusoid table
        T = x.size(1)
        pos = sinusoid(torch.arange(-(T - 1), T).reshape(1, -1), x.size(-1))
    else:
        x = add_posenc(x)
    nb = cfg.get('normalize_before', False)
    for i in range(cfg['n_blocks']):
        x = encoder_layer(_sub(sd, 'blocks.%d.' % i), x, mask.unsqueeze(1), cfg['n_heads'],
                          cfg.get('activation', 'relu'), nb, pos)
    if nb:
        x = _ln(sd, 'norm', x)
    return x, mask


# --------------------------------------------------------------------------- conformer (C4)
def relpos_self_attention(sd, x, mask, pos, h):
    """MultiHeadedSelfAttentionWithRelPos.forward + _RelPosBias: otrans/module/attention.py:196-253.

    scores = ((q+u) k^T + shift((q+v) p^T)) / sqrt(dk), p = pos_proj(sinusoid[-(T-1)..T-1]) (no bias);
    shift by gather index j - i + T - 1 (:209-215).  As shipped (ctor bug at :178) the
    module has NO output projection when slf_attn_dropout == 0: the merged heads are returned."""
    B, T, d = x.shape
    dk = d // h
    q, k, v = torch.split(F.linear(x, sd['qvk_proj.weight'], sd['qvk_proj.bias']), d, dim=-1)
    q = q.reshape(B, T, h, dk)
    k, v = _heads(k, h), _heads(v, h)
    p = F.linear(pos, sd['pos_proj.weight']).reshape(pos.size(0), -1, h, dk).transpose(1, 2)     # [1,h,2T-1,dk]
    ac = torch.matmul((q + sd['posu']).transpose(1, 2), k.transpose(-2, -1))
    bd_full = torch.matmul((q + sd['posv']).transpose(1, 2), p.transpose(-2, -1))                 # [B,h,T,2T-1]
    idx = (torch.arange(T)[None] - torch.arange(T)[:, None] + (T - 1)).reshape(1, 1, T, T)
    bd = torch.gather(bd_full, 3, idx.expand(B, h, T, T))
    scores = (ac + bd) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(~mask.unsqueeze(1), -float('inf'))
    ctx = torch.matmul(torch.softmax(scores, dim=-1), v)
    ctx = ctx.transpose(1, 2).reshape(B, T, d)
    if 'output_proj.weight' in sd:
        ctx = F.linear(ctx, sd['output_proj.weight'], sd['output_proj.bias'])
    return ctx


def conformer_conv_module(sd, x, mask, training=True):
    """ConformerConvolutionModule.forward: otrans/module/conformer.py:36-57.

    Linear C->2C, GLU, zero padded frames, depthwise Conv1d(k, pad (k-1)/2), BatchNorm1d (batch
    statistics over ALL B*T positions in training, padded ones included), swish, Linear C->C, zero
    padded frames."""
    m = mask.unsqueeze(2)
    y = F.glu(F.linear(x, sd['pointwise_conv1.weight'], sd['pointwise_conv1.bias']), dim=-1)
    y = y.masked_fill(~m, 0.0).transpose(1, 2)
    wdw = sd['depthwise_conv.weight']
    y = F.conv1d(y, wdw, sd.get('depthwise_conv.bias'), padding=(wdw.size(-1) - 1) // 2, groups=wdw.size(0))
    y = F.batch_norm(y, sd['batch_norm.running_mean'].clone(), sd['batch_norm.running_var'].clone(),
                     sd['batch_norm.weight'], sd['batch_norm.bias'], training, 0.1, 1e-5)
    y = (y * torch.sigmoid(y)).transpose(1, 2)
    y = F.linear(y, sd['pointwise_conv2.weight'], sd['pointwise_conv2.bias'])
    return y.masked_fill(~m, 0.0)


def conformer_block(sd, x, mask, pos, h, ffn_scale=0.5, training=True):
    """ConformerEncoderBlock.forward: otrans/encoder/conformer.py:75-89 (dropout 0).

    As shipped: post_ffn is NEVER applied -- only post_ffn_norm, then final_norm (:87-89)."""
    x = x + ffn_scale * feed_forward(_sub(sd, 'pre_ffn.'), _ln(sd, 'macaron_ffn_norm', x), 'glu')
    x = x + relpos_self_attention(_sub(sd, 'mha.'), _ln(sd, 'mha_norm', x), mask.unsqueeze(1), pos, h)
    x = x + conformer_conv_module(_sub(sd, 'conv.'), _ln(sd, 'conv_norm', x), mask, training)
    x = _ln(sd, 'post_ffn_norm', x)
    return _ln(sd, 'final_norm', x)


def conformer_encoder(sd, x, mask, cfg, training=True):
    """ConformerEncoder.forward: otrans/encoder/conformer.py:141-164 (relative positional)."""
    T = x.size(1)
    pos = sinusoid(torch.arange(-(T - 1), T).reshape(1, -1), x.size(-1))
    for i in range(cfg['nblocks']):
        x = conformer_block(_sub(sd, 'blocks.%d.' % i), x, mask, pos, cfg['n_heads'], cfg.get('ffn_scale', 0.5), training)
    return x, mask


# --------------------------------------------------------------------------- decoder
def decoder_layer(sd, x, tgt_mask, memory, memory_mask, h, activation, normalize_before=False):
    """TransformerDecoderLayer.forward: otrans/decoder/transformer.py:47-90 (dropout off; concat_after when the state
    holds concat_linear1/2)."""
    if normalize_before:
        x = _ln(sd, 'norm1', x)
    x = _attn_residual(sd, 'concat_linear1', x, self_attention(_sub(sd, 'slf_attn.'), x, tgt_mask, h))
    if not normalize_before:
        x = _ln(sd, 'norm1', x)
    if normalize_before:
        x = _ln(sd, 'norm2', x)
    x = _attn_residual(sd, 'concat_linear2', x, cross_attention(_sub(sd, 'src_attn.'), x, memory, memory_mask, h))
    if not normalize_before:
        x = _ln(sd, 'norm2', x)
    if normalize_before:
        x = _ln(sd, 'norm3', x)
    x = x + feed_forward(_sub(sd, 'feed_forward.'), x, activation)
    if not normalize_before:
        x = _ln(sd, 'norm3', x)
    return x


def transformer_decoder(sd, targets, memory, memory_mask, cfg):
    """TransformerDecoder.forward: otrans/decoder/transformer.py:161-183.

    Causal tril mask only, no target-pad mask (decoder/utils.py:7-11)."""
    x = add_posenc(F.embedding(targets, sd['embedding.weight']))
    L = targets.size(1)
    tgt_mask = torch.tril(torch.ones(targets.size(0), L, L)).bool()
    nb = cfg.get('normalize_before', True)
    for i in range(cfg['n_blocks']):
        x = decoder_layer(_sub(sd, 'blocks.%d.' % i), x, tgt_mask, memory, memory_mask.unsqueeze(1),
                          cfg['n_heads'], cfg.get('activation', 'relu'), nb)
    if nb:
        x = _ln(sd, 'after_norm', x)
    return F.linear(x, sd['output_layer.weight'], sd['output_layer.bias'])


def decoder_inference(sd, preds, memory, memory_mask, cfg):
    """TransformerDecoder.inference: otrans/decoder/transformer.py:185-208 (full re-forward)."""
    logits = transformer_decoder(sd, preds, memory, memory_mask, cfg)
    return F.log_softmax(logits[:, -1, :], dim=-1)


# --------------------------------------------------------------------------- losses
def label_smoothing_loss(logits, target, smoothing, padding_idx=PAD, mask=None, normalize_length=True):
    """LabelSmoothingLoss.forward: otrans/module/loss.py:21-48.

    Off-target mass eps/(V-1); rows with target==PAD (or mask True) zeroed; divided by the number of rows kept, or by all
    rows when normalize_length is False."""
    V = logits.size(-1)
    logits = logits.reshape(-1, V)
    tgt = target.reshape(-1)
    conf = torch.full_like(logits, smoothing / (V - 1))
    conf.scatter_(1, tgt.unsqueeze(1), 1.0 - smoothing)
    logp = F.log_softmax(logits, dim=-1)
    row = torch.sum(conf * (torch.log(conf) - logp), dim=-1)
    pad = tgt == padding_idx
    if mask is not None:
        pad = pad | mask.reshape(-1).bool()
    denom = torch.sum(~pad) if normalize_length else logits.size(0)
    return torch.sum(row.masked_fill(pad, 0.0)) / denom


def ctc_nll(log_probs, targets, in_len, tgt_len, blank=BLK):
    """Per-utterance CTC negative log likelihood (the algorithm behind nn.CTCLoss,
    third-party torch; used by otrans/model/ctc.py:30,50-53).

    log_probs [B,T,V] (log-softmaxed), targets [B,Lmax].  Standard log-domain alpha
    recursion over the blank-extended label sequence (Graves et al. 2006, eq. 6-8)."""
    B = log_probs.size(0)
    out = []
    for b in range(B):
        T, L = int(in_len[b]), int(tgt_len[b])
        lab = targets[b, :L]
        ext = torch.full((2 * L + 1,), blank, dtype=torch.long)
        ext[1::2] = lab
        S = 2 * L + 1
        lp = log_probs[b, :T][:, ext]                      # [T,S]
        neg = torch.full((S,), -1e30)     # finite "log 0": keeps autograd through logsumexp NaN-free
        alpha = neg.clone()
        alpha[0] = lp[0, 0]
        if S > 1:
            alpha[1] = lp[0, 1]
        skip_ok = torch.zeros(S, dtype=torch.bool)
        if S > 2:
            skip_ok[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        for t in range(1, T):
            a1 = torch.cat([neg[:1], alpha[:-1]])
            a2 = torch.cat([neg[:2], alpha[:-2]])
            a2 = torch.where(skip_ok, a2, neg)
            alpha = torch.logsumexp(torch.stack([alpha, a1, a2]), dim=0) + lp[t]
        tail = alpha[-1:] if S == 1 else alpha[-2:]
        out.append(-torch.logsumexp(tail, dim=0))
    return torch.stack(out)


def ctc_look_ahead(sd_ctc, memory):
    """CTCAssistor look-ahead (model/ctc.py:17-24,35-39): right-pad lookahead_steps zero frames, depthwise Conv1d with
    kernel lookahead_steps + 1, no bias.  Identity when the head has no lookahead_conv."""
    if 'lookahead_conv.weight' not in sd_ctc:
        return memory
    w = sd_ctc['lookahead_conv.weight']                    # [C, 1, L+1]
    x = F.pad(memory, (0, 0, 0, w.size(-1) - 1)).transpose(1, 2)
    return F.conv1d(x, w, None, 1, 0, 1, w.size(0)).transpose(1, 2)


def ctc_loss(logits, in_len, targets, tgt_len):
    """CTCAssistor.compute_loss: otrans/model/ctc.py:50-53 with nn.CTCLoss(blank=0,
    zero_infinity=True), default reduction 'mean' = mean_b(nll_b / clamp(tgt_len_b,1))."""
    nll = ctc_nll(F.log_softmax(logits, dim=-1), targets, in_len, tgt_len)
    nll = torch.where(nll > 1e29, torch.zeros_like(nll), nll)      # zero_infinity=True
    return torch.mean(nll / tgt_len.clamp(min=1).to(nll.dtype))


# --------------------------------------------------------------------------- model
def speech2text_forward(sd, params, inputs, targets):
    """SpeechToText.forward: otrans/model/speech2text.py:39-64.

    sd: {'frontend':..,'encoder':..,'decoder':..,['ctc':..]} (checkpoint layout :72-82).
    Returns (loss, aux) with aux = {'logits','memory','memory_mask',['ctc_loss']}."""
    x, mask = conv_frontend(sd['frontend'], inputs['inputs'], inputs['mask'])
    if params.get('encoder_type', 'transformer') == 'conformer':
        memory, memory_mask = conformer_encoder(sd['encoder'], x, mask, params['encoder'])
    else:
        memory, memory_mask = transformer_encoder(sd['encoder'], x, mask, params['encoder'])
    truth = targets['targets']
    logits = transformer_decoder(sd['decoder'], truth[:, :-1], memory, memory_mask, params['decoder'])
    target_out = truth[:, 1:]
    loss = label_smoothing_loss(logits, target_out, params['smoothing'])
    aux = {'logits': logits, 'memory': memory, 'memory_mask': memory_mask}
    w = params.get('ctc_weight', 0.0)
    if w > 0:
        ctc_logits = F.linear(ctc_look_ahead(sd['ctc'], memory), sd['ctc']['output_layer.weight'], sd['ctc']['output_layer.bias'])
        lctc = ctc_loss(ctc_logits, memory_mask.sum(-1), target_out, targets['targets_length'])
        aux['ctc_loss'] = lctc
        loss = (1 - w) * loss + w * lctc
    return loss, aux


def ctc_inference(sd_ctc, memory, memory_mask):
    """CTCAssistor.inference (no look-ahead): otrans/model/ctc.py:55-66."""
    logits = F.linear(ctc_look_ahead(sd_ctc, memory), sd_ctc['output_layer.weight'], sd_ctc['output_layer.bias'])
    return F.log_softmax(logits, dim=-1), memory_mask.sum(-1)


def ctc_greedy(log_probs, lengths):
    """CTCRecognizer.recognize_greedy: otrans/recognize/ctc.py:38-58
    (argmax per frame, collapse repeats, drop id 0)."""
    best = log_probs.argmax(-1)
    res = []
    for b in range(best.size(0)):
        last, out = PAD, []
        for i in range(int(lengths[b])):
            k = int(best[b, i])
            if k != last and k != PAD:
                out.append(k)
            last = k
        res.append(out)
    return res


# --------------------------------------------------------------------------- LM + beam search
def transformer_lm_predict(sd, cfg, tokens):
    """TransformerLanguageModel.predict(last_frame=True): otrans/model/lm.py:143-163.

    embed + posenc + post-norm GLU encoder layers with causal mask + tied output."""
    x = add_posenc(F.embedding(tokens, sd['embedding.weight']))
    L = tokens.size(1)
    mask = torch.tril(torch.ones(tokens.size(0), L, L)).bool()
    for i in range(cfg['num_blocks']):
        x = encoder_layer(_sub(sd, 'blocks.%d.' % i), x, mask, cfg['n_heads'], 'glu', False)
    logits = F.linear(x, sd['output_project.weight'], sd['output_project.bias'])
    return F.log_softmax(logits[:, -1, :], dim=-1)


def rnn_lm_predict(sd, cfg, tokens, hidden=None):
    """RecurrentLanguageModel.predict: otrans/model/lm.py:72-79 -- embedding, nn.LSTM(hidden_size, hidden_size, num_layers,
    batch_first, unidirectional), output_project, log_softmax.  nn.LSTM restated cell by cell (torch's documented equations, gate
    order i | f | g | o in weight_ih_l{k} / weight_hh_l{k} / bias_ih_l{k} / bias_hh_l{k}; inter-layer dropout is off in eval):
        i, f, g, o = split(W_ih x_t + b_ih + W_hh h_{t-1} + b_hh);  c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g);
        h_t = sigmoid(o) tanh(c_t)
    tokens [B, t]; hidden None (zeros) or (h [layers, B, H], c [layers, B, H]).  Returns (log_probs [B, t, V], (h_n, c_n))."""
    nl, H = cfg['num_layers'], cfg['hidden_size']
    B, T = tokens.shape
    x = F.embedding(tokens, sd['embedding.weight'])
    if hidden is None:
        h = [torch.zeros(B, H) for _ in range(nl)]
        c = [torch.zeros(B, H) for _ in range(nl)]
    else:
        h, c = [hidden[0][k] for k in range(nl)], [hidden[1][k] for k in range(nl)]
    outs = []
    for t in range(T):
        inp = x[:, t]
        for k in range(nl):
            gates = (F.linear(inp, sd['rnn.weight_ih_l%d' % k], sd['rnn.bias_ih_l%d' % k])
                     + F.linear(h[k], sd['rnn.weight_hh_l%d' % k], sd['rnn.bias_hh_l%d' % k]))
            gi, gf, gg, go = gates.chunk(4, dim=-1)
            c[k] = torch.sigmoid(gf) * c[k] + torch.sigmoid(gi) * torch.tanh(gg)
            h[k] = torch.sigmoid(go) * torch.tanh(c[k])
            inp = h[k]
        outs.append(inp)
    y = torch.stack(outs, dim=1)
    logits = F.linear(y, sd['output_project.weight'], sd['output_project.bias'])
    return F.log_softmax(logits, dim=-1), (torch.stack(h), torch.stack(c))


def lm_step_log_probs(lm, preds):
    """Recognizer.lm_decode as decode_step calls it (recognize/base.py:26-37 from recognize/speech2text.py:102-105): a transformer LM
    re-reads the whole prefix; a recurrent LM sees ONLY the last token and NO state -- decode_step passes cache['lm'], which is never
    set (the lines that would carry the hidden state forward are commented out, speech2text.py:143-150) -- so every step is one LSTM
    step from zeros."""
    sd, cfg = lm
    if cfg.get('type', 'transformer_lm') == 'recurrent_lm':
        return rnn_lm_predict(sd, cfg, preds[:, -1:], None)[0][:, 0]
    return transformer_lm_predict(sd, cfg, preds)


def beam_search(sd, params, inputs, inputs_mask, beam=5, max_len=50, penalty=0.0, lamda=5,
                nbest=1, lm=None, lm_weight=0.1):
    """SpeechToTextRecognizer.recognize/decode_step + mask_finished_*:
    otrans/recognize/speech2text.py:39-192.  Returns (token lists [B][nbest], scores [B,nbest]).

    lm: optional (lm_state_dict, lm_cfg) for shallow fusion (recognize/base.py:26-37)."""
    with torch.no_grad():
        x, mask = conv_frontend(sd['frontend'], inputs, inputs_mask)
        memory, mmask = transformer_encoder(sd['encoder'], x, mask, params['encoder'])
        B, T, D = memory.shape
        bm = memory.unsqueeze(1).repeat(1, beam, 1, 1).view(B * beam, T, D)
        bmask = mmask.unsqueeze(1).repeat(1, beam, 1).view(B * beam, T)
        preds = torch.full((B * beam, 1), BOS, dtype=torch.long)
        scores = torch.tensor([0.0] + [-float('inf')] * (beam - 1)).repeat(B).unsqueeze(1)
        flag = torch.zeros_like(scores, dtype=torch.bool)
        for _ in range(max_len):
            lp = decoder_inference(sd['decoder'], preds, bm, bmask, params['decoder'])
            if lm is not None:
                lp = lp + lm_weight * lm_step_log_probs(lm, preds)
            k_scores, k_preds = lp.topk(beam)
            # finished beams: one live branch with score 0 that emits EOS (speech2text.py:156-192)
            fin = flag.expand(-1, beam)
            first = torch.zeros_like(fin)
            first[:, 0] = True
            k_scores = torch.where(fin & ~first, torch.full_like(k_scores, -float('inf')), k_scores)
            k_scores = torch.where(fin & first, torch.zeros_like(k_scores), k_scores)
            k_preds = torch.where(fin, torch.full_like(k_preds, EOS), k_preds)
            cand = (scores + k_scores).view(B, beam * beam)
            top, off = torch.topk(cand, k=beam)
            scores = top.view(-1, 1)
            best = (torch.arange(B).view(-1, 1) * beam * beam + off).view(-1)
            tok = k_preds.reshape(-1)[best]
            src = best // beam
            preds = torch.cat([preds[src], tok.view(-1, 1)], dim=1)
            flag = (preds[:, -1] == EOS).view(-1, 1)
            if int(flag.sum()) == B * beam:
                break
        scores = scores.view(B, beam)
        preds = preds.view(B, beam, -1)
        if penalty:
            lengths = (preds != EOS).float().sum(-1)
            scores = scores / torch.pow((lamda + lengths) / (lamda + 1), penalty)
        ss, idx = torch.sort(scores, dim=-1, descending=True)
        preds = torch.gather(preds, 1, idx.unsqueeze(-1).expand_as(preds))[:, :min(beam, nbest), 1:]
        hyps = []
        for b in range(B):
            row = []
            for n in range(preds.size(1)):
                out = []
                for tkn in preds[b, n].tolist():
                    if tkn == EOS:
                        break
                    out.append(tkn)
                row.append(out)
            hyps.append(row)
        return hyps, ss[:, :min(beam, nbest)]


# --------------------------------------------------------------------------- train step (a21)
def noam_lr(step, model_size, warmup_steps, factor=1.0):
    """TransformerScheduler.get_step_lr: otrans/train/scheduler.py:137-138."""
    return factor * model_size ** (-0.5) * min(step ** (-0.5), step * warmup_steps ** (-1.5))


def dp_mean_loss_grads(sd_flat_params, loss_fn, shards):
    """nn.DataParallel semantics (otrans/train/trainer.py:64-66,206-209): loss = mean of
    per-replica losses, each normalised by its own token count => grad = (1/N) * sum_i grad_i."""
    grads = None
    losses = []
    for sh in shards:
        loss = loss_fn(sh)
        g = torch.autograd.grad(loss, sd_flat_params, allow_unused=True)
        losses.append(loss.detach())
        grads = g if grads is None else [a + b if (a is not None and b is not None) else (a if b is None else b)
                                         for a, b in zip(grads, g)]
    n = len(shards)
    return torch.stack(losses).mean(), [None if g is None else g / n for g in grads]


# --------------------------------------------------------------------------- optimizer update (trainer core)


class TrainerUpdate:
    """One parameter update of Trainer.train_one_epoch (otrans/train/trainer.py:221-234) with the baseline's Adam + Noam
    schedule (transformer_baseline.yaml:82-93): grad_norm = clip_grad_norm_(params, clip); a NaN norm skips BOTH
    scheduler.step() and optimizer.step(); Adam is torch.optim.Adam with L2 weight decay (added to the gradient).

    Scheduler quirk (scheduler.py:16-53): BaseScheduler starts at global_step 1 and initial_lr() already calls step()
    once, so the first real update is evaluated at global_step 3."""

    def __init__(self, params, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip=5.0, model_size=256, warmup_steps=12000,
                 factor=1.0):
        self.params = params
        self.b1, self.b2, self.eps, self.wd, self.clip = betas[0], betas[1], eps, weight_decay, clip
        self.noam = (model_size, warmup_steps, factor)
        self.global_step, self.t, self.lr = 2, 0, None
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self, grads):
        """grads: one tensor per parameter.  Returns (grad_norm, skipped)."""
        norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
        if torch.isnan(norm):
            return float(norm), True
        coef = torch.clamp(self.clip / (norm + 1e-6), max=1.0)
        self.global_step += 1
        self.lr = noam_lr(self.global_step, *self.noam)
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                g = g * coef + self.wd * p
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                p.addcdiv_(m, v.sqrt() / math.sqrt(bc2) + self.eps, value=-self.lr / bc1)
        return float(norm), False
