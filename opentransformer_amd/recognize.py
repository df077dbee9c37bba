"""Decode loop: batch beam search with length penalty + LM shallow fusion, CTC greedy
(otrans/recognize/speech2text.py:6-192, recognize/base.py:26-37,104-119, recognize/ctc.py:38-58).

Same constructor arguments and return values as the reference recognizers.  Two decode loops:

* apply_cache=False (the reference's behaviour; it forces this at recognize/speech2text.py:22): per step the
  decoder is re-run over the whole prefix (decoder/transformer.py:185-208), but the scoring is fused on the
  device: the [B*beam, V] log-prob tensor, the LM fusion add, both top-k's, the finished-beam masking and the
  prefix gather are two kernels (otr_beam_topk, otr_beam_prune) instead of ~25 aten launches.
* apply_cache=True (the KV cache is a TODO in the reference, README.md:13): one token per
  hypothesis per step.  Cross-attention K/V are projected once per UTTERANCE (not per beam, not per step);
  self-attention K/V live in write-once caches addressed through a beam-tree ancestor table; the position,
  prefix length and ping-pong state are device-resident, so each step is ONE hipGraph replay
  (decoder + LM + scoring + pruning).  Hypotheses are identical to the re-forward loop.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .nn import (BOS, EOS, PAD, LabelSmoothingLoss, PositionalEncoding, TransformerEncoderLayer)

_DECODE_STEP_FUSED = True   # cached beam step on otr_dec_self_step + the fused tail
_DECODE_FFN16 = True
_DECODE_FORK = True   # cached beam step: the LM chain on a side stream (CachedBeamState) -- only where the pair launches below do not apply
_DECODE_LAGGED_STOP = True   # cached beam search: the all-finished test lags one step behind the launches (CachedBeamState.run); False = sync after every step
_DECODE_PAIR = True   # cached beam step: the LM's layers ride in the decoder's launches (otr_dec_*_pair): one chain, no branch in the graph


class TransformerLanguageModel(nn.Module):
    """model/lm.py:94-171: embed + posenc + post-norm GLU encoder layers (causal) + tied output."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.model_type = 'transformer_lm'
        self.normalize_before = False
        self.smoothing = params['smoothing']
        self.vocab_size = params['vocab_size']
        self.num_blocks = params['num_blocks']
        self.embedding = nn.Embedding(self.vocab_size, params['d_model'])
        self.pos_embedding = PositionalEncoding(params['d_model'], 0.0)
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(params['n_heads'], params['d_model'], params['d_ff'], slf_attn_dropout=0.0,
                                    ffn_dropout=0.0, residual_dropout=params['residual_dropout'],
                                    normalize_before=False, concat_after=False, activation='glu')
            for _ in range(self.num_blocks)])
        self.output_project = nn.Linear(params['d_model'], self.vocab_size)
        if params['share_embedding']:
            self.output_project.weight = self.embedding.weight
        self.crit = LabelSmoothingLoss(size=self.vocab_size, smoothing=self.smoothing, padding_idx=PAD)

    def logits(self, tokens):
        x = ops.embed_posenc(tokens.contiguous(), self.embedding.weight)
        for block in self.blocks:
            x, _ = block(x, None, causal=True)
        return ops.linear(x, self.output_project.weight, self.output_project.bias)

    def forward(self, inputs, targets):
        return self.crit(self.logits(inputs['inputs']), targets['targets']), None

    def predict(self, targets, last_frame=True):
        lg = self.logits(targets)
        if last_frame:
            return ops.log_softmax(lg[:, -1, :]).unsqueeze(1)
        return ops.log_softmax(lg)

    def set_epoch(self, epoch):
        pass


class RecurrentLanguageModel(nn.Module):
    """model/lm.py:33-91: nn.Embedding -> nn.LSTM(hidden, hidden, num_layers, batch_first) -> tied Linear, same constructor dict,
    attribute names and state_dict keys (`embedding.weight`, `rnn.weight_ih_l0` ...: a reference checkpoint loads strict).  On the hot
    path it is an INFERENCE component: the recognizers fuse its log-probabilities into the beam search (recognize/base.py:26-37), and
    the reference hands it the LAST token of every hypothesis with no carried state (recognize/speech2text.py:102-105 passes
    cache['lm'], which is never set) -- `logits_last`.  `predict` is the reference's general form (any prefix length, optional
    state).  The cell runs as two otr_linear_fwd GEMMs + otr_lstm_cell per layer and step; `self.rnn` only holds the parameters.
    Training this LM (model/lm.py:63-70 through nn.LSTM's backward) is not part of the speech path: forward()
    evaluates the loss without building a graph."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.model_type = 'recurrent_lm'
        self.vocab_size = params['vocab_size']
        self.share_embedding = params['share_embedding']
        self.smoothing = params['smoothing']
        self.num_layers = params['num_layers']
        self.hidden_size = params['hidden_size']
        self.embedding = nn.Embedding(params['vocab_size'], params['hidden_size'])
        self.rnn = nn.LSTM(input_size=params['hidden_size'], hidden_size=params['hidden_size'], num_layers=params['num_layers'],
                           batch_first=True, dropout=params['dropout'], bidirectional=False)
        self.output_project = nn.Linear(params['hidden_size'], params['vocab_size'])
        if self.share_embedding:
            assert self.embedding.weight.size() == self.output_project.weight.size()
            self.output_project.weight = self.embedding.weight
        self.crit = LabelSmoothingLoss(size=self.vocab_size, smoothing=self.smoothing, padding_idx=PAD)

    def _layer_params(self, k):
        r = self.rnn
        return (getattr(r, 'weight_ih_l%d' % k), getattr(r, 'weight_hh_l%d' % k), getattr(r, 'bias_ih_l%d' % k), getattr(r, 'bias_hh_l%d' % k))

    def _step(self, x, h, c):
        """one time step through all layers; h / c: lists per layer (entries None = zero state)"""
        for k in range(self.num_layers):
            w_ih, w_hh, b_ih, b_hh = self._layer_params(k)
            ga = ops.linear(x, w_ih, b_ih)
            gb = ops.linear(h[k], w_hh, b_hh) if h[k] is not None else None          # W_hh . 0 = 0: the GEMM is skipped, b_hh stays
            h[k], c[k] = ops.lstm_cell(ga, gb, b_hh if gb is None else None, c[k])
            x = h[k]
        return x

    @torch.no_grad()
    def _run(self, tokens, hidden):
        B, T = tokens.shape
        nl = self.num_layers
        h = [None] * nl if hidden is None else [ops.attach_lp(hidden[0][k].contiguous().float(), hidden[0][k].to(ops.act_dtype()).contiguous())
                                                 if ops.is_half() else hidden[0][k].contiguous().float() for k in range(nl)]
        c = [None] * nl if hidden is None else [hidden[1][k].contiguous().float() for k in range(nl)]
        tokens = tokens.contiguous()
        outs = []
        for t in range(T):
            outs.append(self._step(ops.decode_lookup(tokens[:, t:t + 1], None, self.embedding.weight), h, c))
        y = torch.stack(outs, dim=1)
        return y, (torch.stack(h), torch.stack(c))

    def predict(self, pred, hidden=None):
        """model/lm.py:72-79: (log_probs [B, t, V], (h_n, c_n))"""
        y, hidden = self._run(pred, hidden)
        logits = ops.linear(y, self.output_project.weight, self.output_project.bias)
        return ops.log_softmax(logits), hidden

    @torch.no_grad()
    def logits_last(self, preds, pos=None, out=None):
        """un-normalised scores [R, V] of the token after preds[:, *pos] from the ZERO state: what the fused beam search adds at every
        step (recognize/base.py:35-36 with hidden = None); pos: device int32 scalar (cached search) or None = the last column"""
        if pos is None:
            preds, pos = preds[:, -1:], None
        x = ops.decode_lookup(preds, pos, self.embedding.weight)
        y = self._step(x, [None] * self.num_layers, [None] * self.num_layers)
        if out is not None:                   # recognize._PaddedOutput of output_project: [R, rows8] logits
            return out(y)
        return ops.linear(y, self.output_project.weight, self.output_project.bias)

    def forward(self, inputs, targets):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('opentransformer_amd: training the recurrent LM (nn.LSTM backward) is outside the speech hot path '
                                      '(SURVEY.md section 8); evaluate it under torch.no_grad()')
        y, _ = self._run(inputs['inputs'], None)
        logits = ops.linear(y, self.output_project.weight, self.output_project.bias)
        return self.crit(logits, targets['targets']), None

    def save_checkpoint(self, params, name):
        torch.save({'params': params, 'model': self.state_dict()}, name)

    def init_hidden_states(self, batch_size, device):
        return torch.zeros([self.num_layers, batch_size, self.hidden_size]).to(device)

    def set_epoch(self, epoch):
        pass


LanguageModel = {'rnn_lm': RecurrentLanguageModel, 'transformer_lm': TransformerLanguageModel}     # otrans/model/__init__.py:11-14


class Recognizer:
    """recognize/base.py:5-119."""

    def __init__(self, model, idx2unit=None, lm=None, lm_weight=None, ngpu=1):
        self.ngpu = ngpu
        self.model = model.eval()
        self.lm = lm.eval() if lm is not None else None
        self.idx2unit = idx2unit
        self.lm_weight = lm_weight

    def translate(self, seqs):
        results = []
        for seq in seqs:
            pred = []
            for i in seq:
                if int(i) == EOS:
                    break
                if int(i) == PAD:
                    continue
                pred.append(self.idx2unit[int(i)])
            results.append(' '.join(pred))
        return results

    def nbest_translate(self, nbest_preds):
        assert nbest_preds.dim() == 3
        results = []
        for b in range(nbest_preds.size(0)):
            nbest_list = []
            for n in range(nbest_preds.size(1)):
                pred = []
                for token in nbest_preds[b, n].tolist():
                    if token == EOS:
                        break
                    pred.append(self.idx2unit[token])
                nbest_list.append(' '.join(pred))
            results.append(nbest_list)
        return results


def _ptr(t, off=0):
    return C.c_void_p(t.data_ptr() + off * t.element_size()) if t is not None else None


class _PaddedOutput:
    """The vocabulary projection of a decode step on ROW-PADDED operands (decoder/transformer.py:153, model/lm.py:60; the training
    path gets the same from dp.FlatDataParallel's padded slots): a 4234-row weight puts the rows of the [R, 4234] logits at odd
    addresses and the product on the generic loaders -- 87 + 41 us of a 0.6 ms decode step for the decoder's and the LM's output layers
    (rocprofv3, profiles/r05_decode_kernels.txt).  Here: a 16-bit copy of the weight with the rows padded to a multiple of 8 (zeros),
    the bias likewise, the logits come out as [R, rows8] and otr_beam_topk reads them through their leading dimension.  Built once per
    (weights' version, pointer) fingerprint by the owners below: a changed checkpoint rebuilds it."""

    def __init__(self, weight, bias):
        N, K = weight.shape
        self.N, self.N8 = N, (N + 7) // 8 * 8
        hdt = ops.half_dtype()
        self.w = torch.zeros((self.N8, K), dtype=hdt, device=weight.device)
        self.w[:N].copy_(weight.detach())
        self.b = None
        if bias is not None:
            self.b = torch.zeros((self.N8,), dtype=torch.float32, device=weight.device)
            self.b[:N].copy_(bias.detach())

    def __call__(self, x):
        """x [R, K] fp32 with its 16-bit twin (or 16-bit) -> logits [R, N8] fp32 (columns >= N are zero-weight products: ignored)"""
        x16 = ops.lp_of(x)
        x2 = (x16 if x16 is not None else x).reshape(-1, self.w.shape[1])
        if x2.dtype != self.w.dtype:
            x2 = x2.to(self.w.dtype)
        return ops.linear_fwd_raw(x2.contiguous(), self.w, self.b, torch.float32)


def _padded_output(weight, bias):
    """a _PaddedOutput where it pays (16-bit mode, CUDA, output width not a multiple of 8), else None"""
    if ops.is_half() and weight.is_cuda and weight.dim() == 2 and weight.shape[0] % 8 != 0 and weight.shape[1] % 8 == 0:
        return _PaddedOutput(weight, bias)
    return None


class SpeechToTextRecognizer(Recognizer):
    """recognize/speech2text.py:6-153.  ctc_weight is accepted and unused, as in the reference."""

    def __init__(self, model, lm=None, lm_weight=0.1, ctc_weight=0.0, beam_width=5, nbest=1, max_len=50,
                 idx2unit=None, penalty=0, lamda=5, ngpu=1, apply_cache=False):
        super().__init__(model, idx2unit, lm, lm_weight, ngpu)
        self.beam_width, self.max_len, self.nbest = beam_width, max_len, nbest
        self.penalty, self.lamda, self.ctc_weight, self.lm_weight = penalty, lamda, ctc_weight, lm_weight
        self.attn_weights = {}
        self.apply_cache = bool(apply_cache)
        self.use_hipgraph = True
        self._cached_states = {}
        self.trace = None       # test hook: a list collects (prefixes [B*beam, step+1], cumulative scores [B*beam]) after every step

    def encode(self, inputs, inputs_mask, cache=None):
        x, mask, fe_cache = self.model.frontend.inference(inputs, inputs_mask, None)
        memory, memory_mask, attn = self.model.encoder(x, mask)
        return memory, memory_mask, {'frontend': fe_cache}, attn

    def _nbest(self, scores, preds, steps, b):
        """n-best selection on the host: B*beam scalars (speech2text.py:70-93)"""
        beam = self.beam_width
        scores_h = scores.cpu().view(b, beam)
        preds_h = preds[:, :steps + 1].cpu().view(b, beam, -1)
        lengths = torch.sum(torch.ne(preds_h, EOS).float(), dim=-1)
        if self.penalty:
            scores_h = scores_h / torch.pow((self.lamda + lengths) / (self.lamda + 1), self.penalty)
        sorted_scores, offset = torch.sort(scores_h, dim=-1, descending=True)
        sorted_preds = torch.gather(preds_h, 1, offset.unsqueeze(-1).expand_as(preds_h))
        nbest_preds = sorted_preds[:, :min(beam, self.nbest), 1:]
        nbest_scores = sorted_scores[:, :min(beam, self.nbest)]
        return self.nbest_translate(nbest_preds), nbest_scores

    @torch.no_grad()
    def recognize_cached(self, inputs, inputs_mask):
        memory, memory_mask, _, _ = self.encode(inputs, inputs_mask)
        b, t, _ = memory.size()
        # the captured graphs bake in the device pointers of the stand-alone 16-bit weight shadows (ops.weight_lp caches a
        # cast per parameter version): a checkpoint loaded into the same modules, averaging, ... re-allocates them, so the
        # weights' (version, pointer) fingerprint is part of the key -- stale graphs are dropped, never replayed
        fp = 0
        for mod in (self.model.decoder, self.lm):
            if mod is not None:
                for p_ in mod.parameters():
                    fp = (fp * 1000003 + p_._version * 31 + p_.data_ptr()) & 0xFFFFFFFFFFFF
        key = (b, t, self.beam_width, self.max_len, ops.get_compute_dtype(), str(memory.device),
               self.lm is not None, bool(self.use_hipgraph), fp)
        st = self._cached_states.get(key)
        if st is None:
            if len(self._cached_states) >= 4:                # a few shapes; each holds caches + two graphs
                self._cached_states.pop(next(iter(self._cached_states)))
            st = self._cached_states[key] = CachedBeamState(self, b, t, memory.device)
        st.load_memory(memory, memory_mask)
        cur, steps = st.run()
        return self._nbest(st.scores[cur], st.preds[cur], steps, b)

    @torch.no_grad()
    def recognize(self, inputs, inputs_mask):
        if self.apply_cache:
            return self.recognize_cached(inputs, inputs_mask)
        beam = self.beam_width
        lib = L.load()
        memory, memory_mask, _, _ = self.encode(inputs, inputs_mask)
        dev = memory.device
        b, t, v = memory.size()
        R = b * beam
        # tile the encoder memory over the beam, like the reference (speech2text.py:51-52)
        beam_memory = memory.unsqueeze(1).repeat([1, beam, 1, 1]).view(R, t, v)
        beam_mask = memory_mask.unsqueeze(1).repeat([1, beam, 1]).view(R, t)
        ldp = self.max_len + 2
        preds = [torch.full((R, ldp), EOS, dtype=torch.long, device=dev) for _ in range(2)]
        preds[0][:, 0] = BOS
        scores = [torch.tensor([0.0] + [-float('inf')] * (beam - 1), device=dev).repeat([b]).contiguous(),
                  torch.empty(R, device=dev)]
        flags = [torch.zeros(R, dtype=torch.uint8, device=dev), torch.zeros(R, dtype=torch.uint8, device=dev)]
        k_score = torch.empty((R, beam), dtype=torch.float32, device=dev)
        k_idx = torch.empty((R, beam), dtype=torch.long, device=dev)
        n_fin = torch.zeros(1, dtype=torch.int32, device=dev)
        cur, steps = 0, 0
        stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)    # noqa: E731
        for step in range(1, self.max_len + 1):
            prefix = preds[cur][:, :step].contiguous()
            logits, _ = self.model.decoder(prefix, beam_memory, beam_mask)       # [R, step, V] fp32
            V = logits.size(-1)
            lm_logits, lm_off, lm_ld = None, 0, step * V
            if self.lm is not None and getattr(self.lm, 'model_type', '') == 'recurrent_lm':
                lm_logits, lm_off, lm_ld = self.lm.logits_last(prefix), 0, V        # the last token, no state (recognize/base.py:35-36)
            elif self.lm is not None:
                lm_logits, lm_off = self.lm.logits(prefix), (step - 1) * V
            L.check(lib.otr_beam_topk(_ptr(logits, (step - 1) * V), step * V,
                                      _ptr(lm_logits, lm_off) if lm_logits is not None else None, lm_ld,
                                      float(self.lm_weight or 0.0), R, V, beam, _ptr(k_score), _ptr(k_idx), stream()),
                    'otr_beam_topk')
            L.check(lib.otr_beam_prune(_ptr(k_score), _ptr(k_idx), _ptr(scores[cur]), _ptr(flags[cur]), _ptr(preds[cur]),
                                       ldp, b, beam, step, EOS, _ptr(scores[cur ^ 1]), _ptr(flags[cur ^ 1]),
                                       _ptr(preds[cur ^ 1]), _ptr(n_fin), stream()), 'otr_beam_prune')
            cur ^= 1
            steps = step
            if self.trace is not None:
                self.trace.append((preds[cur][:, :step + 1].clone(), scores[cur].clone()))
            if int(n_fin.item()) == R:           # the reference syncs here every step too (speech2text.py:67)
                break
        return self._nbest(scores[cur], preds[cur], steps, b)


class CachedBeamState:
    """Static device state of the cached beam search for one (batch, T', beam, max_len) shape: ping-pong beam
    buffers, ancestor tables, write-once self-attention caches for the decoder and the LM, per-utterance
    cross-attention K/V, and the two captured step graphs (even / odd ping-pong phase)."""

    def __init__(self, rec, b, Tm, dev):
        self.rec, self.b, self.Tm, self.dev = weakref.proxy(rec), b, Tm, dev     # no rec <-> state cycle: graphs die with rec
        dec, lm = rec.model.decoder, rec.lm
        beam = rec.beam_width
        R = self.R = b * beam
        self.maxlen = rec.max_len + 1
        self.ldp = rec.max_len + 2
        adt = ops.act_dtype()
        new = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)    # noqa: E731
        self.preds = [new((R, self.ldp), torch.long) for _ in range(2)]
        self.scores = [new((R,), torch.float32) for _ in range(2)]
        self.flags = [new((R,), torch.uint8) for _ in range(2)]
        self.pos = [new((1,), torch.int32) for _ in range(2)]
        self.anc = [new((R, self.maxlen), torch.int32) for _ in range(2)]
        self.k_score = new((R, beam), torch.float32)
        self.k_idx = new((R, beam), torch.long)
        self.n_fin = new((2, 2), torch.int32)         # per phase: [finished hypotheses, the prune kernel's arrival word]; the count of a step
                                                      # stays readable while the next step runs (run: the lagged all-finished test)
        self.score0 = torch.tensor([0.0] + [-float('inf')] * (beam - 1), device=dev).repeat([b]).contiguous()
        d = dec.d_model
        self.mem_kv = [new((b, Tm, 2 * d), adt) for _ in dec.blocks]
        self.mem_mask = new((b, Tm), torch.uint8)
        self.dec_cache = [(new((R, self.maxlen, d), adt), new((R, self.maxlen, d), adt)) for _ in dec.blocks]
        self.lm_cache = None
        self.lm_recurrent = lm is not None and getattr(lm, 'model_type', '') == 'recurrent_lm'
        if lm is not None and not self.lm_recurrent:
            dl = lm.embedding.weight.shape[1]
            self.lm_cache = [(new((R, self.maxlen, dl), adt), new((R, self.maxlen, dl), adt)) for _ in lm.blocks]
        self.out_dec = _padded_output(dec.output_layer.weight, dec.output_layer.bias)
        self.out_lm = _padded_output(lm.output_project.weight, lm.output_project.bias) if lm is not None else None
        self.graphs = [None, None]
        self.nf_host = torch.zeros(2, dtype=torch.int32).pin_memory()     # pinned landing pad + events of the lagged all-finished test (run)
        self.nf_ev = [torch.cuda.Event(), torch.cuda.Event()]
        self.warm = [False, False]
        self.fused_dec = (not dec.normalize_before and adt == ops.half_dtype() and all(kv.shape[2] == 512 for kv in self.mem_kv)
                          and self._fused_stack_ok(dec.blocks, True))
        self.fused_lm = (self.lm_cache is not None and not getattr(lm, 'normalize_before', False) and adt == ops.half_dtype()
                         and self._fused_stack_ok(lm.blocks, False))
        # The LM's layers and the decoder's are two independent chains of small launches (24-80 workgroups on 256 CUs) that only
        # meet at the top-k: the LM chain runs on a side stream, forked at the start of the step and joined before the top-k
        # (in the captured graph: two parallel branches).  Its GEMMs get their own split-K workspace.
        # r06: ONE fork anywhere in a hipGraph takes the whole graph off the runtime's fast per-node path (1.6 -> ~3 us per node,
        # profiles/r06_boundary_probe.txt), which cost this 34-node step about as much as the overlap returned.  Where both stacks run on
        # the fused launches, the LM's layers are the SECOND problem of the decoder's own launches instead (otr_dec_self_step_pair,
        # otr_dec_ffn_fwd_pair, otr_dec_ln_pair: _fused_stacks_paired): the same concurrency, one chain, 23 nodes.
        self.paired = bool(_DECODE_PAIR and self.fused_dec and self.fused_lm and not self.lm_recurrent)
        self.side = torch.cuda.Stream(device=dev) if (lm is not None and _DECODE_FORK and dev.type == 'cuda' and not self.paired) else None
        self.side_ws = ops.new_workspace(dev) if self.side is not None else None

    def load_memory(self, memory, memory_mask):
        """Project the encoder memory to cross-attention K|V once per utterance and layer
        (module/attention.py:128-134 does it per hypothesis and per step), then reset the beams."""
        adt = ops.act_dtype()
        for blk, kv in zip(self.rec.model.decoder.blocks, self.mem_kv):
            a = blk.src_attn
            m_kv = ops.linear(memory, a.vk_proj.weight, a.vk_proj.bias, out_dtype=adt)
            kv.copy_(torch.cat((m_kv, m_kv), dim=-1) if a.share_vk_proj else m_kv)
        self.mem_mask.copy_(memory_mask.reshape(self.b, self.Tm))
        self.preds[0].fill_(EOS)
        self.preds[0][:, 0] = BOS
        self.scores[0].copy_(self.score0)
        self.flags[0].zero_()
        self.pos[0].zero_()
        self.n_fin.zero_()       # incl. the prune kernel's arrival word: a launch aborted mid-step would leave it poisoned for every later batch

    @staticmethod
    def _close(blk, concat_linear, norm, x, a, ctx):
        """output projection (+ the concat_after Linear) and the add+LayerNorm that closes an attention sub-layer: norm_k of
        a post-norm layer, norm_k+1 of a pre-norm one (nn.TransformerEncoderLayer)"""
        if not blk.concat_after and x.dtype == torch.float32:
            # output projection + residual + LayerNorm as ONE row-block launch (csrc/rowblock.hip), as in training
            packs = ops.proj_ln_packs(x, ctx, a.output_proj.weight, norm.weight)
            if packs is not None:
                return ops.proj_add_layernorm(x, ctx, a.output_proj.weight, a.output_proj.bias, norm.weight, norm.bias, 0.0,
                                              norm.eps, packs)
        att = ops.linear(ctx, a.output_proj.weight, a.output_proj.bias)
        if blk.concat_after:
            att = ops.linear(torch.cat((x, att), dim=-1), concat_linear.weight, concat_linear.bias)
        return ops.add_layernorm(x, att, norm.weight, norm.bias, 0.0, norm.eps)

    @staticmethod
    def _ffn(blk, norm, x):
        ff = blk.feed_forward(x)
        if blk.normalize_before:
            return ops.residual_add(x, ff, 1.0, 0.0)
        return ops.add_layernorm(x, ff, norm.weight, norm.bias, 0.0, norm.eps)

    def _stack_step(self, x, blk, cache, cur, concat_linear=None):
        """self-attention sub-layer of one layer for the new position (encoder/transformer.py:41-56,
        decoder/transformer.py:56-68)"""
        a = blk.slf_attn
        if blk.normalize_before:
            x = ops.add_layernorm(x, None, blk.norm1.weight, blk.norm1.bias, 0.0, blk.norm1.eps)
        qkv = ops.linear(x, a.qvk_proj.weight, a.qvk_proj.bias, out_dtype=ops.act_dtype())
        if a.share_qvk_proj:
            qkv = torch.cat((qkv, qkv, qkv), dim=-1)
        ctx = ops.decode_self_attention(qkv, cache[0], cache[1], self.anc[cur], self.pos[cur], a.nheads)
        return self._close(blk, concat_linear, blk.norm2 if blk.normalize_before else blk.norm1, x, a, ctx)

    def _fused_tail_ok(self, blk, with_cross):
        """can the cross-attention + FFN sub-layers (decoder) / the FFN sub-layer (LM) of this post-norm layer run on the fused decoder
        launches of training (csrc/declayer.hip)?  16-bit mode, d_model 256, 4 heads, GLU, no concat_after, beam <= 32"""
        ff = blk.feed_forward
        if (not ops._DEC_FUSED or not ops.is_half() or blk.normalize_before or blk.concat_after or ff.activation != 'glu'
                or ff.w_1.bias is None or ff.w_2.bias is None or ff.w_1.weight.shape[1] != 256
                or tuple(ff.w_2.weight.shape) != (256, ff.w_1.weight.shape[0] // 2) or self.rec.beam_width > 32):
            return False
        if ops.dec_ffn_slices(ff.w_2.weight.shape[1]) == 0 or ops.ffn_packs(ff.w_1.weight, ff.w_2.weight) is None:
            return False
        if with_cross:
            ca = blk.src_attn
            if ca.nheads != 4 or tuple(ca.q_proj.weight.shape) != (256, 256) or ca.q_proj.bias is None or ca.output_proj.bias is None:
                return False
            if ops.lin_packs(ca.q_proj.weight) is None or ops.lin_packs(ca.output_proj.weight) is None:
                return False
        return True

    def _fused_tail(self, blk, x1, kv, norm_cross, norm_ffn):
        """x1 = the layer's state after the self-attention sub-layer.  kv given (decoder): [q projection + cross-attention over the
        utterance memory + output projection] as ONE launch per layer cut along (group of 32 // beam utterances, head) -- the beam
        hypotheses of an utterance are the query rows of one attention problem, exactly the training launch with L = beam -- then
        [LayerNorm + w_1 + GLU + w_2] cut along (32-row block, hidden slice), then the closing LayerNorm (otr_dec_ln): 3 launches
        for what took 8-9 (decoder/transformer.py:70-86).  kv None (LM layer, model/lm.py:94-140): the FFN sub-layer alone, 2 launches."""
        lib, R, d = L.load(), self.R, 256
        dev, hdt, st = x1.device, ops.half_dtype(), ops._stream()
        ff = blk.feed_forward
        F = ff.w_2.weight.shape[1]
        S = ops.dec_ffn_slices(F)
        h16 = lambda *sh: torch.empty(sh, dtype=hdt, device=dev)          # noqa: E731
        f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
        xr, x16 = x1.reshape(R, d), ops.lp_of(x1).reshape(R, d)
        if kv is not None:
            ca = blk.src_attn
            beam = self.rec.beam_width
            slB, q16, ctx2, lse2 = h16(4, R, d), h16(R, d), h16(R, d), f32(self.b, 4, beam)
            lnB = ops._dec_ln(None, x16, None, 0)                          # the rows are normalised already: nothing to finish
            W = kv.shape[2]
            L.check(lib.otr_dec_cross_fwd(C.byref(lnB), self.b, beam, ops._p(ops.lin_packs(ca.q_proj.weight)[0]), ops._p(ca.q_proj.bias),
                                          ops._p(ops.lin_packs(ca.output_proj.weight)[0]), ops._p(kv), self.Tm * W, W, 0, W // 2,
                                          ops._p(self.mem_mask), self.Tm, ops._p(q16), ops._p(ctx2), ops._p(lse2), ops._p(slB), st),
                    'otr_dec_cross_fwd')
            y2, y216 = f32(R, d), h16(R, d)
            lnC = ops._dec_ln(xr, None, slB, 4, ca.output_proj.bias, norm_cross.weight, norm_cross.bias, None, 0.0, norm_cross.eps, 0,
                              y2, y216)
        else:
            y2 = xr
            lnC = ops._dec_ln(None, x16, None, 0)
        packs = ops.ffn_packs(ff.w_1.weight, ff.w_2.weight)
        slC = h16(S, R, d)
        L.check(lib.otr_dec_ffn_fwd(C.byref(lnC), R, ops._p(packs[0]), ops._p(ff.w_1.bias), ops._p(packs[1]), F, S, ops._p(slC), None, st),
                'otr_dec_ffn_fwd')
        y3, y316 = f32(R, d), h16(R, d)
        lnF = ops._dec_ln(y2, None, slC, S, ff.w_2.bias, norm_ffn.weight, norm_ffn.bias, None, 0.0, norm_ffn.eps, 0, y3, y316)
        L.check(lib.otr_dec_ln(C.byref(lnF), R, st), 'otr_dec_ln')
        return ops.attach_lp(y3, y316)

    def _fused_stack_ok(self, blocks, with_cross):
        """can every layer of this stack run the step on the fused launches (otr_dec_self_step + the fused tail)?"""
        if not _DECODE_STEP_FUSED:
            return False
        for blk in blocks:
            a = blk.slf_attn
            if (not self._fused_tail_ok(blk, with_cross) or a.nheads != 4 or a.share_qvk_proj or tuple(a.qvk_proj.weight.shape) != (768, 256)
                    or tuple(a.output_proj.weight.shape) != (256, 256) or a.qvk_proj.bias is None or a.output_proj.bias is None
                    or ops.lin_packs(a.qvk_proj.weight) is None or ops.lin_packs(a.output_proj.weight) is None):
                return False
        return True

    def _fused_stack(self, x, blocks, caches, kvs, cur):
        """Every layer of a post-norm stack for the new position on the fused launches of csrc/declayer.hip: per layer
        [LayerNorm of the layer below + q|k|v + cached self-attention + output projection] (otr_dec_self_step), for a decoder layer
        [LayerNorm + q + cross-attention + output projection] (otr_dec_cross_fwd, kvs given), [LayerNorm + w_1 + GLU + w_2]
        (otr_dec_ffn_fwd); the LayerNorm that closes the stack is otr_dec_ln.  3 launches per decoder layer, 2 per LM layer (it was 6
        and 5: decoder/transformer.py:56-86, encoder/transformer.py:41-63)."""
        lib, R, d = L.load(), self.R, 256
        dev, hdt, st = x.device, ops.half_dtype(), ops._stream()
        h16 = lambda *sh: torch.empty(sh, dtype=hdt, device=dev)          # noqa: E731
        f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
        beam = self.rec.beam_width
        yres = x.reshape(R, d)
        ln = ops._dec_ln(None, ops.lp_of(x).reshape(R, d), None, 0)       # the embedded rows: nothing to finish
        out = None

        def closes(slabs, nslab, bias, norm):
            """descriptor of the add + LayerNorm the NEXT launch finishes in its prologue, and its outputs"""
            y, y16 = f32(R, d), h16(R, d)
            return ops._dec_ln(yres, None, slabs, nslab, bias, norm.weight, norm.bias, None, 0.0, norm.eps, 0, y, y16), y, y16

        for li, blk in enumerate(blocks):
            a, ff = blk.slf_attn, blk.feed_forward
            slA = h16(4, R, d)
            L.check(lib.otr_dec_self_step(C.byref(ln), R, ops._p(ops.lin_packs(a.qvk_proj.weight)[0]), ops._p(a.qvk_proj.bias),
                                          ops._p(ops.lin_packs(a.output_proj.weight)[0]), ops._p(caches[li][0]), ops._p(caches[li][1]),
                                          ops._p(self.anc[cur]), ops._p(self.pos[cur]), self.maxlen, ops._p(slA), st), 'otr_dec_self_step')
            ln, yres, _ = closes(slA, 4, a.output_proj.bias, blk.norm1)
            if kvs is not None:
                ca, kv = blk.src_attn, kvs[li]
                slB, q16, ctx2, lse2 = h16(4, R, d), h16(R, d), h16(R, d), f32(self.b, 4, beam)
                W = kv.shape[2]
                L.check(lib.otr_dec_cross_fwd(C.byref(ln), self.b, beam, ops._p(ops.lin_packs(ca.q_proj.weight)[0]), ops._p(ca.q_proj.bias),
                                              ops._p(ops.lin_packs(ca.output_proj.weight)[0]), ops._p(kv), self.Tm * W, W, 0, W // 2,
                                              ops._p(self.mem_mask), self.Tm, ops._p(q16), ops._p(ctx2), ops._p(lse2), ops._p(slB), st),
                        'otr_dec_cross_fwd')
                ln, yres, _ = closes(slB, 4, ca.output_proj.bias, blk.norm2)
            F = ff.w_2.weight.shape[1]
            # a few row blocks only: cut the hidden units 16 ways (a workgroup streams its slice of w_1 / w_2 at what ONE CU ingests,
            # 24 workgroups of 393 KB at S = 8); otr_dec_self_step and otr_dec_ln take up to 16 slabs
            S = 16 if (_DECODE_FFN16 and F % 2048 == 0 and R <= 128) else ops.dec_ffn_slices(F)
            packs = ops.ffn_packs(ff.w_1.weight, ff.w_2.weight)
            slC = h16(S, R, d)
            L.check(lib.otr_dec_ffn_fwd(C.byref(ln), R, ops._p(packs[0]), ops._p(ff.w_1.bias), ops._p(packs[1]), F, S, ops._p(slC), None, st),
                    'otr_dec_ffn_fwd')
            ln, yres, y16 = closes(slC, S, ff.w_2.bias, blk.norm3 if kvs is not None else blk.norm2)
            out = (yres, y16)
        L.check(lib.otr_dec_ln(C.byref(ln), R, st), 'otr_dec_ln')
        return ops.attach_lp(*out)

    def _fused_stacks_paired(self, xd, xl, cur):
        """The decoder stack and the LM stack of one step in LOCKSTEP on pair launches: layer i of both in one otr_dec_self_step_pair and
        one otr_dec_ffn_fwd_pair (the decoder's cross-attention launch between them is its own), the two closing LayerNorms in one
        otr_dec_ln_pair.  The same launches on the same operands as two _fused_stack calls: bit-identical results."""
        lib, R, d = L.load(), self.R, 256
        dev, hdt, st = xd.device, ops.half_dtype(), ops._stream()
        h16 = lambda *sh: torch.empty(sh, dtype=hdt, device=dev)          # noqa: E731
        f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
        beam = self.rec.beam_width
        dec, lm = self.rec.model.decoder, self.rec.lm
        state = self

        class Walk:
            """one stack's position in the chain: the descriptor of the add + LayerNorm the NEXT launch finishes, the residual rows"""

            def __init__(self, x, blocks, caches, kvs):
                self.blocks, self.caches, self.kvs, self.li = blocks, caches, kvs, 0
                self.yres = x.reshape(R, d)
                self.keep = [x, ops.lp_of(x)]                                     # tensors the pending descriptor points at
                self.ln = ops._dec_ln(None, ops.lp_of(x).reshape(R, d), None, 0)

            def done(self):
                return self.li >= len(self.blocks)

            def closes(self, slabs, nslab, bias, norm):
                y, y16 = f32(R, d), h16(R, d)
                self.ln = ops._dec_ln(self.yres, None, slabs, nslab, bias, norm.weight, norm.bias, None, 0.0, norm.eps, 0, y, y16)
                self.keep = [self.yres, slabs, y, y16]
                self.yres, self.y16 = y, y16

            def self_item(self):
                a = self.blocks[self.li].slf_attn
                self.slA = h16(4, R, d)
                it = L.DecSelfStep()
                it.ln, it.R = self.ln, R
                it.wqkv_pack, it.bqkv = ops.lin_packs(a.qvk_proj.weight)[0].data_ptr(), a.qvk_proj.bias.data_ptr()
                it.wo_pack = ops.lin_packs(a.output_proj.weight)[0].data_ptr()
                it.kcache, it.vcache = self.caches[self.li][0].data_ptr(), self.caches[self.li][1].data_ptr()
                it.anc, it.pos, it.maxlen, it.slabs = state.anc[cur].data_ptr(), state.pos[cur].data_ptr(), state.maxlen, self.slA.data_ptr()
                return it

            def after_self(self):
                blk = self.blocks[self.li]
                self.closes(self.slA, 4, blk.slf_attn.output_proj.bias, blk.norm1)

            def cross(self):
                if self.kvs is None:
                    return
                blk = self.blocks[self.li]
                ca, kv = blk.src_attn, self.kvs[self.li]
                slB, q16, ctx2, lse2 = h16(4, R, d), h16(R, d), h16(R, d), f32(state.b, 4, beam)
                W = kv.shape[2]
                L.check(lib.otr_dec_cross_fwd(C.byref(self.ln), state.b, beam, ops._p(ops.lin_packs(ca.q_proj.weight)[0]), ops._p(ca.q_proj.bias),
                                              ops._p(ops.lin_packs(ca.output_proj.weight)[0]), ops._p(kv), state.Tm * W, W, 0, W // 2,
                                              ops._p(state.mem_mask), state.Tm, ops._p(q16), ops._p(ctx2), ops._p(lse2), ops._p(slB), st),
                        'otr_dec_cross_fwd')
                self.closes(slB, 4, ca.output_proj.bias, blk.norm2)

            def ffn_item(self):
                ff = self.blocks[self.li].feed_forward
                F = ff.w_2.weight.shape[1]
                self.S = 16 if (_DECODE_FFN16 and F % 2048 == 0 and R <= 128) else ops.dec_ffn_slices(F)
                packs = ops.ffn_packs(ff.w_1.weight, ff.w_2.weight)
                self.slC = h16(self.S, R, d)
                it = L.DecFfnFwd()
                it.ln, it.R = self.ln, R
                it.w1_pack, it.b1, it.w2_pack = packs[0].data_ptr(), ff.w_1.bias.data_ptr(), packs[1].data_ptr()
                it.F, it.S, it.slabs, it.hsave = F, self.S, self.slC.data_ptr(), None
                return it

            def after_ffn(self):
                blk = self.blocks[self.li]
                self.closes(self.slC, self.S, blk.feed_forward.w_2.bias, blk.norm3 if self.kvs is not None else blk.norm2)
                self.li += 1

        wd, wl = Walk(xd, dec.blocks, self.dec_cache, self.mem_kv), Walk(xl, lm.blocks, self.lm_cache, None)
        while not (wd.done() and wl.done()):
            live = [w for w in (wd, wl) if not w.done()]
            items = [w.self_item() for w in live]
            if len(items) == 2:
                L.check(lib.otr_dec_self_step_pair(C.byref(items[0]), C.byref(items[1]), st), 'otr_dec_self_step_pair')
            else:
                i = items[0]
                L.check(lib.otr_dec_self_step(C.byref(i.ln), R, i.wqkv_pack, i.bqkv, i.wo_pack, i.kcache, i.vcache, i.anc, i.pos, i.maxlen, i.slabs, st),
                        'otr_dec_self_step')
            for w in live:
                w.after_self()
                w.cross()
            items = [w.ffn_item() for w in live]
            if len(items) == 2:
                L.check(lib.otr_dec_ffn_fwd_pair(C.byref(items[0]), C.byref(items[1]), st), 'otr_dec_ffn_fwd_pair')
            else:
                i = items[0]
                L.check(lib.otr_dec_ffn_fwd(C.byref(i.ln), R, i.w1_pack, i.b1, i.w2_pack, i.F, i.S, i.slabs, None, st), 'otr_dec_ffn_fwd')
            for w in live:
                w.after_ffn()
        L.check(lib.otr_dec_ln_pair(C.byref(wd.ln), R, C.byref(wl.ln), R, st), 'otr_dec_ln_pair')
        return ops.attach_lp(wd.yres, wd.y16), ops.attach_lp(wl.yres, wl.y16)

    def _lm_logits(self, cur):
        """the LM's scores of the next token for every hypothesis, [R, V or V padded to 8] (speech2text.py:108-113)"""
        lm = self.rec.lm
        if self.lm_recurrent:
            return lm.logits_last(self.preds[cur], self.pos[cur], out=self.out_lm)     # one LSTM step from zeros on the last token (base.py:35-36)
        y = ops.decode_embed(self.preds[cur], self.pos[cur], lm.embedding.weight)
        if self.fused_lm and ops.lp_of(y) is not None:
            y = self._fused_stack(y, lm.blocks, self.lm_cache, None, cur)
            return self.out_lm(y) if self.out_lm is not None else ops.linear(y, lm.output_project.weight, lm.output_project.bias)
        for blk, cache in zip(lm.blocks, self.lm_cache):
            y = self._stack_step(y, blk, cache, cur, getattr(blk, 'concat_linear', None))
            if ops.lp_of(y) is not None and y.dtype == torch.float32 and self._fused_tail_ok(blk, False):
                y = self._fused_tail(blk, y, None, None, blk.norm2)
                continue
            y = self._ffn(blk, blk.norm2, y)
        return self.out_lm(y) if self.out_lm is not None else ops.linear(y, lm.output_project.weight, lm.output_project.bias)

    def step(self, cur):
        """One beam-search step (recognize/speech2text.py:95-146) reading phase `cur`, writing phase cur^1."""
        rec, lib = self.rec, L.load()
        dec, lm, beam = rec.model.decoder, rec.lm, rec.beam_width
        adt = ops.act_dtype()
        main = torch.cuda.current_stream()
        stream = C.c_void_p(main.cuda_stream)
        lm_logits = None
        paired = False
        if self.paired:
            x = ops.decode_embed(self.preds[cur], self.pos[cur], dec.embedding.weight)
            xl = ops.decode_embed(self.preds[cur], self.pos[cur], lm.embedding.weight)
            if ops.lp_of(x) is not None and ops.lp_of(xl) is not None:
                x, yl = self._fused_stacks_paired(x, xl, cur)
                lm_logits = self.out_lm(yl) if self.out_lm is not None else ops.linear(yl, lm.output_project.weight, lm.output_project.bias)
                paired = True
        if paired:
            pass
        elif self.side is not None:
            self.side.wait_stream(main)                   # fork
            with torch.cuda.stream(self.side), ops.workspace_lane(self.side_ws):
                lm_logits = self._lm_logits(cur)
        elif lm is not None:
            lm_logits = self._lm_logits(cur)
        if not paired:
            x = ops.decode_embed(self.preds[cur], self.pos[cur], dec.embedding.weight)
        fused_dec = (self.fused_dec and ops.lp_of(x) is not None) or paired
        if fused_dec and not paired:
            x = self._fused_stack(x, dec.blocks, self.dec_cache, self.mem_kv, cur)
        for blk, cache, kv in (() if fused_dec else zip(dec.blocks, self.dec_cache, self.mem_kv)):
            x = self._stack_step(x, blk, cache, cur, getattr(blk, 'concat_linear1', None))
            a = blk.src_attn
            if ops.lp_of(x) is not None and x.dtype == torch.float32 and self._fused_tail_ok(blk, True) and kv.shape[2] == 512:
                x = self._fused_tail(blk, x, kv, blk.norm2, blk.norm3)
                continue
            q = ops.linear(x, a.q_proj.weight, a.q_proj.bias, out_dtype=adt)
            # the beam hypotheses of an utterance are the query rows of ONE attention problem over its memory
            ctx = ops.CrossAttentionFn.apply(q.view(self.b, beam, -1), kv, self.mem_mask, a.nheads)
            x = self._close(blk, getattr(blk, 'concat_linear2', None), blk.norm3 if blk.normalize_before else blk.norm2,
                            x, a, ctx.view(self.R, -1))
            x = self._ffn(blk, blk.norm3, x)
        if dec.normalize_before:
            x = ops.add_layernorm(x, None, dec.after_norm.weight, dec.after_norm.bias, 0.0, dec.after_norm.eps)
        V = dec.output_layer.weight.shape[0]
        logits = self.out_dec(x) if self.out_dec is not None else ops.linear(x, dec.output_layer.weight, dec.output_layer.bias)
        ld = logits.size(-1)                             # V, or V padded to a multiple of 8 (_PaddedOutput)
        if self.side is not None:
            main.wait_stream(self.side)                   # join: the LM's logits are ready
        ld_lm = lm_logits.size(-1) if lm_logits is not None else V
        L.check(lib.otr_beam_topk(_ptr(logits), ld, _ptr(lm_logits), ld_lm, float(rec.lm_weight or 0.0), self.R, V, beam,
                                  _ptr(self.k_score), _ptr(self.k_idx), stream), 'otr_beam_topk')
        nxt = cur ^ 1
        L.check(lib.otr_beam_prune_cached(_ptr(self.k_score), _ptr(self.k_idx), _ptr(self.scores[cur]),
                                          _ptr(self.flags[cur]), _ptr(self.preds[cur]), self.ldp, self.b, beam, EOS,
                                          _ptr(self.pos[cur]), _ptr(self.pos[nxt]), _ptr(self.anc[cur]),
                                          _ptr(self.anc[nxt]), self.maxlen, _ptr(self.scores[nxt]), _ptr(self.flags[nxt]),
                                          _ptr(self.preds[nxt]), _ptr(self.n_fin[nxt]), stream), 'otr_beam_prune_cached')

    def _launch(self, cur):
        if not self.rec.use_hipgraph:
            return self.step(cur)
        if self.graphs[cur] is not None:
            return self.graphs[cur].replay()
        if not self.warm[cur]:            # first visit runs eagerly: creates weight shadows, workspace, allocator state
            self.warm[cur] = True
            return self.step(cur)
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            self.step(cur)
        self.graphs[cur] = g
        g.replay()

    def run(self):
        cur, steps = 0, 0
        if self.rec.trace is not None or not _DECODE_LAGGED_STOP:
            for step in range(1, self.rec.max_len + 1):
                self._launch(cur)
                cur ^= 1
                steps = step
                if self.rec.trace is not None:
                    self.rec.trace.append((self.preds[cur][:, :step + 1].clone(), self.scores[cur].clone()))
                if int(self.n_fin[cur, 0].item()) == self.R:  # the reference syncs here every step too (speech2text.py:67)
                    break
            return cur, steps
        # r06: the all-finished count of step s is read AFTER step s + 1 has been queued.  The per-step sync was a host round trip between
        # two replayed steps (rocprofv3: the 4 us device-to-host copy, then 18 us of nothing until the next graph's first kernel: 6 % of a
        # 0.29 ms step).  At most one step runs in vain; it reads phase A and writes phase B, so the finished state in A is what is returned
        # -- the same hypotheses and scores as the synchronous loop (tests/test_gpu_decode.py, test_gpu_round6.py).
        # The count goes home as a 4-byte copy queued behind the step on the SAME stream (4 us + a ~6 us gap per step by rocprofv3).  Measured
        # and not kept: the copy on a side stream behind an event (0.32 ms per step against 0.28: a cross-stream wait per step costs more than
        # the copy, like every fork of DESIGN.md 5.8); the copy as a node of the step's own graph (the same 0.28); the prune launch's last
        # workgroup storing the count straight to pinned host memory, no copy at all (the same 0.28 within the run-to-run noise).
        nf, ev = self.nf_host, self.nf_ev
        for step in range(1, self.rec.max_len + 1):
            self._launch(cur)
            cur ^= 1
            steps = step
            nf[cur:cur + 1].copy_(self.n_fin[cur, :1], non_blocking=True)
            ev[cur].record()
            if step > 1:
                ev[cur ^ 1].synchronize()
                if int(nf[cur ^ 1]) == self.R:
                    return cur ^ 1, step - 1
        return cur, steps


class CTCRecognizer(Recognizer):
    """recognize/ctc.py:7-58, greedy mode (the 'beam' mode needs the un-vendored ctcdecode_edited:
    SURVEY.md 8c, out of scope).  `model` must expose frontend / encoder / assistor."""

    def __init__(self, model, lm=None, lm_weight=0.1, ngram_lm=None, beam_width=5, idx2unit=None, ngpu=1,
                 mode='greedy', alpha=0.1, beta=0.0):
        super().__init__(model, idx2unit, lm, lm_weight, ngpu)
        if mode != 'greedy':
            raise NotImplementedError("CTCRecognizer mode '%s': only 'greedy' is built" % mode)
        self.beam_width, self.mode = beam_width, mode

    @torch.no_grad()
    def recognize_greedy(self, inputs, inputs_mask):
        x, mask, _ = self.model.frontend.inference(inputs, inputs_mask, None)
        memory, memory_mask, _ = self.model.encoder(x, mask)
        logits = self.model.assistor.compute_logits(memory)                 # [B,T',V]; argmax(log_softmax) == argmax
        B, T, V = logits.shape
        best_s = torch.empty((B * T, 1), dtype=torch.float32, device=logits.device)
        best_i = torch.empty((B * T, 1), dtype=torch.long, device=logits.device)
        L.check(L.load().otr_beam_topk(_ptr(logits), V, None, 0, 0.0, B * T, V, 1, _ptr(best_s), _ptr(best_i),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_beam_topk')
        best = best_i.view(B, T).cpu()
        length = memory_mask.sum(-1).cpu()
        results = []
        for b in range(B):
            pred, last_k = [], PAD
            for i in range(int(length[b])):
                k = int(best[b, i])
                if k != last_k and k != PAD:
                    pred.append(k)
                last_k = k
            results.append(pred)
        return results

    def recognize(self, inputs, inputs_mask):
        return self.translate(self.recognize_greedy(inputs, inputs_mask))


def build_recognizer(model_type, model, lm, args, idx2unit):
    """recognize/__init__.py:5-16."""
    if model_type == 'speech2text':
        return SpeechToTextRecognizer(model=model, lm=lm, lm_weight=args.lm_weight, ctc_weight=args.ctc_weight,
                                      beam_width=args.beam_width, nbest=args.nbest, max_len=args.max_len,
                                      idx2unit=idx2unit, penalty=args.penalty, lamda=args.lamda, ngpu=args.ngpu)
    if model_type == 'ctc':
        return CTCRecognizer(model=model, lm=lm, lm_weight=args.lm_weight, ngram_lm=args.ngram_lm,
                             beam_width=args.beam_width, idx2unit=idx2unit, ngpu=args.ngpu, mode=args.mode,
                             alpha=args.alpha, beta=args.beta)
    raise NotImplementedError
