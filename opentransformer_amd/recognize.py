"""Decode loop: batch beam search with length penalty + LM shallow fusion, CTC greedy
(otrans/recognize/speech2text.py:6-192, recognize/base.py:26-37,104-119, recognize/ctc.py:38-58).

Same constructor arguments and return values as the reference recognizers.  Per step the decoder is
re-run over the whole prefix exactly like the reference (decoder/transformer.py:185-208; its KV cache
is a TODO there -- SURVEY.md 8f rank 1), but the scoring is fused on the device: the [B*beam, V]
log-prob tensor, the LM fusion add, both top-k's, the finished-beam masking and the prefix gather are
two kernels (otr_beam_topk, otr_beam_prune) instead of ~25 aten launches.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .nn import (BOS, EOS, PAD, LabelSmoothingLoss, PositionalEncoding, TransformerEncoderLayer)


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


LanguageModel = {'transformer_lm': TransformerLanguageModel}     # otrans/model/__init__.py:11-14 (rnn_lm not built)


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


class SpeechToTextRecognizer(Recognizer):
    """recognize/speech2text.py:6-153.  ctc_weight is accepted and unused, as in the reference."""

    def __init__(self, model, lm=None, lm_weight=0.1, ctc_weight=0.0, beam_width=5, nbest=1, max_len=50,
                 idx2unit=None, penalty=0, lamda=5, ngpu=1, apply_cache=False):
        super().__init__(model, idx2unit, lm, lm_weight, ngpu)
        self.beam_width, self.max_len, self.nbest = beam_width, max_len, nbest
        self.penalty, self.lamda, self.ctc_weight, self.lm_weight = penalty, lamda, ctc_weight, lm_weight
        self.attn_weights = {}
        self.apply_cache = False

    def encode(self, inputs, inputs_mask, cache=None):
        x, mask, fe_cache = self.model.frontend.inference(inputs, inputs_mask, None)
        memory, memory_mask, attn = self.model.encoder(x, mask)
        return memory, memory_mask, {'frontend': fe_cache}, attn

    @torch.no_grad()
    def recognize(self, inputs, inputs_mask):
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
            lm_logits = self.lm.logits(prefix) if self.lm is not None else None
            L.check(lib.otr_beam_topk(_ptr(logits, (step - 1) * V), step * V,
                                      _ptr(lm_logits, (step - 1) * V) if lm_logits is not None else None, step * V,
                                      float(self.lm_weight or 0.0), R, V, beam, _ptr(k_score), _ptr(k_idx), stream()),
                    'otr_beam_topk')
            L.check(lib.otr_beam_prune(_ptr(k_score), _ptr(k_idx), _ptr(scores[cur]), _ptr(flags[cur]), _ptr(preds[cur]),
                                       ldp, b, beam, step, EOS, _ptr(scores[cur ^ 1]), _ptr(flags[cur ^ 1]),
                                       _ptr(preds[cur ^ 1]), _ptr(n_fin), stream()), 'otr_beam_prune')
            cur ^= 1
            steps = step
            if int(n_fin.item()) == R:           # the reference syncs here every step too (speech2text.py:67)
                break
        # n-best selection on the host: B*beam scalars (speech2text.py:70-93)
        scores_h = scores[cur].cpu().view(b, beam)
        preds_h = preds[cur][:, :steps + 1].cpu().view(b, beam, -1)
        lengths = torch.sum(torch.ne(preds_h, EOS).float(), dim=-1)
        if self.penalty:
            scores_h = scores_h / torch.pow((self.lamda + lengths) / (self.lamda + 1), self.penalty)
        sorted_scores, offset = torch.sort(scores_h, dim=-1, descending=True)
        sorted_preds = torch.gather(preds_h, 1, offset.unsqueeze(-1).expand_as(preds_h))
        nbest_preds = sorted_preds[:, :min(beam, self.nbest), 1:]
        nbest_scores = sorted_scores[:, :min(beam, self.nbest)]
        return self.nbest_translate(nbest_preds), nbest_scores


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
