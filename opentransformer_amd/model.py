"""Model assembly: SpeechToText (otrans/model/speech2text.py:15-90) and the CTC head
(otrans/model/ctc.py:12-66) built from the drop-in registries of this package."""
import torch
import torch.nn as nn

from . import ops
from .nn import (BLK, ConformerEncoder, ConvFrontEnd, LabelSmoothingLoss, TransformerDecoder, TransformerEncoder,
                 _unsupported)

BuildFrontEnd = {'conv': ConvFrontEnd}                 # otrans/frontend/__init__.py:8-12
BuildEncoder = {'transformer': TransformerEncoder, 'conformer': ConformerEncoder}     # otrans/encoder/__init__.py:10-13
BuildDecoder = {'transformer': TransformerDecoder}     # otrans/decoder/__init__.py:8-10


class CTCAssistor(nn.Module):
    """model/ctc.py:12-66: Linear(d -> V) + log_softmax + CTC loss (blank 0, zero_infinity, 'mean')."""

    def __init__(self, hidden_size, vocab_size, blank=BLK, lookahead_steps=-1):
        super().__init__()
        self.lookahead_steps, self.apply_look_ahead, self.blank = lookahead_steps, lookahead_steps > 0, blank
        if self.apply_look_ahead:
            if lookahead_steps > 6 or hidden_size % 4:
                _unsupported('CTCAssistor lookahead_steps > 6 (the depthwise-conv kernels hold at most 7 taps)')
            self.lookahead_conv = nn.Conv1d(hidden_size, hidden_size, lookahead_steps + 1, padding=0, stride=1, bias=False,
                                            groups=hidden_size)         # parameter container only (model/ctc.py:17-24)
        self.output_layer = nn.Linear(hidden_size, vocab_size)

    def _look_ahead(self, memory):
        return ops.LookaheadConvFn.apply(memory, self.lookahead_conv.weight) if self.apply_look_ahead else memory

    def compute_logits(self, enc_states):
        return ops.linear(enc_states, self.output_layer.weight, self.output_layer.bias)

    def forward(self, memory, memory_length=None, targets=None, tgt_length=None, return_logits=False):
        logits = self.compute_logits(self._look_ahead(memory))
        if return_logits:
            return logits
        return self.compute_loss(logits, memory_length, targets, tgt_length)

    def compute_loss(self, logits, enc_length, targets, targets_length):
        return ops.CTCLossFn.apply(logits, targets, enc_length, targets_length, self.blank)

    def inference(self, memory, memory_mask):
        logits = self.compute_logits(self._look_ahead(memory))
        memory_length = torch.sum(memory_mask.squeeze(1) if memory_mask.dim() == 3 else memory_mask, dim=-1)
        return ops.log_softmax(logits), memory_length


class SpeechToText(nn.Module):
    """model/speech2text.py:15-90.  forward(inputs: dict, targets: dict) -> (loss, aux | None)."""

    def __init__(self, params):
        super().__init__()
        self.frontend = BuildFrontEnd[params['frontend_type']](**params['frontend'])
        self.encoder = BuildEncoder[params['encoder_type']](**params['encoder'])
        self.decoder = BuildDecoder[params['decoder_type']](**params['decoder'])
        self.crit = LabelSmoothingLoss(size=params['decoder']['vocab_size'], smoothing=params['smoothing'])
        self.ctc_weight = params['ctc_weight']
        if self.ctc_weight > 0.0:
            self.assistor = CTCAssistor(hidden_size=params['encoder_output_size'],
                                        vocab_size=params['decoder']['vocab_size'],
                                        lookahead_steps=params['lookahead_steps'] if 'lookahead_steps' in params else 0)

    def forward(self, inputs, targets):
        enc_inputs, enc_mask = inputs['inputs'], inputs['mask']
        truth, truth_length = targets['targets'], targets['targets_length']
        enc_inputs, enc_mask = self.frontend(enc_inputs, enc_mask)
        memory, memory_mask, _ = self.encoder(enc_inputs, enc_mask)
        memory = ops.early_mark(memory, self) # everything behind this point finishes its backward before the encoder's starts (dp.py)
        if self.ctc_weight > 0 or not truth.is_cuda or truth.stride(-1) != 1:
            shifted = torch.stack((truth[:, :-1], truth[:, 1:]))     # decoder input | loss target (speech2text.py:53,57 clones each) in one launch
            dec_in, target_out = shifted[0], shifted[1]
        else:
            # the embedding and the loss kernels read the two shifted VIEWS of the target matrix in place (row stride L + 1): no launch
            dec_in, target_out = truth[:, :-1], truth[:, 1:]
        logits, _ = self.decoder(dec_in, memory, memory_mask)
        if self.ctc_weight <= 0 and isinstance(self.crit, LabelSmoothingLoss):
            # the loss is the root of the backward pass: the factor it is seeded with (fp16 loss scale, ops.scale_loss_grad) rides in
            # the loss launch's gradient instead of two scalar-multiply launches
            self.crit._otr_grad_scale = ops.loss_scale_of(self)
            try:
                return self.crit(logits, target_out), None
            finally:
                self.crit._otr_grad_scale = None
        loss = self.crit(logits, target_out)
        if self.ctc_weight > 0:
            loss_ctc = self.compute_ctc_loss(memory, memory_mask, target_out, truth_length)
            # the reference returns {'CTCLoss': loss_ctc.item()} (a host sync per step); we keep the tensor
            total = (1 - self.ctc_weight) * loss + self.ctc_weight * loss_ctc
            return ops.scale_loss_grad(total, self), {'CTCLoss': loss_ctc.detach()}
        return ops.scale_loss_grad(loss, self), None      # identity unless a loss scale is registered (fp16 training)

    def compute_ctc_loss(self, memory, memory_mask, targets_out, targets_length):
        memory_length = torch.sum(memory_mask, dim=-1)
        return self.assistor(memory, memory_length, targets_out, targets_length)

    def save_checkpoint(self, params, name):
        checkpoint = {'params': params, 'frontend': self.frontend.state_dict(), 'encoder': self.encoder.state_dict(),
                      'decoder': self.decoder.state_dict()}
        if self.ctc_weight > 0.0:
            checkpoint['ctc'] = self.assistor.state_dict()
        torch.save(checkpoint, name)

    def load_model(self, chkpt):
        self.frontend.load_state_dict(chkpt['frontend'])
        self.encoder.load_state_dict(chkpt['encoder'])
        self.decoder.load_state_dict(chkpt['decoder'])
        if 'ctc' in chkpt and self.ctc_weight > 0.0:
            self.assistor.load_state_dict(chkpt['ctc'])

    def set_epoch(self, epoch):
        pass


class CTCModel(nn.Module):
    """model/ctc.py:69-140: frontend + encoder + CTC head.  forward(inputs, targets) -> (loss, None) with the labels
    truth[:, 1:-1] and lengths targets_length - 1 (:95; SpeechToText's joint CTC term keeps the EOS instead).  The
    reference's inference / recognize / ts_forward call the encoder without the frontend and with a length where a mask
    belongs (:98-121, broken as shipped); inference() here is the working form: frontend -> encoder ->
    CTCAssistor.inference, which is what recognize.CTCRecognizer consumes."""

    def __init__(self, params):
        super().__init__()
        self.frontend = BuildFrontEnd[params['frontend_type']](**params['frontend'])
        self.encoder = BuildEncoder[params['encoder_type']](**params['encoder'])
        self.assistor = CTCAssistor(hidden_size=params['encoder_output_size'], vocab_size=params['vocab_size'],
                                    lookahead_steps=params['lookahead_steps'] if 'lookahead_steps' in params else -1)

    def forward(self, inputs, targets):
        truth, truth_length = targets['targets'], targets['targets_length']
        enc_inputs, enc_mask = self.frontend(inputs['inputs'], inputs['mask'])
        memory, memory_mask, _ = self.encoder(enc_inputs, enc_mask)
        memory_length = torch.sum(memory_mask, dim=-1)
        loss = self.assistor(memory, memory_length, truth[:, 1:-1].contiguous(), truth_length.add(-1))
        return ops.scale_loss_grad(loss, self), None

    def inference(self, inputs, inputs_mask):
        enc_inputs, enc_mask = self.frontend(inputs, inputs_mask)
        memory, memory_mask, _ = self.encoder(enc_inputs, enc_mask)
        return self.assistor.inference(memory, memory_mask)

    recognize = inference

    def save_checkpoint(self, params, name):
        torch.save({'params': params, 'frontend': self.frontend.state_dict(), 'encoder': self.encoder.state_dict(),
                    'ctc': self.assistor.state_dict()}, name)

    def load_model(self, chkpt):
        self.frontend.load_state_dict(chkpt['frontend'])
        self.encoder.load_state_dict(chkpt['encoder'])
        self.assistor.load_state_dict(chkpt['ctc'])

    def set_epoch(self, epoch):
        pass


End2EndModel = {'ctc': CTCModel, 'speech2text': SpeechToText}           # otrans/model/__init__.py:6-9
