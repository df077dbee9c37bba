"""Drop-in modules for the otrans frontend / encoder / decoder / module stack.

Same class names, constructor kwargs, forward signatures and state_dict keys as the reference
(SURVEY.md 8b), so a reference checkpoint loads unchanged and the reference registries
(BuildFrontEnd / BuildEncoder / BuildDecoder) can point here.  torch.nn.Linear / Conv2d / LayerNorm /
Embedding objects are used only as *parameter containers* (identical shapes, names and default
initialisation); their forward() is never called -- all math goes through opentransformer_amd.ops,
i.e. through libotrans_hip.so.

Transformer layers come in the reference's four variants: post-norm (the shipped yamls) or its non-standard pre-norm
(`normalize_before=True`: the residual is taken AFTER the norm, encoder/transformer.py:42-44), each with or without
`concat_after` (a Linear over cat(x, attention) instead of dropout(attention)).

Dropout inside attention (on the projected context, module/attention.py:46), the FFN hidden (module/ffn.py:40) and the
frontend's conv layers (frontend/conv.py:66) runs through ops.dropout (counter RNG, mask regenerated in backward); with p = 0
(the shipped AISHELL yamls) nothing is launched.  The Conformer convolution module takes a dropout rate but, like the reference
(module/conformer.py:58-123), never applies it.  Not built (constructor
raises NotImplementedError): in_channel != 1, pos_dropout > 0 (which in the reference silently switches the formula).
"""
import math

import torch
import torch.nn as nn

from . import ops

_MASK_FOLD = True  # ... and the convolution branch's row mask applied by that launch
_LN2 = True                   # ... and post_ffn_norm + final_norm in one launch each way
_POS_TABLES = True    # ConformerEncoder: the relative-position tables of all blocks in two batched launches (ops.relpos_tables)
_LN16 = True          # ... the inner LayerNorms' outputs leave as the 16-bit tensor alone (ops.ResidualLnFn lp_only): 16-bit input gradients back
_LN3 = True           # ... and the next block's macaron LayerNorm in the closing launches of the block below (ops.ResidualLn3Fn)
_RES_LN = True     # ConformerEncoderBlock: residual adds fused into the LayerNorms that follow them

PAD, BLK, BOS, EOS = 0, 0, 1, 1       # otrans/data/__init__.py:7-12


def _unsupported(what):
    raise NotImplementedError('opentransformer_amd: %s is not built on the HIP path yet' % what)


# ------------------------------------------------------------------------------------- frontend
class Conv2dLayer(nn.Module):
    """Parameter container mirroring frontend/conv.py:15-83 (the compute is fused in ConvFrontEnd)."""

    def __init__(self, input_size, in_channel, out_channel, kernel_size, stride, dropout=0.1, batch_norm=False,
                 residual=False, act_func_type='relu'):
        super().__init__()
        if list(kernel_size) != [3, 3] if not isinstance(kernel_size, int) else kernel_size != 3:
            _unsupported('Conv2dLayer kernel_size != 3x3')
        if stride != 2:
            _unsupported('Conv2dLayer stride != 2')
        if batch_norm or residual:
            _unsupported('Conv2dLayer batch_norm/residual')
        self.input_size, self.in_channel, self.out_channel = input_size, in_channel, out_channel
        self.kernel_size, self.stride, self.padding = kernel_size, stride, (0, 1)
        self.conv_layer = nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=2, padding=(0, 1))
        self.output_size = (input_size + 2 - 3) // 2 + 1     # cal_width_dim_2d, conv.py:11-12

    @staticmethod
    def return_output_mask(mask, t):
        return mask[:, 1::2][:, :t]                          # conv.py:78-83


class ConvFrontEnd(nn.Module):
    """frontend/conv.py:86-158.  forward(x [B,T,F], mask [B,T]) -> (x [B,T',d], mask [B,T'])."""

    def __init__(self, input_size, output_size, in_channel=1, mid_channel=32, out_channel=128,
                 kernel_size=[[3, 3], [3, 3]], stride=[2, 2], dropout=0.0, act_func_type='relu',
                 front_end_layer_norm=False):
        super().__init__()
        if in_channel != 1:
            _unsupported('ConvFrontEnd in_channel != 1')
        self.dropout = dropout
        self.kernel_size, self.stride, self.output_size = kernel_size, stride, output_size
        self.act_func_type, self.front_end_layer_norm = act_func_type, front_end_layer_norm
        self.conv1 = Conv2dLayer(input_size, in_channel, mid_channel, kernel_size[0], stride[0], dropout)
        self.conv2 = Conv2dLayer(self.conv1.output_size, mid_channel, out_channel, kernel_size[1], stride[1], dropout)
        self.conv_output_size = self.conv2.output_size * out_channel
        self.output_layer = nn.Linear(self.conv_output_size, output_size)
        if front_end_layer_norm:
            self.layer_norm = nn.LayerNorm(output_size)           # frontend/conv.py:128-129,150-151

    def regrouped_weights(self):
        """[(parameter, (A, R, S))]: the kernels read these weights as [A, S, R] (channel-last) where the reference stores
        [A, R, S]: conv2's taps [C2, C1, 3*3] -> [C2, 3*3, C1], and the output Linear's columns from c*F2+f (the reference's
        channel-first flatten, frontend/conv.py:145) to f*C2+c.  dp.FlatDataParallel keeps such 16-bit shadows fresh in its
        one transpose launch per optimizer step; without it ops regroups on every forward."""
        c2 = self.conv2.conv_layer
        C2, C1, F2 = c2.out_channels, c2.in_channels, self.conv2.output_size
        return [(c2.weight, (C2, C1, 9)), (self.output_layer.weight, (self.output_layer.out_features, C2, F2))]

    def forward(self, x, mask):
        c1, c2 = self.conv1.conv_layer, self.conv2.conv_layer
        C2, F2 = c2.out_channels, self.conv2.output_size
        if self.dropout and self.training:
            # Conv2dLayer = dropout(relu(conv(x))) (frontend/conv.py:63-66): a mask between the two convolutions breaks their
            # fused chain, so this (never shipped) configuration runs them one at a time
            act2 = ops.conv_subsample_with_dropout(x, c1.weight, c1.bias, c2.weight, c2.bias, self.dropout)
        else:
            act2 = ops.ConvSubsampleFn.apply(x, c1.weight, c1.bias, c2.weight, c2.bias)     # [B,T2,F2*C2] channel-last
        y = ops.linear(act2, self.output_layer.weight, self.output_layer.bias, perm=(C2, F2))
        t1 = (x.size(1) - 3) // 2 + 1
        mask = Conv2dLayer.return_output_mask(mask, t1)
        mask = Conv2dLayer.return_output_mask(mask, act2.size(1))
        if self.front_end_layer_norm:
            y = ops.add_layernorm(y, None, self.layer_norm.weight, self.layer_norm.bias, 0.0, self.layer_norm.eps)
        return y, mask

    def inference(self, x, mask, cache):
        x, mask = self.forward(x, mask)
        return x, mask, cache


# ------------------------------------------------------------------------------------- primitives
class PositionalEncoding(nn.Module):
    """module/pos.py:11-72 (scale_learnable=False branch: x*sqrt(d) + PE).

    NB the reference's callers pass pos_dropout into the scale_learnable slot
    (encoder/transformer.py:102), so pos_dropout > 0 would switch formula; we refuse it."""

    def __init__(self, emb_dim, scale_learnable=False, dropout=0.0):
        super().__init__()
        if scale_learnable or dropout:
            _unsupported('PositionalEncoding scale_learnable / dropout (pos_dropout must be 0.0)')
        self.emb_dim = emb_dim
        self.xscale = math.sqrt(emb_dim)

    def forward(self, x, mask=None):
        """mask (an extra keyword the reference's callers never pass): the [B, T] key mask of the rows; its uint8 cast leaves the same
        launch (ops.PosEncFn) instead of one of its own"""
        return ops.posenc(x, mask), None


def _key_mask(mask, B, Tk):
    """reference masks are bool [B,1,Tk] (key mask) -> uint8 [B,Tk]"""
    if mask is None:
        return None
    if mask.dim() == 3:
        if mask.size(1) != 1:
            raise ValueError('expected a [B,1,T] key mask')
        mask = mask[:, 0]
    return ops._mask_u8(mask, B, Tk)


class MultiHeadedSelfAttention(nn.Module):
    """module/attention.py:49-84."""

    def __init__(self, n_heads, d_model, dropout_rate=0.0, share_qvk_proj=False):
        super().__init__()
        self.dropout_rate = dropout_rate          # on the projected context (module/attention.py:46)
        self.d_model, self.nheads, self.d_k = d_model, n_heads, d_model // n_heads
        self.share_qvk_proj = share_qvk_proj
        self.output_proj = nn.Linear(d_model, d_model)
        self.qvk_proj = nn.Linear(d_model, d_model if share_qvk_proj else d_model * 3)

    def context(self, x, mask, causal=False, link=None):
        """softmax(QK^T/sqrt(dk)) V merged over heads, before output_proj (act dtype)."""
        B, T, _ = x.shape
        if mask is not None and mask.dim() == 3 and mask.size(1) == T and T > 1:
            # decoder-style [B,T,T] mask: only the causal tril mask is built (decoder/utils.py:7-11)
            if not bool(torch.equal(mask, torch.tril(torch.ones_like(mask)))):
                _unsupported('arbitrary [B,T,T] attention masks (only key masks and the causal mask)')
            mask, causal = None, True
        qkv = ops.linear(x, self.qvk_proj.weight, self.qvk_proj.bias, out_dtype=ops.act_dtype(), link=link)
        if self.share_qvk_proj:          # query = key = value = the one projection (module/attention.py:71-72); rare: packed by copy
            qkv = torch.cat((qkv, qkv, qkv), dim=-1)
        ctx = ops.SelfAttentionFn.apply(qkv, _key_mask(mask, B, T), self.nheads, causal)
        if qkv.is_contiguous() and ctx.requires_grad:
            ctx._otr_touch = qkv          # what the attention's backward launch reads first (ops.ProjLnFn has the launch before it touch it)
            wp = ops.lin_packs(self.qvk_proj.weight) if not self.share_qvk_proj else None
            ctx._otr_touch_w = wp[1] if wp is not None else None      # the input-gradient pack the launch after that one streams
        return ctx

    def forward(self, x, mask, causal=False, defer_bias=False, link=None):
        """defer_bias: the caller feeds the result to _post_norm(..., a_bias=self.output_proj.bias); link: ops.ResidualLink
        shared with that _post_norm."""
        ctx = self.context(x, mask, causal, link)
        if self.dropout_rate and self.training:   # dropout(output_proj(ctx)): the bias is inside the mask, so it cannot be deferred
            out = ops.linear(ctx, self.output_proj.weight, self.output_proj.bias, out_dtype=ops.act_dtype() if defer_bias else None)
            return ops.dropout(out, self.dropout_rate), None
        # a branch that feeds the fused add+LayerNorm is written in the activation dtype (bf16 in bf16 mode, like every
        # other GEMM output; the fp32 residual stream adds it in fp32): its gradient then comes back in bf16 too
        return ops.linear(ctx, self.output_proj.weight, self.output_proj.bias, defer_bias=defer_bias,
                          out_dtype=ops.act_dtype() if defer_bias else None), None

    def inference(self, x, mask, cache=None):
        out, w = self.forward(x, mask)
        return out, w, cache


class MultiHeadedCrossAttention(nn.Module):
    """module/attention.py:107-173."""

    def __init__(self, n_heads, d_model, memory_dim, dropout_rate=0.0, share_vk_proj=False):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.d_model, self.nheads, self.d_k = d_model, n_heads, d_model // n_heads
        self.share_vk_proj = share_vk_proj
        self.output_proj = nn.Linear(d_model, d_model)
        self.q_proj = nn.Linear(d_model, d_model)
        self.vk_proj = nn.Linear(memory_dim, d_model if share_vk_proj else d_model * 2)

    def forward(self, query, memory, memory_mask, defer_bias=False, link=None, kv_all=None):
        """kv_all = (projection of the memory by ALL decoder layers' vk_proj in one GEMM, this layer's slice index, shared
        bookkeeping): TransformerDecoder.forward builds it once per pass (ops.CrossKVAllFn)."""
        ctx = self.context(query, memory, memory_mask, link, kv_all)
        if self.dropout_rate and self.training:   # dropout(output_proj(ctx)): bias inside the mask, not deferrable
            out = ops.linear(ctx, self.output_proj.weight, self.output_proj.bias, out_dtype=ops.act_dtype() if defer_bias else None)
            return ops.dropout(out, self.dropout_rate), None
        return ops.linear(ctx, self.output_proj.weight, self.output_proj.bias, defer_bias=defer_bias,
                          out_dtype=ops.act_dtype() if defer_bias else None), None

    def context(self, query, memory, memory_mask, link=None, kv_all=None):
        """merged-head attention context before output_proj (act dtype)"""
        B, T, _ = memory.shape
        adt = ops.act_dtype()
        q = ops.linear(query, self.q_proj.weight, self.q_proj.bias, out_dtype=adt, link=link)
        if kv_all is not None:
            ctx = ops.CrossAttentionSliceFn.apply(q, kv_all[0], _key_mask(memory_mask, B, T), self.nheads, kv_all[1], kv_all[2])
        else:
            kv = ops.linear(memory, self.vk_proj.weight, self.vk_proj.bias, out_dtype=adt)
            if self.share_vk_proj:           # key = value (module/attention.py:131-132)
                kv = torch.cat((kv, kv), dim=-1)
            ctx = ops.CrossAttentionFn.apply(q, kv, _key_mask(memory_mask, B, T), self.nheads)
        return ctx

    def inference(self, query, memory, memory_mask, cache=None):
        out, w = self.forward(query, memory, memory_mask)
        return out, w, cache


class PositionwiseFeedForward(nn.Module):
    """module/ffn.py:24-41."""

    def __init__(self, d_model, d_ff, dropout, activation='relu'):
        super().__init__()
        assert activation in ('relu', 'gelu', 'glu', 'tanh', 'swish')
        self.dropout = dropout                    # on the hidden, between the activation and w_2 (module/ffn.py:40)
        self.activation = activation
        self.w_1 = nn.Linear(d_model, d_ff * 2 if activation == 'glu' else d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)

    def forward(self, x, defer_bias=False, link=None, branch=False):
        """branch=True: the result feeds a residual add that takes the branch in the activation type (its gradient then comes
        back in that type too: the w_2 weight gradient joins the deferred launch and the GLU backward runs in the GEMM epilogue)"""
        odt = ops.act_dtype() if (defer_bias or branch) else None
        if self.dropout and self.training:        # w_2(dropout(act(w_1 x))): the mask sits between the two GEMMs -> unfused chain
            h = ops.linear(x, self.w_1.weight, self.w_1.bias, relu=self.activation == 'relu', out_dtype=ops.act_dtype(), link=link)
            if self.activation == 'glu':
                h = ops.GLUFn.apply(h)
            elif self.activation != 'relu':
                h = ops.activation(h, self.activation)
            h = ops.dropout(h, self.dropout)
            return ops.linear(h, self.w_2.weight, self.w_2.bias, defer_bias=defer_bias, out_dtype=odt)
        if self.activation == 'glu':
            return ops.FeedForwardGLUFn.apply(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias,
                                              defer_bias, odt if odt is not None else torch.float32, link)
        h = ops.linear(x, self.w_1.weight, self.w_1.bias, relu=self.activation == 'relu', out_dtype=ops.act_dtype(), link=link)
        if self.activation != 'relu':
            h = ops.activation(h, self.activation)
        return ops.linear(h, self.w_2.weight, self.w_2.bias, defer_bias=defer_bias, out_dtype=odt)


def _post_norm(norm, x, branch, p, training, a_bias=None, link=None):
    """LN(x + dropout(branch)) in one kernel; a_bias = bias of the Linear that produced `branch` when that Linear
    was called with defer_bias=True (its gradient is then reduced inside the LayerNorm backward)."""
    return ops.add_layernorm(x, branch, norm.weight, norm.bias, p if training else 0.0, norm.eps, a_bias=a_bias, link=link)


def _ffn_post_norm(ff, norm, x, p, defer_ln=False):
    """LN(x + dropout(FFN(x))): one row-block fused launch (ops.FfnLnFn) for GLU FFNs on enough rows, else GEMMs + add+LN.
    defer_ln: see ops.ffn_add_layernorm (the LayerNorm may be left to the launch that reads the result)."""
    if ff.activation == 'glu' and not (ff.dropout and ff.training):
        y = ops.ffn_add_layernorm(x, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias, norm.weight, norm.bias, p, norm.eps,
                                  defer_ln=defer_ln)
        if y is not None:
            return y
    link = ops.new_link()
    return _post_norm(norm, x, ff(x, defer_bias=True, link=link), p, True, ff.w_2.bias, link)


def _norm(norm, x):
    return ops.add_layernorm(x, None, norm.weight, norm.bias, 0.0, norm.eps)


def _deferred_bias(attn):
    """the output_proj bias the closing add+LayerNorm must add (and differentiate) -- None when the attention module applied
    it itself because its own dropout sits behind the projection"""
    return None if (attn.dropout_rate and attn.training) else attn.output_proj.bias


def _attention_branch(layer, concat_linear, x, p, run):
    """The residual branch of an attention sub-layer, bias left to the LayerNorm that closes it: (branch, bias, dropout
    rate, link).  `run(**kw)` calls the attention module.  concat_after (encoder/transformer.py:51-52,
    decoder/transformer.py:63-64,73-74): Linear(cat(x, attention)) and NO dropout on that path."""
    if layer.concat_after:
        att, _ = run()
        cat = torch.cat((x, att.to(x.dtype)), dim=-1)
        return (ops.linear(cat, concat_linear.weight, concat_linear.bias, defer_bias=True, out_dtype=ops.act_dtype()),
                concat_linear.bias, 0.0, None)
    link = ops.new_link()                     # skip-connection gradients are summed in GEMM epilogues
    att, bias = run(defer_bias=True, link=link)
    return att, bias, p, link


def _attn_sublayer(layer, concat_linear, norm, attn, x, p, run, run_ctx):
    """LN(x + dropout(attention branch)).  When the row-block kernels apply (16-bit mode, d_model 256, no concat_after, no
    dropout inside the attention module) output_proj + residual + LayerNorm are ONE launch (ops.ProjLnFn); `run_ctx(link)`
    returns the attention context before output_proj, `run(**kw)` the projected branch of the generic path."""
    if not layer.concat_after and run_ctx is not None and not (attn.dropout_rate and attn.training) and x.is_cuda:
        link = ops.new_link()
        c = run_ctx(link)
        packs = ops.proj_ln_packs(x, c, attn.output_proj.weight, norm.weight)
        if packs is not None:
            ops.touch_ffn_packs_next(getattr(layer, 'feed_forward', None), x)
            return ops.proj_add_layernorm(x, c, attn.output_proj.weight, attn.output_proj.bias, norm.weight, norm.bias, p, norm.eps,
                                          packs, link)
        branch = ops.linear(c, attn.output_proj.weight, attn.output_proj.bias, defer_bias=True, out_dtype=ops.act_dtype())
        return _post_norm(norm, x, branch, p, True, attn.output_proj.bias, link)
    branch, bias, p1, link = _attention_branch(layer, concat_linear, x, p, run)
    return _post_norm(norm, x, branch, p1, True, bias, link)


# ------------------------------------------------------------------------------------- encoder
class TransformerEncoderLayer(nn.Module):
    """encoder/transformer.py:16-90.  Post-norm: x = LN1(x + drop(SA(x))); x = LN2(x + drop(FFN(x))).  Pre-norm as the
    reference wrote it: x = LN1(x); x = x + drop(SA(x)); x = LN2(x); x = x + drop(FFN(x)) -- so "norm_k+1 of (x + branch_k)"
    is the same fused add+LayerNorm kernel in both variants, shifted by one sub-layer."""

    def __init__(self, n_heads, d_model, d_ff, slf_attn_dropout, ffn_dropout, residual_dropout,
                 normalize_before=False, concat_after=False, relative_positional=False, activation='relu'):
        super().__init__()
        self.relative_positional, self.normalize_before, self.concat_after = relative_positional, normalize_before, concat_after
        if relative_positional:       # Transformer-XL style scores; as shipped this module has no output projection (a19)
            self.slf_attn = MultiHeadedSelfAttentionWithRelPos(n_heads, d_model, slf_attn_dropout)
        else:
            self.slf_attn = MultiHeadedSelfAttention(n_heads, d_model, slf_attn_dropout)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, ffn_dropout, activation)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.residual_dropout = residual_dropout
        if concat_after:
            self.concat_linear = nn.Linear(d_model * 2, d_model)

    # set by TransformerEncoder.forward around its calls: the block's closing LayerNorm may stay pending (ops.PendingLn) -- the caller
    # hands the result to another block of this kind, whose first read is the q|k|v projection, or to ops.materialize.  (An attribute,
    # not an argument: forward keeps the reference's signature.)
    _defer_ln = False

    def forward(self, x, mask, pos=None, causal=False):
        defer_ln = self._defer_ln
        p = self.residual_dropout if self.training else 0.0
        pre = self.normalize_before
        if pre:
            x = _norm(self.norm1, x)

        def run(**kw):
            if self.relative_positional:
                if causal:
                    _unsupported('causal relative-positional self-attention')
                return self.slf_attn(x, mask, pos)[0], None
            att, _ = self.slf_attn(x, mask, causal, **kw)
            return att, _deferred_bias(self.slf_attn)
        run_ctx = None if self.relative_positional else (lambda link: self.slf_attn.context(x, mask, causal, link))
        x = _attn_sublayer(self, getattr(self, 'concat_linear', None), self.norm2 if pre else self.norm1, self.slf_attn, x, p, run,
                           run_ctx)
        if pre:
            x = ops.residual_add(x, self.feed_forward(x), 1.0, p)
        else:
            x = _ffn_post_norm(self.feed_forward, self.norm2, x, p, defer_ln and not self.relative_positional)
        return x, {'slf_attn_weights': None}

    def inference(self, x, mask, pos=None, cache=None):
        was = self.training
        self.training = False
        try:
            x, w = self.forward(x, mask, pos)
        finally:
            self.training = was
        return x, cache, w


class TransformerEncoder(nn.Module):
    """encoder/transformer.py:93-134.  forward(inputs, mask [B,T]) -> (out, mask, attn_weights)."""

    def __init__(self, d_model=256, n_heads=4, d_ff=2048, n_blocks=6, pos_dropout=0.0, slf_attn_dropout=0.0,
                 ffn_dropout=0.0, residual_dropout=0.1, normalize_before=False, concat_after=False,
                 relative_positional=False, activation='relu'):
        super().__init__()
        self.normalize_before, self.relative_positional = normalize_before, relative_positional
        self.pos_emb = PositionalEncoding(d_model, pos_dropout)
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(n_heads, d_model, d_ff, slf_attn_dropout, ffn_dropout, residual_dropout,
                                    normalize_before, concat_after, relative_positional, activation)
            for _ in range(n_blocks)])
        if normalize_before:
            self.norm = nn.LayerNorm(d_model)               # encoder/transformer.py:111-112

    def forward(self, inputs, mask):
        if self.relative_positional:                        # encoder/transformer.py:116-120: no sqrt(d) scaling, no absolute PE
            x, pos = inputs.float(), relative_sinusoid(inputs.size(1), inputs.size(2), inputs.device)
        else:
            # the positional-encoding launch also leaves the key mask as bytes (ops.PosEncFn: no cast launch)
            (x, _), pos = self.pos_emb(inputs, mask=mask if mask.dim() == 2 else None), None
        # cast once; every layer's key mask is this uint8 view (and the decoder's memory mask: ops._mask_u8 remembers it on the tensor)
        km = (ops._mask_u8(mask, mask.size(0), mask.size(1)) if mask.dim() == 2 else mask.to(torch.uint8)).unsqueeze(1)
        defer = not self.normalize_before and not self.relative_positional
        for block in self.blocks:
            block._defer_ln = defer
            try:
                x, _ = block(x, km, pos)
            finally:
                block._defer_ln = False
        x = ops.materialize(x)
        if self.normalize_before:
            x = _norm(self.norm, x)
        # the reference returns every layer's [B,h,T,T] weights; nothing reads them
        return x, mask, {}


# ------------------------------------------------------------------------------------- conformer
class MultiHeadedSelfAttentionWithRelPos(nn.Module):
    """module/attention.py:176-253.  As shipped (constructor bug at :178) the module has no
    output projection and no dropout when slf_attn_dropout == 0 -- and crashes otherwise; we mirror that."""

    def __init__(self, n_heads, d_model, dropout_rate=0.0, skip_term_b=False, share_qvk_proj=False):
        super().__init__()
        if dropout_rate:
            _unsupported('MultiHeadedSelfAttentionWithRelPos dropout_rate > 0 (the reference crashes there)')
        if skip_term_b or share_qvk_proj:
            _unsupported('skip_term_b / share_qvk_proj')
        self.d_model, self.nheads, self.d_k = d_model, n_heads, d_model // n_heads
        self.qvk_proj = nn.Linear(d_model, d_model * 3)
        self.pos_proj = nn.Linear(d_model, d_model, bias=False)
        self.posu = nn.Parameter(torch.Tensor(1, 1, n_heads, self.d_k))
        self.posv = nn.Parameter(torch.Tensor(1, 1, n_heads, self.d_k))
        torch.nn.init.xavier_normal_(self.posu)
        torch.nn.init.xavier_normal_(self.posv)

    def forward(self, x, mask, pos):
        B, T, _ = x.shape
        qkv = ops.linear(x, self.qvk_proj.weight, self.qvk_proj.bias, out_dtype=ops.act_dtype())
        ctx = ops.RelPosAttentionFn.apply(qkv, pos, self.pos_proj.weight, self.posu, self.posv, _key_mask(mask, B, T),
                                          self.nheads, getattr(self, 'pos_tables', None))
        return ctx, None

    def inference(self, inputs, mask, pos, cache=None):
        context, w = self.forward(inputs, mask, pos)
        return context, w, cache


class ConformerConvolutionModule(nn.Module):
    """module/conformer.py:12-57."""

    def __init__(self, channels, kernel_size, bias=True, dropout=0.0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.dropout = dropout                    # the reference builds nn.Dropout(dropout) here and never applies it (module/conformer.py:34-57)
        self.pointwise_conv1 = nn.Linear(channels, 2 * channels, bias=bias)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, stride=1, padding=(kernel_size - 1) // 2,
                                        groups=channels, bias=bias)
        self.batch_norm = nn.BatchNorm1d(channels)
        self.pointwise_conv2 = nn.Linear(channels, channels, bias=bias)
        self.tick_later = None                    # a list (not a module attribute of tensors): see ConformerEncoder.forward

    def forward(self, x, mask, mask_out=True):
        """mask_out False: the caller zeroes the padded frames' rows of the result (and of its gradient) itself -- ops.ResidualLnFn(a_mask)"""
        bn = self.batch_norm
        B, T, _ = x.shape
        out = ops.ConformerConvFn.apply(x, ops._mask_u8(mask, B, T).reshape(-1), self.pointwise_conv1.weight,
                                        self.pointwise_conv1.bias, self.depthwise_conv.weight, self.depthwise_conv.bias,
                                        bn.weight, bn.bias, bn.running_mean, bn.running_var, self.pointwise_conv2.weight,
                                        self.pointwise_conv2.bias, self.training, bn.eps, bn.momentum, mask_out)
        if self.training:
            if self.tick_later is not None:
                self.tick_later.append(bn.num_batches_tracked)     # ConformerEncoder adds 1 to all of them in one launch
            else:
                bn.num_batches_tracked += 1
        return out


class ConformerEncoderBlock(nn.Module):
    """encoder/conformer.py:20-114.  As shipped the post-FFN is never applied: forward ends with
    post_ffn_norm then final_norm (:87-89), and F.dropout stays active in eval (:53-72)."""

    def __init__(self, d_model, d_ff, cov_kernel_size, n_heads, slf_attn_dropout=0.0, ffn_dropout=0.0,
                 residual_dropout=0.1, conv_dropout=0.0, macaron_style=True, conv_first=False, ffn_scale=0.5,
                 conv_bias=True, relative_positional=True, activation='glu'):
        super().__init__()
        self.conv_first, self.macaron_style, self.ffn_scale = conv_first, macaron_style, ffn_scale
        self.relative_positional, self.residual_dropout = relative_positional, residual_dropout
        if macaron_style:
            self.pre_ffn = PositionwiseFeedForward(d_model, d_ff, ffn_dropout, activation=activation)
            self.macaron_ffn_norm = nn.LayerNorm(d_model)
        if relative_positional:
            self.mha = MultiHeadedSelfAttentionWithRelPos(n_heads, d_model, slf_attn_dropout)
        else:
            self.mha = MultiHeadedSelfAttention(n_heads, d_model, slf_attn_dropout)
        self.mha_norm = nn.LayerNorm(d_model)
        self.conv = ConformerConvolutionModule(d_model, cov_kernel_size, conv_bias, conv_dropout)
        self.conv_norm = nn.LayerNorm(d_model)
        self.post_ffn = PositionwiseFeedForward(d_model, d_ff, ffn_dropout, activation=activation)   # unused, as shipped
        self.post_ffn_norm = nn.LayerNorm(d_model)
        self.final_norm = nn.LayerNorm(d_model)

    @staticmethod
    def _ln(norm, x, link=None):
        return ops.add_layernorm(x, None, norm.weight, norm.bias, 0.0, norm.eps, link=link)

    def _attn(self, x, mask, pos, p):
        link = ops.new_prenorm_link()                  # x + f(LN(x)): the two gradients of x meet in the LayerNorm backward
        h = self._ln(self.mha_norm, x, link)
        # the key mask as bytes, cast once per batch: ops._mask_u8 remembers the cast ON the mask tensor, and `mask.unsqueeze(1)` is a new
        # tensor object in every block (12 cast launches of 8.6 us per step)
        km = ops._mask_u8(mask, mask.shape[0], mask.shape[1]).unsqueeze(1) if (mask is not None and mask.dim() == 2 and mask.is_cuda) else mask.unsqueeze(1)
        out = self.mha(h, km, pos)[0] if self.relative_positional else self.mha(h, km)[0]
        return ops.residual_add(x, out, 1.0, p, link)

    def _conv(self, x, mask, p):
        link = ops.new_prenorm_link()
        return ops.residual_add(x, self.conv(self._ln(self.conv_norm, x, link), mask), 1.0, p, link)

    def _forward_fused(self, x, mask, pos):
        """the shipped layout (macaron FFN, attention, convolution) with every residual add fused into the LayerNorm that reads its
        result (ops.ResidualLnFn): encoder/conformer.py:50-89.  r06 (set by ConformerEncoder.forward around its calls, not arguments:
        forward keeps the reference's signature): `chain_in` = macaron_ffn_norm(x) already computed by the block BELOW in its closing
        launch, `chain_next` = the macaron_ffn_norm of the block ABOVE, whose output this block's closing launch leaves in `chain_out`."""
        p = self.residual_dropout
        ch = self.__dict__.get('_chain') or {}         # a plain dict: a Module stored as an attribute would be registered as a sub-module
        h_pre = ch.get('in')
        if h_pre is not None:
            a = self.pre_ffn(h_pre, branch=True)       # (x's two gradients meet in the closing launch of the block below: autograd)
            link = None
        else:
            link = ops.new_prenorm_link()
            a = self.pre_ffn(self._ln(self.macaron_ffn_norm, x, link), branch=True)
        n = self.mha_norm
        x, h = ops.residual_layernorm(x, a, self.ffn_scale, p, n.weight, n.bias, n.eps, link, lp_only=_LN16)
        km = ops._mask_u8(mask, mask.shape[0], mask.shape[1]).unsqueeze(1)
        a = self.mha(h, km, pos)[0] if self.relative_positional else self.mha(h, km)[0]
        n = self.conv_norm
        x, h = ops.residual_layernorm(x, a, 1.0, p, n.weight, n.bias, n.eps, lp_only=_LN16)
        am = ops._mask_u8(mask, mask.shape[0], mask.shape[1]).reshape(-1) if _MASK_FOLD else None
        a = self.conv(h, mask, mask_out=am is None)      # the padded frames' rows are zeroed by the residual launch below
        n, n2 = self.post_ffn_norm, self.final_norm
        n3 = ch.get('next')
        if _LN2 and _LN3 and n3 is not None and n.eps == n2.eps == n3.eps and am is not None:
            # ... and the macaron LayerNorm of the NEXT block in the same launches (otr_add_layernorm3_*)
            _, y, ch['out'] = ops.residual_layernorm3(x, a, 1.0, p, n, n2, n3, a_mask=am)
            return y, {'slf_attn_weights': None}
        if _LN2 and n.eps == n2.eps:                    # the two closing LayerNorms in the same launches (otr_add_layernorm2_*)
            _, y = ops.residual_layernorm(x, a, 1.0, p, n.weight, n.bias, n.eps, None, n2.weight, n2.bias, a_mask=am)
            return y, {'slf_attn_weights': None}
        _, y = ops.residual_layernorm(x, a, 1.0, p, n.weight, n.bias, n.eps, a_mask=am)
        return self._ln(self.final_norm, y), {'slf_attn_weights': None}

    def forward(self, x, mask, pos=None):
        p = self.residual_dropout                      # F.dropout(..., p): active in train AND eval in the reference
        if (_RES_LN and self.macaron_style and not self.conv_first and x.is_cuda and x.dtype == torch.float32 and mask is not None
                and mask.dim() == 2 and x.shape[-1] % 4 == 0):
            return self._forward_fused(x, mask, pos)
        if self.macaron_style:
            link = ops.new_prenorm_link()
            x = ops.residual_add(x, self.pre_ffn(self._ln(self.macaron_ffn_norm, x, link), branch=True), self.ffn_scale, p, link)
        if self.conv_first:
            x = self._attn(self._conv(x, mask, p), mask, pos, p)
        else:
            x = self._conv(self._attn(x, mask, pos, p), mask, p)
        x = self._ln(self.post_ffn_norm, x)
        return self._ln(self.final_norm, x), {'slf_attn_weights': None}


_SINUSOID_CACHE = {}


def relative_sinusoid(T, d, device):
    """PositionalEncoding._embedding_from_positions(arange(-(T-1), T)) (module/pos.py:30-42): a constant
    table, built once per (T, d) on the host."""
    key = (T, d, str(device))
    if key not in _SINUSOID_CACHE:
        pos = torch.arange(-(T - 1), T, dtype=torch.float32).unsqueeze(-1)
        div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
        pe = torch.zeros(1, 2 * T - 1, d)
        pe[0, :, 0::2] = torch.sin(pos * div)
        pe[0, :, 1::2] = torch.cos(pos * div)
        _SINUSOID_CACHE[key] = pe.to(device)
    return _SINUSOID_CACHE[key]


class ConformerEncoder(nn.Module):
    """encoder/conformer.py:117-164 (note the kwarg is `nblocks`, not `n_blocks`)."""

    def __init__(self, d_model, d_ff, cov_kernel_size, n_heads, nblocks=12, pos_dropout=0.0, slf_attn_dropout=0.0,
                 ffn_dropout=0.0, residual_dropout=0.1, conv_dropout=0.0, macaron_style=True, ffn_scale=0.5,
                 conv_bias=True, positional_encoding=True, relative_positional=True, conv_first=False, activation='glu'):
        super().__init__()
        self.positional_encoding, self.relative_positional, self.output_size = positional_encoding, relative_positional, d_model
        if relative_positional and not positional_encoding:
            _unsupported('relative_positional without positional_encoding')
        if positional_encoding:
            self.pos_emb = PositionalEncoding(d_model, pos_dropout)
        self.blocks = nn.ModuleList([
            ConformerEncoderBlock(d_model, d_ff, cov_kernel_size, n_heads, slf_attn_dropout, ffn_dropout,
                                  residual_dropout, conv_dropout, macaron_style, conv_first, ffn_scale, conv_bias,
                                  relative_positional, activation) for _ in range(nblocks)])

    def forward(self, inputs, mask):
        if self.positional_encoding and not self.relative_positional:
            x, pos = self.pos_emb(inputs)
        else:
            x = inputs.float()
            pos = relative_sinusoid(inputs.size(1), inputs.size(2), inputs.device) if self.relative_positional else None
        ticks = [] if self.training else None           # BatchNorm1d.num_batches_tracked += 1 (module/conformer.py:33) of every block: one launch
        # r06: pos_proj(sinusoid) of EVERY block (and its transpose) in two batched launches instead of a GEMM + a transposing copy per block
        tables = None
        if self.relative_positional and _POS_TABLES and x.is_cuda:
            tables = ops.relpos_tables(pos, [block.mha.pos_proj.weight for block in self.blocks])
        h_next = None
        for bi, block in enumerate(self.blocks):
            block.conv.tick_later = ticks
            block.mha.pos_tables = tables[bi] if tables is not None else None
            nxt = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
            # r06: this block's closing launch also runs the NEXT block's macaron LayerNorm (ops.ResidualLn3Fn) and hands its output over
            chain = {'in': h_next, 'next': nxt.macaron_ffn_norm if (nxt is not None and getattr(nxt, 'macaron_style', False)) else None, 'out': None}
            block.__dict__['_chain'] = chain
            try:
                x, _ = block(x, mask, pos)
                h_next = chain['out']
            finally:
                block.conv.tick_later = None
                block.mha.pos_tables = None
                block.__dict__.pop('_chain', None)
        if ticks:
            torch._foreach_add_(ticks, 1)
        return x, mask, {}


# ------------------------------------------------------------------------------------- decoder
class TransformerDecoderLayer(nn.Module):
    """decoder/transformer.py:18-126; the same four variants as TransformerEncoderLayer."""

    def __init__(self, n_heads, d_model, d_ff, memory_dim, slf_attn_dropout=0.0, src_attn_dropout=0.0,
                 ffn_dropout=0.0, residual_dropout=0.1, normalize_before=False, concat_after=False,
                 relative_positional=False, activation='relu'):
        super().__init__()
        if relative_positional:
            _unsupported('relative_positional=True')
        self.relative_positional, self.normalize_before, self.concat_after = False, normalize_before, concat_after
        self.slf_attn = MultiHeadedSelfAttention(n_heads, d_model, slf_attn_dropout)
        self.src_attn = MultiHeadedCrossAttention(n_heads, d_model, memory_dim, src_attn_dropout)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, ffn_dropout, activation)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.residual_dropout = residual_dropout
        if concat_after:
            self.concat_linear1 = nn.Linear(d_model * 2, d_model)
            self.concat_linear2 = nn.Linear(d_model * 2, d_model)

    def forward(self, tgt, tgt_mask, memory, memory_mask, pos=None, kv_all=None):
        """tgt_mask: the causal [B,L,L] tril mask of decoder/utils.py:7-11, or None meaning causal."""
        p = self.residual_dropout if self.training else 0.0
        pre = self.normalize_before
        norms = (self.norm2, self.norm3) if pre else (self.norm1, self.norm2)      # the norm that closes sub-layer 1, 2
        x = _norm(self.norm1, tgt) if pre else tgt

        def run_self(**kw):
            att, _ = self.slf_attn(x, tgt_mask, causal=tgt_mask is None, **kw)
            return att, _deferred_bias(self.slf_attn)
        x0 = x
        x = _attn_sublayer(self, getattr(self, 'concat_linear1', None), norms[0], self.slf_attn, x0, p, run_self,
                           lambda link: self.slf_attn.context(x0, tgt_mask, tgt_mask is None, link))

        def run_src(**kw):
            att, _ = self.src_attn(x, memory, memory_mask, kv_all=kv_all, **kw)
            return att, _deferred_bias(self.src_attn)
        x1 = x
        x = _attn_sublayer(self, getattr(self, 'concat_linear2', None), norms[1], self.src_attn, x1, p, run_src,
                           lambda link: self.src_attn.context(x1, memory, memory_mask, link, kv_all))
        if pre:
            x = ops.residual_add(x, self.feed_forward(x), 1.0, p)
        else:
            x = _ffn_post_norm(self.feed_forward, self.norm3, x, p)
        return x, {'slf_attn_weights': None, 'src_attn_weights': None}


class TransformerDecoder(nn.Module):
    """decoder/transformer.py:129-208."""

    def __init__(self, vocab_size, d_model=256, n_heads=4, d_ff=2048, memory_dim=256, n_blocks=6, pos_dropout=0.0,
                 slf_attn_dropout=0.0, src_attn_dropout=0.0, ffn_dropout=0.0, residual_dropout=0.1, activation='relu',
                 normalize_before=True, concat_after=False, share_embedding=False):
        super().__init__()
        self.decoder_type = 'transformer'
        self.normalize_before, self.relative_positional, self.d_model = normalize_before, False, d_model
        self.embedding = nn.Embedding(vocab_size, d_model)
        self.pos_emb = PositionalEncoding(d_model, pos_dropout)
        self.blocks = nn.ModuleList([
            TransformerDecoderLayer(n_heads, d_model, d_ff, memory_dim, slf_attn_dropout, src_attn_dropout,
                                    ffn_dropout, residual_dropout, normalize_before=normalize_before,
                                    concat_after=concat_after, relative_positional=False, activation=activation)
            for _ in range(n_blocks)])
        if normalize_before:
            self.after_norm = nn.LayerNorm(d_model)              # decoder/transformer.py:150-151
        self.output_layer = nn.Linear(d_model, vocab_size)
        if share_embedding:
            self.output_layer.weight = self.embedding.weight      # decoder/transformer.py:156-158

    def forward(self, targets, memory, memory_mask):
        x = ops.embed_posenc(targets, self.embedding.weight)
        kv = None
        if (len(self.blocks) > 1 and memory.is_cuda and not any(b.src_attn.share_vk_proj for b in self.blocks)):
            # keys / values of every layer from ONE GEMM over the shared memory (ops.CrossKVAllFn)
            shared = ops.CrossKVShared(len(self.blocks))
            wb = [t for b in self.blocks for t in (b.src_attn.vk_proj.weight, b.src_attn.vk_proj.bias)]
            kv = (ops.CrossKVAllFn.apply(memory, shared, *wb), shared)
        S = ops.decoder_stack_applies(x, memory, self.blocks, self.normalize_before) if kv is not None else 0
        if S and all(b.residual_dropout == self.blocks[0].residual_dropout and b.norm1.eps == self.blocks[0].norm1.eps
                     and b.norm2.eps == b.norm1.eps and b.norm3.eps == b.norm1.eps for b in self.blocks):
            # the whole stack as three launches per layer (csrc/declayer.hip): cut along (utterance group, head) / (rows, hidden slice)
            x = ops.decoder_stack(x, kv[0], ops._mask_u8(memory_mask, memory.size(0), memory.size(1)), self.blocks, S)
        else:
            mm = memory_mask.to(torch.uint8).unsqueeze(1)
            for i, block in enumerate(self.blocks):
                x, _ = block(x, None, memory, mm, kv_all=(kv[0], i, kv[1]) if kv is not None else None)   # None -> causal self-attention
        if self.normalize_before:
            x = _norm(self.after_norm, x)
        logits = ops.linear(x, self.output_layer.weight, self.output_layer.bias)
        return logits, {}

    def inference(self, preds, memory, memory_mask=None, cache=None):
        """Full re-forward + log_softmax of the last position, exactly the reference's behaviour
        (decoder/transformer.py:185-208; its cache argument is ignored there too)."""
        assert preds.dim() == 2
        logits, attn = self.forward(preds, memory, memory_mask)
        return ops.log_softmax(logits[:, -1, :]), cache, attn


# ------------------------------------------------------------------------------------- losses / heads
class LabelSmoothingLoss(nn.Module):
    """module/loss.py:12-48.  `mask` (True = leave the position out, on top of target == PAD) and normalize_length=False
    (divide by all B*L positions instead of the counted ones) are folded into the one kernel's inputs / result."""

    def __init__(self, size, smoothing=0.1, padding_idx=PAD, normalize_length=True):
        super().__init__()
        self.size, self.smoothing, self.padding_idx, self.normalize_length = size, smoothing, padding_idx, normalize_length

    _otr_grad_scale = None

    def forward(self, logits, target, mask=None):
        assert logits.dim() == 3 and logits.size(-1) == self.size
        if mask is not None:          # a masked row contributes nothing, exactly like a PAD row (loss.py:31-35,45-46)
            target = target.masked_fill(mask.bool(), self.padding_idx)
        if not self.normalize_length:
            loss = ops.LabelSmoothingLossFn.apply(logits.float(), target, float(self.smoothing), int(self.padding_idx))
            kept = (target != self.padding_idx).sum().to(loss.dtype)
            loss = loss * (kept / target.numel())
            return ops.ScaleGradFn.apply(loss, self._otr_grad_scale) if self._otr_grad_scale is not None and loss.requires_grad else loss
        # _otr_grad_scale: set by the model around its call (an attribute, not an argument: forward keeps the reference's signature) --
        # the device scalar its backward pass is seeded with (fp16 loss scale); the fused loss launch folds it into the gradient
        return ops.label_smoothing_loss(logits.float(), target, float(self.smoothing), int(self.padding_idx),
                                        grad_scale=self._otr_grad_scale)
