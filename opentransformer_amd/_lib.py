"""ctypes binding of libotrans_hip.so (include/otrans_hip.h).

This file is the "reference-side binding": the only thing between Python and the C ABI.  There is
no fallback: if the library is missing or a symbol is absent, importing the ops raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# One source, two builds: the 16-bit storage / MFMA-input type is bf16 in libotrans_hip.so and IEEE fp16 in
# libotrans_hip_f16.so (csrc/common.h).  select() picks the build every later load() returns; ops.set_compute_dtype calls it.
_LIB_DIR = os.environ.get('OTR_LIB_DIR') or os.path.join(_HERE, 'lib')     # OTR_LIB_DIR: a second build for A/B runs (csrc/Makefile)
LIB_PATHS = {'bf16': os.path.join(_LIB_DIR, 'libotrans_hip.so'), 'fp16': os.path.join(_LIB_DIR, 'libotrans_hip_f16.so')}
LIB_PATH = LIB_PATHS['bf16']

OTR_F32, OTR_BF16, OTR_F16 = 0, 1, 2
OTR_ABI_VERSION = 601           # include/otrans_hip.h: the header this binding's structures and SIGNATURES were written against
OTR_OPT_STATE_FLOATS = 528      # include/otrans_hip.h: floats of otr_optimizer_step's device state block
ACT_NONE, ACT_RELU = 0, 1


class LinearDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('x_dtype', C.c_int32), ('w_dtype', C.c_int32), ('y_dtype', C.c_int32), ('compute', C.c_int32),
                ('ldx', C.c_int64), ('ldw', C.c_int64), ('ldy', C.c_int64),
                ('act', C.c_int32), ('accumulate', C.c_int32)]


class WgradItem(C.Structure):               # otr_wgrad_item_t
    _fields_ = [('dy', C.c_void_p), ('x', C.c_void_p), ('dw', C.c_void_p), ('M', C.c_int32), ('N', C.c_int32),
                ('K', C.c_int32), ('ldy', C.c_int64), ('ldx', C.c_int64), ('ldw', C.c_int64),
                ('dy_dtype', C.c_int32), ('x_dtype', C.c_int32), ('dbias', C.c_void_p), ('overwrite', C.c_int32)]


class ColsumItem(C.Structure):              # otr_colsum_item_t
    _fields_ = [('a', C.c_void_p), ('out', C.c_void_p), ('M', C.c_int64), ('N', C.c_int64), ('lda', C.c_int64),
                ('dtype', C.c_int32)]


class AttnDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('H', C.c_int32), ('Tq', C.c_int32), ('Tk', C.c_int32), ('dk', C.c_int32),
                ('dtype', C.c_int32),
                ('q_bs', C.c_int64), ('q_ts', C.c_int64), ('k_bs', C.c_int64), ('k_ts', C.c_int64),
                ('v_bs', C.c_int64), ('v_ts', C.c_int64), ('o_bs', C.c_int64), ('o_ts', C.c_int64),
                ('causal', C.c_int32), ('scale', C.c_float)]


class LnDesc(C.Structure):
    _fields_ = [('M', C.c_int64), ('d', C.c_int32), ('a_dtype', C.c_int32),
                ('eps', C.c_float), ('p_drop', C.c_float), ('rng_offset', C.c_uint64), ('a_scale', C.c_float), ('a_row_mask', C.c_void_p),
                ('dy_dtype', C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('F', C.c_int32), ('C1', C.c_int32), ('C2', C.c_int32),
                ('T1', C.c_int32), ('F1', C.c_int32), ('T2', C.c_int32), ('F2', C.c_int32),
                ('act_dtype', C.c_int32), ('compute', C.c_int32), ('w_dtype', C.c_int32)]


class DecLn(C.Structure):                   # otr_dec_ln_t
    _fields_ = [('xres', C.c_void_p), ('x16', C.c_void_p), ('slabs', C.c_void_p), ('nslab', C.c_int32),
                ('bias', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('seed', C.c_void_p),
                ('p_drop', C.c_float), ('eps', C.c_float), ('rng_offset', C.c_uint64),
                ('y', C.c_void_p), ('y16', C.c_void_p), ('z', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p)]


class DecSelfStep(C.Structure):             # otr_dec_self_step_t
    _fields_ = [('ln', DecLn), ('R', C.c_int64), ('wqkv_pack', C.c_void_p), ('bqkv', C.c_void_p), ('wo_pack', C.c_void_p),
                ('kcache', C.c_void_p), ('vcache', C.c_void_p), ('anc', C.c_void_p), ('pos', C.c_void_p), ('maxlen', C.c_int32),
                ('slabs', C.c_void_p)]


class DecFfnFwd(C.Structure):               # otr_dec_ffn_fwd_t
    _fields_ = [('ln', DecLn), ('R', C.c_int64), ('w1_pack', C.c_void_p), ('b1', C.c_void_p), ('w2_pack', C.c_void_p),
                ('F', C.c_int32), ('S', C.c_int32), ('slabs', C.c_void_p), ('hsave', C.c_void_p)]


class DecLnB(C.Structure):                  # otr_dec_lnb_t
    _fields_ = [('dskip', C.c_void_p), ('slabs', C.c_void_p), ('nslab', C.c_int32),
                ('z', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('gamma', C.c_void_p), ('seed', C.c_void_p),
                ('p_drop', C.c_float), ('rng_offset', C.c_uint64),
                ('dz', C.c_void_p), ('da16', C.c_void_p), ('partial', C.c_void_p)]


_P = C.c_void_p
_I32, _I64, _F32 = C.c_int32, C.c_int64, C.c_float

# name -> argtypes (restype is int32 unless listed in _RESTYPE).  Must list every symbol that
# include/otrans_hip.h declares; tests/test_cabi.py cross-checks the two.
SIGNATURES = {
    'otr_version': [],
    'otr_half_type': [],
    'otr_debug_set': [_I32, _I32],
    'otr_set_fault_counter': [_P],
    'otr_last_error_string': [],
    'otr_linear_fwd': [C.POINTER(LinearDesc), _P, _P, _P, _P, _P, _I64, _P],
    'otr_linear_fwd_batched': [C.POINTER(LinearDesc), _P, _P, _P, _I32, _I64, _I64, _I64, _P],
    'otr_linear_dgrad': [C.POINTER(LinearDesc), _P, _P, _P, _P, _I64, _P],
    'otr_linear_wgrad': [C.POINTER(LinearDesc), _P, _P, _P, _P, _I64, _P],
    'otr_ffn_glu_fwd': [_P, _I64, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _P],
    'otr_ffn_glu_bwd': [_P, _I32, _I64, _P, _I64, _P, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, _P],
    'otr_pack_frags': [_P, _P, _P, _I32, _I64, _P],
    'otr_ffn_ln_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, C.c_uint64, _F32, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_ffn_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_ffn_split_scratch_bytes': [_I64],
    'otr_ffn_split_sync_ints': [_I64],
    'otr_ffn_split_hsave_bytes': [_I64, _I32],
    'otr_ffn_split_padded_rows': [_I64],
    'otr_ffn_ln_fwd_split': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, C.c_uint64, _F32, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _P],
    'otr_ffn_bwd_split': [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _P],
    'otr_linear_wgrad_grouped': [_P, _I32, _I32, _P, _I64, _P],
    'otr_colsum_grouped': [_P, _I32, _P],
    'otr_colsum': [_P, _I32, _I64, _I64, _I64, _P, _I32, _P],
    'otr_attention_fwd': [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P],
    'otr_attention_bwd': [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_attention_bias_fwd': [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _P, _P, _P],
    'otr_attention_bias_bwd': [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _I32, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_add_layernorm_fwd': [C.POINTER(LnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_add_layernorm_bwd': [C.POINTER(LnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_add_layernorm_bwd_skip': [C.POINTER(LnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_add_layernorm2_fwd': [C.POINTER(LnDesc)] + [_P] * 15,
    'otr_add_layernorm2_bwd': [C.POINTER(LnDesc)] + [_P] * 15,
    'otr_add_layernorm3_fwd': [C.POINTER(LnDesc)] + [_P] * 20,
    'otr_add_layernorm3_bwd': [C.POINTER(LnDesc), _P, _P, _I32] + [_P] * 18,
    'otr_add_layernorm_bwd_partial_rows': [_I64],
    'otr_rb_linear': [_P, _I64, _P, _P, _P, _I64, _P, _I32, _I64, _I64, _I32, _I32, _P],
    'otr_proj_ln_fwd': [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _F32, _F32, C.c_uint64, _P],
    'otr_ln_bwd_proj_partial_rows': [_I64],
    'otr_rb_linear_ln_bwd': [_P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _F32, C.c_uint64, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_touch': [_P, _I64, _P],
    'otr_zero_tick': [_P, _I64, _P, _I64, _P],
    'otr_touch_hint': [_P, _I64, _P, _I64],
    'otr_rb_linear_ln_bwd_pf': [_P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _F32, C.c_uint64, _P, _P, _P, _I64, _I32, _I32, _P, _I64, _P],
    'otr_ln_bwd_proj': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I32, _F32, C.c_uint64, _P],
    'otr_debug_trace': [_P],
    'otr_wgrad256_takes': [C.POINTER(WgradItem), _I32],
    'otr_debug_ffn_split_map': [_I64, _I32, _P, _I32],
    'otr_debug_attention_grid': [_I32, _I32, _I32, _P, _I32],
    'otr_debug_wgrad256_plan': [C.POINTER(WgradItem), _I32, _I32, _P],
    'otr_debug_wgrad256_errors': [_P],
    'otr_debug_trread': [_P, _P, _P, _P],
    'otr_spec_mask': [_P, _P, _I32, _I32, _I32, _I32, _P],
    'otr_transpose_batched': [_P, _P, _P, _I32, _I64, _I32, _P],
    'otr_glu_fwd': [_P, _P, _I32, _I64, _I64, _P, _P],
    'otr_glu_bwd': [_P, _P, _P, _P, _I32, _I64, _I64, _P, _I32, _P],
    'otr_posenc_fwd': [_P, _P, _P, _I64, _I32, _I32, _F32, _P],
    'otr_posenc_mask_fwd': [_P, _P, _P, _I64, _I32, _I32, _F32, _P, _I64, _I64, _P, _P],
    'otr_embed_posenc_fwd': [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _F32, _P],
    'otr_embed_posenc_fwd_ld': [_P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _F32, _P],
    'otr_embed_bwd_ld': [_P, _I64, _I32, _P, _P, _I32, _P, _I64, _I32, _I32, _F32, _P],
    'otr_embed_bwd': [_P, _P, _P, _I64, _I32, _I32, _F32, _P],
    'otr_scale': [_P, _P, _I64, _P, _F32, _P],
    'otr_scale_cast': [_P, _P, _I64, _F32, _P],
    'otr_cast_f32_to_bf16': [_P, _P, _I64, _P],
    'otr_conv1_fwd': [C.POINTER(ConvDesc), _P, _P, _P, _P, _P],
    'otr_conv1_wgrad': [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P],
    'otr_conv1_wgrad_partial_rows': [],
    'otr_conv2_fwd': [C.POINTER(ConvDesc), _P, _P, _P, _P, _P],
    'otr_conv12_fwd': [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_conv2_dgrad_cols': [C.POINTER(ConvDesc), _P, _P, _P, _P],
    'otr_conv2_col2im': [C.POINTER(ConvDesc), _P, _P, _P, _P],
    'otr_conv2_dgrad': [C.POINTER(ConvDesc), _P, _P, _P, _P, _P],
    'otr_conv2_wide_scratch_bytes': [],
    'otr_conv2_dgrad_wide': [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I64, _P],
    'otr_debug_conv2_dgrad_plan': [C.POINTER(ConvDesc), _P],
    'otr_conv2_wgrad': [C.POINTER(ConvDesc), _P, _P, _P, _P, _I64, _P],
    'otr_relu_bwd': [_P, _P, _P, _I32, _I64, _P],
    'otr_relu_bwd_colsum_partial_rows': [_I64, _I32, _I32],
    'otr_relu_bwd_colsum': [_P, _P, _P, _P, _I32, _I64, _I32, _P],
    'otr_act_fwd': [_P, _P, _I32, _I64, _I32, _P],
    'otr_act_bwd': [_P, _P, _P, _I32, _I64, _I32, _P],
    'otr_label_smoothing_loss': [_P, _P, _I64, _I32, _F32, _I32, _P, _P, _P, _P],
    'otr_label_smoothing_loss_ld': [_P, _I64, _P, _I64, _I32, _F32, _I32, _P, _P, _I64, _P, _P],
    'otr_label_smoothing_loss_fused': [_P, _I64, _P, _I64, _I32, _I64, _I32, _F32, _I32, _P, _P, _P, _I32, _I64, _P, _P, _P],
    'otr_log_softmax': [_P, _P, _I64, _I32, _P],
    'otr_ctc_loss': [_P, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    'otr_ffn_fwd_split_slab': [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_ffn_bwd_split_slab': [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_ln_bwd_proj_slabs': [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I32, _F32, C.c_uint64, _P],
    'otr_rb_linear_ln': [C.POINTER(DecLn), _P, _P, _P, _I32, _I64, _I64, _I32, _I32, _P],
    'otr_dec_self_fwd': [C.POINTER(DecLn), _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_dec_self_step': [C.POINTER(DecLn), _I64, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P],
    'otr_dec_cross_fwd': [C.POINTER(DecLn), _I32, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _I32, _P, _P, _P, _P, _P],
    'otr_dec_ffn_hsave_bytes': [_I64, _I32],
    'otr_dec_ffn_fwd': [C.POINTER(DecLn), _I64, _P, _P, _P, _I32, _I32, _P, _P, _P],
    'otr_dec_ln': [C.POINTER(DecLn), _I64, _P],
    'otr_dec_self_step_pair': [C.POINTER(DecSelfStep), C.POINTER(DecSelfStep), _P],
    'otr_dec_ffn_fwd_pair': [C.POINTER(DecFfnFwd), C.POINTER(DecFfnFwd), _P],
    'otr_dec_ln_pair': [C.POINTER(DecLn), _I64, C.POINTER(DecLn), _I64, _P],
    'otr_dec_ffn_bwd': [C.POINTER(DecLnB), _I64, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P],
    'otr_dec_cross_bwd': [C.POINTER(DecLnB), _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _I32, _P, _P, _P],
    'otr_dec_self_bwd': [C.POINTER(DecLnB), _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    'otr_dec_sum': [_P, _P, _I32, _I64, _P, _P],
    'otr_dec_group_size': [_I32, _I32],
    'otr_optimizer_step': [_P, _P, _P, _P, _I64, _P, _I32, _P] + [_F32] * 12 + [_P],
    'otr_allreduce_unique_id': [_P],
    'otr_allreduce_init': [C.POINTER(C.c_void_p), _P, _I32, _I32],
    'otr_allreduce_run': [_P, _P, _I64, _I32, _P],
    'otr_allreduce_destroy': [_P],
    'otr_beam_topk': [_P, _I64, _P, _I64, _F32, _I64, _I32, _I32, _P, _P, _P],
    'otr_beam_prune': [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    'otr_beam_prune_cached': [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P],
    'otr_decode_embed': [_P, _I64, _P, _P, _P, _P, _I64, _I32, _I32, _F32, _P],
    'otr_decode_lookup': [_P, _I64, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_lstm_cell': [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    'otr_decode_self_attention': [_P, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _I32, _F32, _P],
    'otr_residual_add_fwd': [_P, _P, _I32, _P, _I64, _F32, _F32, _P, C.c_uint64, _P],
    'otr_residual_add_bwd': [_P, _P, _I32, _I64, _F32, _F32, _P, C.c_uint64, _P],
    'otr_dropout': [_P, _P, _I32, _I64, _F32, _P, C.c_uint64, _P],
    'otr_head_bias_add': [_P, _I64, _P, _P, _P, _I32, _I64, _I32, _P],
    'otr_add2_strided': [_P, _I64, _P, _I64, _P, _I64, _I32, _I64, _I32, _P],
    'otr_add2_colsum_partial_rows': [_I64],
    'otr_add2_strided_colsum': [_P, _I64, _P, _I64, _P, _I64, _I32, _I64, _I32, _P, _P],
    'otr_regroup_add': [_P, _P, _I64, _I32, _I32, _I32, _P],
    'otr_row_mask': [_P, _P, _P, _I64, _I32, _P],
    'otr_row_mask_cast': [_P, _I32, _P, _P, _I32, _I64, _I32, _P],
    'otr_dwconv_fwd': [_P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    'otr_dwconv_bwd': [_P, _P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    'otr_dwconv_bwd_part': [_P, _P, _I32, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    'otr_dwconv_bwd_partial_rows': [_I64],
    'otr_bn_swish_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I64, _I32, _F32, _F32, _I32, _P],
    'otr_bn_swish_fwd_part': [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _I32, _I64, _I32, _F32, _F32, _P],
    'otr_dwconv_fwd_part': [_P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    'otr_dwconv_fwd_partial_rows': [_I64],
    'otr_bn_swish_bwd_partial_rows': [_I64],
    'otr_bn_swish_bwd': [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    'otr_bn_swish_bwd_sums': [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    'otr_conformer_conv_bwd_mid': [_P] * 13 + [_I32] * 7 + [_P],
}
_RESTYPE = {'otr_last_error_string': C.c_char_p, 'otr_dec_ffn_hsave_bytes': C.c_int64, 'otr_ffn_split_scratch_bytes': C.c_int64, 'otr_ffn_split_sync_ints': C.c_int64, 'otr_ffn_split_hsave_bytes': C.c_int64,
            'otr_ffn_split_padded_rows': C.c_int64, 'otr_add_layernorm_bwd_partial_rows': C.c_int64,
            'otr_ln_bwd_proj_partial_rows': C.c_int64, 'otr_dwconv_bwd_partial_rows': C.c_int64, 'otr_dwconv_fwd_partial_rows': C.c_int64,
            'otr_conv2_wide_scratch_bytes': C.c_int64, 'otr_add2_colsum_partial_rows': C.c_int64}

_libs = {}
_kind = 'bf16'


class OtransHipError(RuntimeError):
    pass


def select(kind):
    """choose the build (16-bit type 'bf16' or 'fp16') that load() returns from now on"""
    global _kind
    assert kind in LIB_PATHS
    _kind = kind


def load(kind=None):
    """Load the HIP library (the selected build) or raise -- never falls back to anything else."""
    kind = kind or _kind
    lib = _libs.get(kind)
    if lib is not None:
        return lib
    path = LIB_PATHS[kind]
    if not os.path.exists(path):
        raise OtransHipError(
            '%s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C opentransformer_amd/csrc`). There is no CPU/PyTorch fallback.' % path)
    lib = C.CDLL(path)
    lib.otr_version.restype = C.c_int32
    if lib.otr_version() != OTR_ABI_VERSION:      # a stale build: structures would be read at the wrong size (no call is safe)
        raise OtransHipError('%s answers ABI version %d, this binding was written against %d (include/otrans_hip.h): rebuild it with '
                             '`make -C opentransformer_amd/csrc`' % (path, lib.otr_version(), OTR_ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int32)
    assert lib.otr_half_type() == (OTR_F16 if kind == 'fp16' else OTR_BF16), 'library / 16-bit type mismatch'
    for kv in filter(None, os.environ.get('OTR_DEBUG_SET', '').split(',')):   # tuning hook for A/B runs: "key=value,..."
        k, v = kv.split('=')
        if lib.otr_debug_set(int(k), int(v)) != 0:
            raise OtransHipError('OTR_DEBUG_SET: bad entry %r' % kv)
    _libs[kind] = lib
    return lib


def check(ret, what):
    if ret != 0:
        msg = load().otr_last_error_string()
        raise OtransHipError('%s failed (%d): %s' % (what, ret, msg.decode() if msg else '?'))
