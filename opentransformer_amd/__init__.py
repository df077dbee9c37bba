"""opentransformer_amd -- MI355X (gfx950) native hot path for ZhengkunTian/OpenTransformer (otrans).

Drop-in for otrans' frontend / encoder / decoder / module stack: same registries, constructor
kwargs, forward signatures and state_dict keys, with every kernel a hand-written HIP kernel behind
the C ABI of include/otrans_hip.h (opentransformer_amd/lib/libotrans_hip.so).  No CPU fallback.
"""
from . import synthetic                                   # noqa: F401  (numpy/torch host helpers only)
from . import data, tools                                # noqa: F401  (batch assembly / SpecAugment; checkpoint + WER tooling)
from .ops import set_compute_dtype, get_compute_dtype     # noqa: F401
from .model import (BuildFrontEnd, BuildEncoder, BuildDecoder, End2EndModel, SpeechToText, CTCModel,   # noqa: F401
                    CTCAssistor)
from .nn import (ConvFrontEnd, ConformerEncoder, ConformerEncoderBlock, ConformerConvolutionModule,   # noqa: F401
                 MultiHeadedSelfAttentionWithRelPos)
from .nn import (ConvFrontEnd, TransformerEncoder, TransformerEncoderLayer, TransformerDecoder,   # noqa: F401
                 TransformerDecoderLayer, MultiHeadedSelfAttention, MultiHeadedCrossAttention,
                 PositionwiseFeedForward, PositionalEncoding, LabelSmoothingLoss)



def _apply_switch_overrides():
    """The ONE environment hook of the Python side (r06: the 39 `OTR_*` reads scattered over ops / nn / recognize are module constants
    now): OTR_SWITCHES="ops._FFN_SLAB=0,nn._LN2=0,recognize._DECODE_FORK=0" sets module-level switches for an A/B run
    (tools/gpu_ab.sh TAG OTR_SWITCHES ops._X=1 ops._X=0).  Values are integers; a switch that is a bool stays a bool.  An unknown
    name is an error: a typo must not silently measure the default twice.  The C side has the same single hook: OTR_DEBUG_SET."""
    import os
    spec = os.environ.get('OTR_SWITCHES', '')
    if not spec:
        return
    from . import nn as _nn, ops as _ops, recognize as _rec
    mods = {'ops': _ops, 'nn': _nn, 'recognize': _rec}
    for item in filter(None, spec.split(',')):
        name, val = item.split('=')
        mod, attr = name.split('.')
        cur = getattr(mods[mod], attr)            # AttributeError / KeyError on an unknown switch
        assert attr.startswith('_') and isinstance(cur, (bool, int)), 'OTR_SWITCHES: %s is not a switch' % name
        setattr(mods[mod], attr, bool(int(val)) if isinstance(cur, bool) else int(val))


_apply_switch_overrides()

__all__ = ['BuildFrontEnd', 'BuildEncoder', 'BuildDecoder', 'End2EndModel', 'SpeechToText', 'CTCAssistor',
           'set_compute_dtype', 'get_compute_dtype']
