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

__all__ = ['BuildFrontEnd', 'BuildEncoder', 'BuildDecoder', 'End2EndModel', 'SpeechToText', 'CTCAssistor',
           'set_compute_dtype', 'get_compute_dtype']
