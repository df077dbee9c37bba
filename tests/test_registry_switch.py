"""CPU (-m "not gpu"; skipped when /root/reference is absent, i.e. on the GPU box): INTEGRATION.md section 4 EXECUTED -- the
reference's own registries (otrans/frontend/__init__.py:8-12, encoder/__init__.py:10-13, decoder/__init__.py:8-10) are pointed
at the drop-in classes and the UNMODIFIED otrans.model.SpeechToText / CTCModel (model/speech2text.py:19-23, model/ctc.py:72-79)
build the HIP-backed model from the yaml dict; constructor and forward / inference signatures of every drop-in equal the
reference class's; parameter names and shapes are identical, so a reference checkpoint loads strict=True.  No compute is run
(the product has no CPU path)."""
import inspect
import os
import sys

import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'otrans')), reason='the reference tree is not present here')


@pytest.fixture(scope='module')
def otrans():
    for p in (os.path.join(REF, 'otrans', 'module'), REF):     # the first entry only serves `from activation import Swish` (otrans/module/ffn.py:9)
        if p not in sys.path:
            sys.path.insert(0, p)
    import otrans as pkg
    import otrans.model                                          # noqa: F401
    return pkg


def _sig(fn):
    """parameter names, kinds and defaults (annotations / return types ignored)"""
    return [(p.name, p.kind, p.default) for p in inspect.signature(fn).parameters.values()]


PAIRS = [  # (reference module path, class name, methods whose signature a caller relies on)
    ('otrans.frontend.conv', 'ConvFrontEnd', ('__init__', 'forward', 'inference')),
    ('otrans.encoder.transformer', 'TransformerEncoder', ('__init__', 'forward')),
    ('otrans.encoder.transformer', 'TransformerEncoderLayer', ('__init__', 'forward')),
    ('otrans.encoder.conformer', 'ConformerEncoder', ('__init__', 'forward')),
    ('otrans.decoder.transformer', 'TransformerDecoder', ('__init__', 'forward', 'inference')),
    ('otrans.decoder.transformer', 'TransformerDecoderLayer', ('__init__',)),
    ('otrans.module.attention', 'MultiHeadedSelfAttention', ('__init__',)),
    ('otrans.module.attention', 'MultiHeadedCrossAttention', ('__init__',)),
    ('otrans.module.ffn', 'PositionwiseFeedForward', ('__init__',)),
    ('otrans.module.loss', 'LabelSmoothingLoss', ('__init__', 'forward')),
    ('otrans.model.speech2text', 'SpeechToText', ('__init__', 'forward', 'save_checkpoint', 'load_model')),
    ('otrans.model.ctc', 'CTCModel', ('__init__', 'forward', 'save_checkpoint', 'set_epoch')),
    ('otrans.model.ctc', 'CTCAssistor', ('__init__', 'forward', 'compute_logits', 'compute_loss', 'inference')),
]


@pytest.mark.parametrize('modpath,cls,methods', PAIRS)
def test_dropin_signatures_equal_the_reference(otrans, modpath, cls, methods):
    import importlib
    import opentransformer_amd as ota
    import opentransformer_amd.nn as onn
    ref_cls = getattr(importlib.import_module(modpath), cls)
    mine = getattr(ota, cls, None) or getattr(onn, cls)
    for m in methods:
        a, b = _sig(getattr(ref_cls, m)), _sig(getattr(mine, m))
        if m == 'forward' and cls in ('TransformerEncoderLayer',):
            b = [p for p in b if p[0] != 'causal']                # one extra keyword with a default: callers of the reference never pass it
        assert [p[0] for p in a] == [p[0] for p in b], (cls, m, a, b)
        for pa, pb in zip(a, b):
            assert pa[1] == pb[1], (cls, m, pa, pb)
            same_default = (pa[2] is pb[2]) or (pa[2] == pb[2])
            assert same_default, (cls, m, pa, pb)


def test_registry_switch_builds_the_dropin_model_inside_the_reference(otrans):
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn
    import otrans.frontend as rf
    import otrans.encoder as re_
    import otrans.decoder as rd
    import otrans.model.speech2text as rs2t
    import otrans.model.ctc as rctc
    cfg = syn.c1_model(0.1, ctc_weight=0.3)
    ref_model = otrans.model.End2EndModel['speech2text'](cfg)                # the reference as shipped
    saved = (dict(rf.BuildFrontEnd), dict(re_.BuildEncoder), dict(rd.BuildDecoder))
    try:
        # ---- INTEGRATION.md section 4, verbatim
        rf.BuildFrontEnd['conv'] = ota.ConvFrontEnd
        re_.BuildEncoder['transformer'] = ota.TransformerEncoder
        re_.BuildEncoder['conformer'] = ota.ConformerEncoder
        rd.BuildDecoder['transformer'] = ota.TransformerDecoder
        # the reference's model files bound the registries by name at import: they are the same dict objects
        assert rs2t.BuildFrontEnd is rf.BuildFrontEnd and rs2t.BuildEncoder is re_.BuildEncoder and rs2t.BuildDecoder is rd.BuildDecoder
        hip_model = otrans.model.End2EndModel['speech2text'](cfg)            # UNMODIFIED otrans.model.SpeechToText
        assert type(hip_model) is rs2t.SpeechToText
        assert type(hip_model.frontend) is ota.ConvFrontEnd and type(hip_model.encoder) is ota.TransformerEncoder
        assert type(hip_model.decoder) is ota.TransformerDecoder
        # identical parameter names / shapes: the reference's checkpoint dict loads strict=True through its own load_model()
        a = {k: tuple(v.shape) for k, v in ref_model.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in hip_model.state_dict().items()}
        assert a == b
        chk = {'frontend': ref_model.frontend.state_dict(), 'encoder': ref_model.encoder.state_dict(),
               'decoder': ref_model.decoder.state_dict(), 'ctc': ref_model.assistor.state_dict()}
        hip_model.load_model(chk)
        for (k, v), (_, w) in zip(sorted(ref_model.state_dict().items()), sorted(hip_model.state_dict().items())):
            if not k.startswith('assistor.'):                    # the reference's load_model leaves the CTC head alone (speech2text.py:84-87)
                assert torch.equal(v, w), k
        # the same through the CTC model and the Conformer encoder (conformer_baseline.yaml's keys)
        ccfg = syn.conformer_model(True, 0.1)
        cm = otrans.model.End2EndModel['speech2text'](ccfg)
        assert type(cm.encoder) is ota.ConformerEncoder
        # and there is no CPU fallback hiding behind the switch: compute on CPU tensors raises
        inputs, targets = syn.synthetic_batch(batch=2, frames=64, feat_dim=cfg['frontend']['input_size'], vocab=cfg['decoder']['vocab_size'],
                                              tgt_len=4, seed=0)
        from opentransformer_amd._lib import OtransHipError
        with pytest.raises((OtransHipError, OSError, RuntimeError)):
            hip_model(inputs, targets)
    finally:
        for reg, old in zip((rf.BuildFrontEnd, re_.BuildEncoder, rd.BuildDecoder), saved):
            reg.clear()
            reg.update(old)
