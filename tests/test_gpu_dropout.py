"""GPU (-m gpu): the dropout options the shipped yamls leave at 0 (VERDICT r01 missing #7): slf_attn / src_attn dropout on
the projected context (module/attention.py:46), ffn_dropout on the hidden (module/ffn.py:40), the frontend's conv dropout
(frontend/conv.py:63-66).  Bit-equal masks with torch's Philox stream are impossible, so each case is checked against a
plain fp32 torch evaluation that uses the masks RECOVERED from the kernels' own outputs, plus keep-rate and determinism."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(autouse=True)
def fp32_mode():
    from opentransformer_amd import ops
    ops.set_compute_dtype('fp32')
    yield
    ops.set_compute_dtype('bf16')


def test_dropout_kernel_keep_rate_determinism_and_backward():
    from opentransformer_amd import ops
    x = torch.randn(64, 1000, device=DEV).requires_grad_(True)
    ops.next_dropout_step(DEV)
    y = ops.dropout(x, 0.3)
    kept = y != 0
    assert abs(float(kept.float().mean()) - 0.7) < 0.01
    assert rel(y[kept], x[kept] / 0.7) < 1e-6
    (dx,) = torch.autograd.grad(y, x, torch.ones_like(y))
    assert torch.equal(dx != 0, kept) and rel(dx[kept], torch.full_like(dx[kept], 1 / 0.7)) < 1e-6
    assert ops.dropout(x, 0.3, training=False) is x and ops.dropout(x, 0.0) is x
    ops._state['rng_offset'] = 0                          # same step, same offset -> same mask
    assert torch.equal(ops.dropout(x, 0.3) != 0, kept)


def test_attention_dropout_on_projected_context():
    import opentransformer_amd.nn as onn
    from opentransformer_amd import ops
    torch.manual_seed(0)
    att = onn.MultiHeadedSelfAttention(4, 64, dropout_rate=0.2).to(DEV).train()
    B, T = 3, 40
    x = torch.randn(B, T, 64, device=DEV, requires_grad=True)
    mask = torch.ones(B, 1, T, dtype=torch.bool, device=DEV)
    ops.next_dropout_step(DEV)
    y, _ = att(x, mask)
    att.eval()
    y0, _ = att(x, mask)                                   # no dropout: output_proj(context)
    att.train()
    m = (y != 0).float() / 0.8                             # recovered mask (y0 == 0 exactly has measure zero)
    assert abs(float((y != 0).float().mean()) - 0.8) < 0.03
    assert rel(y, y0 * m) < 1e-5
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, [x] + list(att.parameters()), g)
    want = torch.autograd.grad(y0, [x] + list(att.parameters()), g * m)
    for a, b in zip(got, want):
        assert rel(a, b) < 1e-4


@pytest.mark.parametrize('act', ['glu', 'relu', 'swish'])
def test_ffn_dropout_on_hidden(act):
    import opentransformer_amd.nn as onn
    from opentransformer_amd import ops
    torch.manual_seed(1)
    ff = onn.PositionwiseFeedForward(64, 128, 0.25, activation=act).to(DEV).train()
    x = torch.randn(5, 30, 64, device=DEV, requires_grad=True)
    ops.next_dropout_step(DEV)
    y = ff(x)
    w1, b1, w2, b2 = ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias
    h = F.linear(x, w1, b1)
    h = {'glu': lambda t: F.glu(t, -1), 'relu': F.relu, 'swish': lambda t: t * torch.sigmoid(t)}[act](h)
    # recover the mask by solving y = w_2 (h * m) + b_2 is not possible directly; regenerate it instead: same step, offset 0
    ops._state['rng_offset'] = 0
    m = (ops.dropout(torch.ones_like(h), 0.25) != 0).float() / 0.75
    yr = F.linear(h * m, w2, b2)
    assert abs(float((m != 0).float().mean()) - 0.75) < 0.02
    assert rel(y, yr) < 1e-5
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, [x, w1, b1, w2, b2], g)
    want = torch.autograd.grad(yr, [x, w1, b1, w2, b2], g)
    for a, b in zip(got, want):
        assert rel(a, b) < 1e-4


def test_frontend_conv_dropout():
    import opentransformer_amd.nn as onn
    from opentransformer_amd import ops
    torch.manual_seed(2)
    fe = onn.ConvFrontEnd(80, 64, mid_channel=32, out_channel=64, dropout=0.2).to(DEV).train()
    x = torch.randn(2, 120, 80, device=DEV)
    mask = torch.ones(2, 120, dtype=torch.bool, device=DEV)
    ops.next_dropout_step(DEV)
    y, _ = fe(x, mask)
    c1, c2, lin = fe.conv1.conv_layer, fe.conv2.conv_layer, fe.output_layer
    # regenerate the two masks: act1 is [B,T1,F1,C1] channel-last with offset 0, act2 [B,T2,F2*C2] follows it
    T1, F1, T2, F2 = ops.conv_geometry(120, 80)
    ops._state['rng_offset'] = 0
    m1 = (ops.dropout(torch.ones(2, T1, F1, 32, device=DEV), 0.2) != 0).float() / 0.8
    m2 = (ops.dropout(torch.ones(2, T2, F2 * 64, device=DEV), 0.2) != 0).float() / 0.8
    h1 = F.relu(F.conv2d(x.unsqueeze(1), c1.weight, c1.bias, stride=2, padding=(0, 1))) * m1.permute(0, 3, 1, 2)
    h2 = F.relu(F.conv2d(h1, c2.weight, c2.bias, stride=2, padding=(0, 1)))            # [B,C2,T2,F2]
    h2 = h2 * m2.view(2, T2, F2, 64).permute(0, 3, 1, 2)
    yr = F.linear(h2.transpose(1, 2).reshape(2, T2, -1), lin.weight, lin.bias)
    assert rel(y, yr) < 1e-5
    params = [c1.weight, c1.bias, c2.weight, c2.bias, lin.weight, lin.bias]
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, params, g)
    want = torch.autograd.grad(yr, params, g)
    for a, b in zip(got, want):
        assert rel(a, b) < 1e-4


def test_layer_with_all_dropouts_trains():
    """an encoder + decoder layer pair with every dropout switched on: finite loss and gradients, masks change per step"""
    import opentransformer_amd as ota
    from opentransformer_amd import synthetic as syn, ops
    cfg = syn.c1_model(0.1, ctc_weight=0.3)
    cfg['encoder'].update(slf_attn_dropout=0.1, ffn_dropout=0.1)
    cfg['decoder'].update(slf_attn_dropout=0.1, src_attn_dropout=0.1, ffn_dropout=0.1)
    cfg['frontend'].update(dropout=0.1)
    model = ota.SpeechToText(cfg)
    syn.fill_state_dict_(model.state_dict(), 5)
    model = model.to(DEV).train()
    inputs, targets = syn.synthetic_batch(4, 200, 80, 100, 10, seed=0)
    inputs = {k: v.to(DEV) for k, v in inputs.items()}
    targets = {k: v.to(DEV) for k, v in targets.items()}
    losses = []
    for _ in range(2):
        ops.next_dropout_step(DEV)
        model.zero_grad()
        loss, _ = model(inputs, targets)
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        losses.append(loss.item())
    assert math.isfinite(losses[0]) and losses[0] != losses[1]
