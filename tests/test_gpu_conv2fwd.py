"""GPU (-m gpu): the weight-stationary conv2 forward kernel (csrc/conv2fwd.hip; frontend/conv.py:63-66, second Conv2dLayer, 64 -> 128
channels, 80- or 40-bin inputs, 16-bit) against the implicit-GEMM path (otr_debug_set(22, 0)) on the same operands and against torch's
conv2d in fp32.  Shapes: the bench batch, frame counts that leave a partial last work item (T2 % 8 = 1 .. 7), one output row, 40 bins."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
HALF = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('mode', ['fp16', 'bf16'])
@pytest.mark.parametrize('B,T,Fdim', [(32, 1000, 80), (3, 97, 40), (1, 7, 80), (5, 331, 80), (2, 1023, 80), (4, 643, 40), (3, 35, 80), (2, 71, 80), (2, 75, 80), (1, 139, 40)])
def test_conv2_forward_direct_kernel(mode, B, T, Fdim):
    from opentransformer_amd import ops, _lib as L
    ops.set_compute_dtype(mode)
    lib = L.load()
    try:
        gen = torch.Generator().manual_seed(B * T + Fdim)
        C1, C2 = 64, 128
        x = torch.randn(B, T, Fdim, generator=gen).to(DEV)
        w1 = (torch.randn(C1, 1, 3, 3, generator=gen) / 3).to(DEV)
        b1 = (0.1 * torch.randn(C1, generator=gen)).to(DEV)
        w2 = (torch.randn(C2, C1, 3, 3, generator=gen) / math.sqrt(9 * C1)).to(DEV)
        b2 = (0.1 * torch.randn(C2, generator=gen)).to(DEV)
        outs, a1s = [], []
        w1g = w1.clone().requires_grad_(True)
        for direct in (1, 2, 0):                 # 1: both layers in one launch, 2: conv2 alone on the new kernel, 0: the generic paths
            L.check(lib.otr_debug_set(22, direct), 'debug_set')
            y = ops.ConvSubsampleFn.apply(x, w1g, b1, w2, b2)
            outs.append(y.detach().float())
            a1s.append(y.grad_fn.saved_tensors[2].clone())
        a, a2, g = outs
        assert a.shape == g.shape and torch.isfinite(a).all()
        # act1 (saved for the backward pass) is the same tensor bit for bit whoever computed it, hence also conv2's output of the two
        # forms of the new kernel
        assert torch.equal(a1s[0], a1s[2]) and torch.equal(a1s[1], a1s[2])
        assert torch.equal(a, a2)
        # same 16-bit operands, fp32 accumulation in a different order, one rounding at the end on both sides
        assert rel(a, g) < (3e-3 if mode == 'bf16' else 4e-4), rel(a, g)
        assert float((a - g).abs().max()) < (0.1 if mode == 'bf16' else 0.02)
        h1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=(0, 1)))
        h2 = F.relu(F.conv2d(h1, w2, b2, stride=2, padding=(0, 1)))
        ref = h2.permute(0, 2, 3, 1).reshape(B, h2.size(2), -1)
        assert rel(a, ref) < (1.5e-2 if mode == 'bf16' else 2e-3), rel(a, ref)
        assert rel(a, ref) < 1.3 * rel(g, ref) + 1e-5
    finally:
        lib.otr_debug_set(22, 1)
        ops.set_compute_dtype('bf16')
