import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from tests.test_gpu_round6 import _small_model, rel
from opentransformer_amd import ops
from opentransformer_amd.dp import FlatDataParallel, FusedAdam
ops.set_compute_dtype('fp16')
model, inputs, targets = _small_model()
dp = FlatDataParallel(model)
FusedAdam(dp, lr=1e-3, loss_scale=256.0)
def fwd_bwd():
    loss, _ = dp(inputs, targets); ops.backward(loss)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        dp.zero_grad(); fwd_bwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
one = {n: p.grad.clone() for n, p in model.named_parameters()}
dp.zero_grad()
g = torch.cuda.CUDAGraph()
with ops.graph_capture(g):
    fwd_bwd()
dp.zero_grad(); g.replay(); g.replay(); torch.cuda.synchronize()
for n, p in model.named_parameters():
    r = rel(p.grad, 2 * one[n])
    if r > 1e-4:
        print(n, tuple(p.shape), 'rel vs 2x %.3f' % r, 'rel vs 1x %.3f' % rel(p.grad, one[n]), 'vs 3x %.3f' % rel(p.grad, 3 * one[n]))
# eager accumulation for comparison
dp.zero_grad(); fwd_bwd(); fwd_bwd(); torch.cuda.synchronize()
for n, p in model.named_parameters():
    r = rel(p.grad, 2 * one[n])
    if r > 1e-4:
        print('EAGER', n, 'rel vs 2x %.3f' % r)
