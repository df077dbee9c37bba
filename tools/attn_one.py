"""Run the encoder self-attention shape repeatedly (for rocprofv3 --pmc passes).  usage: attn_one.py [fwd|bwd] [iters]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops
ops.set_compute_dtype('bf16')
kind = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, T, H, d = 32, 249, 4, 256
qkv = torch.randn(B, T, 3 * d, device='cuda').to(torch.bfloat16).requires_grad_(kind == 'bwd')
mask = torch.ones(B, T, dtype=torch.uint8, device='cuda')
g = torch.randn(B, T, d, device='cuda').to(torch.bfloat16)
for _ in range(iters):
    out = ops.SelfAttentionFn.apply(qkv, mask, H, False)
    if kind == 'bwd':
        out.backward(g)
torch.cuda.synchronize()
