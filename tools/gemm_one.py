"""Run ONE GEMM shape repeatedly (for rocprofv3 --pmc passes).  usage: gemm_one.py kind M N K [xdt wdt ydt] [iters]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opentransformer_amd import ops
ops.set_compute_dtype('bf16')
kind, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dts = {'bf': torch.bfloat16, 'f32': torch.float32}
xdt, wdt, ydt = [dts[a] for a in (sys.argv[5:8] if len(sys.argv) >= 8 else ['bf', 'bf', 'bf'])]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
x = torch.randn(m, k, device='cuda').to(xdt)
w = (torch.randn(n, k, device='cuda') / 16).to(wdt)
dy = torch.randn(m, n, device='cuda').to(ydt)
b = torch.randn(n, device='cuda')
for _ in range(iters):
    if kind == 'fwd':
        ops.linear_fwd_raw(x, w, b, ydt)
    elif kind == 'dgrad':
        ops.linear_dgrad_raw(dy, w, xdt)
    else:
        ops.linear_wgrad_raw(dy, x, w)
torch.cuda.synchronize()
