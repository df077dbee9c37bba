#!/usr/bin/env python3
"""Where does the error of the layer-0 attention gradients come from (review r05, "Tighten the parity instruments" (b))?
numpy / fp64 emulation of one attention head's backward pass (T = 200, head dim 64) on inputs scaled like the tests' (x ~ N(0, s^2),
nn.Linear default init), three ways against exact arithmetic on the UNROUNDED q, k, v:
  do_o          the kernels' arithmetic: 16-bit q, k, v, O, dO; delta = rowsum(dO . O) from the rounded operands; P, dS rounded
  p_dp          the same with delta = rowsum(P . dP) from the fp32 P and dP (sum_j dS_ij = 0 exactly)
  operands_only exact arithmetic on the 16-bit-rounded q, k, v (nothing else rounded)
Relative errors of (dq, dk, dv).  At s = 16 (layer 0: sqrt(d) x embedding) all three read 1.6e-2: the error is made by rounding the
OPERANDS in front of a saturated softmax, before any backward arithmetic runs; the choice of delta moves the third digit.
Output kept in profiles/r06_delta_study.txt."""
import numpy as np
rng = np.random.default_rng(0)
T, d, dk = 200, 256, 64
def h(a): return a.astype(np.float16).astype(np.float64)   # fp16 rounding
def run(xscale):
    x = rng.standard_normal((T, d)) * xscale
    W = rng.uniform(-1/16, 1/16, (3*dk, d))
    qkv = x @ W.T
    q, k, v = qkv[:, :dk], qkv[:, dk:2*dk], qkv[:, 2*dk:]
    dO = rng.standard_normal((T, dk))
    sc = 1/np.sqrt(dk)
    def bwd(q, k, v, dO, mode):
        S = (q @ k.T) * sc
        m = S.max(1, keepdims=True)
        lse = m + np.log(np.exp(S - m).sum(1, keepdims=True))
        P = np.exp(S - lse)
        O = P @ v
        if mode == 'exact':
            dP = dO @ v.T
            delta = (dO * O).sum(1, keepdims=True)
            dS = P * (dP - delta)
            return dS @ k * sc, dS.T @ q * sc, P.T @ dO
        # 16-bit kernels: operands rounded, fp32 accumulate (emulated in fp64), P / dS rounded before their MFMA
        O16 = h(O); dO16 = h(dO)
        dP = dO16 @ v.T
        if mode == 'do_o':
            delta = (dO16 * O16).sum(1, keepdims=True)
        else:
            delta = (P * dP).sum(1, keepdims=True)
        dS = h(P * (dP - delta))
        return h(dS @ k * sc), h(dS.T @ q * sc), h(h(P).T @ dO16)
    ref = bwd(q, k, v, dO, 'exact')
    q16, k16, v16 = h(q), h(k), h(v)
    res = {}
    for mode in ('do_o', 'p_dp'):
        got = bwd(q16, k16, v16, dO, mode)
        res[mode] = [np.linalg.norm(g - r) / np.linalg.norm(r) for g, r in zip(got, ref)]
    # same 16-bit operands but otherwise exact: what the operand rounding alone costs
    got = bwd(q16, k16, v16, dO, 'exact')
    res['operands_only'] = [np.linalg.norm(g - r) / np.linalg.norm(r) for g, r in zip(got, ref)]
    return res
for xs in (1.0, 4.0, 16.0):
    r = run(xs)
    print('x scale', xs, {k: ['%.2e' % e for e in v] for k, v in r.items()})
