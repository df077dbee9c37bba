#!/usr/bin/env python3
"""CPU study: which operand rounding drives the bf16-mode logit drift?

Runs the fp32 oracle on the C2 model (B=2 x 1000 frames) with torch.nn.functional.linear / conv2d / matmul patched
to round selected operands the way the HIP path does, and reports rel-Frobenius error of the logits / relative loss
error against the unrounded oracle.  No GPU involved: it answers "what would precision policy X measure" before a
kernel is written.

    python tools/precision_study.py [--batch 2] [--layers 12 6]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opentransformer_amd import synthetic as syn      # noqa: E402
from oracle import otrans_oracle as orc               # noqa: E402
from tests import helpers as H                        # noqa: E402

_lin, _mm, _conv = F.linear, torch.matmul, F.conv2d


def rnd(t, kind):
    if kind == 'f32':
        return t
    if kind == 'bf16':
        return t.to(torch.bfloat16).float()
    if kind == 'f16':
        return t.to(torch.float16).float()
    if kind == 'bf16x2':                 # hi + lo split: 16 mantissa bits
        hi = t.to(torch.bfloat16).float()
        return hi + (t - hi).to(torch.bfloat16).float()
    raise ValueError(kind)


class Policy:
    def __init__(self, act='f32', wgt='f32', attn='f32', out='f32'):
        self.act, self.wgt, self.attn, self.out = act, wgt, attn, out

    def __enter__(self):
        pol = self

        def linear(x, w, b=None):
            y = _lin(rnd(x, pol.act), rnd(w, pol.wgt), b)
            return y

        def matmul(a, b):
            return _mm(rnd(a, pol.attn), rnd(b, pol.attn))

        def conv2d(x, w, b=None, **kw):
            return _conv(rnd(x, pol.act), rnd(w, pol.wgt), b, **kw)
        F.linear, torch.matmul, F.conv2d = linear, matmul, conv2d
        orc.F.linear, orc.torch.matmul, orc.F.conv2d = linear, matmul, conv2d
        return self

    def __exit__(self, *a):
        F.linear, torch.matmul, F.conv2d = _lin, _mm, _conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--frames', type=int, default=1000)
    args = ap.parse_args()
    torch.set_num_threads(8)
    cfg = syn.c2_model(0.0)
    parts = H.filled_state(cfg)
    inputs, targets = syn.synthetic_batch(args.batch, args.frames, 80, 4234, 15, seed=0)
    with torch.no_grad():
        ref_loss, ref = orc.speech2text_forward(parts, cfg, inputs, targets)
    rl = ref['logits']

    def report(name, **kw):
        with torch.no_grad(), Policy(**kw):
            loss, aux = orc.speech2text_forward(parts, cfg, inputs, targets)
        e = ((aux['logits'] - rl).norm() / rl.norm()).item()
        m = ((aux['memory'] - ref['memory']).norm() / ref['memory'].norm()).item()
        print('%-46s logits_rel %.2e  memory_rel %.2e  loss_rel %.2e' % (name, e, m, abs(loss.item() - ref_loss.item()) / abs(ref_loss.item())))

    report('all fp32 (sanity)')
    report('bf16 act + bf16 wgt + bf16 attn (today)', act='bf16', wgt='bf16', attn='bf16')
    report('bf16 act only', act='bf16')
    report('bf16 wgt only', wgt='bf16')
    report('bf16 attn (q,k,v,P) only', attn='bf16')
    report('bf16 act + bf16 wgt, fp32 attn', act='bf16', wgt='bf16')
    report('f16 everywhere', act='f16', wgt='f16', attn='f16')
    report('f16 act + f16 wgt, bf16 attn', act='f16', wgt='f16', attn='bf16')
    report('bf16x2 act + bf16 wgt + bf16 attn', act='bf16x2', wgt='bf16', attn='bf16')
    report('bf16x2 act + bf16x2 wgt + bf16 attn', act='bf16x2', wgt='bf16x2', attn='bf16')
    report('bf16x2 act + bf16x2 wgt + bf16x2 attn', act='bf16x2', wgt='bf16x2', attn='bf16x2')
    report('bf16 act + bf16x2 wgt + bf16 attn', act='bf16', wgt='bf16x2', attn='bf16')


if __name__ == '__main__':
    main()
