#!/usr/bin/env python3
"""How the CPU oracle (a port, oracle/otrans_oracle.py) compares in SPEED with the real reference (otrans.model.SpeechToText
imported from /root/reference), on the same cores, same model (transformer_baseline.yaml + input_size 80), same batch: train
fwd+bwd, fp32.  The GPU box has no /root/reference, so bench.py's cpu_baseline times the port there (kind "port") and quotes the
ratio measured HERE (profiles/r04_cpu_ref_vs_port.json) next to it.  Runs only where /root/reference exists.

usage: cpu_ref_vs_port.py [--batch 4] [--threads 8] [--iters 5]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--threads', type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--ref-dropout', type=float, default=0.0, help='residual_dropout of the REFERENCE model (the yaml ships 0.1; the port is timed without dropout)')
    a = ap.parse_args()
    from opentransformer_amd import synthetic as syn
    from tests import helpers as H
    from oracle import otrans_oracle as orc
    torch.set_num_threads(a.threads)
    cfg = syn.c2_model(0.0)
    inputs, targets = syn.synthetic_batch(a.batch, 1000, 80, 4234, 15, seed=0)

    def med(fn):
        fn()
        fn()
        ts = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    parts = H.require_grad(H.filled_state(cfg))
    flat = [t for sd in parts.values() for t in sd.values()]

    def port():
        for t in flat:
            t.grad = None
        loss, _ = orc.speech2text_forward(parts, cfg, inputs, targets)
        loss.backward()
    t_port = med(port)

    sys.path[:0] = [REF, os.path.join(REF, 'otrans', 'module')]      # second entry: the bare `from activation import Swish` of otrans/module/ffn.py:9
    from otrans.model import End2EndModel
    ref = End2EndModel['speech2text'](syn.c2_model(a.ref_dropout))
    for name, sd in (('frontend', ref.frontend), ('encoder', ref.encoder), ('decoder', ref.decoder)):
        sd.load_state_dict({k: v.detach() for k, v in parts[name].items()}, strict=True)
    ref.train()

    def reference():
        ref.zero_grad(set_to_none=True)
        loss, _ = ref(inputs, targets)
        loss.backward()
    t_ref = med(reference)
    out = {'batch': a.batch, 'threads': a.threads, 'iters': a.iters, 'host_cores': os.cpu_count(),
           'reference_s_per_batch': t_ref, 'port_s_per_batch': t_port,
           'reference_utt_per_s': a.batch / t_ref, 'port_utt_per_s': a.batch / t_port,
           'port_over_reference': t_ref / t_port, 'reference_residual_dropout': a.ref_dropout,
           'note': 'median of %d iterations after 2 warm-ups, torch %s, same weights / batch; dropout 0' % (a.iters, torch.__version__)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
