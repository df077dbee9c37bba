"""Which torch (aten) operators still launch kernels inside a Conformer training step, and from which line of this package?
One eager step under torch.profiler with stacks; prints aten ops with device time grouped by the innermost opentransformer_amd frame."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opentransformer_amd as ota                                   # noqa: E402
from opentransformer_amd import ops, synthetic as syn               # noqa: E402
from opentransformer_amd.dp import FlatDataParallel, FusedAdam      # noqa: E402


def main():
    ops.set_compute_dtype('fp16')
    model_kind = sys.argv[1] if len(sys.argv) > 1 else 'conformer'
    cfg = syn.conformer_model(False, 0.1) if model_kind == 'conformer' else syn.c2_model(0.1)
    model = ota.SpeechToText(cfg).cuda().train()
    dp = FlatDataParallel(model)
    opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0,
                    noam=dict(model_size=256, warmup_steps=12000, factor=1.0))
    inputs, targets = syn.synthetic_batch(batch=32, frames=1000, feat_dim=80, vocab=4234, tgt_len=15, seed=1)
    di = {k: v.cuda() for k, v in inputs.items()}
    dt = {k: v.cuda() for k, v in targets.items()}

    def step():
        dp.zero_grad(next_dropout_step=True)
        loss, _ = dp(di, dt)
        ops.backward(loss)
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith('aten::'):
            continue
        dev = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
        if dev <= 0:
            continue
        where = '?'
        for fr in (ev.stack or []):
            if 'opentransformer_amd' in fr or 'bench.py' in fr:
                where = fr.strip()[-90:]
                break
        k = (ev.name, where)
        agg[k][0] += 1
        agg[k][1] += dev
    tot = sum(v[1] for v in agg.values())
    print('aten ops with device time in one step: %d launches, %.1f us' % (sum(v[0] for v in agg.values()), tot))
    for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print('%8.1f us %4d x  %-28s %s' % (t, n, name, where))


if __name__ == '__main__':
    main()
