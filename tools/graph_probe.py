"""Diagnostic: does torch.cuda.CUDAGraph capture/replay the ctypes-launched train step?  Prints a
timestamped line per phase (flushes), dumps all Python stacks and exits if a phase stalls."""
import faulthandler
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.time()


def say(*a):
    print('[%7.2fs]' % (time.time() - T0), *a, flush=True)


def main():
    import opentransformer_amd as ota
    from opentransformer_amd import ops, synthetic as syn
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    dev = torch.device('cuda', 0)
    if which == 'c1':
        cfg, B, T, V = syn.c1_model(0.1), 4, 200, 100
    else:
        cfg, B, T, V = syn.c2_model(0.1), 32, 1000, 4234
    model = ota.SpeechToText(cfg)
    syn.fill_state_dict_(model.state_dict(), 1234)
    model = model.to(dev).train()
    dp = FlatDataParallel(model)
    opt = FusedAdam(dp, noam=dict(model_size=256, warmup_steps=12000, factor=1.0))
    inputs, targets = syn.synthetic_batch(B, T, 80, V, 15 if which != 'c1' else 8, seed=0)
    inputs = {k: v.to(dev) for k, v in inputs.items()}
    targets = {k: v.to(dev) for k, v in targets.items()}
    loss_buf = torch.zeros((), device=dev)

    def fwd_bwd():
        dp.zero_grad()
        ops.next_dropout_step(dev)
        loss, _ = dp(inputs, targets)
        loss.backward()
        loss_buf.copy_(loss.detach())

    say('model built', which)
    for i in range(3):
        faulthandler.dump_traceback_later(120, exit=True)
        t = time.time()
        fwd_bwd()
        torch.cuda.synchronize()
        say('eager fwd_bwd %d: %.1f ms loss %.5f grad_norm %.5f' % (i, (time.time() - t) * 1e3, loss_buf.item(),
                                                                float(dp.flat_grad.double().norm())))
    faulthandler.dump_traceback_later(120, exit=True)
    t = time.time()
    opt.step(1.0)
    torch.cuda.synchronize()
    say('optimizer step: %.2f ms' % ((time.time() - t) * 1e3), opt.stats())

    faulthandler.dump_traceback_later(180, exit=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    say('side-stream warmup done')
    g = torch.cuda.CUDAGraph()
    t = time.time()
    with torch.cuda.graph(g):
        fwd_bwd()
    say('capture done in %.1f ms' % ((time.time() - t) * 1e3))
    for i in range(5):
        faulthandler.dump_traceback_later(120, exit=True)
        t = time.time()
        g.replay()
        torch.cuda.synchronize()
        say('replay %d: %.2f ms loss %.5f grad_norm %.5f' % (i, (time.time() - t) * 1e3, loss_buf.item(),
                                                           float(dp.flat_grad.double().norm())))
        bad = [(k, float(p.grad.abs().max()), int((p.grad.abs() > 1e6).sum()), p.grad.numel())
               for k, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 1e6]
        if bad:
            say('HUGE grads (name, max, count>1e6, numel):', bad[:10], '(%d tensors)' % len(bad))
            k, p = [(k, p) for k, p in model.named_parameters() if k == bad[0][0]][0]
            idx = (p.grad.abs() > 1e6).nonzero()[:8].tolist()
            say('  first bad indices:', idx, 'values', [float(p.grad[tuple(i)]) for i in idx])
    faulthandler.dump_traceback_later(120, exit=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
        opt.step(1.0)
    e1.record()
    torch.cuda.synchronize()
    say('10x (replay + optimizer): %.2f ms / step' % (e0.elapsed_time(e1) / 10))
    faulthandler.cancel_dump_traceback_later()
    if which == 'cpu':
        from oracle import otrans_oracle as orc
        from tests import helpers as H
        parts = H.require_grad(H.filled_state(syn.c2_model(0.0)))
        ci, ct = syn.synthetic_batch(4, 1000, 80, 4234, 15, seed=0)
        for nt in (16, 64, os.cpu_count()):
            torch.set_num_threads(nt)
            for it in range(3):
                t = time.time()
                loss, _ = orc.speech2text_forward(parts, syn.c2_model(0.0), ci, ct)
                loss.backward()
                say('cpu oracle threads=%d iter %d: %.2f s' % (nt, it, time.time() - t))


if __name__ == '__main__':
    main()
