#!/usr/bin/env python3
"""bench.py -- utterances/sec of the otrans speech-transformer TRAIN step on MI355X.

Workload (BASELINE.json configs[1]): egs/aishell/conf/transformer_baseline.yaml with
frontend.input_size 80, residual_dropout 0.1, synthetic 80-d fbank x 1000 frames, 15 decoder rows,
B = 32 utterances per GPU, bf16 MFMA / fp32 accumulate.  One step = zero grads, forward, backward,
ONE gradient all-reduce (N>1), clip + Adam + Noam update: nothing is skipped in the timed region.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) following the driver's contract, plus `roofline` (dominant kernel,
timed live with events on the launch stream) and `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from opentransformer_amd import synthetic as syn          # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--mode', default='fp16', choices=['fp16', 'bf16', 'fp32'],
                    help='16-bit MFMA operand type: fp16 (default; logits within 1e-3 of the fp32 reference) or bf16; fp32 = exact-fp32 MFMA')
    ap.add_argument('--model', default='transformer', choices=['transformer', 'conformer'],
                    help='transformer = BASELINE configs[1] (the metric); conformer = configs[3] (informative)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=4)
    ap.add_argument('--cpu-iters', type=int, default=8)
    return ap.parse_args()


def cpu_baseline(args):
    """The CPU oracle (a port of the reference path, oracle/otrans_oracle.py) timed on the host
    cores: train fwd+bwd, fp32, same model config, bounded sample."""
    from oracle import otrans_oracle as orc
    from tests import helpers as H
    # torch CPU kernels stop scaling long before the host's 256 hardware threads (measured on the GPU
    # box: 0.24 s/iter at 16 threads, 0.80 s at 64, 173 s at 256), so the baseline uses 16 threads.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = syn.c2_model(0.0)
    parts = H.require_grad(H.filled_state(cfg))
    B = args.cpu_batch
    inputs, targets = syn.synthetic_batch(B, args.frames, 80, 4234, 15, seed=0)
    flat = [t for sd in parts.values() for t in sd.values()]
    times = []
    for it in range(2 + args.cpu_iters):
        for t in flat:
            t.grad = None
        t0 = time.perf_counter()
        loss, _ = orc.speech2text_forward(parts, cfg, inputs, targets)
        loss.backward()
        times.append(time.perf_counter() - t0)
    times = sorted(times[2:])
    med = times[len(times) // 2]
    return {'value': B / med, 'unit': 'utterances/s', 'cores': cores, 'kind': 'port',
            'sample': 'CPU oracle fwd+bwd fp32, B=%d x %d frames, median of %d iters (2 warm-up), %d torch threads'
                      % (B, args.frames, args.cpu_iters, cores)}


def time_dominant_kernel(model, mode):
    """Dominant kernel = the FFN w_1 GEMM (52%% of encoder FLOPs incl. bwd twins): time the forward
    instance [M=B*T', N=2*d_ff, K=d] live with events on the launch stream."""
    from opentransformer_amd import ops
    w1 = model.encoder.blocks[0].feed_forward.w_1
    M = time_dominant_kernel.rows
    adt = ops.act_dtype()
    x = torch.randn(M, w1.in_features, device=w1.weight.device).to(adt)      # the model feeds the bf16 twin
    wq = ops.weight_lp(w1.weight)
    wq = wq if wq is not None else w1.weight
    for _ in range(5):
        ops.linear_fwd_raw(x, wq, w1.bias, adt)
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.linear_fwd_raw(x, wq, w1.bias, adt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * M * w1.out_features * w1.in_features
    peak = PEAK_F32_TFLOPS if mode == 'fp32' else PEAK_BF16_TFLOPS
    ach = flops / (ms * 1e-3) / 1e12
    traffic = None
    try:     # HBM bytes per launch measured offline with rocprofv3 PMC passes (profiles/r01_pmc_traffic.json)
        key = 'gemm_fwd_%dx%dx%d_%s' % (M, w1.out_features, w1.in_features, mode)
        traffic = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))[key]['hbm_bytes']
    except Exception:                                          # noqa: BLE001
        pass
    return {'bound': 'mfma', 'kernel': 'gemm_kernel (FFN w_1 forward, M=%d N=%d K=%d)' % (M, w1.out_features, w1.in_features),
            'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
            'avg_launch_ms': ms}


def time_grouped_wgrad(ops, mode):
    """Largest single kernel of the step: the grouped weight-gradient launch (every dW of the backward pass).  Re-run
    it on the operands of the last backward pass (accumulating into the already-consumed gradient buffer)."""
    w, b = ops._wq.get('last', ([], []))
    if not w:
        return None
    flops = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _ in w)
    byts = sum(dy.numel() * dy.element_size() + x.numel() * x.element_size() + 2 * out.numel() * 4 for dy, x, out in w)

    def run():
        ops._wq['w'], ops._wq['b'] = list(w), []
        ops.flush_weight_grads()
    ops._wq['keep_last'] = False
    for _ in range(2):
        run()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    peak = PEAK_F32_TFLOPS if mode == 'fp32' else PEAK_BF16_TFLOPS
    ach = flops / (ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'kernel': 'gemm_grouped_kernel (all %d weight gradients of one backward pass, one launch per '
                                       'operand-type group)' % len(w),
            'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
            'algorithmic_bytes': byts, 'avg_launch_ms': ms}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', init_method='env://')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product has no CPU path)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    import opentransformer_amd as ota
    from opentransformer_amd import ops
    from opentransformer_amd.dp import FlatDataParallel, FusedAdam
    ops.set_compute_dtype(args.mode)

    cfg = syn.c2_model(residual_dropout=0.1) if args.model == 'transformer' else syn.conformer_model(False, 0.1)
    model = ota.SpeechToText(cfg)
    syn.fill_state_dict_(model.state_dict(), 1234)           # identical replicas on every rank
    model = model.to(dev).train()
    dp = FlatDataParallel(model)
    opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0,
                    noam=dict(model_size=cfg['encoder']['d_model'], warmup_steps=12000, factor=1.0))   # *_baseline.yaml train section
    inputs, targets = syn.synthetic_batch(args.batch, args.frames, 80, 4234, 15, seed=rank)
    inputs = {k: v.to(dev) for k, v in inputs.items()}
    targets = {k: v.to(dev) for k, v in targets.items()}
    loss_buf = torch.zeros((), device=dev)

    def fwd_bwd():
        dp.zero_grad()
        ops.next_dropout_step(dev)
        loss, _ = dp(inputs, targets)
        loss.backward()
        loss_buf.copy_(loss.detach())

    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with ops.graph_capture(graph):
                fwd_bwd()
        except Exception as e:                                # noqa: BLE001
            if rank == 0:
                print('hipGraph capture failed (%s: %s); running eagerly' % (type(e).__name__, e), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        scale, _ = dp.all_reduce_gradients()
        opt.step(scale)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # untimed extras (rank 0): where the step time goes, and the operands of one backward pass for the wgrad roofline
    final_loss, final_stats = float(loss_buf.item()), opt.stats()      # state at the end of the timed region
    parts = None
    if True:                                                 # every rank: the collectives must match
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        acc = [0.0, 0.0]
        for _ in range(5):
            ev[0].record()
            if graph is not None:
                graph.replay()
            else:
                fwd_bwd()
            ev[1].record()
            scale, _ = dp.all_reduce_gradients()
            opt.step(scale)
            ev[2].record()
            torch.cuda.synchronize()
            acc[0] += ev[0].elapsed_time(ev[1])
            acc[1] += ev[1].elapsed_time(ev[2])
        parts = {'fwd_bwd_ms': acc[0] / 5, 'allreduce_optimizer_ms': acc[1] / 5}
    if rank == 0:
        ops._wq['keep_last'] = True
        fwd_bwd()                                            # eager pass: queues and flushes once, keeping the items
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()

    if rank == 0:
        global_batch = args.batch * world
        utt_s = global_batch * args.steps / elapsed
        flops_utt = syn.flops_per_utt(cfg, args.frames, 15) if args.model == 'transformer' else 63207.57e6
        time_dominant_kernel.rows = args.batch * (((args.frames - 3) // 2 + 1 - 3) // 2 + 1)
        out = {
            'metric': 'utterances/sec (80-d fbank, ~1000 frames) train fwd+bwd', 'value': utt_s,
            'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.mode, 'data': 'synthetic',
            'config': {'workload': 'AISHELL-1 transformer_baseline.yaml (+input_size 80), 12 enc / 6 dec layers, '
                                   'B=%d/GPU x %d frames x 80-d fbank, 15 decoder rows, V=4234, residual_dropout 0.1; '
                                   'step = fwd + bwd + grad all-reduce + clip/Adam/Noam' % (args.batch, args.frames),
                       'global_batch': global_batch, 'frames': args.frames, 'parallelism': 'dp%d' % world,
                       'hipgraph': graph is not None},
            'loss': final_loss, 'optimizer': final_stats,
            'model_tflops_per_s': utt_s * flops_utt / 1e12,
            'model_mfma_frac': utt_s * flops_utt / 1e12 / world / (PEAK_F32_TFLOPS if args.mode == 'fp32' else PEAK_BF16_TFLOPS),
        }
        st = final_stats
        if st['skipped'] != 0 or not (st['grad_sqnorm'] == st['grad_sqnorm'] and st['grad_sqnorm'] < float('inf')):
            out['INVALID'] = 'non-finite gradient norm: %d optimizer updates were skipped' % int(st['skipped'])
        out['step_breakdown'] = parts
        if args.model == 'transformer':
            out['roofline'] = time_dominant_kernel(model, args.mode)
            rw = time_grouped_wgrad(ops, args.mode)
            if rw is not None:
                out['roofline_wgrad_grouped'] = rw
        else:
            out['config']['workload'] = out['config']['workload'].replace('transformer_baseline.yaml (+input_size 80), 12 enc / 6 dec layers', 'conformer_baseline.yaml, 12 conformer blocks / 6 dec layers')
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
