"""CPU (-m "not gpu"): the data-parallel engine with world_size 2 over gloo.

FlatDataParallel is model-agnostic host logic, so it is exercised here on CPU tensors with a small
torch module whose loss is normalised per replica by its own token count -- exactly the property
that makes nn.DataParallel's result (mean of per-replica losses: trainer.py:208) equal to
all-reduce-sum / N."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opentransformer_amd.dp import FlatDataParallel


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.emb = torch.nn.Embedding(11, 8)
        self.l1 = torch.nn.Linear(8, 16)
        self.out = torch.nn.Linear(16, 11)
        self.out2 = torch.nn.Linear(8, 11, bias=False)
        self.out2.weight = self.emb.weight          # tied, like decoder embedding/output_layer

    def forward(self, tok, tgt):
        h = torch.relu(self.l1(self.emb(tok)))
        logits = self.out(h) + self.out2(self.emb(tok))
        keep = tgt != 0
        nll = torch.nn.functional.cross_entropy(logits.view(-1, 11), tgt.view(-1), reduction='none')
        return (nll * keep.view(-1)).sum() / keep.sum()      # normalised by THIS replica's token count


def _data():
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(1, 11, (8, 6), generator=g)
    tgt = torch.randint(1, 11, (8, 6), generator=g)
    tgt[1, 3:] = 0
    tgt[6, 1:] = 0
    return tok, tgt


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = Tiny()
    if rank == 1:                                   # replicas start different; broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp = FlatDataParallel(model)
    dp.broadcast_parameters(0)
    tok, tgt = _data()
    shard = slice(rank * 4, rank * 4 + 4)           # contiguous utterance shards, like DataParallel.scatter
    dp.zero_grad()
    loss = dp(tok[shard], tgt[shard])
    loss.backward()
    scale, _ = dp.all_reduce_gradients()
    grads = dp.packed_grads() * scale
    if rank == 0:
        torch.save({'grad': grads.clone(), 'loss': loss.detach(), 'n': dp.param_numel,
                    'views_ok': all(p.grad.data_ptr() >= dp.flat_grad.data_ptr() for p in dp.params)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_equals_dataparallel_semantics(tmp_path):
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # reference semantics: loss = mean_i(loss_i), each shard normalised by its own token count
    model = Tiny()
    tok, tgt = _data()
    loss = 0.5 * (model(tok[:4], tgt[:4]) + model(tok[4:], tgt[4:]))
    params, seen = [], set()
    for p in model.parameters():
        if id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    ref = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, params)])
    assert got['n'] == ref.numel() and got['views_ok']
    torch.testing.assert_close(got['grad'], ref, rtol=1e-5, atol=1e-6)
    # and it is NOT the globally token-normalised mean (shards have different token counts)
    glob = torch.cat([g.reshape(-1) for g in torch.autograd.grad(model(tok, tgt), params)])
    assert (glob - ref).abs().max() > 1e-4


def test_flat_buffers_alias_parameters():
    model = Tiny()
    before = [p.detach().clone() for p in model.parameters()]
    dp = FlatDataParallel(model)
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b)                   # flattening preserves values
    dp.flat_param.add_(1.0)
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b + 1.0)             # parameters are views into the flat buffer
    tok, tgt = _data()
    dp.zero_grad()
    dp(tok, tgt).backward()
    assert float(dp.flat_grad.abs().sum()) > 0              # autograd accumulated into the flat views
    assert model.emb.weight.grad.data_ptr() == model.out2.weight.grad.data_ptr()


def _worker_lp(rank, world, port, out):
    """same as _worker with a bf16 gradient payload and non-flattened parameters (the per-tensor broadcast path)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = Tiny()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp = FlatDataParallel(model, flatten_params=False, grad_comm_dtype=torch.bfloat16)
    dp.broadcast_parameters(0)                       # this path used to call a method that did not exist
    tok, tgt = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    dp.zero_grad()
    dp(tok[shard], tgt[shard]).backward()
    scale, _ = dp.all_reduce_gradients()
    if rank == 1:
        torch.save({'grad': (dp.packed_grads() * scale).clone(),
                    'params': torch.cat([p.detach().reshape(-1) for p in dp.params])}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_payload_and_per_tensor_broadcast(tmp_path):
    out = str(tmp_path / 'r1.pt')
    mp.spawn(_worker_lp, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = Tiny()
    tok, tgt = _data()
    loss = 0.5 * (model(tok[:4], tgt[:4]) + model(tok[4:], tgt[4:]))
    params, seen = [], set()
    for p in model.parameters():
        if id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    ref = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, params)])
    assert torch.equal(got['params'], torch.cat([p.detach().reshape(-1) for p in params]))     # rank 1 received rank 0's weights
    # each rank's gradient is rounded to bf16 before the sum: 2^-9 relative per element
    assert float((got['grad'] - ref).norm() / ref.norm()) < 6e-3


def test_dropped_gradient_views_are_detected():
    """torch's default zero_grad(set_to_none=True) drops the flat views; the engine must notice instead of
    training on an all-zero flat buffer."""
    model = Tiny()
    dp = FlatDataParallel(model)
    tok, tgt = _data()
    model.zero_grad()                                        # torch default: .grad = None
    assert all(p.grad is None for p in model.parameters())
    dp.zero_grad()                                           # puts the views back
    dp(tok, tgt).backward()
    assert float(dp.flat_grad.abs().sum()) > 0
    dp.all_reduce_gradients()
    model.l1.weight.grad = torch.zeros_like(model.l1.weight)  # a foreign gradient tensor
    with pytest.raises(RuntimeError, match='flat'):
        dp.all_reduce_gradients()


def test_gradient_hand_over_left_behind_is_an_error():
    """ops links (ResidualLink / PreNormLink / LnOutLink) pass gradients between autograd nodes outside autograd's bookkeeping; a
    hand-over nobody picked up by the end of the backward pass must raise, not vanish (ops._park / ops._check_parked)."""
    import pytest
    from opentransformer_amd import ops
    link, lo = ops.ResidualLink(), ops.LnOutLink()
    link.buf, lo.result = torch.zeros(1), (torch.zeros(1),)
    done = ops.ResidualLink()                     # picked up: buf is None again
    ops._parked.extend([link, lo, done])
    with pytest.raises(RuntimeError, match='never picked up'):
        ops._check_parked()
    assert link.buf is None and lo.result is None and not ops._parked
    ops._parked.append(done)
    ops._check_parked()                           # nothing left: no error


def test_fused_adam_state_dict_is_torch_adams_layout():
    """FusedAdam.state_dict / load_state_dict (train/trainer.py:280-290, run.py:49-62): the layout is torch.optim.Adam's, so an
    optimizer checkpoint of the reference resumes here and the other way round.  Host logic only (no update is run)."""
    from opentransformer_amd.dp import FusedAdam
    tok, tgt = _data()
    ref = Tiny()
    adam = torch.optim.Adam(filter(lambda p: p.requires_grad, ref.parameters()), lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    for _ in range(3):
        adam.zero_grad()
        ref(tok, tgt).backward()
        adam.step()
    model = Tiny()
    dp = FlatDataParallel(model)
    opt = FusedAdam(dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    opt.load_state_dict(adam.state_dict())                       # the reference's checkpoint -> flat buffers
    assert float(opt.state[0]) == 3.0 and opt.global_step == 5   # scheduler.py: global_step starts at 1, +1 at build, +1 per update
    assert abs(float(opt.state[2]) - (1 - 0.9 ** 3)) < 1e-6 and abs(float(opt.state[3]) - (1 - 0.98 ** 3)) < 1e-6
    ref_params = [p for p in ref.parameters()]
    for p, q, off in zip(dp.params, ref_params, dp.offsets):
        st = adam.state[q]
        assert torch.equal(opt.exp_avg[off:off + p.numel()].view(p.shape), st['exp_avg'])
        assert torch.equal(opt.exp_avg_sq[off:off + p.numel()].view(p.shape), st['exp_avg_sq'])
    sd = opt.state_dict()
    adam2 = torch.optim.Adam(filter(lambda p: p.requires_grad, Tiny().parameters()), lr=1.0)
    adam2.load_state_dict(sd)                                    # and back: torch accepts it as its own
    for (k, a), (_, b) in zip(sorted(adam.state_dict()['state'].items()), sorted(adam2.state_dict()['state'].items())):
        assert float(a['step']) == float(b['step']) and torch.equal(a['exp_avg'], b['exp_avg']) and torch.equal(a['exp_avg_sq'], b['exp_avg_sq'])
    assert adam2.param_groups[0]['betas'] == (0.9, 0.98) and adam2.param_groups[0]['weight_decay'] == 1e-6
    opt2 = FusedAdam(FlatDataParallel(Tiny()), lr=1e-3)
    opt2.load_state_dict(sd)                                     # own round trip incl. the extra state block
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.state, opt.state)
    # the reference checkpoint's scheduler counter ('global_step', trainer.py:284) is separate from Adam's t
    opt3 = FusedAdam(FlatDataParallel(Tiny()), lr=1e-3)
    opt3.load_state_dict(adam.state_dict(), global_step=100)     # e.g. resumed with --from_step
    assert float(opt3.state[0]) == 3.0 and opt3.global_step == 100 and opt3.step_offset == 97.0
    assert opt3.state_dict()['otr']['global_step'] == 100
    opt4 = FusedAdam(FlatDataParallel(Tiny()), lr=1e-3)
    opt4.load_state_dict(opt3.state_dict())                      # ... and it survives the own round trip
    assert opt4.global_step == 100
    sd_c = adam.state_dict()
    sd_c['param_groups'][0]['lr'] = 0.25
    opt5 = FusedAdam(FlatDataParallel(Tiny()), lr=1e-3, noam=None)
    opt5.load_state_dict(sd_c)
    assert opt5.lr == 0.25                                       # constant-lr run: the next tick must not fall back to the constructor lr
    with pytest.raises(ValueError):
        bad = adam.state_dict()
        bad['param_groups'][0]['params'] = bad['param_groups'][0]['params'][:-1]
        opt2.load_state_dict(bad)


class TinySplit(Tiny):
    """Tiny with the early-group mark between the trunk (emb, l1) and the head (out)"""

    def forward(self, tok, tgt):
        from opentransformer_amd import ops
        h = torch.relu(self.l1(self.emb(tok)))
        h = ops.early_mark(h, self)
        logits = self.out(h) + self.out2(self.emb(tok))
        keep = tgt != 0
        nll = torch.nn.functional.cross_entropy(logits.view(-1, 11), tgt.view(-1), reduction='none')
        return (nll * keep.view(-1)).sum() / keep.sum()


def _split_worker(rank, world, port, out, payload):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    tok, tgt = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    res = {}
    from opentransformer_amd import ops
    for name, early in (('single', False), ('split', True), ('staged', True)):
        model = TinySplit()
        dp = FlatDataParallel(model, early_modules=[model.out] if early else None, grad_comm_dtype=payload)
        dp.zero_grad()
        ops.set_stage_split(name == 'staged')
        try:
            loss = dp(tok[shard], tgt[shard])
            if name == 'staged':
                # the graph-cut form (bench.py at N > 1): loss.backward() stops at the mark, the early group's collective starts,
                # the trunk's backward follows as a second pass
                issued = []
                stages = dp.backward_staged(loss, between=lambda: (dp.start_early_reduce(), issued.append(dp._early_state is not None)))
                res[name + '_stages'] = len(stages)
                res[name + '_early_issued'] = issued[0]
            else:
                loss.backward()
                res[name + '_early_issued'] = dp._early_state is not None
        finally:
            ops.set_stage_split(False)
        scale, _ = dp.all_reduce_gradients()
        res[name] = (dp.packed_grads() * scale).clone()
        res[name + '_early_end'] = dp.early_end
    ops.set_early_callback(None)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def _accum_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    tok, tgt = _data()
    res = {}
    from opentransformer_amd import ops
    micro = [slice(rank * 4, rank * 4 + 2), slice(rank * 4 + 2, rank * 4 + 4)]
    for name, early in (('single', False), ('split', True)):
        model = TinySplit()
        dp = FlatDataParallel(model, early_modules=[model.out] if early else None)
        dp.zero_grad()
        with dp.no_sync():                                   # every backward pass of the step except the last
            dp(tok[micro[0]], tgt[micro[0]]).backward()
            res[name + '_issued_inside_no_sync'] = dp._early_state is not None
        dp(tok[micro[1]], tgt[micro[1]]).backward()
        res[name + '_issued_by_last'] = dp._early_state is not None
        scale, _ = dp.all_reduce_gradients()
        res[name] = (dp.packed_grads() * scale).clone()
        if early:
            # a further backward pass of a step whose early collective has started would write into the buffer that is being
            # reduced: refused, not raced (its own step here: the refused pass leaves the head's gradients half accumulated)
            dp.zero_grad()
            dp(tok[micro[0]], tgt[micro[0]]).backward()
            try:
                dp(tok[micro[1]], tgt[micro[1]]).backward()
                res['second_backward'] = 'ran'
            except RuntimeError as e:
                res['second_backward'] = 'refused' if 'no_sync' in str(e) else repr(e)
            dp.all_reduce_gradients()
        # an abandoned step: the early collective is in flight, zero_grad() joins it before it clears the buffer
        if early:
            dp.zero_grad()
            dp(tok[micro[1]], tgt[micro[1]]).backward()
            assert dp._early_state is not None
            dp.zero_grad()
            res['abandoned_left'] = (dp._early_state is not None, float(dp._grad_store.abs().sum()))
    # two engines in one process (an ASR model and an LM): each mark fires ITS engine
    a, b = TinySplit(), TinySplit()
    dpa, dpb = FlatDataParallel(a, early_modules=[a.out]), FlatDataParallel(b, early_modules=[b.out])   # b registered last
    dpa.zero_grad(); dpb.zero_grad()
    dpa(tok[micro[0]], tgt[micro[0]]).backward()
    res['two_engines'] = (dpa._early_state is not None, dpb._early_state is not None)
    dpa.all_reduce_gradients(); dpb.all_reduce_gradients()
    ops.set_early_callback(None)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_early_group_with_gradient_accumulation_and_two_engines(tmp_path):
    """ADVICE r04 (dp.py early reduce): (1) with accumulation the early collective starts in the LAST backward pass only
    (dp.no_sync() around the others) and a backward pass after it has started is refused; (2) the callback belongs to the model
    that placed the mark, not to the engine that registered last; (3) zero_grad() joins a collective still in flight."""
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_accum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert not got['split_issued_inside_no_sync'] and got['split_issued_by_last'] and not got['single_issued_by_last']
    assert got['second_backward'] == 'refused', got['second_backward']
    torch.testing.assert_close(got['split'], got['single'], rtol=0, atol=0)
    assert got['abandoned_left'] == (False, 0.0)
    assert got['two_engines'] == (True, False)


@pytest.mark.parametrize('payload', [None, torch.bfloat16])
def test_two_group_allreduce_equals_the_single_collective(tmp_path, payload):
    """the gradient all-reduce split into an EARLY group (the modules behind ops.early_mark: reduced while the
    rest of the backward pass still runs) and the rest gives the same reduced gradient as one collective -- world size 2, gloo."""
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_split_worker, args=(2, _free_port(), out, payload), nprocs=2, join=True)
    got = torch.load(out)
    assert got['split_early_end'] > 0 and got['single_early_end'] == 0
    assert got['split_early_issued'] and not got['single_early_issued']        # the early collective was started INSIDE backward
    torch.testing.assert_close(got['split'], got['single'], rtol=0, atol=0)
    assert got['staged_stages'] == 1 and got['staged_early_issued']            # ... and BETWEEN the two passes of the staged form
    torch.testing.assert_close(got['staged'], got['single'], rtol=0, atol=0)


def test_staged_backward_left_half_done_is_an_error():
    """ops.set_stage_split(True) + a plain loss.backward(): the trunk in front of the cut gets no gradient -- the gradient consumers refuse"""
    from opentransformer_amd import ops
    tok, tgt = _data()
    model = TinySplit()
    dp = FlatDataParallel(model)
    ops.set_stage_split(True)
    try:
        dp.zero_grad()
        dp(tok[:4], tgt[:4]).backward()
        with pytest.raises(RuntimeError, match='never taken'):
            dp.all_reduce_gradients()
        dp.zero_grad()
        dp.backward_staged(dp(tok[:4], tgt[:4]))          # the supported form leaves nothing behind
        dp.all_reduce_gradients()
        assert float(model.l1.weight.grad.abs().sum()) > 0
    finally:
        ops.set_stage_split(False)
        ops.set_early_callback(None)


def test_stacked_views_of_a_flat_buffer_launch_nothing():
    """ops.stack_rows / stack_cols (the decoder layers' vk_proj weights and biases, ops.CrossKVAllFn): a VIEW when the pieces sit one
    after the other in ONE allocation -- also for parameters whose .data was pointed into the flat buffer (they are not autograd
    views: `_base` is None) -- and a real concatenation otherwise."""
    from opentransformer_amd import ops
    flat = torch.arange(64, dtype=torch.float32)
    ps = [torch.nn.Parameter(torch.empty(2, 4)) for _ in range(3)]
    for i, p in enumerate(ps):
        p.data = flat[8 * i:8 * i + 8].view(2, 4)
    v = ops.stack_rows(ps)
    assert v.data_ptr() == flat.data_ptr() and v.shape == (6, 4) and not v.requires_grad
    assert torch.equal(v, flat[:24].view(6, 4))
    bs = [torch.nn.Parameter(torch.empty(4)) for _ in range(2)]
    for i, b in enumerate(bs):
        b.data = flat[32 + 4 * i:36 + 4 * i]
    vb = ops.stack_rows(bs)
    assert vb.data_ptr() == bs[0].data_ptr() and torch.equal(vb, flat[32:40])
    gap = ops.stack_rows([ps[0], ps[2]])                     # not adjacent: a copy
    assert gap.data_ptr() != flat.data_ptr() and torch.equal(gap, torch.cat([ps[0], ps[2]]))
    other = ops.stack_rows([flat[:8].view(2, 4), torch.zeros(2, 4)])        # different allocations: a copy
    assert other.shape == (4, 4) and other.data_ptr() != flat.data_ptr()
    big = flat[:48].view(4, 12)
    cols = ops.stack_cols([big[:, 0:4], big[:, 4:8], big[:, 8:12]])
    assert cols.data_ptr() == flat.data_ptr() and torch.equal(cols, big)
    assert torch.equal(ops.stack_cols([big[:, 0:4], big[:, 8:12]]), torch.cat([big[:, 0:4], big[:, 8:12]], 1))


def test_mask_cast_is_remembered_on_the_tensor():
    from opentransformer_amd import ops
    frames = torch.rand(3, 40) > 0.3
    m = frames[:, 1::2][:, :19][:, 1::2][:, :9]              # frontend/conv.py:78-83, twice
    a = ops._mask_u8(m, 3, 9)
    assert a.dtype == torch.uint8 and a.is_contiguous() and torch.equal(a.bool(), m)
    assert ops._mask_u8(m, 3, 9) is a                        # the decoder's memory mask: same tensor, no second cast
    frames.fill_(False)                                      # in-place change of the base: the version counter moves
    b = ops._mask_u8(m, 3, 9)
    assert b is not a and not b.any()
    assert ops._mask_u8(b, 3, 9) is b and ops._mask_u8(None, 3, 9) is None


def test_row_padded_slots_in_the_flat_layout():
    """dp.py `slot_numel`: a 2-D parameter whose row count is not a multiple of 8 (the 4234-token output layer,
    decoder/transformer.py:153) owns the rows up to the next multiple of 8 inside its slot of the flat buffers; the parameter and
    its gradient stay the [rows, K] head of the slot, everything behind starts 64-element aligned, the extra rows are zero and take
    no gradient (host layout only here: the padded operand images themselves are device-side, tests/test_gpu_dp.py)."""
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.a = torch.nn.Linear(16, 11)         # 11 rows -> 16 in the slot
            self.b = torch.nn.Linear(16, 8)          # already a multiple of 8
            self.c = torch.nn.Linear(12, 13)         # 12 columns: not a multiple of 8, left alone

        def forward(self, x):
            return (self.a(x).sum() + self.b(x).sum() + self.c(x[:, :12]).sum())
    m = M()
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    dp = FlatDataParallel(m)
    off = {id(p): o for p, o in zip(dp.params, dp.offsets)}
    order = sorted(dp.params, key=lambda p: off[id(p)])
    assert all(off[id(p)] % 64 == 0 for p in order)
    for p, q in zip(order, order[1:]):
        room = off[id(q)] - off[id(p)]
        want = 16 * 16 if p is m.a.weight else p.numel()
        assert room == (want + 63) // 64 * 64, (tuple(p.shape), room)
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), ref[n]) and p.data_ptr() == dp.flat_param[off[id(p)]:].data_ptr()
        assert p.grad.data_ptr() == dp.flat_grad[off[id(p)]:].data_ptr() and p.grad.shape == p.shape
    o = off[id(m.a.weight)]
    assert not dp.flat_param[o + 11 * 16:o + 16 * 16].any()
    dp.zero_grad()
    m(torch.randn(5, 16)).backward()
    assert m.a.weight.grad.abs().sum() > 0 and not dp.flat_grad[o + 11 * 16:o + 16 * 16].any()
    assert dp.packed_grads().numel() == sum(p.numel() for p in m.parameters())
