"""Utterance-level data parallelism: one process per GPU, persistent replicas, ONE flat gradient
buffer, ONE all-reduce (RCCL over xGMI; backend "nccl" on ROCm) per step.

Replaces torch.nn.DataParallel in the reference (otrans/train/trainer.py:56-66): no per-step
parameter broadcast (146 MB), no scatter/gather through GPU0, no GIL-bound replica threads.
Equivalence: the reference's loss is mean_i(loss_i) with each replica's loss
normalised by its own token count, so grad = (1/N) * sum_i grad_i  ==  all-reduce-sum, then 1/N
folded into the clip/Adam step.

The engine is model-agnostic host logic (it only touches .grad/.data of nn.Parameters), so the
gloo/CPU tests in tests/test_dp.py exercise exactly this code with world_size 2.
"""
import contextlib
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class FlatDataParallel:
    def __init__(self, module, process_group=None, flatten_params=True, comm='torch', grad_comm_dtype=None, early_modules=None):
        """comm: 'torch' = torch.distributed all_reduce (backend "nccl" IS RCCL on ROCm; gloo in the CPU tests), or
        'rccl' = the library's own communicator (otr_allreduce_*, include/otrans_hip.h): the collective is issued on the
        compute stream through the C ABI, torch.distributed only ships the 128-byte unique id at start-up.
        grad_comm_dtype: None = all-reduce the fp32 flat buffer (bit-exact sum order aside); torch.bfloat16 / torch.float16
        = all-reduce a 16-bit copy (half the xGMI bytes: 73 MB instead of 146 MB for the AISHELL transformer) and widen the
        sum back into the fp32 buffer -- bf16 keeps fp32's exponent range, so loss-scaled fp16-mode gradients cannot
        overflow in the payload.
        early_modules: sub-modules whose gradients are COMPLETE before the rest of the backward pass runs -- for SpeechToText the
        decoder (+ the CTC head): their backward precedes the encoder's.  Their parameters take the front of the flat buffers, and
        when the model signals that point (ops.early_mark on the encoder output, see model.SpeechToText.forward) their slice is
        all-reduced on a side stream WHILE the encoder / frontend backward runs; all_reduce_gradients() then reduces only the rest
        and joins.  Two collectives instead of one; same sums (the reference's nn.DataParallel reduces per parameter,
        train/trainer.py:56-66).  None = one collective at the end."""
        self.module = module
        self.group = process_group
        assert comm in ('torch', 'rccl')
        self.comm, self.grad_comm_dtype = comm, grad_comm_dtype
        self._rccl = None
        self._payload = None
        seen, params = set(), []
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:      # tied weights appear once
                seen.add(id(p))
                params.append(p)
        self.params = params
        # every parameter starts on a 64-element (256-byte) boundary of the flat buffers: gradient kernels then see
        # 16-byte aligned outputs whatever the sizes before them (a 4234-wide bias would misalign everything after it)
        ALIGN = 64
        early_ids = set()
        for m in (early_modules or []):
            early_ids.update(id(p) for p in m.parameters())
        # parameters that one GEMM reads as ONE matrix sit next to each other: the decoder layers' cross-attention key / value
        # projections (module/attention.py:128-134 `vk_proj`, all applied to the same encoder memory: ops.CrossKVAllFn) -- their
        # concatenation is then a VIEW of the flat buffers instead of three torch.cat launches per step
        name_of = {id(p): n for n, p in module.named_parameters()}
        def group_key(p):
            n = name_of.get(id(p), '')
            for suffix in ('src_attn.vk_proj.weight', 'src_attn.vk_proj.bias'):
                if n.endswith(suffix) and p.numel() % ALIGN == 0:
                    return suffix
            return None
        # A Linear whose row count is not a multiple of 8 (the 4234-token output layer, decoder/transformer.py:153) gets the missing
        # rows as part of its slot: the GEMMs of that layer then see [rows8, K] operands with aligned rows everywhere (weight,
        # transposed shadow, gradient; the logits / their gradient with a leading dimension of rows8) and run on the branch-free
        # kernels.  The extra rows are zero and stay zero (zero gradient, zero weight: Adam and the decay leave them alone); the
        # parameter itself is the [rows, K] head of the slot.  A bias finds its 8-padding inside the alignment gap it has anyway.
        def slot_numel(q):
            if q.dim() == 2 and q.shape[0] % 8 != 0 and q.shape[1] % 8 == 0 and q.shape[0] > 8:
                return (q.shape[0] + 7) // 8 * 8 * q.shape[1]
            return q.numel()
        offs, total = [None] * len(params), 0
        self._row_groups = []                       # [(indices of params stacked along dim 0)]
        for want_early in (True, False):            # the early group first: [0, early_end), then everything else
            todo = [i for i, p in enumerate(params) if (id(p) in early_ids) == want_early]
            groups = {}
            for i in todo:
                k = group_key(params[i])
                if k is not None:
                    groups.setdefault((k, tuple(params[i].shape)), []).append(i)
            done = set()
            for i in todo:
                if i in done:
                    continue
                k = group_key(params[i])
                members = groups.get((k, tuple(params[i].shape)), [i]) if k is not None else [i]
                if len(members) > 1:
                    self._row_groups.append(list(members))
                for j in members:
                    offs[j] = total
                    total += (slot_numel(params[j]) + ALIGN - 1) // ALIGN * ALIGN
                    done.add(j)
            if want_early:
                self.early_end = total
        self.offsets = offs
        self._early_state = None                    # None: not issued this step; else (work handle or None, payload slice or None)
        self._side = None
        self.skip_collectives = False               # measurement only (bench.py: exposed all-reduce time = step - step without)
        self.numel = total                                  # flat length incl. alignment gaps
        self.param_numel = sum(p.numel() for p in params)   # true parameter count
        padded = total
        dev, dt = params[0].device, params[0].dtype
        # the gradient buffer carries one extra cell behind the parameters' range (its own 64-element block): the collective
        # sums it like any gradient, and all_reduce_gradients() uses it to make the fault word GLOBAL (see there).  Optimizer
        # and norm only ever see the first `padded` elements.
        # ... and behind that, staging images for the gradients of weights a kernel reads with two axes swapped
        # (nn.ConvFrontEnd.regrouped_weights): the weight-gradient launch accumulates there in the kernel's order, one strided add
        # regroups it into the parameter's layout (ops.LinearFn.backward).  Part of the one allocation zero_grad() clears; NOT part of
        # the buffer the collectives move.
        stage_of, stage_total = {}, 0
        if dev.type == 'cuda' and dt == torch.float32:
            for mod in module.modules():
                regroup = getattr(mod, 'regrouped_weights', None)
                for p, (A, R, S) in (regroup() if callable(regroup) else []):
                    if any(p is q for q in params) and p.dim() == 2 and p.numel() == A * R * S and id(p) not in stage_of:
                        stage_of[id(p)] = (stage_total, (A, S, R))
                        stage_total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self._store_all = torch.zeros(padded + ALIGN + stage_total, device=dev, dtype=dt)
        self._grad_store = self._store_all[:padded + ALIGN]
        for p in params:
            if id(p) in stage_of:
                o, shp = stage_of[id(p)]
                p._otr_regroup_grad = self._store_all[padded + ALIGN + o:padded + ALIGN + o + p.numel()].view(shp)
                p._otr_regroup_state = {'dirty': False}      # True once a backward pass has added the image to the gradient
        self._staged = [p for p in params if id(p) in stage_of]
        self.flat_grad = self._grad_store[:padded]
        self._fault_cell = self._grad_store[padded:padded + 1]
        self.flat_param = torch.empty(padded, device=dev, dtype=dt) if flatten_params else None
        if flatten_params:
            self.flat_param.zero_()         # alignment gaps stay zero: their gradient is zero, Adam leaves them at zero
        for p, off in zip(params, offs):
            n = p.numel()
            if flatten_params:
                self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            p._otr_grad_inplace = True      # ops.grad_target(): backward kernels accumulate straight into the view
            if flatten_params and dev.type == 'cuda' and ((p.dim() == 1 and n % 8 != 0) or slot_numel(p) != n):
                # the row-padded images of this parameter and of its gradient (ops.padded_rows): [rows8, K] / [n8]
                shape8 = ((p.shape[0] + 7) // 8 * 8,) + tuple(p.shape[1:])
                n8 = shape8[0] * (p.shape[1] if p.dim() == 2 else 1)
                if p.dim() <= 2 and n8 <= (slot_numel(p) + ALIGN - 1) // ALIGN * ALIGN:
                    p._otr_pad = {'param': self.flat_param[off:off + n8].view(shape8), 'grad': self.flat_grad[off:off + n8].view(shape8)}
        # gradient buffers with ONE writer per backward pass (ops.register_single_writer_grads): the 2-D weights whose gradient is a
        # deferred Linear weight-gradient product and nothing else -- not an embedding matrix (its scatter-add, and with
        # share_embedding the output layer's product, land in one buffer) -- plus the staging images above
        self._single_writer = set()
        if dev.type == 'cuda' and flatten_params:
            emb = {id(m.weight) for m in module.modules() if isinstance(m, torch.nn.Embedding)}
            for p in params:
                if p.dim() == 2 and id(p) not in emb:
                    self._single_writer.add(p.grad.data_ptr())
                if id(p) in stage_of:
                    self._single_writer.add(p._otr_regroup_grad.data_ptr())
            from . import ops
            ops.register_single_writer_grads(self._single_writer)
        # bf16 shadow of every parameter (GEMM operand form), kept fresh by FusedAdam in the same pass
        self.flat_param_lp = None
        if flatten_params and dev.type == 'cuda' and dt == torch.float32:
            from . import ops
            if ops.is_half():
                self.flat_param_lp = torch.empty(padded, device=dev, dtype=ops.half_dtype())
                for p, off in zip(params, offs):
                    n = p.numel()
                    p._otr_lp_view = self.flat_param_lp[off:off + n].view(p.shape)
                    pad = getattr(p, '_otr_pad', None)
                    if pad is not None:
                        pad['lp'] = self.flat_param_lp[off:off + pad['param'].numel()].view(pad['param'].shape)
                self._build_ffn_packs(module, dev)
                # transposed bf16 shadows of the 2-D weights ([K,N]: dgrad becomes a forward-type GEMM) -- except the weights that
                # have fragment-major packs (the FFNs' w_1 / w_2, the 256- and 768-wide projections: 33 of the 36.5 M parameters
                # of the AISHELL model): their input gradients run on the packs, and refreshing 66 MB of transposes nobody reads
                # cost 35 us of every step.  A path that does not take the packs (fewer than 1024 rows) finds no transposed
                # shadow for them and uses the plain input-gradient GEMM (ops.weight_lpt returns None).
                self.flat_param_lpt = torch.empty(padded, device=dev, dtype=ops.half_dtype())
                table, tiles = [], 0
                grouped = set()
                for members in self._row_groups:     # a stacked group is transposed as ONE matrix: its members' shadows are column slices
                    ps = [params[j] for j in members]
                    if ps[0].dim() != 2:
                        continue
                    rows, cols, o0 = sum(q.shape[0] for q in ps), ps[0].shape[1], offs[members[0]]
                    big = self.flat_param_lpt[o0:o0 + rows * cols].view(cols, rows)
                    r0 = 0
                    for q in ps:
                        q._otr_lpt_view = big[:, r0:r0 + q.shape[0]]
                        r0 += q.shape[0]
                        grouped.add(id(q))
                    table.append([o0, rows, cols, tiles])
                    tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
                # weights a kernel reads with two axes swapped (nn.ConvFrontEnd.regrouped_weights): [A, R, S] -> [A, S, R] is A
                # small transposes, done in the same launch instead of a copy per forward pass
                off_of = {id(p): o for p, o in zip(params, offs)}
                for mod in module.modules():
                    regroup = getattr(mod, 'regrouped_weights', None)
                    for p, (A, R, S) in (regroup() if callable(regroup) else []):
                        if id(p) not in off_of or id(p) in grouped or p.numel() != A * R * S or not p.is_contiguous():
                            continue
                        o0 = off_of[id(p)]
                        p._otr_regroup_view = self.flat_param_lpt[o0:o0 + A * R * S].view(A, S, R)
                        for a in range(A):
                            table.append([o0 + a * R * S, R, S, tiles])
                            tiles += ((R + 63) // 64) * ((S + 63) // 64)
                        grouped.add(id(p))
                for p, off in zip(params, offs):
                    n = p.numel()
                    if id(p) in grouped:
                        continue
                    if p.dim() == 2 and getattr(p, '_otr_lin_packs', None) is None and id(p) not in self._ffn_packed:
                        pad = getattr(p, '_otr_pad', None)
                        rows = pad['param'].shape[0] if pad is not None else p.shape[0]       # the padded matrix: [K, rows8], zero tail columns
                        full = self.flat_param_lpt[off:off + rows * p.shape[1]].view(p.shape[1], rows)
                        p._otr_lpt_view = full[:, :p.shape[0]]
                        if pad is not None:
                            pad['lpt'] = full
                        table.append([off, rows, p.shape[1], tiles])
                        tiles += ((rows + 63) // 64) * ((p.shape[1] + 63) // 64)
                # one launch transposes every such shadow (include/otrans_hip.h: otr_transpose_batched)
                self._lpt_table = torch.tensor(table, dtype=torch.int64, device=dev).reshape(-1, 4)
                self._lpt_tiles = tiles
                self.refresh_lp()
        if dev.type == 'cuda':
            from . import ops
            ops.defer_weight_grads(True)    # weight / bias gradients run as grouped launches at the end of backward
        self._accumulating = False                  # inside no_sync(): backward passes accumulate, no collective may start
        if self.early_end > 0:
            from . import ops
            ops.set_early_callback(self._on_early_mark, modules=list(module.modules()))

    def _build_ffn_packs(self, module, dev):
        """Fragment-major copies of every GLU FFN's weights for the row-block fused FFN kernels (ops.FfnLnFn): one flat
        buffer, one device table, ONE otr_pack_frags launch per optimizer step (csrc/ffn_fused.hip)."""
        from . import ops
        self.flat_pack, self._pack_table, self._pack_blocks, self._ffn_packed = None, None, 0, set()
        off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        rows, total, views = [], 0, []
        for mod in module.modules():
            w1, w2 = getattr(getattr(mod, 'w_1', None), 'weight', None), getattr(getattr(mod, 'w_2', None), 'weight', None)
            if w1 is None or w2 is None or getattr(mod, 'activation', None) != 'glu':
                continue
            if id(w1) not in off_of or id(w2) not in off_of or w1.shape[1] != 256 or (w1.shape[0] // 2) % 256 != 0:
                continue
            F2, d = w1.shape
            r, offs, n = ops.ffn_pack_items(off_of[id(w1)], off_of[id(w2)], F2, d, F2 // 2, total)
            rows += r
            views.append((w1, offs, F2 * d, d * (F2 // 2)))
            total += n
        # every Linear weight the row-block kernels can take (ops.lin_packs / ops._RB_SHAPES): forward + input-gradient pack
        # each.  By SHAPE, not by attribute name: any such weight that reached ops.LinearFn without registered packs would
        # otherwise be served from a cache keyed by (_version, data_ptr), which FusedAdam's raw-pointer update never changes
        lin_views = []
        for w in self.params:
            if w.dim() != 2 or tuple(w.shape) not in ops._RB_SHAPES or not ops._RB:
                continue
            r, n = ops.lin_pack_items(off_of[id(w)], w.shape[0], w.shape[1], total)
            rows += r
            lin_views.append((w, total, w.numel()))
            total += n
        if not rows:
            return
        self.flat_pack = torch.empty(total, device=dev, dtype=self.flat_param_lp.dtype)
        table, blocks = [], 0
        for r in rows:
            table.append(list(r) + [blocks])
            blocks += ((r[3] // 32) * (r[4] // 16) + 3) // 4
        self._pack_table = torch.tensor(table, dtype=torch.int64, device=dev)
        self._pack_blocks = blocks
        for mod in module.modules():
            w1, w2 = getattr(getattr(mod, 'w_1', None), 'weight', None), getattr(getattr(mod, 'w_2', None), 'weight', None)
            if w1 is not None and w2 is not None and any(v[0] is w1 for v in views):
                self._ffn_packed.update((id(w1), id(w2)))
        for w1, offs, n1, n2 in views:
            w1._otr_ffn_packs = (self.flat_pack[offs[0]:offs[0] + n1], self.flat_pack[offs[1]:offs[1] + n2],
                                 self.flat_pack[offs[2]:offs[2] + n2], self.flat_pack[offs[3]:offs[3] + n1])
        for w, o, n in lin_views:
            w._otr_lin_packs = (self.flat_pack[o:o + n], self.flat_pack[o + n:o + 2 * n])

    def refresh_lp(self):
        """re-cast the 16-bit shadows after any out-of-band parameter change (load_state_dict, broadcast, ...)."""
        if self.flat_param_lp is not None:
            from . import ops
            ops.cast_bf16(self.flat_param, self.flat_param_lp)
            self.refresh_transposed()

    def refresh_transposed(self):
        """W^T shadows and the packed FFN weights follow the 16-bit shadows (call after every optimizer step)."""
        if self.flat_param_lp is not None and self._lpt_tiles:
            L.check(L.load().otr_transpose_batched(
                C.c_void_p(self.flat_param_lp.data_ptr()), C.c_void_p(self.flat_param_lpt.data_ptr()),
                C.c_void_p(self._lpt_table.data_ptr()), self._lpt_table.shape[0], self._lpt_tiles, 2,
                C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_transpose_batched')
        if self.flat_param_lp is not None and getattr(self, '_pack_blocks', 0):
            L.check(L.load().otr_pack_frags(
                C.c_void_p(self.flat_param_lp.data_ptr()), C.c_void_p(self.flat_pack.data_ptr()),
                C.c_void_p(self._pack_table.data_ptr()), self._pack_table.shape[0], self._pack_blocks,
                C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_pack_frags')

    def packed_grads(self):
        """gradients without the alignment gaps, in parameter order (tests / checkpointing)"""
        return torch.cat([self.flat_grad[o:o + p.numel()] for p, o in zip(self.params, self.offsets)])

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def __call__(self, *args, **kw):
        return self.module(*args, **kw)

    def _check_grad_views(self, reinstall=False):
        """Every parameter's .grad must still be its view of the flat buffer: the backward kernels write there
        (ops.grad_target) and the all-reduce / optimizer read there.  torch's default `module.zero_grad()` /
        `optimizer.zero_grad(set_to_none=True)` drop the views (.grad = None; the next backward would then allocate
        ordinary gradients OUTSIDE the flat buffer and training would silently become weight decay only)."""
        es = self.flat_grad.element_size()
        base = self.flat_grad.data_ptr()
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is not None and g.data_ptr() == base + off * es:
                continue
            if g is None and reinstall:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p.data)
                continue
            raise RuntimeError('FlatDataParallel: the .grad of a parameter of shape %s is no longer its view of the flat '
                               'gradient buffer (was module.zero_grad() / optimizer.zero_grad(set_to_none=True) called?). '
                               'Use FlatDataParallel.zero_grad().' % (tuple(p.shape),))

    def _join_early(self):
        """wait for an early collective that is still in flight (work handle + side stream) and forget it"""
        st, self._early_state = self._early_state, None
        if st is not None:
            work, _ = st
            if work is not None:
                work.wait()
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)

    def zero_grad(self, next_dropout_step=False):
        """next_dropout_step: also advance the dropout seed (ops.next_dropout_step) -- on the GPU in the same launch as the fill"""
        self._check_grad_views(reinstall=True)      # a view dropped by set_to_none is put back; a foreign .grad raises
        # a step that is abandoned (NaN loss -> zero_grad without all_reduce_gradients) may still have the early group's collective
        # in flight on the side stream: zeroing the buffer under it would race
        self._join_early()
        for p in self._staged:
            p._otr_regroup_state['dirty'] = False
        if self.flat_grad.is_cuda:
            from . import ops
            if next_dropout_step and self._store_all.dtype == torch.float32:
                ops.zero_and_next_dropout_step(self._store_all)
            else:
                self._store_all.zero_()
                if next_dropout_step:
                    ops.next_dropout_step(self._store_all.device)
            ops.discard_pending_weight_grads()      # nothing queued survives into a new step (e.g. after an exception)
            ops.gradients_cleared(self._single_writer)
        else:
            self._store_all.zero_()

    def broadcast_parameters(self, src=0):
        """one-time replica sync at start-up (the reference re-broadcasts every step)."""
        if self.world_size > 1:
            if self.flat_param is not None:
                dist.broadcast(self.flat_param, src, group=self.group)
            else:                       # per-parameter storage: broadcast each tensor in place
                for p in self.params:
                    dist.broadcast(p.data, src, group=self.group)
            self.refresh_lp()           # the GEMMs read the 16-bit shadows: they must follow the broadcast masters

    def _rccl_handle(self):
        """lazily build the library-owned RCCL communicator (rank 0's unique id travels through torch.distributed)"""
        if self._rccl is None:
            lib = L.load()
            rank, ws = (dist.get_rank(self.group), self.world_size) if dist.is_available() and dist.is_initialized() else (0, 1)
            uid = C.create_string_buffer(128)
            if rank == 0:
                L.check(lib.otr_allreduce_unique_id(uid), 'otr_allreduce_unique_id')
            if ws > 1:
                box = [uid.raw]
                # `src` of a collective is a GLOBAL rank: group rank 0 of a sub-group need not be global rank 0
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast_object_list(box, src=src, group=self.group)
                uid = C.create_string_buffer(box[0], 128)
            h = C.c_void_p()
            L.check(lib.otr_allreduce_init(C.byref(h), uid, rank, ws), 'otr_allreduce_init')
            self._rccl = h
        return self._rccl

    def close(self):
        if getattr(self, '_single_writer', None):
            from . import ops
            ops.unregister_single_writer_grads(self._single_writer)
            self._single_writer = set()
        if self._rccl is not None:
            L.check(L.load().otr_allreduce_destroy(self._rccl), 'otr_allreduce_destroy')
            self._rccl = None

    def __del__(self):
        try:
            self.close()
        except Exception:                   # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass

    def _reduce_slice(self, lo, hi, stream=None):
        """in-place sum over ranks of _grad_store[lo:hi] (through the 16-bit payload if configured); returns a work handle or None"""
        buf, pay = self._grad_store[lo:hi], None
        if self.grad_comm_dtype is not None:
            if self._payload is None:
                self._payload = torch.empty_like(self._grad_store, dtype=self.grad_comm_dtype)
            pay = self._payload[lo:hi]
            pay.copy_(buf)
            buf = pay
        work = None
        if self.comm == 'rccl':
            code = {torch.float32: L.OTR_F32, torch.bfloat16: L.OTR_BF16, torch.float16: L.OTR_F16}[buf.dtype]
            L.check(L.load().otr_allreduce_run(self._rccl_handle(), C.c_void_p(buf.data_ptr()), buf.numel(), code,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_allreduce_run')
        elif self.world_size > 1:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work, (pay if self.grad_comm_dtype is not None else None)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (the reference's accum_steps, train/trainer.py:206-221): wrap every backward pass of a step EXCEPT
        the last one.  Inside, the gradients accumulate in the flat buffer and the early group's all-reduce does not start; the
        last backward (outside) starts it.  A backward pass that crosses the mark AFTER the early collective of this step has
        started would write into a buffer that is being reduced: that raises instead of racing."""
        was, self._accumulating = self._accumulating, True
        try:
            yield self
        finally:
            self._accumulating = was

    def _on_early_mark(self):
        """ops.EarlyMarkFn's callback: the backward pass crossed the mark"""
        if self._accumulating or self.early_end <= 0 or self.skip_collectives or self.world_size <= 1:
            return
        if self._early_state is not None:
            raise RuntimeError('FlatDataParallel: a backward pass crossed ops.early_mark after the early group\'s all-reduce of this step '
                               'had started (gradient accumulation / a second backward()).  Its gradients would be written into a '
                               'buffer that is being reduced.  Wrap every backward pass of the step except the last in dp.no_sync().')
        self._on_early_ready()

    def _on_early_ready(self, force=False):
        """called once per step, when the early group's gradients are final and their deferred weight-gradient launches are queued on
        the compute stream (from inside the LAST backward pass: _on_early_mark; or by hand: start_early_reduce): start their
        all-reduce on a side stream"""
        if self.early_end <= 0 or self._early_state is not None or not (self.world_size > 1 or force) or self.skip_collectives:
            return
        if torch.cuda.is_available() and self._grad_store.is_cuda and torch.cuda.is_current_stream_capturing():
            return                                   # a captured step keeps the single collective after the graph
        if self._grad_store.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                work, pay = self._reduce_slice(0, self.early_end)
        else:
            work, pay = self._reduce_slice(0, self.early_end)
        self._early_state = (work, pay)

    def start_early_reduce(self, force=False):
        """Staged step (ops.set_stage_split): call between stage 1 (forward + the backward of everything behind ops.early_mark, whose
        deferred weight gradients are flushed when that backward ends) and stage 2 (the rest of the backward pass): the early group's
        all-reduce starts on the side stream, ordered behind whatever the compute stream holds now, and runs beside stage 2;
        all_reduce_gradients() reduces the rest and joins.  Nothing happens at world size 1 (unless forced), with no early group,
        or when it was already started this step."""
        self._on_early_ready(force)

    def backward_staged(self, loss, between=None):
        """loss.backward() cut at the marks of ops.set_stage_split(True); `between` (default: start_early_reduce) runs after stage 1"""
        from . import ops
        ops.backward(loss)
        stages = ops.take_stages()
        (between or self.start_early_reduce)()
        for x, leaf in reversed(stages):
            if leaf.grad is not None:
                x.backward(leaf.grad)
        return stages

    def all_reduce_gradients(self, async_op=False, force=False):
        """single collective over the flat buffer; returns 1/world_size for the optimizer to fold in.
        force: run the collective even at world_size 1 (self-test of the RCCL path on a one-GPU box)."""
        ws = self.world_size
        work = None
        from . import ops as _ops
        _ops.check_no_pending_stages('all_reduce_gradients')
        self._check_grad_views()
        if self.skip_collectives:
            return 1.0 / ws, None
        if ws > 1 or force:
            # The sticky fault word (a bounded in-kernel wait gave up on THIS rank: its gradient sums may be wrong) is local, but
            # the gradient it taints is about to be summed into every replica.  It rides in the cell behind the gradients
            # through the same collective, and comes back as the sum over ranks: every rank's optimizer step then sees a
            # non-zero word and skips together -- the replicas stay identical (a rank skipping alone would diverge for good).
            fault = None
            if self.flat_grad.is_cuda:
                from . import ops
                fault = ops.fault_counter(self.flat_grad.device)
                self._fault_cell.copy_(fault)
            if self._early_state is not None:
                # the early group is already on its way (side stream): reduce the rest [early_end, end) incl. the fault cell here,
                # then join
                work_e, pay_e = self._early_state
                self._early_state = None
                work_l, pay_l = self._reduce_slice(self.early_end, self._grad_store.numel())
                if work_l is not None:
                    work_l.wait()
                if pay_l is not None:
                    self._grad_store[self.early_end:].copy_(pay_l)
                if work_e is not None:
                    work_e.wait()
                if self._side is not None:
                    torch.cuda.current_stream().wait_stream(self._side)
                if pay_e is not None:
                    self._grad_store[:self.early_end].copy_(pay_e)
                if fault is not None:
                    fault.copy_(self._fault_cell)
                return 1.0 / ws, None
            buf = self._grad_store
            if self.grad_comm_dtype is not None:            # 16-bit payload: half the bytes over xGMI
                if self._payload is None:
                    self._payload = torch.empty_like(self._grad_store, dtype=self.grad_comm_dtype)
                self._payload.copy_(self._grad_store)
                buf = self._payload
            if self.comm == 'rccl':
                code = {torch.float32: L.OTR_F32, torch.bfloat16: L.OTR_BF16, torch.float16: L.OTR_F16}[buf.dtype]
                L.check(L.load().otr_allreduce_run(self._rccl_handle(), C.c_void_p(buf.data_ptr()), buf.numel(), code,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'otr_allreduce_run')
            elif ws > 1:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group,
                                       async_op=async_op and buf is self._grad_store and fault is None)
            if buf is not self._grad_store:
                self._grad_store.copy_(buf)
            if fault is not None:
                fault.copy_(self._fault_cell)               # float -> int32: the number of give-ups over all ranks
        return 1.0 / ws, work


class FusedAdam:
    """clip_grad_norm_(clip) + NaN guard + Adam(L2 wd) + Noam lr in three launches over the flat
    buffers (include/otrans_hip.h: otr_optimizer_step).  State tensors are caller-owned."""

    def __init__(self, dp, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, clip_grad=5.0,
                 noam=None, grad_noise=0.0, loss_scale=None, loss_scale_growth=2000):
        """noam: dict(model_size, warmup_steps, factor) of train/scheduler.py:129-138, or None.
        grad_noise: std of the Gaussian gradient noise (train.grad_noise / accum_steps, trainer.py:223-227).
        loss_scale: initial dynamic loss scale; None = 4096 in fp16 mode, off otherwise (bf16 / fp32 need none).  The
        scale lives in the device state block; the model's backward pass is seeded with it (ops.ScaleGradFn)."""
        assert dp.flat_param is not None, 'FusedAdam needs FlatDataParallel(flatten_params=True)'
        self.dp = dp
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_grad
        self.noam = noam
        self.exp_avg = torch.zeros_like(dp.flat_param)
        self.exp_avg_sq = torch.zeros_like(dp.flat_param)
        # include/otrans_hip.h: f32[OTR_OPT_STATE_FLOATS]; [0..15] is the state proper, the rest the norm kernel's scratch
        self._state_store = torch.zeros(L.OTR_OPT_STATE_FLOATS, dtype=torch.float32, device=dp.flat_param.device)
        self.state = self._state_store[:16]
        self.grad_noise = float(grad_noise)
        self.step_offset = 2.0          # Noam step of update t+1 = t + 1 + step_offset (scheduler.py:16-53); load_state_dict may move it
        if dp.flat_param.is_cuda:
            from . import ops
            ops.fault_counter(dp.flat_param.device)     # the update skips when a spin-bounded kernel of the step gave up
            if loss_scale is None:
                loss_scale = 4096.0 if ops.get_compute_dtype() == 'fp16' else 0.0
            if loss_scale:
                self.state[6] = float(loss_scale)
                self.state[9] = float(loss_scale_growth)
                dp.module._otr_loss_scale = self.state[6:7]     # the model seeds its backward pass with this device scalar
            else:
                dp.module._otr_loss_scale = None

    def step(self, grad_scale=1.0):
        if not self.dp.flat_param.is_cuda:
            raise L.OtransHipError('FusedAdam runs on the GPU only')
        n = self.dp.flat_param.numel()
        nm = self.noam or {}
        from . import ops as _ops
        _ops.check_no_pending_stages('FusedAdam.step')
        self._last_grad_scale = float(grad_scale)
        ret = L.load().otr_optimizer_step(
            C.c_void_p(self.dp.flat_param.data_ptr()), C.c_void_p(self.dp.flat_grad.data_ptr()),
            C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()), n,
            C.c_void_p(self.state.data_ptr()), self._state_store.numel(),
            C.c_void_p(self.dp.flat_param_lp.data_ptr()) if self.dp.flat_param_lp is not None else None,
            self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
            grad_scale, self.clip, float(nm.get('model_size', 1.0)), float(nm.get('warmup_steps', 0.0)),
            float(nm.get('factor', 1.0)), self.step_offset,   # scheduler.py:41-53: the first update sees global_step 3
            self.grad_noise,
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.check(ret, 'otr_optimizer_step')
        self.dp.refresh_transposed()

    # ---- checkpoint / resume (train/trainer.py:280-290 save_optimizer_state_dict, run.py:49-62 --init_optim_state) ----------
    # The layout is torch.optim.Adam's own state_dict -- {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups':
    # [{'lr', 'betas', 'eps', 'weight_decay', ..., 'params': [0 .. n-1]}]} with parameter i = the i-th of
    # filter(requires_grad, model.parameters()), the order the reference builds its optimizer in (run.py:42-44) -- so an optimizer
    # checkpoint written by the reference resumes here and the other way round.  What torch's Adam does not have (dynamic loss
    # scale, skip / fault counters) travels under the extra key 'otr', which torch's load_state_dict ignores.
    @property
    def global_step(self):
        """the reference scheduler's global_step after the updates applied so far (scheduler.py:16-53: it starts at 1, the stepwise
        initial_lr() advances it once, every update once more -- the first update computes its lr at 3)"""
        return int(self.state[0].item() + self.step_offset)

    def state_dict(self):
        st = self.state.tolist()
        state = {}
        for i, (p, off) in enumerate(zip(self.dp.params, self.dp.offsets)):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(st[0])),
                        'exp_avg': self.exp_avg[off:off + n].view(p.shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {'lr': float(st[1]) if st[0] > 0 else self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.wd,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(self.dp.params)))}
        return {'state': state, 'param_groups': [group],
                'otr': {'state_block': self.state.detach().cpu().clone(), 'global_step': int(st[0] + self.step_offset)}}

    def load_state_dict(self, sd, global_step=None):
        """accepts state_dict() of this class or of the torch.optim.Adam the reference trains with; moments are copied into the flat
        buffers, Adam's update count t is taken from the per-parameter 'step' entries.
        global_step: the reference checkpoint's separate scheduler counter (train/trainer.py:284 'global_step', run.py:59-60); it
        legitimately differs from t + 2 after --from_step or a mismatched resume.  Given (or found under sd['otr']), the Noam step
        of the next update is global_step + 1 instead of t + 3: the difference is kept in `self.step_offset`."""
        state, groups = sd['state'], sd['param_groups']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(self.dp.params):
            raise ValueError('optimizer checkpoint holds %d parameters, the model has %d' % (len(order), len(self.dp.params)))
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for key, p, off in zip(order, self.dp.params, self.dp.offsets):
            ent = state.get(key)
            if ent is None:                 # torch leaves parameters that never received a gradient without state
                continue
            n = p.numel()
            if tuple(ent['exp_avg'].shape) != tuple(p.shape):
                raise ValueError('optimizer checkpoint: parameter %s has shape %s, the model %s'
                                 % (key, tuple(ent['exp_avg'].shape), tuple(p.shape)))
            self.exp_avg[off:off + n].copy_(ent['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(ent['exp_avg_sq'].reshape(-1))
            steps.add(float(ent['step']))
        if len(steps) > 1:
            raise ValueError('optimizer checkpoint: parameters disagree about the step count (%s); one flat update needs one' % sorted(steps))
        t = steps.pop() if steps else 0.0
        blk = sd.get('otr', {}).get('state_block')
        if blk is not None:                 # loss scale, good-step count, skip / fault counters
            self.state.copy_(blk.to(self.state.device))
        g0 = groups[0]
        self.betas, self.eps, self.wd = tuple(g0.get('betas', self.betas)), g0.get('eps', self.eps), g0.get('weight_decay', self.wd)
        b1, b2 = self.betas
        self.state[0] = t
        self.state[1] = float(g0.get('lr', self.lr))
        if self.noam is None:               # constant-lr runs: the checkpoint's lr IS the lr of the next update
            self.lr = float(g0.get('lr', self.lr))
        if global_step is None:
            global_step = sd.get('otr', {}).get('global_step')
        self.step_offset = 2.0 if global_step is None else float(global_step) - t
        self.state[2] = 1.0 - b1 ** t
        self.state[3] = 1.0 - b2 ** t

    def stats(self):
        s = self.state.tolist()
        # state[4] is the squared norm of the gradients as they sat in memory (loss-scaled, summed over ranks); state[8] is
        # the grad_scale / loss_scale of THAT update -- not the current state[6], which the same update may have halved or
        # doubled.  A skipped update (NaN unscale after a fault, non-finite norm) reports the raw value over the current scale.
        us, gs = s[8], getattr(self, '_last_grad_scale', 1.0)
        ls = gs / us if (us == us and 0 < us < float('inf')) else (s[6] if s[6] > 0 else 1.0)
        return {'step': s[0], 'lr': s[1], 'grad_sqnorm': s[4] / (ls * ls), 'skipped': s[5], 'loss_scale': s[6], 'faults': s[10]}
