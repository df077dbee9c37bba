"""torch.autograd wrappers over the C ABI (include/otrans_hip.h).

PyTorch is used for device memory, streams and autograd bookkeeping only: every forward/backward
below hands raw data_ptr()s and torch's *current stream* to libotrans_hip.so.  There is no CPU path:
a non-CUDA tensor raises.

Precision policy (set_compute_dtype):
  'bf16' / 'fp16': MFMA inputs in that 16-bit type, fp32 accumulate; the residual stream, LayerNorm/softmax
          statistics, losses and every parameter/gradient stay fp32; q/k/v, attention context, conv
          activations and the FFN hidden are stored 16-bit.  The two modes run the same kernels built for the
          other 16-bit type (libotrans_hip.so / libotrans_hip_f16.so).  fp16 has 3 more mantissa bits -- the
          full-size model's logits land 5e-4 from the fp32 reference instead of 3.9e-3 in bf16
          (tools/precision_study.py; the north-star bar is 1e-3) -- and needs the loss scaling below.
  'fp32': exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) everywhere, all activations fp32 -- parity mode.
"""
import contextlib
import ctypes as C
import gc
import math

import torch

from . import _lib as L

_state = {'compute': 'bf16', 'rng_offset': 0, 'seed': None}


@contextlib.contextmanager
def graph_capture(graph, **kw):
    """`torch.cuda.graph(graph)` with Python's cyclic garbage collector parked.  A collection that fires while a stream
    is capturing can finalize ANOTHER CUDAGraph (or tensors of its private pool) that sat in a dead reference cycle; the
    hipGraphExecDestroy / hipFree inside that finalizer is illegal during capture and aborts the process (seen in a
    sequential pytest run; torch >= 2.10 no longer collects on entry).  So: collect first, keep the collector off until
    the capture has ended."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    _capture['serial'] += 1
    _capture['active'] = _capture['serial']
    try:
        with torch.cuda.graph(graph, **kw):
            yield graph
    finally:
        _capture['active'] = None
        if was_enabled:
            gc.enable()


# Which recording are we in?  Host-side facts about device memory ("this gradient buffer was just cleared") are only true for the
# launches issued in the SAME context: eagerly, or inside the same capture -- a captured launch is replayed later, any number of
# times, whatever the host knew when it was recorded.  ('eager',) outside a capture, ('capture', n) inside the n-th graph_capture of
# this process, None inside a capture somebody else started (torch.cuda.graph used directly): nothing can be assumed there.
_capture = {'serial': 0, 'active': None}


def capture_context():
    if _capture['active'] is not None:
        return ('capture', _capture['active'])
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return None
    return ('eager',)


# ---------------------------------------------------------------------------------------- kernel timing hook (bench.py)
# bench.py's `roofline` times the dominant kernels INSIDE a real training step: with a timer installed the wrappers of
# the hot launches bracket them with events on the launch stream (torch's current stream is the stream handed to the
# library).  Off (None) in production: one dict lookup per launch.
def set_kernel_timer(records):
    """records: a list that receives (name, meta, start_event, end_event) tuples, or None to switch timing off"""
    _state['ktimer'] = records


def _timed(name, meta, call):
    rec = _state.get('ktimer')
    if rec is None:
        return call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = call()
    e1.record()
    rec.append((name, meta, e0, e1, call))
    return r


def set_compute_dtype(name):
    assert name in ('bf16', 'fp16', 'fp32')
    _state['compute'] = name
    L.select('fp16' if name == 'fp16' else 'bf16')      # fp32 mode lives in both builds; keep the bf16 one
    _register_fault_counter()                           # each build keeps its own pointer to the one device word


def fault_counter(device):
    """The sticky device fault word (include/otrans_hip.h: otr_set_fault_counter): int32[1], one per process, registered with
    the library build in use.  Kernels whose inter-workgroup waits are bounded (the turnstile of the 256-wide weight-gradient
    launch) add 1 when a wait gives up; FusedAdam's update reads and clears it and skips the update -- a wrong-but-finite
    gradient never reaches the parameters."""
    f = _state.get('fault')
    if f is None or f.device != device:
        f = torch.zeros(1, dtype=torch.int32, device=device)
        _state['fault'] = f
        _register_fault_counter()
    return f


def _register_fault_counter():
    f = _state.get('fault')
    if f is not None:
        L.check(L.load().otr_set_fault_counter(C.c_void_p(f.data_ptr())), 'otr_set_fault_counter')


def is_half():
    return _state['compute'] != 'fp32'


def half_dtype():
    """the 16-bit tensor dtype of the current mode (GEMM-operand form of activations and weights)"""
    return torch.float16 if _state['compute'] == 'fp16' else torch.bfloat16


def get_compute_dtype():
    return _state['compute']


def act_dtype():
    return half_dtype() if is_half() else torch.float32


def _compute_code():
    return {'bf16': L.OTR_BF16, 'fp16': L.OTR_F16, 'fp32': L.OTR_F32}[_state['compute']]


def _code(dt):
    if dt == torch.float32:
        return L.OTR_F32
    if dt == torch.bfloat16:
        return L.OTR_BF16
    if dt == torch.float16:
        return L.OTR_F16
    raise TypeError('opentransformer_amd: unsupported dtype %s' % dt)


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.OtransHipError('opentransformer_amd ops need CUDA/HIP tensors (got a %s tensor); '
                                   'there is no CPU fallback' % t.device)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, offset_elems=0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())


def rng_seed_tensor(device):
    """Device-resident dropout seed (uint64 stored in an int64 tensor); bump it once per step with
    `next_dropout_step()` so fwd and bwd of one step regenerate identical masks (graph-capture safe)."""
    s = _state['seed']
    device = torch.device(device)
    if s is None or s.device.type != device.type or (device.index is not None and s.device.index != device.index):
        s = torch.full((1,), 0x5EED, dtype=torch.int64, device=device)
        _state['seed'] = s
    return s


_DROPOUT_STEP_INC = 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF


def next_dropout_step(device):
    rng_seed_tensor(device).add_(_DROPOUT_STEP_INC)
    _state['rng_offset'] = 0


def zero_and_next_dropout_step(buf):
    """the start of a training step in ONE launch (otr_zero_tick): clear the flat gradient buffer `buf` (fp32, CUDA) and advance the
    dropout seed like next_dropout_step -- a fill launch + an 8-byte add launch otherwise"""
    _cuda(buf)
    assert buf.dtype == torch.float32 and buf.is_contiguous()
    seed = rng_seed_tensor(buf.device)
    L.check(L.load().otr_zero_tick(_p(buf), buf.numel(), _p(seed), _DROPOUT_STEP_INC, _stream()), 'otr_zero_tick')
    _state['rng_offset'] = 0


def _next_rng_offset(n):
    off = _state['rng_offset']
    _state['rng_offset'] = off + n
    return off


# ---------------------------------------------------------------------------------------- loss scaling (fp16)
# fp16 activation gradients underflow without it (d loss / d logit is ~1e-7 for most of a 4234-way softmax).  The scale
# is a DEVICE scalar owned by the optimizer (slot 6 of otr_optimizer_step's state block): the model multiplies the
# gradient that enters its backward pass by it (ScaleGradFn, one otr_scale launch), every gradient in the flat buffer
# comes out that much too large, and the optimizer divides it out, halves it on overflow and grows it back -- all on the
# device, so a captured hipGraph stays valid.  No scale registered (reference-style use, bf16 / fp32 modes) = identity.
def set_loss_scale_tensor(t):
    """t: 1-element fp32 device tensor (a view of the optimizer state) or None."""
    _state['loss_scale'] = t


def loss_scale_tensor():
    return _state.get('loss_scale')


class ScaleGradFn(torch.autograd.Function):
    """identity forward; backward multiplies the incoming gradient by the registered device-side loss scale"""

    @staticmethod
    def forward(ctx, loss, scale):
        ctx.save_for_backward(scale)
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty_like(g)
        L.check(L.load().otr_scale(_p(g), _p(out), g.numel(), _p(scale), 1.0, _stream()), 'otr_scale')
        return out, None


def loss_scale_of(owner=None):
    """the device scalar the backward pass of `owner`'s loss is seeded with, or None (see scale_loss_grad)"""
    s = getattr(owner, '_otr_loss_scale', None) if owner is not None else None
    return s if s is not None else _state.get('loss_scale')


def scale_loss_grad(loss, owner=None):
    """owner: the model whose optimizer registered a scale on it (FusedAdam sets `_otr_loss_scale` on the wrapped module, so
    two models trained in one process keep separate scales); the process-wide tensor is the fallback (parity tests)."""
    s = getattr(owner, '_otr_loss_scale', None) if owner is not None else None
    if s is None:
        s = _state.get('loss_scale')
    if s is None or not loss.requires_grad:
        return loss
    return ScaleGradFn.apply(loss, s)


def backward(loss):
    """loss.backward() for a scalar loss, seeded with a cached device-side 1.0 instead of the ones_like(loss) the autograd engine
    would fill on every call (one launch per step; the cached scalar is allocated by the first eager call, so a capture only
    ever reads it)."""
    seed = None
    if loss.dim() == 0 and loss.is_cuda:
        key = ('unit_grad', loss.device, loss.dtype)
        seed = _state.get(key)
        if seed is None and not torch.cuda.is_current_stream_capturing():
            seed = _state[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    torch.autograd.backward(loss, grad_tensors=seed)


# ---------------------------------------------------------------------------------------- bf16 shadows
# In bf16 mode every GEMM operand is read as bf16 from memory:
#  * activations of the fp32 residual stream carry a bf16 twin produced by the kernel that wrote them
#    (add+LayerNorm, pos-enc, embedding); it rides along as a Python attribute of the fp32 tensor;
#  * weights have bf16 shadows: a slice of FlatDataParallel's flat bf16 buffer that the fused optimizer
#    refreshes in the same pass as the fp32 master, or (stand-alone modules) a cached cast keyed by
#    the parameter's version counter.
_LP_ATTR = '_otr_bf16'
_PAD_ROWS = True      # A/B switch of ops.padded_rows


def lp_of(t):
    if not is_half() or t is None:
        return None
    return getattr(t, _LP_ATTR, None)


def attach_lp(t, lp):
    if lp is not None:
        setattr(t, _LP_ATTR, lp)
    return t


def cast_bf16(src, dst=None):
    src = src.contiguous()
    if dst is None:
        dst = torch.empty(src.shape, dtype=half_dtype(), device=src.device)
    L.check(L.load().otr_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), 'otr_cast_f32_to_bf16')
    return dst


def weight_lp(w):
    """bf16 shadow of an fp32 weight (None in fp32 mode)."""
    if not is_half() or w.dtype != torch.float32:
        return None
    view = getattr(w, '_otr_lp_view', None)
    if view is not None:
        return view
    cache = getattr(w, '_otr_lp_cache', None)
    if cache is not None and cache[0] == w._version and cache[1] == w.data_ptr():
        return cache[2]
    lp = cast_bf16(w.detach())
    w._otr_lp_cache = (w._version, w.data_ptr(), lp)
    return lp


# ---------------------------------------------------------------------------------------- in-place gradients
def grad_target(p):
    """FlatDataParallel pre-installs every parameter's .grad as a view of ONE flat buffer (zeroed once per
    step).  Backward kernels then accumulate straight into that view and return None to autograd: no
    temporary dw tensors, no AccumulateGrad add kernel per parameter, no zero-fills."""
    if p is not None and getattr(p, '_otr_grad_inplace', False) and p.grad is not None:
        return p.grad
    return None


# ---------------------------------------------------------------------------------------- split-K workspace
_WS_BYTES = 64 << 20


def _workspace(device):
    """Caller-owned split-K workspace (include/otrans_hip.h): one fp32 buffer per device, shared by all
    GEMM calls (they are ordered on the launch stream)."""
    ws = _state.get('ws')
    if ws is None or ws.device != device:
        ws = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
        _state['ws'] = ws
        fault_counter(device)           # registered before the first launch that could report through it (and before any capture)
        _zero_placeholder(device, ())   # likewise allocated outside any capture's private pool (LnOutLink)
        _ffn_sync_pool(device)          # and the split FFN kernels' arrival counters (one row per launch stream)
    lane = _state.get('ws_lane')
    return lane if lane is not None and lane.device == device else ws


def new_workspace(device):
    """the second split-K workspace of this device, for launches that run on ANOTHER stream concurrently with the default lane
    (workspace_lane).  One per device, shared by everything that forks ONE side stream at a time (recognize.CachedBeamState: the
    states of different batch shapes never run concurrently)."""
    _workspace(device)
    side = _state.get('ws_side')
    if side is None or side.device != device:
        side = _state['ws_side'] = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
    return side


@contextlib.contextmanager
def workspace_lane(ws):
    """GEMM launches issued inside take `ws` (from new_workspace) as their split-K workspace: the shared one is only safe for
    launches ordered on one stream (recognize.CachedBeamState runs the LM branch of a beam step on a side stream)."""
    prev = _state.get('ws_lane')
    _state['ws_lane'] = ws
    try:
        yield
    finally:
        _state['ws_lane'] = prev


def padded_rows(w, b=None):
    """FlatDataParallel's row-padded images of a Linear whose output width is not a multiple of 8 (dp.py `slot_numel`): dicts
    {'param', 'grad', 'lp', 'lpt'} for the weight and {'param', 'grad'} for the bias, or None when this weight has none (or the
    bias lacks its own).  The GEMMs of such a layer run on [rows8, K] operands: aligned rows, branch-free loaders."""
    if not is_half() or not _PAD_ROWS:
        return None
    pw = getattr(w, '_otr_pad', None)
    if pw is None or 'lp' not in pw or 'lpt' not in pw or pw['lp'].dtype != half_dtype():
        return None
    pb = None
    if b is not None:
        pb = getattr(b, '_otr_pad', None)
        if pb is None or pb['param'].shape[0] != pw['param'].shape[0]:
            return None
    return pw, pb


# Gradient buffers whose columns behind the logical width are known to be zero (written by the loss kernel that made them) carry
# that fact ON the buffer: `_otr_zero_tail = (rows, width8)` on the padded [rows, width8] tensor whose head the gradient is a view
# of.  LinearFn.backward widens exactly these -- found through `dy._base` -- to their padded width.  (Not a process-wide map keyed
# by data_ptr: a recycled allocation with the same address would have been read at the padded width with an unverified tail, and
# two models trained in one process cleared each other's entries: ADVICE r04.)
def _zero_tail_of(t):
    base = t._base if t._base is not None else t
    tail = getattr(base, '_otr_zero_tail', None)
    if tail is None or base.data_ptr() != t.data_ptr() or tuple(base.shape) != tuple(tail):
        return None
    return tail


def regrouped_lp(w, shape):
    """the 16-bit shadow of w with its two inner axes swapped ([A, R, S] kept as [A, S, R] = `shape`), where FlatDataParallel
    registered one (it refreshes it after every optimizer step, dp.refresh_transposed); None otherwise"""
    v = getattr(w, '_otr_regroup_view', None) if is_half() else None
    if v is None or tuple(v.shape) != tuple(shape) or v.dtype != half_dtype():
        return None
    return v


def weight_lpt(w):
    """TRANSPOSED bf16 shadow [K, N] of an fp32 weight [N, K] (None in fp32 mode): with it the input
    gradient dx = dy . w becomes a forward-type GEMM (both operands k-contiguous)."""
    if not is_half() or w.dtype != torch.float32 or w.dim() != 2:
        return None
    view = getattr(w, '_otr_lpt_view', None)
    if view is not None:
        return view
    if getattr(w, '_otr_grad_inplace', False):
        # a replica parameter without a registered transposed shadow (FlatDataParallel keeps none for weights that have packs):
        # FusedAdam rewrites it through raw pointers, so a version-keyed cache would go stale after the first update
        return None
    cache = getattr(w, '_otr_lpt_cache', None)
    if cache is not None and cache[0] == w._version and cache[1] == w.data_ptr():
        return cache[2]
    lpt = weight_lp(w).t().contiguous()
    w._otr_lpt_cache = (w._version, w.data_ptr(), lpt)
    return lpt


# ---------------------------------------------------------------------------------------- linear
def _linear_desc(M, N, K, xdt, wdt, ydt, ldx, ldw, ldy, act=L.ACT_NONE, accumulate=0):
    return L.LinearDesc(M, N, K, _code(xdt), _code(wdt), _code(ydt), _compute_code(), ldx, ldw, ldy, act, accumulate)


class ResidualLink:
    """Pairs the first Linear of a residual branch with the add+LayerNorm that closes it: y = LN(x + f(x)).
    x receives two gradients, one through the skip connection (written by the LayerNorm backward) and one through
    f's first Linear; autograd would add them with an extra elementwise kernel per sub-layer.  With a link the
    LayerNorm backward hands its dx over instead of returning it, and the Linear's input-gradient GEMM accumulates
    into that buffer in its epilogue and returns the sum.  (The Linear's backward always runs after the LayerNorm's:
    it depends on it through f.)"""
    __slots__ = ('armed', 'buf')

    def __init__(self):
        self.armed = False
        self.buf = None


def new_link():
    return ResidualLink() if torch.is_grad_enabled() else None


class PreNormLink:
    """The mirror image for a pre-norm residual y = x + f(LN(x)) (encoder/conformer.py:50-73): the residual add's backward
    runs first and parks dy (the gradient through the skip connection) here instead of returning it for x; the LayerNorm's
    backward -- which always runs later, it is reached through f -- adds it to its input gradient in the same kernel
    (otr_add_layernorm_bwd_skip).  armed: set by the LayerNorm when its input wants a gradient."""
    __slots__ = ('armed', 'buf')

    def __init__(self):
        self.armed = False
        self.buf = None


def new_prenorm_link():
    return PreNormLink() if torch.is_grad_enabled() else None


class LnOutLink:
    """Pairs the LayerNorm that closes a post-norm FFN sub-layer, y = LN(z), with the first Linear that reads y (the q|k|v
    projection of the next layer).  That Linear's input-gradient launch produces dy row by row, complete, so the LayerNorm backward
    runs in its epilogue (otr_rb_linear_ln_bwd) instead of a launch of its own that would read dy back: the Linear's backward
    leaves (d z, d a 16-bit, partial affine / bias sums) in `result` and returns a zero PLACEHOLDER for y (a stride-0 view of one
    device scalar); the FFN sub-layer's backward recognises the placeholder and takes `result`.  Any other gradient that reaches
    y is added to the placeholder by autograd, arrives as a real tensor and goes through the ordinary LayerNorm backward on top
    (the operation is linear in dy).  saved = what the LayerNorm backward needs, params = the parameters whose gradients it sums."""
    __slots__ = ('armed', 'saved', 'params', 'result', 'prefetch')

    def __init__(self):
        self.armed, self.saved, self.params, self.result = False, None, None, None
        self.prefetch = None       # (tensor, bytes): what the launch AFTER the linked Linear's backward streams first (otr_rb_linear_ln_bwd_pf)


_LNOUT = True


class LnInLink:
    """Pairs the attention sub-layer's closing launch, y1 = LN(x + proj(c)) (ProjLnFn), with the split FFN that reads y1 in SLAB mode:
    the FFN's backward launch leaves the four hidden slices' shares of d y1 as 16-bit slabs instead of summing them, so FfnLnFn.backward
    leaves (skip-path gradient f32, slabs) in `result` and returns the zero placeholder; ProjLnFn.backward sums them while it loads
    its rows (otr_ln_bwd_proj_slabs).  A gradient from any other consumer of y1 arrives as a real tensor and is added on top."""
    __slots__ = ('armed', 'result', 'z')

    def __init__(self):
        self.armed, self.result, self.z = False, None, None


class PendingLn:
    """A LayerNorm output nobody has computed yet: the split FFN in slab mode returns y / y16 / z / mean / rstd as ALLOCATED buffers
    plus this record; the first Linear that reads y (the next layer's q|k|v projection) finishes the LayerNorm in its prologue and
    fills them (otr_rb_linear_ln), anything else calls materialize() first (otr_dec_ln: the LayerNorm alone).  Only callers that
    control who reads y next ask for this (nn.TransformerEncoder)."""
    __slots__ = ('kw', 'M', 'done')

    def __init__(self):
        self.kw, self.M, self.done = None, 0, True

    def desc(self):
        return _dec_ln(**self.kw)

    def finish(self):
        self.done, self.kw = True, None      # the slabs go back to the allocator


def materialize(x):
    """x with its pending LayerNorm (if any) computed"""
    pend = getattr(x, '_otr_pending', None)
    if pend is not None and not pend.done:
        M = pend.M
        d = pend.desc()
        L.check(_timed('dec_ln', {'bytes': M * 256 * (4 + 8 + 4 + 2 + 4)}, lambda: L.load().otr_dec_ln(C.byref(d), M, _stream())), 'otr_dec_ln')
        pend.finish()
    return x


def _zero_placeholder(device, shape):
    z = _state.setdefault('zero_scalar', {}).get(device)
    if z is None:
        z = _state['zero_scalar'][device] = torch.zeros((), dtype=torch.float32, device=device)
    return z.expand(shape)


def _is_zero_placeholder(t):
    z = _state.get('zero_scalar', {}).get(t.device)
    return z is not None and t.data_ptr() == z.data_ptr() and all(st == 0 for st in t.stride())


def linear_fwd_raw(x2, w, b, out_dtype, act=L.ACT_NONE, out=None):
    """y = act(x w^T + b); with `out` the product is ACCUMULATED into that [M,N] buffer."""
    M, K = x2.shape
    N = w.shape[0]
    y = out if out is not None else torch.empty((M, N), dtype=out_dtype, device=x2.device)
    d = _linear_desc(M, N, K, x2.dtype, w.dtype, y.dtype, x2.stride(0), w.stride(0), y.stride(0), act,
                     accumulate=int(out is not None))
    ws = _workspace(x2.device)
    L.check(_timed('linear_fwd %dx%dx%d' % (M, N, K), {'flops': 2.0 * M * N * K},
                   lambda: L.load().otr_linear_fwd(C.byref(d), _p(x2), _p(w), _p(b), _p(y), _p(ws), _WS_BYTES, _stream())),
            'otr_linear_fwd')
    return y


def linear_dgrad_raw(dy2, w, dx_dtype, out=None):
    M, N = dy2.shape
    K = w.shape[1]
    dx = out if out is not None else torch.empty((M, K), dtype=dx_dtype, device=dy2.device)
    d = _linear_desc(M, N, K, dx.dtype, w.dtype, dy2.dtype, dx.stride(0), w.stride(0), dy2.stride(0),
                     accumulate=int(out is not None))
    ws = _workspace(dy2.device)
    L.check(L.load().otr_linear_dgrad(C.byref(d), _p(dy2), _p(w), _p(dx), _p(ws), _WS_BYTES, _stream()), 'otr_linear_dgrad')
    return dx


# ---------------------------------------------------------------------------------------- deferred weight gradients
# With in-place gradient buffers (FlatDataParallel) the weight / bias gradients of a backward pass do not have to
# be launched where autograd reaches them: nothing downstream reads them before the optimizer.  They are queued
# and, when the autograd engine finishes, run as a few GROUPED launches (otr_linear_wgrad_grouped /
# otr_colsum_grouped): ~150 latency-bound GEMM + split-K-reduce + column-sum launches per step become ~6.
# The queue belongs to ONE backward pass: it is filled and flushed inside a single autograd-engine run (the flush is an
# engine callback of that run), so two models trained in one process -- an ASR model and an LM, say -- never see each
# other's items as long as their backward passes do not interleave on one thread; a pass that dies half way leaves its
# items behind, which the owner's zero_grad() / the next pass's first enqueue (different graph-task id) discards.
_wq = {'on': False, 'w': [], 'b': [], 'post': [], 'armed': False, 'task': None}    # post: callables run behind the grouped launches
# Store instead of accumulate (otr_wgrad_item_t.overwrite): a replica engine registers the gradient buffers that have exactly ONE
# writer per backward pass -- the deferred weight-gradient product of a Linear (not: an embedding that is also an output layer, whose
# scatter-add lands in the same buffer) -- and tells when it has cleared them.  The first grouped launch that writes such a buffer
# after a clear may store its sums; every write is recorded, so a second backward pass before the next clear (gradient accumulation)
# accumulates as before.  Addresses, not tensors: a gradient view and its row-padded image are the same buffer.
# The store is safe by construction, not by convention: it is taken only when the clear and the launch were issued in the SAME
# context (capture_context(): both eagerly, or both inside one captured graph).  A launch captured WITHOUT its clear always
# accumulates -- replaying a captured forward + backward twice between two eager zero_grad() calls (graph-based gradient accumulation)
# therefore adds up, like the biases and LayerNorm gradients do.  Anything else that adds into a registered buffer between the clear and
# the first grouped launch (an autograd AccumulateGrad of a torch-native use of the weight, a hook, user code) must say so with
# gradients_written(ptrs); FlatDataParallel registers only buffers whose gradient never passes through AccumulateGrad.
_wq_excl = {'single_writer': set(), 'written': set(), 'cleared_in': {}}
_WG_OVERWRITE = True        # tests flip it to compare the store with accumulation on zeros


def register_single_writer_grads(ptrs):
    _wq_excl['single_writer'].update(ptrs)


def unregister_single_writer_grads(ptrs):
    _wq_excl['single_writer'].difference_update(ptrs)
    _wq_excl['written'].difference_update(ptrs)
    for q in ptrs:
        _wq_excl['cleared_in'].pop(q, None)


def gradients_cleared(ptrs):
    """the buffers at these addresses were just zeroed by their owner (FlatDataParallel.zero_grad), in the current context"""
    _wq_excl['written'].difference_update(ptrs)
    ctx = capture_context()
    _wq_excl['cleared_in'].update((q, ctx) for q in ptrs)


def gradients_written(ptrs):
    """somebody other than the grouped weight-gradient launch added into the buffers at these addresses: the next grouped launch
    accumulates instead of storing"""
    _wq_excl['written'].update(ptrs)

_FUSE_BIAS_COLSUM = True
_DEBUG_WQ = False


def defer_weight_grads(on):
    _wq['on'] = bool(on)


def discard_pending_weight_grads():
    """Drop queued (never launched) weight-gradient work, e.g. left behind by a backward pass that raised.  Called by
    FlatDataParallel.zero_grad(): a stale queue must not leak into the next step's gradients."""
    _wq['w'], _wq['b'], _wq['post'], _wq['armed'], _wq['task'] = [], [], [], False, None


def _arm_flush():
    task = torch._C._current_graph_task_id()
    if _wq['task'] is not None and _wq['task'] != task:      # leftovers of a backward pass that never finished
        discard_pending_weight_grads()
    _wq['task'] = task
    if not _wq['armed']:
        _wq['armed'] = True
        torch.autograd.Variable._execution_engine.queue_callback(flush_weight_grads)


def flush_weight_grads():
    """Launch everything queued by linear_wgrad_raw / colsum_raw (called by the autograd engine at the end of
    backward; safe to call by hand)."""
    _wq['armed'], _wq['task'] = False, None
    _wq['dp_pool'] = None                # (the queued items and post hooks keep their slices alive)
    w, b, post = _wq['w'], _wq['b'], _wq['post']
    _wq['w'], _wq['b'], _wq['post'] = [], [], []
    # a gradient buffer that appears twice (one Linear applied twice in the forward pass): the problems of ONE grouped launch run
    # side by side and would read-modify-write the same tiles unordered -- the repeats go to a flush of their own behind this one
    if w:
        first, again, seen_ptr = [], [], set()
        for item in w:
            (again if item[2].data_ptr() in seen_ptr else first).append(item)
            seen_ptr.add(item[2].data_ptr())
        if again:
            w = first
            _wq['w'] = again
            post = post + [flush_weight_grads]
    if _wq.get('keep_last'):            # bench.py re-times the grouped launch on the items of the last backward
        _wq['last'] = (list(w), list(b))
    lib = L.load()
    if _DEBUG_WQ and (w or b) and _wq.get('dumped', 0) < 4:   # listing of the first flushes' deferred work (tuning aid)
        _wq['dumped'] = _wq.get('dumped', 0) + 1
        print('wq flush', _wq['dumped'], flush=True)
        for dy2, x2, out in w:
            print('wq w', tuple(dy2.shape), dy2.dtype, dy2.stride(0), tuple(x2.shape), x2.dtype, flush=True)
        for a2, out in b:
            print('wq b', tuple(a2.shape), a2.dtype, a2.stride(0), flush=True)
    if w:
        items = (L.WgradItem * len(w))()
        seen = {}
        for _, _, out in w:
            seen[out.data_ptr()] = seen.get(out.data_ptr(), 0) + 1
        sw, written, cleared_in, here = _wq_excl['single_writer'], _wq_excl['written'], _wq_excl['cleared_in'], capture_context()
        for it, (dy2, x2, out) in zip(items, w):
            it.dy, it.x, it.dw = dy2.data_ptr(), x2.data_ptr(), out.data_ptr()
            it.M, it.N, it.K = dy2.shape[0], dy2.shape[1], x2.shape[1]
            it.ldy, it.ldx, it.ldw = dy2.stride(0), x2.stride(0), out.stride(0)
            it.dy_dtype, it.x_dtype = _code(dy2.dtype), _code(x2.dtype)
            it.dbias = None
            ptr = out.data_ptr()
            it.overwrite = int(_WG_OVERWRITE and ptr in sw and ptr not in written and seen[ptr] == 1
                               and here is not None and cleared_in.get(ptr) == here)
        written.update(seen)
        if b and _FUSE_BIAS_COLSUM:
            # a bias gradient whose matrix is the dy operand of a weight gradient the 256-wide kernel takes rides along with
            # it (the kernel reads that matrix anyway): one less pass over it by the column-sum launch
            by_dy = {}
            for i, (dy2, x2, out) in enumerate(w):
                by_dy.setdefault((dy2.data_ptr(), dy2.shape[0], dy2.shape[1], dy2.stride(0)), i)
            rest = []
            for a2, out in b:
                i = by_dy.get((a2.data_ptr(), a2.shape[0], a2.shape[1], a2.stride(0)))
                if (i is not None and not items[i].dbias and a2.dtype == w[i][0].dtype and out.dtype == torch.float32
                        and out.is_contiguous() and lib.otr_wgrad256_takes(C.byref(items[i]), _compute_code())):
                    items[i].dbias = out.data_ptr()
                else:
                    rest.append((a2, out))
            b = rest
        ws = _workspace(w[0][0].device)
        meta = None
        if _state.get('ktimer') is not None:
            meta = {'problems': len(w),
                    'flops': sum(2.0 * wi[0].shape[0] * wi[0].shape[1] * wi[1].shape[1] for wi in w),
                    'bytes': sum(wi[0].numel() * wi[0].element_size() + wi[1].numel() * wi[1].element_size() + 2 * wi[2].numel() * 4 for wi in w)}
        L.check(_timed('linear_wgrad_grouped', meta, lambda: lib.otr_linear_wgrad_grouped(
            items, len(w), _compute_code(), _p(ws), _WS_BYTES, _stream())), 'otr_linear_wgrad_grouped')
    if b:
        items = (L.ColsumItem * len(b))()
        for it, (a2, out) in zip(items, b):
            it.a, it.out = a2.data_ptr(), out.data_ptr()
            it.M, it.N, it.lda, it.dtype = a2.shape[0], a2.shape[1], a2.stride(0), _code(a2.dtype)
        L.check(lib.otr_colsum_grouped(items, len(b), _stream()), 'otr_colsum_grouped')
    for fn in post:                      # work that reads what the grouped launches just wrote (LinearFn: a regrouped weight gradient)
        fn()
    if post and (_wq['w'] or _wq['b']):  # ... and may queue weight gradients of its own (RelPosAttentionFn: dW_pos from the finished dp)
        keep, _wq['keep_last'] = _wq.get('keep_last'), False      # bench.py re-times the MAIN grouped launch, not this tail
        try:
            flush_weight_grads()
        finally:
            _wq['keep_last'] = keep


# ---------------------------------------------------------------------------------------- early gradient groups
# The decoder's (and the CTC head's) backward runs BEFORE the encoder's: their gradients are final long before the pass ends, and a
# data-parallel engine can start reducing them while the encoder backward runs (dp.FlatDataParallel(early_modules=...)).  The
# model marks the point on the encoder output: every autograd node created after the mark (decoder, embedding, heads) has a higher
# sequence number than the mark, so the engine runs their backward first; the mark's backward then (i) launches the weight / bias
# gradients queued so far -- all of them belong to modules behind the mark -- and (ii) tells the engine.
def set_early_callback(fn, modules=None):
    """fn: a bound method (kept by weak reference) or None.  `modules`: the modules of the engine that registers -- the callback is
    stored ON them (`_otr_early_cb`), and a mark placed with `early_mark(x, owner)` resolves it through its owner, so two engines
    in one process (an ASR model and an LM) each get their own marks.  The process-wide slot written as well is only the fallback
    of marks placed WITHOUT an owner: it belongs to whichever engine registered last."""
    import weakref
    ref = weakref.WeakMethod(fn) if fn is not None else None
    _state['early_cb'] = ref
    for m in (modules or []):
        object.__setattr__(m, '_otr_early_cb', ref)


def _early_cb_of(owner):
    if owner is not None:                    # a model without an engine of its own never borrows another model's
        return getattr(owner, '_otr_early_cb', None)
    return _state.get('early_cb')


class EarlyMarkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ref):
        ctx.cb_ref = ref                     # the engine of the model that placed THIS mark (resolved at forward time)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ref = ctx.cb_ref
        cb = ref() if ref is not None else None          # a weak reference: a dead engine leaves nothing behind
        if cb is not None:
            if _wq['w'] or _wq['b']:
                flush_weight_grads()         # re-arms itself at the next enqueue (the encoder's items)
            cb()
        return g, None


def set_stage_split(on):
    """Staged backward (dp.py, bench.py): with the switch on, ops.early_mark CUTS the autograd graph -- it returns a detached leaf
    that shares x's storage and records (x, leaf).  `loss.backward()` then stops at the leaf (stage 1: everything behind the mark);
    `for x, leaf in reversed(ops.take_stages()): x.backward(leaf.grad)` runs the rest (stage 2).  Each stage can be captured as its
    own hipGraph, and between two replays the host can start the all-reduce of the gradients that are already final (the
    collective itself is never captured)."""
    _state['stage_split'] = bool(on)
    _state['stages'] = []


def take_stages():
    st = _state.get('stages') or []
    _state['stages'] = []
    return st


def check_no_pending_stages(who):
    """With the graph cut on, `loss.backward()` alone leaves everything in front of the cut WITHOUT gradients.  The consumers of the
    gradients (dp.all_reduce_gradients, FusedAdam.step) call this: cuts recorded by a forward pass that nobody took (ops.take_stages /
    dp.backward_staged) mean stage 2 never ran."""
    st = _state.get('stages') or []
    if st:
        _state['stages'] = []
        raise RuntimeError('%s: ops.set_stage_split(True) is on and %d graph cut(s) of the last forward pass were never taken -- '
                           'loss.backward() stopped at ops.early_mark and the rest of the backward pass did not run; use '
                           'dp.backward_staged(loss) (or ops.take_stages() + x.backward(leaf.grad)), or switch the split off' % (who, len(st)))


def early_mark(x, owner=None):
    """identity; see EarlyMarkFn (keeps the 16-bit twin of x).  In stage-split mode: a graph cut (set_stage_split).
    owner: the module that places the mark (the model); its engine's callback fires in backward, nobody else's."""
    if _state.get('stage_split') and x.requires_grad and torch.is_grad_enabled():
        x = materialize(x)
        leaf = x.detach().requires_grad_(True)
        lp = getattr(x, _LP_ATTR, None)
        if lp is not None:
            setattr(leaf, _LP_ATTR, lp)
        _state['stages'].append((x, leaf))
        return leaf
    ref = _early_cb_of(owner)
    if ref is None or not x.requires_grad or not torch.is_grad_enabled():
        return x
    y = EarlyMarkFn.apply(x, ref)
    lp = getattr(x, _LP_ATTR, None)
    if lp is not None:
        setattr(y, _LP_ATTR, lp)
    return y


# Links (ResidualLink, PreNormLink, LnOutLink) hand a gradient from one autograd node to a later one OUTSIDE autograd's own
# bookkeeping.  If the receiving node never runs in that pass (its branch detached or frozen after the forward pass), the gradient
# would be dropped without a trace: every hand-over is registered, and the engine's end-of-pass callback raises if one is left.
_parked = []
_parked_task = [None]      # graph-task id of the backward pass whose end-of-pass check is queued


def _drop_handovers(links):
    for k in links:
        if hasattr(k, 'buf'):
            k.buf = None
        if hasattr(k, 'result'):
            k.result = None


def _park(link):
    if not _in_backward():
        return
    task = torch._C._current_graph_task_id()
    if _parked_task[0] != task:
        # first hand-over of a NEW backward pass.  Whatever is still listed belongs to a pass that raised before its callback
        # ran (the engine drops queued callbacks then): clear those links -- their gradients are stale, and they pin tensors --
        # and queue the check for THIS pass (keyed by the pass, not by list emptiness, so a dead pass cannot switch it off)
        _drop_handovers(_parked)
        del _parked[:]
        _parked_task[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_check_parked)
    _parked.append(link)


def _check_parked():
    _parked_task[0] = None
    left = [k for k in _parked if getattr(k, 'buf', None) is not None or getattr(k, 'result', None) is not None]
    del _parked[:]
    _drop_handovers(left)
    if left:
        raise RuntimeError('%d gradient hand-over(s) between linked autograd nodes (%s) were never picked up: the receiving node did '
                           'not run in this backward pass; results of the pass are incomplete'
                           % (len(left), ', '.join(sorted({type(k).__name__ for k in left}))))


def _in_backward():
    """True while the autograd engine is executing a backward pass on this thread (queue_callback needs it)."""
    try:
        return torch._C._current_graph_task_id() != -1
    except AttributeError:          # very old torch: no way to tell, never defer
        return False


def linear_wgrad_raw(dy2, x2, w_like, out=None):
    """dw = dy^T x; with `out` (an fp32 [N,K] gradient buffer) the result is ACCUMULATED into it."""
    M, N = dy2.shape
    K = x2.shape[1]
    if out is not None and _wq['on'] and out.stride(1) == 1 and dy2.stride(1) == 1 and x2.stride(1) == 1 and _in_backward():
        _wq['w'].append((dy2, x2, out))
        _arm_flush()
        return out
    dw = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=dy2.device)
    if out is not None:
        _wq_excl['written'].add(out.data_ptr())          # written outside the grouped launches: whoever comes next accumulates
    d = _linear_desc(M, N, K, x2.dtype, torch.float32, dy2.dtype, x2.stride(0), K, dy2.stride(0),
                     accumulate=int(out is not None))
    ws = _workspace(dy2.device)
    L.check(L.load().otr_linear_wgrad(C.byref(d), _p(dy2), _p(x2), _p(dw), _p(ws), _WS_BYTES, _stream()), 'otr_linear_wgrad')
    return dw


def colsum_raw(a2, out=None, defer=True):
    """out[N] (+)= column sums of a2[M, N].  With `out` (a gradient buffer that outlives the backward pass) the sum may be
    queued for the grouped launch at the end of backward; defer=False forces it now (out is a temporary autograd returns)."""
    M, N = a2.shape
    acc = out is not None
    if acc and defer and _wq['on'] and a2.stride(1) == 1 and a2.data_ptr() % 16 == 0 and _in_backward():
        _wq['b'].append((a2, out))
        _arm_flush()
        return out
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=a2.device)
    L.check(L.load().otr_colsum(_p(a2), _code(a2.dtype), M, N, a2.stride(0), _p(out), int(acc), _stream()), 'otr_colsum')
    return out


def _rows(x):
    """View [..., K] as [M, K] with unit inner stride (copy only if the caller handed us a strange view)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    return x2


# ---------------------------------------------------------------------------------------- row-block projections
# The skinny projections of the attention sub-layers (q|k|v, q, output_proj; N, K in {256, 768}) run as row-block kernels
# (csrc/rowblock.hip) on fragment-major packs of the weight: `fwd` = pack(W as A[n][k]) and `dgrad` = pack(W as A[k][n]),
# slices of FlatDataParallel's pack buffer (refreshed with the FFN packs after every optimizer step) or, for stand-alone
# modules, a cache keyed by the parameter version.
_RB = True
_RB_SHAPES = ((256, 256), (768, 256))          # (N, K) of the Linear
_RB_LINEAR_MIN_ROWS = 1024   # below: 16 workgroups each streaming the whole weight lose to the 64-wide tile GEMM


def lin_pack_items(w_off, N, K, dst_off):
    """otr_pack_frags table rows for one Linear weight [N, K]: forward pack, input-gradient pack; elements used"""
    return [[w_off, K, 1, N, K, 0, dst_off], [w_off, 1, K, K, N, 0, dst_off + N * K]], 2 * N * K


def lin_packs(w):
    """(fwd_pack, dgrad_pack) of a Linear weight, or None when the row-block kernels do not apply"""
    if not _RB or _state['compute'] == 'fp32' or w is None or w.dim() != 2 or not w.is_cuda or tuple(w.shape) not in _RB_SHAPES:
        return None
    views = getattr(w, '_otr_lin_packs', None)
    if views is not None:
        return views
    if getattr(w, '_otr_grad_inplace', False):
        # a parameter of a FlatDataParallel replica without registered packs: FusedAdam rewrites it through raw pointers
        # (neither _version nor data_ptr changes), so a version-keyed cache would go stale after the first update
        return None
    key = (w._version, w.data_ptr(), _state['compute'])
    cache = getattr(w, '_otr_lin_pack_cache', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    N, K = w.shape
    src = weight_lp(w).reshape(-1)
    rows, total = lin_pack_items(0, N, K, 0)
    dst = torch.empty(total, dtype=src.dtype, device=src.device)
    pack_frags(src, dst, rows)
    packs = (dst[:N * K], dst[N * K:])
    w._otr_lin_pack_cache = (key, packs)
    return packs


def _rb_rows_ok(t2):
    """[M, C] 16-bit rows the row-block kernels can read: unit column stride, 16-byte aligned rows"""
    return (t2 is not None and t2.dtype == half_dtype() and t2.dim() == 2 and t2.stride(1) == 1 and t2.stride(0) % 8 == 0
            and t2.data_ptr() % 16 == 0)


def rb_linear_raw(x2, pack, N, bias, out_dtype, skip=None):
    """out[M,N] = x2 . W^T (+ bias) (+ skip, which may be the output buffer itself) through otr_rb_linear"""
    M, K = x2.shape
    out = skip if (skip is not None and skip.dtype == out_dtype) else torch.empty((M, N), dtype=out_dtype, device=x2.device)
    L.check(_timed('rb_linear %dx%dx%d' % (M, N, K), {'flops': 2.0 * M * N * K},
                   lambda: L.load().otr_rb_linear(_p(x2), x2.stride(0), _p(pack), _p(bias), _p(skip),
                                                  skip.stride(0) if skip is not None else 0, _p(out), _code(out_dtype), out.stride(0),
                                                  M, N, K, _stream())), 'otr_rb_linear')
    return out


def rb_linear_pending_raw(pend, pack, N, bias, out_dtype):
    """out[M,N] = LN(...) . W^T + bias with the pending LayerNorm finished in the launch's prologue (otr_rb_linear_ln)"""
    M = pend.M
    out = torch.empty((M, N), dtype=out_dtype, device=bias.device if bias is not None else pack.device)
    d = pend.desc()
    L.check(_timed('rb_linear_ln %dx%dx256' % (M, N), {'flops': 2.0 * M * N * 256},
                   lambda: L.load().otr_rb_linear_ln(C.byref(d), _p(pack), _p(bias), _p(out), _code(out_dtype), out.stride(0), M, N, 256,
                                                     _stream())), 'otr_rb_linear_ln')
    pend.finish()
    return out


def rb_linear_ln_bwd_raw(g2, wt_pack, skip, saved, prefetch=None):
    """(d z f32, d a 16-bit, partial [blocks, 3*256]) of the LayerNorm y = LN(z) whose output gradient is skip + g2 . W"""
    z, mean, rstd, gamma, seed, p_drop, off = saved
    pf_t, pf_n, pf_h = prefetch if prefetch is not None else (None, 0, None)
    if pf_h is not None:
        L.check(L.load().otr_touch_hint(_p(pf_h), pf_h.numel() * pf_h.element_size(), None, 0), 'otr_touch_hint')
    M, K = g2.shape
    d = 256
    lib = L.load()
    dx = torch.empty((M, d), dtype=torch.float32, device=g2.device)
    da = torch.empty((M, d), dtype=g2.dtype, device=g2.device)
    part = torch.empty((lib.otr_ln_bwd_proj_partial_rows(M), 3 * d), dtype=torch.float32, device=g2.device)
    L.check(_timed('rb_linear_ln_bwd %dx%dx%d' % (M, d, K), {'flops': 2.0 * M * d * K},
                   lambda: lib.otr_rb_linear_ln_bwd_pf(_p(g2), g2.stride(0), _p(wt_pack), _p(skip), skip.stride(0) if skip is not None else 0,
                                                       _p(z), _p(mean), _p(rstd), _p(gamma), _p(seed), p_drop, off, _p(dx), _p(da), _p(part),
                                                       M, d, K, _p(pf_t), pf_n, _stream())), 'otr_rb_linear_ln_bwd')
    return dx, da, part


class LinearFn(torch.autograd.Function):
    """y = act(x w^T + b): nn.Linear of the reference (e.g. module/attention.py:43,68).

    perm = (C, F): the GEMM uses the weight with its columns regrouped from c*F+f to f*C+c (the
    channel-last flatten of the conv frontend, frontend/conv.py:145); dw is regrouped back."""

    @staticmethod
    def forward(ctx, x, w, b, relu, out_dtype, perm, defer_bias=False, link=None, g16=None):
        _cuda(x, w, b)
        ctx.defer_bias = defer_bias      # the consumer (AddLayerNormFn, a_bias=b) produces the bias gradient
        ctx.link = link
        if link is not None:             # see ResidualLink
            link.armed = bool(ctx.needs_input_grad[0]) and x.dtype == torch.float32 and perm is None and not relu
        xc = lp_of(x)
        x2 = _rows(xc if xc is not None else x)
        wl = weight_lp(w)
        wc = wl if wl is not None else w
        if perm is not None:
            C_, F_ = perm
            rg = regrouped_lp(w, (w.shape[0], F_, C_))
            wc = rg.view(w.shape[0], F_ * C_) if rg is not None else wc.view(-1, C_, F_).permute(0, 2, 1).reshape(w.shape[0], F_ * C_).contiguous()
        packs = lin_packs(w) if (perm is None and not relu and _rb_rows_ok(x2) and x2.shape[0] >= _RB_LINEAR_MIN_ROWS
                                 and (b is None or b.data_ptr() % 16 == 0)) else None
        ctx.rb = packs
        pend = getattr(x, '_otr_pending', None)
        if pend is not None and pend.done:
            pend = None
        if pend is not None and not (packs is not None and tuple(w.shape) == (768, 256) and out_dtype in (torch.float32, half_dtype())):
            materialize(x)
            pend = None
        lo = getattr(x, '_otr_lnout', None)      # x is the output of a LayerNorm whose backward can run in this Linear's dgrad launch
        ctx.lnout = lo if (lo is not None and lo.armed and packs is not None and ctx.needs_input_grad[0] and x.dtype == torch.float32
                           and w.shape[1] == 256 and w.shape[0] in (256, 768)) else None
        ctx.pad = None
        pad = padded_rows(w, b) if (packs is None and pend is None and perm is None and not relu and out_dtype == torch.float32) else None
        if pend is not None:
            y = rb_linear_pending_raw(pend, packs[0], w.shape[0], b, out_dtype)
        elif packs is not None:
            y = rb_linear_raw(x2, packs[0], w.shape[0], b, out_dtype)
        elif pad is not None:
            # output width not a multiple of 8: the product is taken against the row-padded weight into a [M, rows8] buffer (the
            # tail columns come out 0) and the result is its [M, N] head -- rows start 16-byte aligned, whoever reads them
            pw, pb = pad
            y = linear_fwd_raw(x2, pw['lp'], pb['param'] if pb is not None else None, out_dtype)[:, :w.shape[0]]
            ctx.pad = pad
        else:
            y = linear_fwd_raw(x2, wc, b, out_dtype, L.ACT_RELU if relu else L.ACT_NONE)
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.perm = perm
        # 16-bit output-gradient hand-over (_Grad16Link): the Linear with regrouped columns (the frontend's output layer), fp32 output,
        # in-place gradient buffers with a kernel-order staging image of this weight's gradient (dp.FlatDataParallel), 16-bit operands
        ctx.g16 = None
        if (g16 is not None and x2.dtype == half_dtype() and (ctx.needs_input_grad[1] or ctx.needs_input_grad[0])
                and (perm is not None or ctx.pad is not None)):
            ctx.g16 = g16
            g16.armed = True
        ctx.wt = weight_lpt(w) if (perm is None and ctx.needs_input_grad[0]) else None
        ctx.w_ref, ctx.b_ref = w, b
        ctx.save_for_backward(x2, wc, y if relu else None)
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wc, y = ctx.saved_tensors
        g16 = ctx.g16
        dyp16 = None
        if g16 is not None and g16.buf is not None and _is_zero_placeholder(dy):
            if ctx.pad is not None:
                dyp16, g16.buf = g16.buf, None       # the loss launch's [M, rows8] 16-bit gradient, zero behind column N
            else:
                dy, g16.buf = g16.buf, None          # the gradient as the 16-bit operand PosEncFn.backward wrote
        dy2 = _rows(dy) if dyp16 is None else dyp16
        if ctx.relu:
            dy2 = relu_bwd_raw(y, dy2.contiguous())
        if (_DY16_WIDE and dyp16 is None and dy2.dtype == torch.float32 and is_half() and ctx.perm is not None and x2.dtype == half_dtype()
                and dy2.shape[0] >= 1024 and _wq['on'] and _in_backward() and getattr(ctx.w_ref, '_otr_regroup_grad', None) is not None
                and grad_target(ctx.w_ref) is not None):
            # r06: an fp32 gradient in front of the frontend's output layer (the Conformer: a LayerNorm sits behind it; C2 gets the 16-bit
            # operand from PosEncFn) is cast ONCE: the weight gradient then joins the grouped 256-wide launch through the staging image
            # instead of an fp32-operand GEMM + split-K reduce + a regrouping add (112 us), the input gradient reads 16-bit rows
            dy2 = cast_bf16(dy2)
        if ctx.pad is not None:
            # the gradient of a row-padded product: when it arrives as the head of a zero-tailed [M, rows8] buffer (the loss
            # kernel wrote it that way), all three GEMMs of this layer run on the padded operands
            pw, pb = ctx.pad
            N8 = pw['param'].shape[0]
            tail = _zero_tail_of(dy2) if dyp16 is None else None
            if dyp16 is not None and tuple(dyp16.shape) != (x2.shape[0], N8):
                raise RuntimeError('LinearFn.backward: the parked 16-bit gradient has shape %s, expected %s' % (tuple(dyp16.shape), (x2.shape[0], N8)))
            if dyp16 is not None or (tail == (dy2.shape[0], N8) and dy2.dim() == 2 and dy2.stride() == (N8, 1) and dy2.dtype == torch.float32
                                     and dy2.data_ptr() % 16 == 0):
                dyp = dyp16 if dyp16 is not None else dy2.as_strided((dy2.shape[0], N8), (N8, 1))
                dx = None
                if ctx.needs_input_grad[0]:
                    skip = None
                    if ctx.link is not None and ctx.link.buf is not None:
                        skip, ctx.link.buf = ctx.link.buf, None
                    dx = linear_fwd_raw(dyp, pw['lpt'], None, ctx.xdtype, out=skip).view(ctx.xshape)
                if ctx.needs_input_grad[1]:
                    linear_wgrad_raw(dyp, x2, pw['lp'], out=pw['grad'])
                if ctx.has_bias and ctx.needs_input_grad[2] and not ctx.defer_bias:
                    colsum_raw(dyp, out=pb['grad'])
                return dx, None, None, None, None, None, None, None, None
        dx = None
        if ctx.needs_input_grad[0]:
            skip = None
            if ctx.link is not None and ctx.link.buf is not None:      # skip-connection gradient handed over by the LN
                skip, ctx.link.buf = ctx.link.buf, None
            rb_ok = ctx.rb is not None and _rb_rows_ok(dy2) and (skip is None or (skip.dtype == ctx.xdtype and skip.stride(0) % 4 == 0))
            lo = ctx.lnout
            if (rb_ok and lo is not None and lo.result is None and dy2.dtype == half_dtype() and _wq['on'] and _in_backward()
                    and all(grad_target(q) is not None for q in lo.params)):
                lo.result = rb_linear_ln_bwd_raw(dy2, ctx.rb[1], skip, lo.saved, lo.prefetch)
                _park(lo)
                dx = _zero_placeholder(dy2.device, ctx.xshape)
            elif rb_ok:
                dx = rb_linear_raw(dy2, ctx.rb[1], wc.shape[1], None, ctx.xdtype, skip=skip).view(ctx.xshape)
            elif ctx.wt is not None:      # dx = dy . w as a forward-type GEMM on the transposed shadow
                dx = linear_fwd_raw(dy2, ctx.wt, None, ctx.xdtype, out=skip).view(ctx.xshape)
            else:
                dx = linear_dgrad_raw(dy2, wc, ctx.xdtype, out=skip).view(ctx.xshape)
        dw = None
        if ctx.needs_input_grad[1]:
            gt = grad_target(ctx.w_ref) if ctx.perm is None else None
            if gt is not None:
                linear_wgrad_raw(dy2, x2, wc, out=gt)
            elif (ctx.perm is not None and dy2.dtype == half_dtype() and x2.dtype == half_dtype() and _wq['on'] and _in_backward()
                  and getattr(ctx.w_ref, '_otr_regroup_grad', None) is not None and grad_target(ctx.w_ref) is not None):
                # 16-bit operands: the product joins the grouped 256-wide launch, accumulating into the KERNEL-order staging image of
                # this weight's gradient (zeroed with the gradient buffer); behind that launch one strided add regroups it into the
                # parameter's layout (flush_weight_grads runs the hook)
                C_, F_ = ctx.perm
                stage = ctx.w_ref._otr_regroup_grad.view(wc.shape[0], F_ * C_)
                ctx.w_ref._otr_regroup_state['dirty'] = True        # (tests read it: "the hand-over really ran")
                linear_wgrad_raw(dy2, x2, wc, out=stage)
                gt_w = grad_target(ctx.w_ref)
                # the regrouping launch leaves the image ZERO: every pass starts from zeros whoever issues it (a second eager pass, a
                # replay of a graph captured without the clear), with no host-side state baked into anything
                _wq['post'].append(lambda gt_w=gt_w, stage=stage, C_=C_, F_=F_: regroup_add(gt_w, stage, stage.shape[0], C_, F_, True))
            else:
                dw = linear_wgrad_raw(dy2, x2, wc)
                if ctx.perm is not None:
                    C_, F_ = ctx.perm
                    dw = dw.view(-1, F_, C_).permute(0, 2, 1)
                    gt = grad_target(ctx.w_ref)
                    if gt is not None:       # regrouped back while it is added to the gradient buffer: one launch, not copy + add
                        gt.view(-1, C_, F_).add_(dw)
                        dw = None
                    else:
                        dw = dw.reshape(dw.shape[0], C_ * F_)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2] and not ctx.defer_bias:
            gt = grad_target(ctx.b_ref)
            if gt is not None:
                colsum_raw(dy2, out=gt)
            else:
                db = colsum_raw(dy2)
        return dx, dw, db, None, None, None, None, None, None


def regroup_add(dst, src, rows, C_, F_, clear):
    """dst[r, c, f] += src[r, f, c] in one launch (otr_regroup_add); clear: src is left zero"""
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous()
    assert dst.numel() == src.numel() == rows * C_ * F_
    L.check(L.load().otr_regroup_add(_p(dst), _p(src), rows, C_, F_, int(bool(clear)), _stream()), 'otr_regroup_add')


def linear(x, w, b=None, relu=False, out_dtype=None, perm=None, defer_bias=False, link=None):
    """defer_bias=True: the caller hands `b` to add_layernorm(..., a_bias=b), whose backward reduces the bias
    gradient in the same pass that produces the branch gradient."""
    out_dtype = out_dtype if out_dtype is not None else torch.float32
    g16 = None
    if _G16 and not relu and out_dtype == torch.float32 and is_half() and _wq['on'] and torch.is_grad_enabled():
        if perm is not None and getattr(w, '_otr_regroup_grad', None) is not None:
            g16 = _Grad16Link()                       # the frontend's output layer <- PosEncFn.backward
        elif perm is None and _G16_LOSS and getattr(w, '_otr_pad', None) is not None:
            g16 = _Grad16Link()                       # the row-padded output layer <- LabelSmoothingLossFusedFn.backward
    y = LinearFn.apply(x, w, b, relu, out_dtype, perm, defer_bias, link, g16)
    if g16 is not None and g16.armed:
        y._otr_g16 = g16                 # read by PosEncFn.forward when y is its input
    return y


def relu_bwd_raw(y, g):
    out = torch.empty_like(g)
    L.check(L.load().otr_relu_bwd(_p(y), _p(g), _p(out), _code(g.dtype), g.numel(), _stream()), 'otr_relu_bwd')
    return out


def relu_bwd_colsum_raw(y2, g2):
    """(g2 * (y2 > 0), per-workgroup column sums of it [nblk, cols] f32) in one pass; None when the shape is not served"""
    rows, cols = y2.shape
    lib = L.load()
    nblk = lib.otr_relu_bwd_colsum_partial_rows(rows, cols, _code(g2.dtype))
    if nblk <= 0 or not (y2.is_contiguous() and g2.is_contiguous()) or (y2.data_ptr() | g2.data_ptr()) % 16:
        return None
    out = torch.empty_like(g2)
    part = torch.empty((nblk, cols), dtype=torch.float32, device=g2.device)
    L.check(lib.otr_relu_bwd_colsum(_p(y2), _p(g2), _p(out), _p(part), _code(g2.dtype), rows, cols, _stream()), 'otr_relu_bwd_colsum')
    return out, part


ACT_KINDS = {'gelu': 1, 'tanh': 2, 'swish': 3}       # otr_act_fwd / otr_act_bwd


class ActivationFn(torch.autograd.Function):
    """gelu / tanh / swish of module/ffn.py:15-21 on the pre-activation (relu lives in the GEMM epilogue, glu has its own path)."""

    @staticmethod
    def forward(ctx, x, kind):
        _cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.load().otr_act_fwd(_p(x), _p(y), _code(x.dtype), x.numel(), ACT_KINDS[kind], _stream()), 'otr_act_fwd')
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous().to(x.dtype)
        dx = torch.empty_like(x)
        L.check(L.load().otr_act_bwd(_p(x), _p(dy), _p(dx), _code(x.dtype), x.numel(), ACT_KINDS[ctx.kind], _stream()),
                'otr_act_bwd')
        return dx, None


def activation(x, kind):
    return ActivationFn.apply(x, kind)


# ---------------------------------------------------------------------------------------- attention
def _attn_desc(B, H, Tq, Tk, dk, dt, qs, ks, vs, os_, causal):
    return L.AttnDesc(B, H, Tq, Tk, dk, _code(dt), qs[0], qs[1], ks[0], ks[1], vs[0], vs[1], os_[0], os_[1],
                      int(causal), 1.0 / math.sqrt(dk))


def _mask_u8(mask, B, Tk):
    """[B,Tk] bool key mask -> uint8 (None if there is no mask).  The cast is remembered ON the mask tensor (with its version
    counter): the encoder's key mask and the decoder's memory mask are one tensor (model/speech2text.py:50-54), a strided
    view of the batch's frame mask, and each cast of it is a launch."""
    if mask is None:
        return None
    if mask.dtype == torch.uint8 and mask.dim() == 2 and mask.is_contiguous() and tuple(mask.shape) == (B, Tk):
        return mask
    hit = getattr(mask, '_otr_u8', None)
    if hit is not None and hit[0] == mask._version and tuple(hit[1].shape) == (B, Tk):
        return hit[1]
    u8 = mask.reshape(B, Tk).to(torch.uint8).contiguous()
    mask._otr_u8 = (mask._version, u8)
    return u8


class SelfAttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dk) + mask) v on a packed [B,T,3d] projection (columns q|k|v, the
    split order of module/attention.py:73); returns the merged-head context [B,T,d]."""

    @staticmethod
    def forward(ctx, qkv, key_mask_u8, n_heads, causal):
        _cuda(qkv)
        B, T, d3 = qkv.shape
        d = d3 // 3
        dk = d // n_heads
        qkv = qkv.contiguous()
        out = torch.empty((B, T, d), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, n_heads, T), dtype=torch.float32, device=qkv.device)
        s3 = (T * d3, d3)
        desc = _attn_desc(B, n_heads, T, T, dk, qkv.dtype, s3, s3, s3, (T * d, d), causal)
        L.check(_timed('self_attention_fwd', {'flops': 4.0 * B * T * T * d * (0.5 if causal else 1.0), 'bytes': B * T * d * 2 * 4},
                       lambda: L.load().otr_attention_fwd(C.byref(desc), _p(qkv), _p(qkv, d), _p(qkv, 2 * d), _p(key_mask_u8),
                                                          _p(out), _p(lse), _stream())), 'otr_attention_fwd')
        ctx.save_for_backward(qkv, out, lse, key_mask_u8)
        ctx.cfg = (n_heads, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, km = ctx.saved_tensors
        n_heads, causal = ctx.cfg
        B, T, d3 = qkv.shape
        d = d3 // 3
        dk = d // n_heads
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        s3 = (T * d3, d3)
        desc = _attn_desc(B, n_heads, T, T, dk, qkv.dtype, s3, s3, s3, (T * d, d), causal)
        # flops: the five products of the backward pass (S, dP, dQ, dK, dV); both orientations of the kernel form S and dP (7 products run)
        L.check(_timed('self_attention_bwd', {'flops': 10.0 * B * T * T * d * (0.5 if causal else 1.0), 'bytes': B * T * d * 2 * 8},
                       lambda: L.load().otr_attention_bwd(C.byref(desc), _p(qkv), _p(qkv, d), _p(qkv, 2 * d), _p(km), _p(out),
                                                          _p(dout), _p(lse), _p(delta), _p(dqkv), _p(dqkv, d), _p(dqkv, 2 * d),
                                                          _stream())), 'otr_attention_bwd')
        return dqkv, None, None, None


class CrossAttentionFn(torch.autograd.Function):
    """q [B,L,d] against packed kv [B,T,2d] (columns k|v: module/attention.py:134)."""

    @staticmethod
    def forward(ctx, q, kv, key_mask_u8, n_heads):
        _cuda(q, kv)
        B, Lq, d = q.shape
        T = kv.shape[1]
        dk = d // n_heads
        q = q.contiguous()
        kv = kv.contiguous()
        out = torch.empty((B, Lq, d), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, n_heads, Lq), dtype=torch.float32, device=q.device)
        sq, skv = (Lq * d, d), (T * 2 * d, 2 * d)
        desc = _attn_desc(B, n_heads, Lq, T, dk, q.dtype, sq, skv, skv, sq, False)
        L.check(L.load().otr_attention_fwd(C.byref(desc), _p(q), _p(kv), _p(kv, d), _p(key_mask_u8), _p(out), _p(lse),
                                           _stream()), 'otr_attention_fwd')
        ctx.save_for_backward(q, kv, out, lse, key_mask_u8)
        ctx.n_heads = n_heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse, km = ctx.saved_tensors
        n_heads = ctx.n_heads
        B, Lq, d = q.shape
        T = kv.shape[1]
        dk = d // n_heads
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        delta = torch.empty_like(lse)
        sq, skv = (Lq * d, d), (T * 2 * d, 2 * d)
        desc = _attn_desc(B, n_heads, Lq, T, dk, q.dtype, sq, skv, skv, sq, False)
        L.check(L.load().otr_attention_bwd(C.byref(desc), _p(q), _p(kv), _p(kv, d), _p(km), _p(out), _p(dout), _p(lse),
                                           _p(delta), _p(dq), _p(dkv), _p(dkv, d), _stream()), 'otr_attention_bwd')
        return dq, dkv, None, None


# ---------------------------------------------------------------------------------------- batched cross-attention K/V
# The decoder's cross-attention projects the SAME encoder memory to keys / values in every layer (module/attention.py:
# 128-134 `vk_proj`, six times per step at [B*T', 256] x [512, 256]^T).  One 512-wide GEMM on 7968 rows runs at 65 TFLOP/s
# (32 us); all layers at once -- [B*T', 256] x [L*512, 256]^T -- is one well-filled GEMM, and the backward pass needs ONE
# input-gradient GEMM (K = L*512) instead of L plus L-1 adds of [B*T', 256] fp32 tensors.
class CrossKVShared:
    """bookkeeping shared by the L cross-attention slices of one decoder pass: the gradient buffer [B,T,L*2d] every slice's
    backward writes its columns into, and how many have done so"""
    __slots__ = ('n', 'dkv', 'done')

    def __init__(self, n):
        self.n, self.dkv, self.done = n, None, 0


def _same_storage(ts):
    """the tensors are windows of ONE allocation (a flat buffer): a view may then span them"""
    s0 = ts[0].untyped_storage()
    return all(t.untyped_storage().data_ptr() == s0.data_ptr() for t in ts)


def stack_rows(ts):
    """torch.cat(ts, 0) -- as a VIEW when the tensors already sit one after the other in one allocation (FlatDataParallel lays the
    decoder layers' vk_proj parameters out that way): no launch"""
    t0 = ts[0]
    if all(t.is_contiguous() and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] for t in ts):
        nb, ok, p = t0.element_size(), True, t0.data_ptr()
        for t in ts:
            ok = ok and t.data_ptr() == p
            p += t.numel() * nb
        if ok and (len(ts) == 1 or _same_storage(ts)):
            rows = sum(t.shape[0] for t in ts)
            return t0.detach().as_strided((rows,) + tuple(t0.shape[1:]), t0.stride())
    return torch.cat(ts, dim=0)


def stack_cols(ts):
    """torch.cat(ts, 1) of 2-D tensors -- as a view when they are adjacent column slices of one row-major matrix"""
    t0 = ts[0]
    if all(t.dim() == 2 and t.stride() == t0.stride() and t.stride(1) == 1 and t.shape[0] == t0.shape[0] and t.dtype == t0.dtype for t in ts):
        nb, ok, p, cols = t0.element_size(), True, t0.data_ptr(), sum(t.shape[1] for t in ts)
        for t in ts:
            ok = ok and t.data_ptr() == p
            p += t.shape[1] * nb
        if ok and cols <= t0.stride(0) and _same_storage(ts):
            return t0.detach().as_strided((t0.shape[0], cols), t0.stride())
    return torch.cat(ts, dim=1)


class CrossKVAllFn(torch.autograd.Function):
    """kv_all[B,T,L*2d] = memory . cat_l(vk_proj_l.weight)^T + cat_l(bias): columns [l*2d, l*2d+d) are layer l's keys, the
    next d its values (split order k, v: module/attention.py:134)."""

    @staticmethod
    def forward(ctx, memory, shared, *wb):
        ws, bs = wb[0::2], wb[1::2]
        _cuda(memory, *ws)
        ctx.shared, ctx.refs = shared, (ws, bs)
        mc = lp_of(memory)
        m2 = _rows(mc if mc is not None else memory)
        adt = act_dtype()
        wl = [weight_lp(w) if weight_lp(w) is not None else w for w in ws]
        wcat = stack_rows(wl)                                         # [L*2d, dm]
        bcat = stack_rows([b for b in bs])
        y = linear_fwd_raw(m2, wcat, bcat, adt)
        wt = [weight_lpt(w) for w in ws]
        ctx.wcat_t = stack_cols(wt) if all(t is not None for t in wt) else None      # [dm, L*2d]
        ctx.save_for_backward(m2, wcat)
        ctx.mshape, ctx.mdtype = memory.shape, memory.dtype
        return y.view(*memory.shape[:-1], wcat.shape[0])

    @staticmethod
    def backward(ctx, dkv):
        m2, wcat = ctx.saved_tensors
        ws, bs = ctx.refs
        d2 = _rows(dkv)
        if ctx.wcat_t is not None:
            dmem = linear_fwd_raw(d2, ctx.wcat_t, None, ctx.mdtype)
        else:
            dmem = linear_dgrad_raw(d2, wcat, ctx.mdtype)
        grads = []
        n2 = ws[0].shape[0]
        for i, (w, b) in enumerate(zip(ws, bs)):
            sl = d2[:, i * n2:(i + 1) * n2]
            gw, gb = grad_target(w), grad_target(b)
            dw = linear_wgrad_raw(sl, m2, None, out=gw)
            db = colsum_raw(sl, out=gb)
            grads += [None if gw is not None else dw, None if gb is not None else db]
        return (dmem.view(ctx.mshape), None, *grads)


class CrossAttentionSliceFn(torch.autograd.Function):
    """CrossAttentionFn on slice `idx` of a CrossKVAllFn output: keys / values are read in place by stride, and the backward
    pass writes d k | d v into the shared [B,T,L*2d] gradient buffer; the last slice to finish hands that buffer to autograd
    (the others return None), so no per-layer gradient tensors are allocated or added."""

    @staticmethod
    def forward(ctx, q, kv_all, key_mask_u8, n_heads, idx, shared):
        _cuda(q, kv_all)
        B, Lq, d = q.shape
        T, W = kv_all.shape[1], kv_all.shape[2]
        dk = d // n_heads
        q = q.contiguous()
        assert kv_all.is_contiguous() and W == shared.n * 2 * d
        out = torch.empty((B, Lq, d), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, n_heads, Lq), dtype=torch.float32, device=q.device)
        sq, skv = (Lq * d, d), (T * W, W)
        desc = _attn_desc(B, n_heads, Lq, T, dk, q.dtype, sq, skv, skv, sq, False)
        L.check(L.load().otr_attention_fwd(C.byref(desc), _p(q), _p(kv_all, idx * 2 * d), _p(kv_all, idx * 2 * d + d),
                                           _p(key_mask_u8), _p(out), _p(lse), _stream()), 'otr_attention_fwd')
        ctx.save_for_backward(q, kv_all, out, lse, key_mask_u8)
        ctx.cfg = (n_heads, idx, shared)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv_all, out, lse, km = ctx.saved_tensors
        n_heads, idx, shared = ctx.cfg
        B, Lq, d = q.shape
        T, W = kv_all.shape[1], kv_all.shape[2]
        dk = d // n_heads
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        if shared.dkv is None:
            shared.dkv = torch.empty_like(kv_all)
        dkv = shared.dkv
        delta = torch.empty_like(lse)
        sq, skv = (Lq * d, d), (T * W, W)
        desc = _attn_desc(B, n_heads, Lq, T, dk, q.dtype, sq, skv, skv, sq, False)
        L.check(L.load().otr_attention_bwd(C.byref(desc), _p(q), _p(kv_all, idx * 2 * d), _p(kv_all, idx * 2 * d + d), _p(km),
                                           _p(out), _p(dout), _p(lse), _p(delta), _p(dq), _p(dkv, idx * 2 * d),
                                           _p(dkv, idx * 2 * d + d), _stream()), 'otr_attention_bwd')
        shared.done += 1
        if shared.done == shared.n:                    # every slice is written: release the whole gradient to the projection
            shared.dkv, shared.done = None, 0
            return dq, dkv, None, None, None, None
        return dq, None, None, None, None, None


# ---------------------------------------------------------------------------------------- add + LayerNorm
class AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(a)) (post-norm residual: encoder/transformer.py:54-56,61-63)."""

    @staticmethod
    def forward(ctx, x, a, gamma, beta, p_drop, eps, a_bias=None, link=None):
        _cuda(x, a, gamma, beta)
        ctx.link = link
        if isinstance(link, PreNormLink):
            link.armed = bool(ctx.needs_input_grad[0]) and a is None and x.dtype == torch.float32
        ctx.set_materialize_grads(False)      # no zero-filled bf16 'gradient' for the non-differentiable twin
        ctx.ab_ref = a_bias
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        a2 = a.reshape(-1, d).contiguous() if a is not None else None
        M = x2.shape[0]
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(x2)
        ylp = torch.empty(x2.shape, dtype=half_dtype(), device=x.device) if is_half() else None
        z = torch.empty_like(x2) if need_grad else None
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        seed = rng_seed_tensor(x.device) if p_drop > 0 else None
        off = _next_rng_offset(M * d) if p_drop > 0 else 0
        desc = L.LnDesc(M, d, _code(a2.dtype) if a2 is not None else L.OTR_F32, eps, p_drop, off)
        L.check(L.load().otr_add_layernorm_fwd(C.byref(desc), _p(x2), _p(a2), _p(gamma), _p(beta), _p(seed), _p(y),
                                               _p(ylp), _p(z), _p(mean), _p(rstd), _stream()), 'otr_add_layernorm_fwd')
        ctx.save_for_backward(z, mean, rstd, gamma, seed)
        ctx.g_ref, ctx.b_ref = gamma, beta
        ctx.cfg = (M, d, a2.dtype if a2 is not None else None, eps, p_drop, off, x.shape,
                   a.shape if a is not None else None)
        if ylp is None:
            return y.view(x.shape), None
        ylp = ylp.view(x.shape)
        ctx.mark_non_differentiable(ylp)
        return y.view(x.shape), ylp

    @staticmethod
    def backward(ctx, dy, _dylp=None):
        if dy is None:
            if isinstance(ctx.link, PreNormLink) and ctx.link.buf is not None:     # nothing came back through the branch
                skip, ctx.link.buf = ctx.link.buf, None
                return (skip.view(ctx.cfg[6]),) + (None,) * 7
            return (None,) * 8
        z, mean, rstd, gamma, seed = ctx.saved_tensors
        M, d, adt, eps, p_drop, off, xshape, ashape = ctx.cfg
        dy2 = dy.reshape(-1, d).contiguous()
        dx = torch.empty_like(dy2)
        da = torch.empty((M, d), dtype=adt, device=dy.device) if adt is not None else None
        gg, gb = grad_target(ctx.g_ref), grad_target(ctx.b_ref)
        inplace = gg is not None and gb is not None
        if not inplace:
            dgb = torch.zeros((2, d), dtype=torch.float32, device=dy.device)
            gg, gb = dgb[0], dgb[1]
        gab, dab = None, None
        want_ab = ctx.ab_ref is not None and da is not None and ctx.needs_input_grad[6]
        if want_ab:
            gab = grad_target(ctx.ab_ref)
            if gab is None:
                gab = dab = torch.zeros((d,), dtype=torch.float32, device=dy.device)
        desc = L.LnDesc(M, d, _code(adt) if adt is not None else L.OTR_F32, eps, p_drop, off)
        part = None
        if inplace and dab is None and _wq['on'] and _in_backward() and d % 4 == 0:
            # in-place gradient buffers + deferred reductions: the kernel writes per-workgroup partial sums and the
            # three column sums join the grouped launch at the end of backward (no atomics, deterministic)
            nrow = L.load().otr_add_layernorm_bwd_partial_rows(M)
            part = torch.empty((nrow, 3 * d), dtype=torch.float32, device=dy.device)
        skip = None
        if isinstance(ctx.link, PreNormLink) and ctx.link.buf is not None:     # gradient through the skip connection of x + f(LN(x))
            skip, ctx.link.buf = ctx.link.buf, None
        L.check(L.load().otr_add_layernorm_bwd_skip(C.byref(desc), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(seed),
                                                    _p(skip), _p(dx), _p(da), _p(gg), _p(gb), _p(gab), _p(part), _stream()),
                'otr_add_layernorm_bwd')
        if part is not None:
            colsum_raw(part[:, :d], out=gg)
            colsum_raw(part[:, d:2 * d], out=gb)
            if want_ab:
                colsum_raw(part[:, 2 * d:], out=gab)
        dx_ret = dx.view(xshape)
        if isinstance(ctx.link, ResidualLink) and ctx.link.armed and ctx.needs_input_grad[0]:
            ctx.link.buf = dx           # the branch's first Linear adds its input gradient into this and returns the sum
            _park(ctx.link)
            dx_ret = None
        return (dx_ret, (da.view(ashape) if da is not None else None),
                None if inplace else gg, None if inplace else gb, None, None, dab, None)


class ResidualLnFn(torch.autograd.Function):
    """(z, y) = (x + scale * dropout(a), LayerNorm(z)): a pre-norm residual add (encoder/conformer.py:53-72) together with the LayerNorm
    that reads its result -- the norm at the head of the NEXT branch, or post_ffn_norm (:87) -- in one launch forward
    (otr_add_layernorm_fwd with a_scale, z is the launch's saved pre-norm sum) and one backward (otr_add_layernorm_bwd_skip: the
    gradient that reaches z from the residual stream is the `skip` operand; dx = skip + LayerNorm input gradient, da = scale *
    dropout'(dx)).  It was residual_add + add_layernorm: 36 + 36 launches of 5-7 us per Conformer step more.  `link`: x is also
    the input of the LayerNorm at the head of the branch that produced a (PreNormLink); its gradient goes there, as in ResidualAddFn.
    gamma2 / beta2: a SECOND LayerNorm on the first one's output in the same launches, y = LN2(LN1(z)) (encoder/conformer.py:87-89:
    post_ffn_norm, then final_norm; otr_add_layernorm2_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, a, scale, p_drop, gamma, beta, eps, link=None, gamma2=None, beta2=None, a_mask=None, lp_only=False):
        """lp_only (r06, 16-bit modes, one LayerNorm): the normalised rows leave as the 16-bit tensor ALONE and that tensor is the
        differentiable output -- its only reader is a Linear (q|k|v, pointwise_conv1), whose input gradient then comes back 16-bit: the
        input-gradient GEMM writes half the bytes and this launch's backward reads half (otr_ln_desc_t.dy_dtype)"""
        _cuda(x, a, gamma, beta)
        ctx.set_materialize_grads(False)
        ctx.a_mask = a_mask                # uint8 [M]: rows with 0 take no branch (module/conformer.py:109), forward and backward
        ctx.link = link if (link is not None and link.armed and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]) else None
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        a2 = a.reshape(-1, d).contiguous()
        M = x2.shape[0]
        lp_only = bool(lp_only and is_half() and gamma2 is None)
        z = torch.empty_like(x2)
        y = None if lp_only else torch.empty_like(x2)
        ylp = torch.empty(x2.shape, dtype=half_dtype(), device=x.device) if is_half() else None
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        seed = rng_seed_tensor(x.device) if p_drop > 0 else None
        off = _next_rng_offset(M * d) if p_drop > 0 else 0
        desc = L.LnDesc(M, d, _code(a2.dtype), eps, p_drop, off, scale, a_mask.data_ptr() if a_mask is not None else None)
        two = gamma2 is not None
        if two:
            mean2, rstd2 = torch.empty_like(mean), torch.empty_like(mean)
            L.check(L.load().otr_add_layernorm2_fwd(C.byref(desc), _p(x2), _p(a2), _p(gamma), _p(beta), _p(gamma2), _p(beta2), _p(seed), _p(y),
                                                    _p(ylp), _p(z), _p(mean), _p(rstd), _p(mean2), _p(rstd2), _stream()),
                    'otr_add_layernorm2_fwd')
            ctx.save_for_backward(z, mean, rstd, gamma, seed, beta, mean2, rstd2, gamma2)
        else:
            L.check(L.load().otr_add_layernorm_fwd(C.byref(desc), _p(x2), _p(a2), _p(gamma), _p(beta), _p(seed), _p(y), _p(ylp), _p(z),
                                                   _p(mean), _p(rstd), _stream()), 'otr_add_layernorm_fwd')
            ctx.save_for_backward(z, mean, rstd, gamma, seed)
        ctx.g_ref, ctx.b_ref, ctx.g2_ref, ctx.b2_ref = gamma, beta, gamma2, beta2
        ctx.cfg = (M, d, a2.dtype, eps, p_drop, off, scale, x.shape, a.shape, two)
        if lp_only:
            return z.view(x.shape), ylp.view(x.shape), None
        if ylp is None:
            return z.view(x.shape), y.view(x.shape), None
        ylp = ylp.view(x.shape)
        ctx.mark_non_differentiable(ylp)
        return z.view(x.shape), y.view(x.shape), ylp

    @staticmethod
    def backward(ctx, dz, dy, _dylp=None):
        if dz is None and dy is None:
            return (None,) * 12
        M, d, adt, eps, p_drop, off, scale, xshape, ashape, two = ctx.cfg
        if two:
            z, mean, rstd, gamma, seed, beta, mean2, rstd2, gamma2 = ctx.saved_tensors
        else:
            z, mean, rstd, gamma, seed = ctx.saved_tensors
        dy2 = dy.reshape(-1, d).contiguous() if dy is not None else torch.zeros((M, d), dtype=torch.float32, device=z.device)
        if dy2.dtype not in (torch.float32, half_dtype()) or (dy2.dtype != torch.float32 and two):
            dy2 = dy2.float()
        skip = dz.reshape(-1, d).contiguous() if dz is not None else None
        dx = torch.empty((M, d), dtype=torch.float32, device=z.device)
        da = torch.empty((M, d), dtype=adt, device=z.device)
        refs = (ctx.g_ref, ctx.b_ref) + ((ctx.g2_ref, ctx.b2_ref) if two else ())
        targets = [grad_target(r) for r in refs]
        inplace = all(t is not None for t in targets)
        lib = L.load()
        desc = L.LnDesc(M, d, _code(adt), eps, p_drop, off, scale, ctx.a_mask.data_ptr() if ctx.a_mask is not None else None, _code(dy2.dtype))
        ret = [None] * len(refs)
        if two:
            part = torch.empty((lib.otr_add_layernorm_bwd_partial_rows(M), 5 * d), dtype=torch.float32, device=z.device)
            L.check(lib.otr_add_layernorm2_bwd(C.byref(desc), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(mean2), _p(rstd2),
                                               _p(gamma2), _p(seed), _p(skip), _p(dx), _p(da), _p(part), _stream()), 'otr_add_layernorm2_bwd')
            cols = (0, 1, 3, 4)                       # dgamma | dbeta | (da sums) | dgamma2 | dbeta2
            if inplace and _wq['on'] and _in_backward() and d % 4 == 0:
                for t, c in zip(targets, cols):
                    colsum_raw(part[:, c * d:(c + 1) * d], out=t)
            else:
                sums = part.sum(0)
                for i, (t, c) in enumerate(zip(targets, cols)):
                    if inplace:
                        t.add_(sums[c * d:(c + 1) * d])
                    else:
                        ret[i] = sums[c * d:(c + 1) * d]
        else:
            gg, gb = targets
            if not inplace:
                dgb = torch.zeros((2, d), dtype=torch.float32, device=z.device)
                gg, gb = dgb[0], dgb[1]
                ret = [gg, gb]
            part = None
            if inplace and _wq['on'] and _in_backward() and d % 4 == 0:
                part = torch.empty((lib.otr_add_layernorm_bwd_partial_rows(M), 3 * d), dtype=torch.float32, device=z.device)
            L.check(lib.otr_add_layernorm_bwd_skip(C.byref(desc), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(seed), _p(skip),
                                                   _p(dx), _p(da), _p(gg), _p(gb), None, _p(part), _stream()), 'otr_add_layernorm_bwd')
            if part is not None:
                colsum_raw(part[:, :d], out=gg)
                colsum_raw(part[:, d:2 * d], out=gb)
        dx_ret = dx.view(xshape)
        if ctx.link is not None:
            ctx.link.buf = dx
            _park(ctx.link)
            dx_ret = None
        g2 = (ret[2], ret[3]) if two else (None, None)
        return dx_ret, da.view(ashape), None, None, ret[0], ret[1], None, None, g2[0], g2[1], None, None


class ResidualLn3Fn(torch.autograd.Function):
    """(z, y2, y3) = (x + scale * dropout(a), LN2(LN1(z)), LN3(y2)): the residual add that closes a Conformer block's convolution branch, the
    block's two closing LayerNorms (encoder/conformer.py:87-89) AND the macaron LayerNorm at the head of the NEXT block (:50) in one launch
    each way (otr_add_layernorm3_fwd / _bwd, r06).  y2 is the residual stream the next block adds to, y3 (with its 16-bit twin) the input
    of the next block's first feed-forward; autograd hands the backward both gradients, so no link is needed: d y2 += LayerNorm-3
    backward of d y3, then the two-LayerNorm chain as in ResidualLnFn."""

    @staticmethod
    def forward(ctx, x, a, scale, p_drop, gamma, beta, eps, gamma2, beta2, gamma3, beta3, a_mask=None):
        _cuda(x, a, gamma, beta, gamma2, beta2, gamma3, beta3)
        ctx.set_materialize_grads(False)
        ctx.a_mask = a_mask
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        a2 = a.reshape(-1, d).contiguous()
        M = x2.shape[0]
        z, y2 = torch.empty_like(x2), torch.empty_like(x2)
        half = is_half()
        y3 = None if half else torch.empty_like(x2)                   # a 16-bit consumer reads the twin only
        y3lp = torch.empty(x2.shape, dtype=half_dtype(), device=x.device) if half else None
        st = [torch.empty((M,), dtype=torch.float32, device=x.device) for _ in range(6)]
        seed = rng_seed_tensor(x.device) if p_drop > 0 else None
        off = _next_rng_offset(M * d) if p_drop > 0 else 0
        desc = L.LnDesc(M, d, _code(a2.dtype), eps, p_drop, off, scale, a_mask.data_ptr() if a_mask is not None else None)
        L.check(L.load().otr_add_layernorm3_fwd(C.byref(desc), _p(x2), _p(a2), _p(gamma), _p(beta), _p(gamma2), _p(beta2), _p(gamma3), _p(beta3),
                                                _p(seed), _p(y2), _p(y3), _p(y3lp), _p(z), *[_p(t) for t in st], _stream()),
                'otr_add_layernorm3_fwd')
        ctx.save_for_backward(z, gamma, beta, gamma2, beta2, gamma3, seed, *st)
        ctx.refs = (gamma, beta, gamma2, beta2, gamma3, beta3)
        ctx.cfg = (M, d, a2.dtype, eps, p_drop, off, scale, x.shape, a.shape)
        if half:
            # the fp32 y3 is never materialised: hand out a [.., d] view of nothing but the twin's owner -- a zero-stride placeholder would
            # break consumers that ask for rows; the 16-bit tensor upcast lazily is not needed either, every consumer takes lp_of()
            y3 = y3lp
        y3 = y3.view(x.shape)
        return z.view(x.shape), y2.view(x.shape), y3

    @staticmethod
    def backward(ctx, dz, dy2, dy3):
        if dz is None and dy2 is None and dy3 is None:
            return (None,) * 12
        z, gamma, beta, gamma2, beta2, gamma3, seed, mean, rstd, mean2, rstd2, mean3, rstd3 = ctx.saved_tensors
        M, d, adt, eps, p_drop, off, scale, xshape, ashape = ctx.cfg
        zeros = None
        def rows(t):
            nonlocal zeros
            if t is None:
                if zeros is None:
                    zeros = torch.zeros((M, d), dtype=torch.float32, device=z.device)
                return zeros
            return t.reshape(-1, d).float().contiguous()
        g2 = rows(dy2)
        g3 = dy3.reshape(-1, d).contiguous() if (dy3 is not None and dy3.dtype in (torch.float32, half_dtype())) else rows(dy3)
        skip = dz.reshape(-1, d).contiguous() if dz is not None else None
        dx = torch.empty((M, d), dtype=torch.float32, device=z.device)
        da = torch.empty((M, d), dtype=adt, device=z.device)
        lib = L.load()
        desc = L.LnDesc(M, d, _code(adt), eps, p_drop, off, scale, ctx.a_mask.data_ptr() if ctx.a_mask is not None else None)
        part = torch.empty((lib.otr_add_layernorm_bwd_partial_rows(M), 7 * d), dtype=torch.float32, device=z.device)
        L.check(lib.otr_add_layernorm3_bwd(C.byref(desc), _p(g2), _p(g3), _code(g3.dtype), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(mean2), _p(rstd2),
                                           _p(gamma2), _p(beta2), _p(mean3), _p(rstd3), _p(gamma3), _p(seed), _p(skip), _p(dx), _p(da), _p(part),
                                           _stream()), 'otr_add_layernorm3_bwd')
        targets = [grad_target(r) for r in ctx.refs]
        inplace = all(t is not None for t in targets)
        cols = (0, 1, 3, 4, 5, 6)                     # dgamma | dbeta | (da sums) | dgamma2 | dbeta2 | dgamma3 | dbeta3
        ret = [None] * 6
        if inplace and _wq['on'] and _in_backward() and d % 4 == 0:
            for t, c in zip(targets, cols):
                colsum_raw(part[:, c * d:(c + 1) * d], out=t)
        else:
            sums = part.sum(0)
            for i, (t, c) in enumerate(zip(targets, cols)):
                if inplace:
                    t.add_(sums[c * d:(c + 1) * d])
                else:
                    ret[i] = sums[c * d:(c + 1) * d]
        return dx.view(xshape), da.view(ashape), None, None, ret[0], ret[1], None, ret[2], ret[3], ret[4], ret[5], None


def residual_layernorm3(x, a, scale, p_drop, n1, n2, n3, a_mask=None):
    """(x + scale * dropout(a), y2 = LN2(LN1(sum)), y3 = LN3(y2)) in one launch (ResidualLn3Fn); n1..n3: the nn.LayerNorm modules (one eps)"""
    return ResidualLn3Fn.apply(x, a, float(scale), float(p_drop), n1.weight, n1.bias, float(n1.eps), n2.weight, n2.bias, n3.weight, n3.bias, a_mask)


def residual_layernorm(x, a, scale, p_drop, gamma, beta, eps=1e-5, link=None, gamma2=None, beta2=None, a_mask=None, lp_only=False):
    """(x + scale * dropout(a), LayerNorm of that sum [with its 16-bit twin]) in one launch: ResidualLnFn; with gamma2 / beta2 the
    second value is LN2(LN1(sum)); a_mask (uint8, one per row): rows with 0 take no branch; lp_only: the second value is the 16-bit
    tensor alone (its reader is a Linear)"""
    z, y, ylp = ResidualLnFn.apply(x, a, float(scale), float(p_drop), gamma, beta, float(eps), link, gamma2, beta2, a_mask, lp_only)
    return z, (attach_lp(y, ylp) if ylp is not None else y)


def add_layernorm(x, a, gamma, beta, p_drop=0.0, eps=1e-5, a_bias=None, link=None):
    """a_bias: the bias parameter of the Linear that produced `a` (called with defer_bias=True); its gradient
    (column sums of d loss / d a) is then reduced inside the LayerNorm backward kernel."""
    y, ylp = AddLayerNormFn.apply(x, a, gamma, beta, float(p_drop), float(eps), a_bias, link)
    return attach_lp(y, ylp)


class ProjLnFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(c . W^T + b)): output projection + residual + LayerNorm of an attention sub-layer
    (module/attention.py:75,140 + encoder/transformer.py:54-56, decoder/transformer.py:66-80) in ONE launch forward;
    backward = LayerNorm backward + the projection's input gradient in one launch (csrc/rowblock.hip), the weight / bias /
    affine gradients join the grouped launches at the end of the pass."""

    @staticmethod
    def forward(ctx, x, c, w, b, gamma, beta, p_drop, eps, packs, link, ilink=None, touch=None, touch_w=None):
        _cuda(x, c, w, gamma, beta)
        ctx.set_materialize_grads(False)
        materialize(x)
        ctx.link = link
        ctx.ilink = ilink
        ctx.touch = touch           # the attention launch's saved q|k|v: this Function's backward launch touches it for the one that follows
        ctx.touch_w = touch_w       # the q|k|v projection's input-gradient pack (read two launches later)
        if ilink is not None:
            ilink.armed = any(ctx.needs_input_grad)
        if link is not None:            # the branch's first Linear armed it under ITS conditions (fp32 x, no perm, no relu): keep them
            link.armed = link.armed and bool(ctx.needs_input_grad[0])
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        c2 = _rows(c)
        M = x2.shape[0]
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(x2)
        ylp = torch.empty(x2.shape, dtype=half_dtype(), device=x.device)
        z = torch.empty_like(x2) if need_grad else None
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        seed = rng_seed_tensor(x.device) if p_drop > 0 else None
        off = _next_rng_offset(M * d) if p_drop > 0 else 0
        L.check(_timed('proj_ln_fwd', {'flops': 2.0 * M * d * d},
                       lambda: L.load().otr_proj_ln_fwd(_p(x2), _p(c2), c2.stride(0), _p(packs[0]), _p(b), _p(gamma), _p(beta), _p(seed),
                                                        _p(y), _p(ylp), _p(z), _p(mean), _p(rstd), M, d, eps, p_drop, off, _stream())),
                'otr_proj_ln_fwd')
        ctx.save_for_backward(z, mean, rstd, gamma, seed, c2)
        if ilink is not None:
            ilink.z = z
        ctx.refs = (w, b, gamma, beta)
        ctx.cfg = (M, d, p_drop, off, x.shape, c.shape, packs)
        ylp = ylp.view(x.shape)
        ctx.mark_non_differentiable(ylp)
        return y.view(x.shape), ylp

    @staticmethod
    def backward(ctx, dy, _dylp=None):
        stash = None
        if ctx.ilink is not None:
            stash, ctx.ilink.result = ctx.ilink.result, None
        if dy is None and stash is None:
            return (None,) * 13
        z, mean, rstd, gamma, seed, c2 = ctx.saved_tensors
        w, b, g_ref, b_ref = ctx.refs
        M, d, p_drop, off, xshape, cshape, packs = ctx.cfg
        dev = z.device
        dx = torch.empty((M, d), dtype=torch.float32, device=dev)
        da = torch.empty((M, d), dtype=half_dtype(), device=dev)
        dc = torch.empty((M, d), dtype=half_dtype(), device=dev)
        nrow = L.load().otr_ln_bwd_proj_partial_rows(M)
        part = torch.empty((nrow, 3 * d), dtype=torch.float32, device=dev)
        if ctx.touch_w is not None and _QKV_W_TOUCH and not (ctx.touch is not None and _ATTN_PREFETCH):
            tw = ctx.touch_w
            L.check(L.load().otr_touch_hint(_p(tw), tw.numel() * tw.element_size(), None, 0), 'otr_touch_hint')
        if ctx.touch is not None and _ATTN_PREFETCH:
            # the attention backward launch runs next and would fetch its saved q|k|v and context (c2) cold: csrc/rowblock.hip RbTouch
            t = ctx.touch
            L.check(L.load().otr_touch_hint(_p(t), t.numel() * t.element_size(), _p(c2), c2.numel() * c2.element_size()), 'otr_touch_hint')
        if stash is not None:           # see LnInLink: the FFN's backward launch left (skip-path gradient, four slabs)
            dskip, bslabs = stash
            if dy is not None and not _is_zero_placeholder(dy):
                dskip = dskip + dy.reshape(-1, d)
            L.check(_timed('ln_bwd_proj', {'flops': 2.0 * M * d * d},
                           lambda: L.load().otr_ln_bwd_proj_slabs(_p(dskip), _p(bslabs), bslabs.shape[0], _p(z), _p(mean), _p(rstd),
                                                                  _p(gamma), _p(seed), _p(packs[1]), _p(dx), _p(da), _p(dc), dc.stride(0),
                                                                  _p(part), M, d, p_drop, off, _stream())), 'otr_ln_bwd_proj_slabs')
        else:
            dy2 = dy.reshape(-1, d).contiguous()
            L.check(_timed('ln_bwd_proj', {'flops': 2.0 * M * d * d},
                           lambda: L.load().otr_ln_bwd_proj(_p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(seed), _p(packs[1]), _p(dx),
                                                            _p(da), _p(dc), dc.stride(0), _p(part), M, d, p_drop, off, _stream())),
                    'otr_ln_bwd_proj')
        outs = []
        for k, ref in ((0, g_ref), (1, b_ref), (2, b)):
            if ref is None:
                outs.append(None)
                continue
            gt = grad_target(ref)
            sl = part[:, k * d:(k + 1) * d]
            if gt is not None:
                colsum_raw(sl, out=gt)
                outs.append(None)
            else:
                outs.append(colsum_raw(sl))
        dgamma, dbeta, dbias = outs
        gw = grad_target(w)
        dw = linear_wgrad_raw(da, c2, w, out=gw)
        dx_ret = dx.view(xshape)
        if isinstance(ctx.link, ResidualLink) and ctx.link.armed and ctx.needs_input_grad[0]:
            ctx.link.buf = dx           # the branch's first Linear adds its input gradient into this and returns the sum
            _park(ctx.link)
            dx_ret = None
        return (dx_ret, dc.view(cshape), None if gw is not None else dw, dbias, dgamma, dbeta, None, None, None, None, None, None, None)


def proj_ln_packs(x, c, w, gamma):
    """packs of `w` when LN(x + drop(c . w^T + b)) can run as the row-block kernels, else None"""
    if not is_half() or x.dtype != torch.float32 or x.shape[-1] != 256 or tuple(w.shape) != (256, 256) or gamma is None:
        return None
    if c.dtype != half_dtype() or not _rb_rows_ok(_rows(c)):
        return None
    return lin_packs(w)


_FFN_FWD_TOUCH = True


def touch_ffn_packs_next(ff, x):
    """the next otr_proj_ln_fwd launch (the attention sub-layer's closing launch) touches the two forward packs of the split FFN that
    follows it (3 MB the FFN launch's 252 workgroups would all wait for, cold, at their first phases): 4.43 -> 4.37 ms per step"""
    if not _FFN_FWD_TOUCH or ff is None or getattr(ff, 'activation', None) != 'glu' or x.numel() // 256 < _FFN_SPLIT_MIN_ROWS:
        return
    packs = ffn_packs(ff.w_1.weight, ff.w_2.weight)
    if packs is None or packs[1].data_ptr() != packs[0].data_ptr() + packs[0].numel() * packs[0].element_size():
        return
    L.check(L.load().otr_touch_hint(_p(packs[0]), (packs[0].numel() + packs[1].numel()) * packs[0].element_size(), None, 0), 'otr_touch_hint')


def proj_add_layernorm(x, c, w, b, gamma, beta, p_drop, eps, packs, link=None):
    ilink = LnInLink() if (_FFN_SLAB and torch.is_grad_enabled()) else None
    y, ylp = ProjLnFn.apply(x, c, w, b, gamma, beta, float(p_drop), float(eps), packs, link, ilink, getattr(c, '_otr_touch', None),
                            getattr(c, '_otr_touch_w', None))
    if ilink is not None and ilink.armed:
        y._otr_inlink = ilink
    return attach_lp(y, ylp)


# ---------------------------------------------------------------------------------------- FFN (fused Function)
class FeedForwardGLUFn(torch.autograd.Function):
    """w_2(glu(w_1 x)) of module/ffn.py:38-41 with activation 'glu': three launches forward
    (GEMM, GLU, GEMM); backward fuses the w_1 bias gradient into the GLU-backward kernel."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, defer_b2=False, out_dtype=torch.float32, link=None):
        _cuda(x, w1, w2)
        ctx.defer_b2 = defer_b2          # see linear(defer_bias=True)
        ctx.link = link
        if link is not None:
            link.armed = bool(ctx.needs_input_grad[0]) and x.dtype == torch.float32
        ctx.refs = (w1, b1, w2, b2)
        xc = lp_of(x)
        x2 = _rows(xc if xc is not None else x)
        adt = act_dtype()
        w1l, w2l = weight_lp(w1), weight_lp(w2)
        ctx.w1t, ctx.w2t = weight_lpt(w1), weight_lpt(w2)
        w1 = w1l if w1l is not None else w1
        w2 = w2l if w2l is not None else w2
        M, F2 = x2.shape[0], w1.shape[0]
        F = F2 // 2
        h = None
        u = torch.empty((M, F), dtype=adt, device=x.device)
        if adt != torch.float32 and x2.dtype == adt and w1.dtype == adt and _FUSED_GLU_FWD:
            h = torch.empty((M, F2), dtype=adt, device=x.device)
            rc = L.load().otr_ffn_glu_fwd(_p(x2), x2.stride(0), _p(w1), w1.stride(0), _p(b1), _p(h), _p(u), M, F,
                                          x2.shape[1], _stream())
            if rc == 1:
                h = None                         # operands do not qualify: two-kernel path below
            else:
                L.check(rc, 'otr_ffn_glu_fwd')
        ctx.h_sig = h is not None               # fused forward keeps (value | sigmoid(gate)) in h
        if h is None:
            h = linear_fwd_raw(x2, w1, b1, adt)
            L.check(L.load().otr_glu_fwd(_p(h), _p(u), _code(adt), M, F, None, _stream()), 'otr_glu_fwd')
        y = linear_fwd_raw(u, w2, b2, out_dtype)
        ctx.save_for_backward(x2, w1, w2, h, u)
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, u = ctx.saved_tensors
        w1p, b1p, w2p, b2p = ctx.refs
        dy2 = _rows(dy)
        M, F = u.shape
        gw2, gb2, gw1, gb1 = grad_target(w2p), grad_target(b2p), grad_target(w1p), grad_target(b1p)
        dw2 = linear_wgrad_raw(dy2, u, w2, out=gw2)
        db2 = None if ctx.defer_b2 else colsum_raw(dy2, out=gb2)
        dh = torch.empty_like(h)
        part = None
        if ctx.w2t is not None and dy2.dtype == half_dtype() and h.dtype == dy2.dtype and is_half() and _FUSED_GLU_BWD:
            # one launch: du = dy . w2 stays in registers / LDS, GLU backward and the bias partials in the GEMM epilogue
            cap = (M + 63) // 64
            part = torch.empty((cap, 2 * F), dtype=torch.float32, device=dy.device)
            rows = C.c_int32(0)
            rc = L.load().otr_ffn_glu_bwd(_p(dy2), _code(dy2.dtype), dy2.stride(0), _p(ctx.w2t), ctx.w2t.stride(0), _p(h),
                                          int(ctx.h_sig), _p(dh), _p(part), cap, C.byref(rows), M, F, dy2.shape[1], _stream())
            if rc == 1:
                part = None                     # operands do not qualify: two-kernel path below
            else:
                L.check(rc, 'otr_ffn_glu_bwd')
                part = part[:rows.value]
        if part is None:
            du = linear_fwd_raw(dy2, ctx.w2t, None, u.dtype) if ctx.w2t is not None else linear_dgrad_raw(dy2, w2, u.dtype)
            nblk = (M + GLU_RPB - 1) // GLU_RPB
            part = torch.empty((nblk, 2 * F), dtype=torch.float32, device=dy.device)
            L.check(L.load().otr_glu_bwd(_p(h), _p(du), _p(dh), _p(part), _code(h.dtype), M, F, None, int(ctx.h_sig), _stream()),
                    'otr_glu_bwd')
        db1 = colsum_raw(part, out=gb1)
        skip = None
        if ctx.link is not None and ctx.link.buf is not None:
            skip, ctx.link.buf = ctx.link.buf, None
        if ctx.w1t is not None:
            dx = linear_fwd_raw(dh, ctx.w1t, None, ctx.xdtype, out=skip).view(ctx.xshape)
        else:
            dx = linear_dgrad_raw(dh, w1, ctx.xdtype, out=skip).view(ctx.xshape)
        dw1 = linear_wgrad_raw(dh, x2, w1, out=gw1)
        return (dx, None if gw1 is not None else dw1, None if gb1 is not None else db1,
                None if gw2 is not None else dw2, None if (gb2 is not None or ctx.defer_b2) else db2, None, None, None)


# ---------------------------------------------------------------------------------------- row-block fused FFN sub-layer
# y = LN(x + dropout(FFN(x))) in ONE launch forward and ONE launch (+ the LayerNorm backward) backward; the d_ff-wide hidden
# is never stored: the backward pass recomputes it (csrc/ffn_fused.hip).  Weights are consumed "fragment-major": four packed
# copies per FFN (w_1, w_2, w_2^T, w_1^T in MFMA-operand order) that FlatDataParallel refreshes after every optimizer step
# (one otr_pack_frags launch for the whole model) or, for stand-alone modules, a cache keyed by the parameter versions.
_FUSED_FFN = True
_FUSED_FFN_MIN_ROWS = 1024   # below: too few 32-row workgroups to fill the chip


# 128-row workgroups with the hidden units split four ways and the partial sums exchanged inside the launch (csrc/ffn3.hip):
# the default from 2048 rows up; OTR_FFN_SPLIT=0 keeps the 32-row kernels
_FFN_SPLIT = True
# the split kernels in slab mode (no in-launch exchange; the LayerNorm moves into the next launch's prologue) where the caller allows it
_FFN_SLAB = True
_FFN_PREFETCH = True
_Z_TOUCH = False
_QKV_W_TOUCH = False              # experiment: ln_bwd_proj touches the q|k|v input-gradient pack
_FFN_HSAVE_TOUCH = False      # experiment: the saved tiles (65 MB) as well
# the same for the attention backward launch's saved q|k|v + context, touched by the LayerNorm-backward launch before it (otr_touch_hint):
# -3.6 us per launch in tools/encattn_prefetch_probe.py, nothing measurable in the step (4.478 vs 4.471 / 4.500 ms on one box): off
_ATTN_PREFETCH = False
_FFN_SPLIT_MIN_ROWS = 2048
_FFN_SYNC_INTS = 1 << 14


_FFN_SYNC_STREAMS = 16


def _ffn_sync_pool(device):
    p = _state.get('ffn_sync_pool')
    if p is None or p[0].device != device:
        p = (torch.zeros((_FFN_SYNC_STREAMS, _FFN_SYNC_INTS), dtype=torch.int32, device=device), {})
        _state['ffn_sync_pool'] = p
    return p


def _ffn_sync(device):
    """Arrival counters of the split FFN kernels (include/otrans_hip.h): zero ONCE, then owned by the kernels -- the counters are
    monotonic (a row block gains 4 per launch, unsigned, wrap-safe), and launches that share a buffer must be stream-ordered.  So
    the buffer is keyed by (device, current stream): training on one stream and an eval / recognizer graph on another never mix
    their arrivals.  The pool is allocated by _workspace() before the first launch, i.e. outside any graph capture's private pool;
    a stream takes the next free row at its first split-FFN launch (a capture's stream included: captured launches replay in
    capture order whatever stream replays them)."""
    _workspace(device)
    pool, rows = _ffn_sync_pool(device)
    sid = torch.cuda.current_stream().cuda_stream
    i = rows.get(sid)
    if i is None:
        if len(rows) >= _FFN_SYNC_STREAMS:
            raise L.OtransHipError('opentransformer_amd: split-FFN launches from more than %d streams on one device' % _FFN_SYNC_STREAMS)
        i = rows[sid] = len(rows)
    return pool[i]


def _ffn_split(M, F):
    return (_FFN_SPLIT and M >= _FFN_SPLIT_MIN_ROWS and F % 256 == 0 and F // 32 // 4 <= 32
            and 8 * ((M + 127) // 128) <= _FFN_SYNC_INTS)


def ffn_pack_items(w1_off, w2_off, F2, d, F, dst_off):
    """otr_pack_frags table rows {src_off, rs, cs, rows, cols, perm, dst_off} for one FFN (w_1 [2F,d], w_2 [d,F]) and the
    element offsets of its four packs inside the destination buffer."""
    n1, n2 = F2 * d, d * F
    o1, o2, o3, o4 = dst_off, dst_off + n1, dst_off + n1 + n2, dst_off + n1 + 2 * n2
    rows = [[w1_off, d, 1, F2, d, 0, o1],        # P1: A[f'][k]  = w_1[f'][k]          (h = w_1 x)
            [w2_off, F, 1, d, F, 1, o2],         # P2: A[n][f]   = w_2[n][f], perm     (y = w_2 u, u from accumulators)
            [w2_off, 1, F, F, d, 0, o3],         # P3: A[f][n]   = w_2[n][f]           (du = dy . w_2)
            [w1_off, 1, d, d, F2, 1, o4]]        # P4: A[k][f']  = w_1[f'][k], perm    (dx = dh . w_1, dh from accumulators)
    return rows, (o1, o2, o3, o4), 2 * (n1 + n2)


def pack_frags(src, dst, rows):
    """run otr_pack_frags for table rows built by ffn_pack_items (first-block column appended here)"""
    table, blocks = [], 0
    for r in rows:
        table.append(list(r) + [blocks])
        blocks += ((r[3] // 32) * (r[4] // 16) + 3) // 4
    t = torch.tensor(table, dtype=torch.int64, device=src.device)
    L.check(L.load().otr_pack_frags(_p(src), _p(dst), _p(t), len(table), blocks, _stream()), 'otr_pack_frags')
    return t, blocks


def ffn_packs(w1, w2):
    """(P1, P2, P3, P4) 16-bit packed weights of one FFN, or None when the fused kernels do not apply."""
    if not _FUSED_FFN or _state['compute'] == 'fp32' or w1.dim() != 2 or w2.dim() != 2:
        return None
    F2, d = w1.shape
    F = F2 // 2
    if d != 256 or F % 256 != 0 or tuple(w2.shape) != (d, F) or not w1.is_cuda:
        return None
    views = getattr(w1, '_otr_ffn_packs', None)
    if views is not None:                      # slices of FlatDataParallel's pack buffer (kept fresh by the optimizer)
        return views
    if getattr(w1, '_otr_grad_inplace', False) or getattr(w2, '_otr_grad_inplace', False):
        return None                            # replica parameters without registered packs: no version-keyed cache (see lin_packs)
    key = (w1._version, w1.data_ptr(), w2._version, w2.data_ptr(), _state['compute'])
    cache = getattr(w1, '_otr_ffn_pack_cache', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    l1, l2 = weight_lp(w1), weight_lp(w2)
    src = torch.cat((l1.reshape(-1), l2.reshape(-1)))
    rows, offs, total = ffn_pack_items(0, l1.numel(), F2, d, F, 0)
    dst = torch.empty(total, dtype=src.dtype, device=src.device)
    pack_frags(src, dst, rows)
    n1, n2 = F2 * d, d * F
    packs = (dst[offs[0]:offs[0] + n1], dst[offs[1]:offs[1] + n2], dst[offs[2]:offs[2] + n2], dst[offs[3]:offs[3] + n1])
    w1._otr_ffn_pack_cache = (key, packs)
    return packs


class FfnLnFn(torch.autograd.Function):
    """LN(x + dropout(w_2(glu(w_1 x + b_1)) + b_2)): the FFN sub-layer of a post-norm layer (encoder/transformer.py:58-63,
    decoder/transformer.py:82-86) on the row-block fused kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, p_drop, eps, packs, olink=None, pend=None, ilink=None):
        _cuda(x, w1, b1, w2, b2, gamma, beta)
        ctx.set_materialize_grads(False)
        materialize(x)
        d = x.shape[-1]
        x2 = x.reshape(-1, d)
        x16 = lp_of(x).reshape(-1, d)
        M, F = x2.shape[0], w2.shape[1]
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(x2)
        y16 = torch.empty(x2.shape, dtype=x16.dtype, device=x.device)
        z = torch.empty_like(x2) if need_grad else None
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        seed = rng_seed_tensor(x.device) if p_drop > 0 else None
        off = _next_rng_offset(M * d) if p_drop > 0 else 0
        ctx.split = _ffn_split(M, F)
        ctx.slab = ctx.split and pend is not None and x2.is_contiguous()
        ctx.ilink = ilink if ctx.slab else None
        hsave = usave = None
        if ctx.slab:
            # slab mode: the four hidden slices leave 16-bit partial sums, whoever reads y finishes the LayerNorm (PendingLn)
            lib = L.load()
            if need_grad:
                hsave = torch.empty(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=x16.dtype, device=x.device)
                usave = torch.empty((lib.otr_ffn_split_padded_rows(M), F), dtype=x16.dtype, device=x.device)
            slabs = torch.empty((4, M, d), dtype=x16.dtype, device=x.device)
            L.check(_timed('ffn_fwd_slab', {'flops': 6.0 * M * F * d, 'bytes': M * d * (2 + 8) + 6 * F * d + (M * F * 6 if need_grad else 0)},
                           lambda: lib.otr_ffn_fwd_split_slab(_p(x16), _p(packs[0]), _p(b1), _p(packs[1]), _p(hsave), _p(usave), _p(slabs),
                                                              M, F, d, _stream())), 'otr_ffn_fwd_split_slab')
            pend.kw = dict(xres=x2, slabs=slabs, nslab=4, bias=b2, gamma=gamma, beta=beta, seed=seed, p_drop=p_drop, eps=eps, off=off,
                           y=y, y16=y16, z=z, mean=mean, rstd=rstd)
            pend.M, pend.done = M, False
        elif ctx.split:
            lib = L.load()
            nb = lib.otr_ffn_split_scratch_bytes(M)
            scratch = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
            sync = _ffn_sync(x.device)
            if need_grad:       # what the backward kernel reads back instead of recomputing the hidden (csrc/ffn3.hip)
                hsave = torch.empty(lib.otr_ffn_split_hsave_bytes(M, F) // 2, dtype=x16.dtype, device=x.device)
                usave = torch.empty((lib.otr_ffn_split_padded_rows(M), F), dtype=x16.dtype, device=x.device)
            L.check(_timed('ffn_ln_fwd_split', {'flops': 6.0 * M * F * d,
                                                'bytes': M * d * (4 + 2 + 4 + 2 + 4) + 6 * F * d + (M * F * 6 if need_grad else 0)},
                           lambda: lib.otr_ffn_ln_fwd_split(_p(x2), _p(x16), _p(packs[0]), _p(b1), _p(packs[1]), _p(b2), _p(gamma),
                                                            _p(beta), _p(seed), p_drop, off, eps, _p(y), _p(y16), _p(z), _p(mean),
                                                            _p(rstd), _p(hsave), _p(usave), _p(scratch), nb, _p(sync), sync.numel(),
                                                            M, F, d, _stream())),
                    'otr_ffn_ln_fwd_split')
        else:
            L.check(_timed('ffn_ln_fwd', {'flops': 6.0 * M * F * d, 'bytes': M * d * (4 + 2 + 4 + 2 + 4) + 6 * F * d},
                           lambda: L.load().otr_ffn_ln_fwd(_p(x2), _p(x16), _p(packs[0]), _p(b1), _p(packs[1]), _p(b2), _p(gamma),
                                                           _p(beta), _p(seed), p_drop, off, eps, _p(y), _p(y16), _p(z), _p(mean),
                                                           _p(rstd), M, F, d, _stream())), 'otr_ffn_ln_fwd')
        ctx.save_for_backward(x16, z, mean, rstd, gamma, seed, b1, hsave, usave)
        ctx.packs = packs
        ctx.refs = (w1, b1, w2, b2, gamma, beta)
        ctx.cfg = (M, d, F, eps, p_drop, off, x.shape)
        ctx.olink = olink
        if olink is not None and need_grad:        # see LnOutLink: the next layer's first Linear may run this LayerNorm's backward
            olink.saved, olink.params, olink.armed = (z, mean, rstd, gamma, seed, p_drop, off), (gamma, beta, b2), True
            if ctx.split and _FFN_PREFETCH and packs[3].data_ptr() == packs[2].data_ptr() + packs[2].numel() * packs[2].element_size():
                # that launch is followed by THIS sub-layer's backward launch, whose two packs (adjacent in the pack buffer) it
                # would fetch cold: have them touched on the way (otr_rb_linear_ln_bwd_pf)
                extra = hsave if (_FFN_HSAVE_TOUCH and hsave is not None) else None
                if extra is None and _Z_TOUCH and ilink is not None:
                    extra = ilink.z          # experiment: the saved pre-norm sums of the attention sub-layer's LayerNorm (read two launches later)
                olink.prefetch = (packs[2], (packs[2].numel() + packs[3].numel()) * packs[2].element_size(), extra)
        y16 = y16.view(x.shape)
        ctx.mark_non_differentiable(y16)
        return y.view(x.shape), y16

    @staticmethod
    def backward(ctx, dy, _dy16=None):
        if dy is None:
            return (None,) * 13
        x16, z, mean, rstd, gamma, seed, b1, hsave, usave = ctx.saved_tensors
        M, d, F, eps, p_drop, off, xshape = ctx.cfg
        w1p, b1p, w2p, b2p, gp, bp = ctx.refs
        P1, _, P3, P4 = ctx.packs
        lib = L.load()
        gg, gb, gb2 = grad_target(gp), grad_target(bp), grad_target(b2p)
        stash = None
        if ctx.olink is not None:
            # `saved` stays: a second backward through a retained graph reaches the linked Linear again (ops.LinearFn.backward)
            stash, ctx.olink.result = ctx.olink.result, None
        part, dgb = None, None
        if stash is not None and _is_zero_placeholder(dy):
            # the LayerNorm backward already ran in the epilogue of the next layer's q|k|v input-gradient launch (LnOutLink)
            dx, da, part = stash
            stash = None
        else:
            dy2 = dy.reshape(-1, d).contiguous()
            dx = torch.empty_like(dy2)
            da = torch.empty((M, d), dtype=x16.dtype, device=dy.device)
            # LayerNorm backward: dx = skip-connection gradient, da = gradient of the FFN output (dropout mask regenerated)
            inplace = gg is not None and gb is not None and gb2 is not None
            desc = L.LnDesc(M, d, _code(da.dtype), eps, p_drop, off)
            if inplace and _wq['on'] and _in_backward():
                part = torch.empty((lib.otr_add_layernorm_bwd_partial_rows(M), 3 * d), dtype=torch.float32, device=dy.device)
            else:
                dgb = torch.zeros((3, d), dtype=torch.float32, device=dy.device)
            L.check(lib.otr_add_layernorm_bwd(C.byref(desc), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(seed), _p(dx), _p(da),
                                              _p(dgb[0]) if dgb is not None else None, _p(dgb[1]) if dgb is not None else None,
                                              _p(dgb[2]) if dgb is not None else None, _p(part), _stream()), 'otr_add_layernorm_bwd')
            if stash is not None:       # y had another consumer besides the linked Linear: add the part that Linear's launch produced
                dx.add_(stash[0])
                da = (da.float() + stash[1].float()).to(da.dtype)
                colsum_raw(stash[2][:, :d], out=gg)
                colsum_raw(stash[2][:, d:2 * d], out=gb)
                colsum_raw(stash[2][:, 2 * d:], out=gb2)
        ret_g = ret_b = ret_b2 = None
        if part is not None:
            colsum_raw(part[:, :d], out=gg)
            colsum_raw(part[:, d:2 * d], out=gb)
            colsum_raw(part[:, 2 * d:], out=gb2)
        else:
            if gg is not None:
                gg.add_(dgb[0])
            else:
                ret_g = dgb[0]
            if gb is not None:
                gb.add_(dgb[1])
            else:
                ret_b = dgb[1]
            if gb2 is not None:
                gb2.add_(dgb[2])
            else:
                ret_b2 = dgb[2]
        if ctx.split and hsave is not None:
            # 128-row workgroups, the hidden read back from the forward pass's tiles: dh for the w_1 weight gradient; dx += dh . w_1
            dh = torch.empty((usave.shape[0], 2 * F), dtype=x16.dtype, device=dy.device)[:M]
            if ctx.slab:
                bslabs = torch.empty((4, M, d), dtype=x16.dtype, device=dy.device)
                L.check(_timed('ffn_bwd_slab', {'flops': 6.0 * M * F * d, 'bytes': M * d * (2 + 8) + M * F * 8 + 6 * F * d},
                               lambda: lib.otr_ffn_bwd_split_slab(_p(da), _p(hsave), _p(P3), _p(P4), _p(dh), _p(bslabs), M, F, d,
                                                                  _stream())), 'otr_ffn_bwd_split_slab')
                il = ctx.ilink
                if il is not None and il.armed and il.result is None and _in_backward():
                    il.result = (dx, bslabs)          # see LnInLink: the attention sub-layer's closing launch sums them
                    _park(il)
                    dx = _zero_placeholder(dy.device, xshape)
                else:
                    tot = torch.empty_like(dx)
                    L.check(lib.otr_dec_sum(_p(dx), _p(bslabs), 4, M, _p(tot), _stream()), 'otr_dec_sum')
                    dx = tot
            else:
                nb = lib.otr_ffn_split_scratch_bytes(M)
                scratch = torch.empty(nb // 4, dtype=torch.float32, device=dy.device)
                sync = _ffn_sync(dy.device)
                L.check(_timed('ffn_bwd_split', {'flops': 6.0 * M * F * d, 'bytes': M * d * (2 + 4 + 4) + M * F * 8 + 6 * F * d},
                               lambda: lib.otr_ffn_bwd_split(_p(da), _p(hsave), _p(P3), _p(P4), _p(dh), _p(dx), _p(dx), _p(scratch), nb,
                                                             _p(sync), sync.numel(), M, F, d, _stream())), 'otr_ffn_bwd_split')
            gw1, gb1, gw2 = grad_target(w1p), grad_target(b1p), grad_target(w2p)
            dw1 = linear_wgrad_raw(dh, x16, None, out=gw1)
            dw2 = linear_wgrad_raw(da, usave[:M], None, out=gw2)
            db1 = colsum_raw(dh, out=gb1)               # rides along with the w_1 weight-gradient launch (same matrix)
            return (dx.view(xshape), None if gw1 is not None else dw1, None if gb1 is not None else db1,
                    None if gw2 is not None else dw2, ret_b2, ret_g, ret_b, None, None, None, None, None, None)
        # FFN backward with recompute: dh, u for the weight gradients; dx += dh . w_1
        dh = torch.empty((M, 2 * F), dtype=x16.dtype, device=dy.device)
        u = torch.empty((M, F), dtype=x16.dtype, device=dy.device)
        bpart = torch.empty(((M + 31) // 32, 2 * F), dtype=torch.float32, device=dy.device)    # d b_1 per 32-row block
        L.check(_timed('ffn_bwd', {'flops': 10.0 * M * F * d, 'bytes': M * d * (2 + 2 + 4 + 4) + M * F * 6 + 10 * F * d},
                       lambda: lib.otr_ffn_bwd(_p(x16), _p(da), _p(P1), _p(b1), _p(P3), _p(P4), _p(dh), _p(u), _p(bpart), _p(dx),
                                               _p(dx), M, F, d, _stream())), 'otr_ffn_bwd')
        gw1, gb1, gw2 = grad_target(w1p), grad_target(b1p), grad_target(w2p)
        dw1 = linear_wgrad_raw(dh, x16, None, out=gw1)
        dw2 = linear_wgrad_raw(da, u, None, out=gw2)
        db1 = colsum_raw(bpart, out=gb1)             # per-workgroup column sums of dh, written by the backward kernel
        return (dx.view(xshape), None if gw1 is not None else dw1, None if gb1 is not None else db1,
                None if gw2 is not None else dw2, ret_b2, ret_g, ret_b, None, None, None, None, None, None)


def ffn_add_layernorm(x, w1, b1, w2, b2, gamma, beta, p_drop=0.0, eps=1e-5, defer_ln=False):
    """Fused FFN sub-layer when it applies (GLU, d_model 256, a 16-bit twin of x, enough rows); else None.
    defer_ln: the caller guarantees that whoever reads the result next goes through ops.linear / ops.materialize (PendingLn)."""
    if lp_of(x) is None or x.dtype != torch.float32 or x.shape[-1] != 256:
        return None
    if x.numel() // 256 < _FUSED_FFN_MIN_ROWS:
        return None
    packs = ffn_packs(w1, w2)
    if packs is None or b1 is None or b2 is None:
        return None
    olink = LnOutLink() if (_LNOUT and torch.is_grad_enabled()) else None
    pend = PendingLn() if (defer_ln and _FFN_SLAB) else None
    y, y16 = FfnLnFn.apply(x, w1, b1, w2, b2, gamma, beta, float(p_drop), float(eps), packs, olink, pend,
                           getattr(x, '_otr_inlink', None))
    if olink is not None and olink.armed:
        y._otr_lnout = olink
    if pend is not None and not pend.done:
        y._otr_pending = pend
    return attach_lp(y, y16)


class GLUFn(torch.autograd.Function):
    """F.glu(h, -1) as its own autograd node (ffn_dropout > 0 puts a dropout between the GLU and w_2, module/ffn.py:40)"""

    @staticmethod
    def forward(ctx, h):
        _cuda(h)
        F2 = h.shape[-1]
        h2 = h.reshape(-1, F2).contiguous()
        u = torch.empty((h2.shape[0], F2 // 2), dtype=h2.dtype, device=h.device)
        L.check(L.load().otr_glu_fwd(_p(h2), _p(u), _code(h2.dtype), h2.shape[0], F2 // 2, None, _stream()), 'otr_glu_fwd')
        ctx.save_for_backward(h2)
        ctx.hshape = h.shape
        return u.view(*h.shape[:-1], F2 // 2)

    @staticmethod
    def backward(ctx, du):
        (h2,) = ctx.saved_tensors
        du2 = du.reshape(-1, du.shape[-1]).contiguous().to(h2.dtype)
        dh = torch.empty_like(h2)
        L.check(L.load().otr_glu_bwd(_p(h2), _p(du2), _p(dh), None, _code(h2.dtype), h2.shape[0], du2.shape[1], None, 0, _stream()),
                'otr_glu_bwd')
        return dh.view(ctx.hshape)


GLU_RPB = 32        # rows per workgroup of otr_glu_bwd (csrc/elementwise.hip)
_FUSED_GLU_BWD = True     # A/B switches for tuning runs
_FUSED_GLU_FWD = True


# ---------------------------------------------------------------------------------------- fused decoder stack
# The post-norm Transformer decoder on few rows (B x L = 480 at the AISHELL batch; decoder/transformer.py:47-90,161-183) as three
# launches per layer and direction, cut along (utterance group, head) / (row block, hidden slice) instead of along operators
# (csrc/declayer.hip).  A sub-layer leaves PARTIAL sums ("slabs") and the next launch finishes the LayerNorm in its prologue.
_DEC_FUSED = True
_DEC_TOUCH = True
_DEC_FFN_SLICES = 8
_EMBED_SINK = True     # A/B: the decoder stack's input gradient summed by the embedding's backward
DEC_LAYER_PARAMS = 18      # qvk w,b | out w,b | norm1 w,b | q w,b | out w,b | norm2 w,b | w_1 w,b | w_2 w,b | norm3 w,b


def dec_ffn_slices(F):
    """hidden slices of the fused decoder's FFN launches: the largest S <= OTR_DEC_FFN_SLICES with F % (128 S) == 0 (0: none)"""
    for s in range(min(_DEC_FFN_SLICES, 64), 0, -1):
        if F % (128 * s) == 0:
            return s
    return 0


def _dec_ln(xres=None, x16=None, slabs=None, nslab=0, bias=None, gamma=None, beta=None, seed=None, p_drop=0.0, eps=1e-5, off=0,
            y=None, y16=None, z=None, mean=None, rstd=None):
    pv = lambda t: t.data_ptr() if t is not None else None
    return L.DecLn(pv(xres), pv(x16), pv(slabs), nslab, pv(bias), pv(gamma), pv(beta), pv(seed), p_drop, eps, off,
                   pv(y), pv(y16), pv(z), pv(mean), pv(rstd))


def _dec_lnb(dskip, slabs, nslab, saved, gamma, seed, p_drop, dz, da16, partial):
    z, mean, rstd, off = saved
    pv = lambda t: t.data_ptr() if t is not None else None
    return L.DecLnB(pv(dskip), pv(slabs), nslab, pv(z), pv(mean), pv(rstd), pv(gamma), pv(seed), p_drop, off, pv(dz), pv(da16), pv(partial))


def _grad_w(dy2, x2, w):
    gt = grad_target(w)
    r = linear_wgrad_raw(dy2, x2, None, out=gt)
    return None if gt is not None else r


def _grad_b(a2, b):
    gt = grad_target(b)
    r = colsum_raw(a2, out=gt)
    return None if gt is not None else r


class DecoderStackFn(torch.autograd.Function):
    """TransformerDecoder.forward between the embedding and the output layer (decoder/transformer.py:172-176): n post-norm layers of
    [causal self-attention, cross-attention over the encoder memory, GLU feed-forward], each closed by dropout + residual + LayerNorm.
    x0 [B,L,256] f32 with its 16-bit twin; kv_all [B,T,n*512] 16-bit = every layer's keys | values of the memory (CrossKVAllFn)."""

    @staticmethod
    def forward(ctx, x0, kv_all, kmask, n_layers, p_drop, eps, S, *params):
        _cuda(x0, kv_all)
        ctx.set_materialize_grads(False)
        lib = L.load()
        B, Lq, d = x0.shape
        R, H = B * Lq, 4
        T, W = kv_all.shape[1], kv_all.shape[2]
        dev, hdt = x0.device, half_dtype()
        assert d == 256 and kv_all.is_contiguous() and kv_all.dtype == hdt and W == n_layers * 512 and len(params) == DEC_LAYER_PARAMS * n_layers
        xres, x16 = x0.reshape(R, d).contiguous(), lp_of(x0).reshape(R, d).contiguous()
        seed = rng_seed_tensor(dev) if p_drop > 0 else None
        f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        h16 = lambda *sh: torch.empty(sh, dtype=hdt, device=dev)
        slA, slB, slC = h16(4, R, d), h16(4, R, d), h16(S, R, d)       # partial sums travel 16-bit (csrc/declayer.hip: dl_store_slab)
        need = any(ctx.needs_input_grad)
        st = _stream()

        def ln_out(xres_, slabs, nslab, bias, gamma, beta):
            """descriptor + outputs of one LayerNorm finished in a prologue"""
            y, y16, z, mean, rstd = f32(R, d), h16(R, d), f32(R, d), f32(R), f32(R)
            off = _next_rng_offset(R * d) if p_drop > 0 else 0
            desc = _dec_ln(xres_, None, slabs, nslab, bias, gamma, beta, seed, p_drop, eps, off, y, y16, z, mean, rstd)
            return desc, y, y16, (z, mean, rstd, off)

        layers, packs_all = [], []
        y_in, y_in16, pending = xres, x16, None          # pending = (slabs, nslab, bias, gamma, beta) of the FFN sub-layer below
        F = params[12].shape[0] // 2
        if _DEC_TOUCH and need:
            # every packed weight of the stack (forward and input-gradient packs, ~40 MB) read once: the 36 launches of the forward and
            # backward pass then find them in the memory-side cache instead of HBM (csrc/elementwise.hip: otr_touch)
            rng = []
            for l in range(n_layers):
                pr = params[DEC_LAYER_PARAMS * l:DEC_LAYER_PARAMS * (l + 1)]
                for t in (*lin_packs(pr[0]), *lin_packs(pr[2]), *lin_packs(pr[6]), *lin_packs(pr[8]), *ffn_packs(pr[12], pr[14])):
                    rng.append((t.data_ptr(), t.numel() * t.element_size()))
            rng.sort()
            merged = [list(rng[0])]
            for a, n in rng[1:]:
                if a <= merged[-1][0] + merged[-1][1] + 4096:
                    merged[-1][1] = max(merged[-1][1], a + n - merged[-1][0])
                else:
                    merged.append([a, n])
            for a, n in merged:
                L.check(lib.otr_touch(C.c_void_p(a), n, st), 'otr_touch')
        for l in range(n_layers):
            (wqkv, bqkv, wo, bo, g1, be1, wq, bq, wo2, bo2, g2, be2, w1, b1, w2, b2, g3, be3) = params[DEC_LAYER_PARAMS * l:DEC_LAYER_PARAMS * (l + 1)]
            pk = (lin_packs(wqkv), lin_packs(wo), lin_packs(wq), lin_packs(wo2), ffn_packs(w1, w2))
            packs_all.append(pk)
            rec = {}
            if pending is None:
                lnA, y0, y016 = _dec_ln(None, x16, None, 0), xres, x16
            else:
                lnA, y0, y016, rec_prev = ln_out(y_in, *pending)
                layers[-1]['ln3'] = rec_prev
            qkv16, ctx1, lse1 = h16(R, 3 * d), h16(R, d), f32(B, H, Lq)
            fl_self = 2.0 * R * d * (3 * d + d) + 4.0 * R * Lq * d
            L.check(_timed('dec_self_fwd', {'flops': fl_self}, lambda lnA=lnA, pk=pk, bqkv=bqkv, qkv16=qkv16, ctx1=ctx1, lse1=lse1: lib.otr_dec_self_fwd(
                C.byref(lnA), B, Lq, _p(pk[0][0]), _p(bqkv), _p(pk[1][0]), _p(qkv16), _p(ctx1), _p(lse1), _p(slA), _stream())), 'otr_dec_self_fwd')
            lnB, y1, y116, rec['ln1'] = ln_out(y0, slA, 4, bo, g1, be1)
            q16, ctx2, lse2 = h16(R, d), h16(R, d), f32(B, H, Lq)
            fl_cross = 2.0 * R * d * (d + d) + 4.0 * R * T * d
            L.check(_timed('dec_cross_fwd', {'flops': fl_cross}, lambda lnB=lnB, pk=pk, bq=bq, q16=q16, ctx2=ctx2, lse2=lse2, l=l: lib.otr_dec_cross_fwd(
                C.byref(lnB), B, Lq, _p(pk[2][0]), _p(bq), _p(pk[3][0]), _p(kv_all), T * W, W, l * 512, l * 512 + 256, _p(kmask), T, _p(q16), _p(ctx2),
                _p(lse2), _p(slB), _stream())), 'otr_dec_cross_fwd')
            lnC, y2, y216, rec['ln2'] = ln_out(y1, slB, 4, bo2, g2, be2)
            hsave = torch.empty(lib.otr_dec_ffn_hsave_bytes(R, F) // 2, dtype=hdt, device=dev) if need else None
            L.check(_timed('dec_ffn_fwd', {'flops': 6.0 * R * F * d}, lambda lnC=lnC, pk=pk, b1=b1, hsave=hsave: lib.otr_dec_ffn_fwd(
                C.byref(lnC), R, _p(pk[4][0]), _p(b1), _p(pk[4][1]), F, S, _p(slC), _p(hsave), _stream())), 'otr_dec_ffn_fwd')
            rec.update(hsave=hsave, y016=y016, qkv16=qkv16, ctx1=ctx1, lse1=lse1, y116=y116, q16=q16, ctx2=ctx2, lse2=lse2, y216=y216)
            layers.append(rec)
            y_in, pending = y2, (slC, S, b2, g3, be3)
        lnF, y3, y316, rec_last = ln_out(y_in, *pending)
        layers[-1]['ln3'] = rec_last
        L.check(lib.otr_dec_ln(C.byref(lnF), R, st), 'otr_dec_ln')
        ctx.layers, ctx.packs, ctx.params = layers, packs_all, params
        ctx.cfg = (B, Lq, d, T, W, n_layers, p_drop, S, F, seed)
        ctx.kv, ctx.kmask = kv_all, kmask
        ctx.sink = getattr(x0, '_otr_embed_sink', None) if _EMBED_SINK else None
        y3, y316 = y3.view(B, Lq, d), y316.view(B, Lq, d)
        ctx.mark_non_differentiable(y316)
        return y3, y316

    @staticmethod
    def backward(ctx, dy, _dy16=None):
        n_in = 7
        if dy is None:
            return (None,) * (n_in + len(ctx.params))
        lib = L.load()
        B, Lq, d, T, W, n_layers, p_drop, S, F, seed = ctx.cfg
        R, G = B * Lq, max(1, lib.otr_dec_group_size(B, Lq))     # utterances per (group, head) workgroup: the library's choice
        ngrp, nblk = (B + G - 1) // G, (R + 31) // 32
        dev, hdt = dy.device, half_dtype()
        f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        h16 = lambda *sh: torch.empty(sh, dtype=hdt, device=dev)
        kv_all, kmask = ctx.kv, ctx.kmask
        dkv = torch.empty_like(kv_all) if ctx.needs_input_grad[1] else None
        st = _stream()
        dskip, slabs, nslab = dy.reshape(R, d).contiguous().float(), None, 0
        grads = [None] * len(ctx.params)
        for l in reversed(range(n_layers)):
            (wqkv, bqkv, wo, bo, g1, be1, wq, bq, wo2, bo2, g2, be2, w1, b1, w2, b2, g3, be3) = ctx.params[DEC_LAYER_PARAMS * l:DEC_LAYER_PARAMS * (l + 1)]
            pk, rec, o = ctx.packs[l], ctx.layers[l], DEC_LAYER_PARAMS * l
            # ---- FFN sub-layer
            dz3, da3, part3 = f32(R, d), h16(R, d), f32(nblk, 3 * d)
            dh, u, bpart, slCb = h16(R, 2 * F), h16(R, F), f32(nblk, 2 * F), h16(S, R, d)
            lnb = _dec_lnb(dskip, slabs, nslab, rec['ln3'], g3, seed, p_drop, dz3, da3, part3)
            _, _, P3, P4 = pk[4]
            L.check(_timed('dec_ffn_bwd', {'flops': 6.0 * R * F * d}, lambda lnb=lnb, rec=rec, P3=P3, P4=P4, dh=dh, u=u, bpart=bpart, slCb=slCb:
                           lib.otr_dec_ffn_bwd(C.byref(lnb), R, _p(rec['hsave']), _p(P3), _p(P4), F, S, _p(dh), _p(u), _p(bpart), _p(slCb), _stream())),
                    'otr_dec_ffn_bwd')
            grads[o + 16], grads[o + 17], grads[o + 15] = _grad_b(part3[:, :d], g3), _grad_b(part3[:, d:2 * d], be3), _grad_b(part3[:, 2 * d:], b2)
            grads[o + 12], grads[o + 13], grads[o + 14] = _grad_w(dh, rec['y216'], w1), _grad_b(bpart, b1), _grad_w(da3, u, w2)
            # ---- cross-attention sub-layer
            dz2, da2, part2, dq16, slBb = f32(R, d), h16(R, d), f32(ngrp, 3 * d), h16(R, d), h16(4, R, d)
            if dkv is None:
                dkv = torch.empty_like(kv_all)
            lnb = _dec_lnb(dz3, slCb, S, rec['ln2'], g2, seed, p_drop, dz2, da2, part2)
            L.check(_timed('dec_cross_bwd', {'flops': 2.0 * R * d * (d + d) + 10.0 * R * T * d},
                           lambda lnb=lnb, pk=pk, rec=rec, l=l, dq16=dq16, slBb=slBb: lib.otr_dec_cross_bwd(
                               C.byref(lnb), B, Lq, _p(pk[3][1]), _p(pk[2][1]), _p(rec['q16']), _p(rec['ctx2']), _p(rec['lse2']), _p(kv_all), _p(dkv),
                               T * W, W, l * 512, l * 512 + 256, _p(kmask), T, _p(dq16), _p(slBb), _stream())), 'otr_dec_cross_bwd')
            grads[o + 10], grads[o + 11], grads[o + 9] = _grad_b(part2[:, :d], g2), _grad_b(part2[:, d:2 * d], be2), _grad_b(part2[:, 2 * d:], bo2)
            grads[o + 8], grads[o + 6], grads[o + 7] = _grad_w(da2, rec['ctx2'], wo2), _grad_w(dq16, rec['y116'], wq), _grad_b(dq16, bq)
            # ---- self-attention sub-layer
            dz1, da1, part1, dqkv16, slAb = f32(R, d), h16(R, d), f32(ngrp, 3 * d), h16(R, 3 * d), h16(4, R, d)
            lnb = _dec_lnb(dz2, slBb, 4, rec['ln1'], g1, seed, p_drop, dz1, da1, part1)
            L.check(_timed('dec_self_bwd', {'flops': 2.0 * R * d * (3 * d + d) + 10.0 * R * Lq * d},
                           lambda lnb=lnb, pk=pk, rec=rec, dqkv16=dqkv16, slAb=slAb: lib.otr_dec_self_bwd(
                               C.byref(lnb), B, Lq, _p(pk[1][1]), _p(pk[0][1]), _p(rec['qkv16']), _p(rec['ctx1']), _p(rec['lse1']), _p(dqkv16),
                               _p(slAb), _stream())), 'otr_dec_self_bwd')
            grads[o + 4], grads[o + 5], grads[o + 3] = _grad_b(part1[:, :d], g1), _grad_b(part1[:, d:2 * d], be1), _grad_b(part1[:, 2 * d:], bo)
            grads[o + 2], grads[o + 0], grads[o + 1] = _grad_w(da1, rec['ctx1'], wo), _grad_w(dqkv16, rec['y016'], wqkv), _grad_b(dqkv16, bqkv)
            dskip, slabs, nslab = dz1, slAb, 4
        dx0 = None
        if ctx.needs_input_grad[0]:
            sink = ctx.sink
            if sink is not None and sink.buf is None and _in_backward() and slabs is not None:
                # x0 is the embedding's output and nothing else reads its gradient: the embedding's backward adds skip + slabs itself
                sink.buf = (dskip, slabs, nslab)
                _park(sink)
                dx0 = _zero_placeholder(dev, (B, Lq, d))
            else:
                dx0 = f32(R, d)
                L.check(lib.otr_dec_sum(_p(dskip), _p(slabs), nslab, R, _p(dx0), st), 'otr_dec_sum')
                dx0 = dx0.view(B, Lq, d)
        return (dx0, dkv if ctx.needs_input_grad[1] else None, None, None, None, None, None, *grads)


def decoder_stack_applies(x0, memory, blocks, normalize_before):
    """S (the FFN launches' hidden slices) when the fused decoder stack can run this forward pass, else 0"""
    if not _DEC_FUSED or not is_half() or not x0.is_cuda or normalize_before or lp_of(x0) is None or x0.dim() != 3:
        return 0
    B, Lq, d = x0.shape
    if d != 256 or Lq > 32 or Lq < 1 or len(blocks) < 1:
        return 0
    F = None
    for b in blocks:
        sa, ca, ff = b.slf_attn, b.src_attn, b.feed_forward
        if (b.concat_after or b.normalize_before or sa.nheads != 4 or ca.nheads != 4 or sa.share_qvk_proj or ca.share_vk_proj
                or ff.activation != 'glu' or (b.training and (sa.dropout_rate or ca.dropout_rate or ff.dropout))):
            return 0
        if ff.w_1.bias is None or ff.w_2.bias is None or tuple(ff.w_2.weight.shape) != (256, ff.w_1.weight.shape[0] // 2):
            return 0
        if F is None:
            F = ff.w_2.weight.shape[1]
        if ff.w_2.weight.shape[1] != F or tuple(ca.vk_proj.weight.shape) != (512, memory.shape[-1]):
            return 0
        for w in (sa.qvk_proj.weight, sa.output_proj.weight, ca.q_proj.weight, ca.output_proj.weight):
            if lin_packs(w) is None:
                return 0
        if ffn_packs(ff.w_1.weight, ff.w_2.weight) is None:
            return 0
    return dec_ffn_slices(F)


def decoder_stack(x0, kv_all, kmask_u8, blocks, S):
    p = blocks[0].residual_dropout if blocks[0].training else 0.0
    params = []
    for b in blocks:
        sa, ca, ff = b.slf_attn, b.src_attn, b.feed_forward
        params += [sa.qvk_proj.weight, sa.qvk_proj.bias, sa.output_proj.weight, sa.output_proj.bias, b.norm1.weight, b.norm1.bias,
                   ca.q_proj.weight, ca.q_proj.bias, ca.output_proj.weight, ca.output_proj.bias, b.norm2.weight, b.norm2.bias,
                   ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias, b.norm3.weight, b.norm3.bias]
    y, y16 = DecoderStackFn.apply(x0, kv_all, kmask_u8, len(blocks), float(p), float(blocks[0].norm1.eps), S, *params)
    return attach_lp(y, y16)


# ---------------------------------------------------------------------------------------- positional encoding
class _Grad16Link:
    """The gradient of a Linear's fp32 OUTPUT handed to that Linear's backward as a 16-bit tensor, outside autograd's own bookkeeping
    (which would cast it back to the output's dtype).  The Linear in front of the encoder's positional encoding (frontend/conv.py:146)
    reads its output gradient only as a GEMM operand -- dx, dw and db are all products with it -- so PosEncFn.backward writes
    sqrt(d) dy straight in the 16-bit operand type, parks it here and returns a stride-0 zero placeholder; LinearFn.backward picks it
    up.  With 16-bit operands the weight gradient joins the 256-wide grouped launch instead of a split-K GEMM of its own."""
    buf = None
    armed = False


_G16 = True
_G16_LOSS = True      # A/B: the loss launch's gradient as a 16-bit operand of the output layer


class PosEncFn(torch.autograd.Function):
    """x*sqrt(d) + PE (module/pos.py:44-57, scale_learnable=False)."""

    @staticmethod
    def forward(ctx, x, mask=None):
        """mask: optional [B, T] bool / uint8 key mask (any strides): its uint8 cast leaves the same launch and is remembered on the
        mask tensor, where ops._mask_u8 finds it (the encoder's attention launches and the decoder's memory mask read that)"""
        _cuda(x)
        ctx.set_materialize_grads(False)
        B, T, d = x.shape
        x = x.contiguous().float()
        y = torch.empty_like(x)
        ylp = torch.empty(x.shape, dtype=half_dtype(), device=x.device) if is_half() else None
        ctx.scale = math.sqrt(d)
        ctx.g16 = getattr(x, '_otr_g16', None)         # x is the output of a LinearFn that takes its gradient as a 16-bit operand
        u8 = None
        if (mask is not None and d % 4 == 0 and mask.is_cuda and mask.dim() == 2 and tuple(mask.shape) == (B, T)
                and mask.dtype in (torch.bool, torch.uint8) and getattr(mask, '_otr_u8', (None,))[0] != mask._version):
            u8 = torch.empty((B, T), dtype=torch.uint8, device=x.device)
            L.check(L.load().otr_posenc_mask_fwd(_p(x), _p(y), _p(ylp), B * T, T, d, ctx.scale, _p(mask), mask.stride(0), mask.stride(1),
                                                 _p(u8), _stream()), 'otr_posenc_mask_fwd')
            mask._otr_u8 = (mask._version, u8)
        else:
            L.check(L.load().otr_posenc_fwd(_p(x), _p(y), _p(ylp), B * T, T, d, ctx.scale, _stream()), 'otr_posenc_fwd')
        if ylp is not None:
            ctx.mark_non_differentiable(ylp)
        return y, ylp

    @staticmethod
    def backward(ctx, dy, _dylp=None):
        if dy is None:
            return None, None
        dy = dy.contiguous()
        link = ctx.g16
        if (link is not None and link.armed and link.buf is None and dy.dtype == torch.float32 and is_half() and _in_backward()
                and dy.data_ptr() % 16 == 0):
            g16 = torch.empty(dy.shape, dtype=half_dtype(), device=dy.device)
            L.check(L.load().otr_scale_cast(_p(dy), _p(g16), dy.numel(), ctx.scale, _stream()), 'otr_scale_cast')
            link.buf = g16
            _park(link)
            return _zero_placeholder(dy.device, dy.shape), None
        dx = torch.empty_like(dy)
        L.check(L.load().otr_scale(_p(dy), _p(dx), dy.numel(), None, ctx.scale, _stream()), 'otr_scale')
        return dx, None


class _EmbedSink:
    """hand-over from the fused decoder stack to the embedding's backward: the stack's input gradient as (skip, slabs, nslab) -- the
    embedding adds the partial sums itself (otr_embed_bwd_ld) instead of a launch that only sums them (otr_dec_sum)"""
    buf = None


def _token_view(tokens):
    """(tokens as the kernels read them, row stride, L): a [B, L] view with unit inner stride is passed through (no copy launch)"""
    if tokens.dim() == 2 and tokens.stride(1) == 1 and tokens.stride(0) >= tokens.shape[1]:
        return tokens, tokens.stride(0), tokens.shape[1]
    tokens = tokens.contiguous()
    return tokens, tokens.shape[-1], tokens.shape[-1]


class EmbedPosEncFn(torch.autograd.Function):
    """embedding(tokens)*sqrt(d) + PE (decoder/transformer.py:163-169)."""

    @staticmethod
    def forward(ctx, tokens, E, sink=None):
        _cuda(tokens, E)
        ctx.set_materialize_grads(False)
        B, Lq = tokens.shape
        V, d = E.shape
        tokens, ldt, _ = _token_view(tokens)
        y = torch.empty((B, Lq, d), dtype=torch.float32, device=E.device)
        ylp = torch.empty((B, Lq, d), dtype=half_dtype(), device=E.device) if is_half() else None
        ctx.scale = math.sqrt(d)
        L.check(L.load().otr_embed_posenc_fwd_ld(_p(tokens), ldt, _p(E), _p(y), _p(ylp), B * Lq, Lq, d, V, ctx.scale, _stream()),
                'otr_embed_posenc_fwd_ld')
        ctx.save_for_backward(tokens)
        ctx.eshape, ctx.ldt, ctx.Lq = (V, d), ldt, Lq
        ctx.e_ref, ctx.sink = E, sink
        if ylp is not None:
            ctx.mark_non_differentiable(ylp)
        return y, ylp

    @staticmethod
    def backward(ctx, dy, _dylp=None):
        sink = ctx.sink
        parked = sink.buf if sink is not None else None
        if dy is None and parked is None:
            return None, None, None
        (tokens,) = ctx.saved_tensors
        V, d = ctx.eshape
        gt = grad_target(ctx.e_ref)
        dev = tokens.device
        dE = gt if gt is not None else torch.zeros((V, d), dtype=torch.float32, device=dev)
        if parked is not None:            # the decoder stack's input gradient arrives as skip + 16-bit partial sums (DecoderStackFn.backward)
            sink.buf = None
            dskip, slabs, nslab = parked
            L.check(L.load().otr_embed_bwd_ld(_p(tokens), ctx.ldt, ctx.Lq, _p(dskip), _p(slabs), nslab, _p(dE), tokens.shape[0] * ctx.Lq, d, V,
                                              ctx.scale, _stream()), 'otr_embed_bwd_ld')
        else:
            dy = dy.contiguous()
            L.check(L.load().otr_embed_bwd_ld(_p(tokens), ctx.ldt, ctx.Lq, _p(dy), None, 0, _p(dE), tokens.shape[0] * ctx.Lq, d, V, ctx.scale,
                                              _stream()), 'otr_embed_bwd_ld')
        return None, (None if gt is not None else dE), None


def posenc(x, mask=None):
    y, ylp = PosEncFn.apply(x, mask)
    return attach_lp(y, ylp)


def embed_posenc(tokens, E):
    sink = _EmbedSink()
    y, ylp = EmbedPosEncFn.apply(tokens, E, sink)
    y._otr_embed_sink = sink              # read by DecoderStackFn when y is its input (and only then)
    return attach_lp(y, ylp)


# ---------------------------------------------------------------------------------------- incremental decoding
def decode_embed(preds, pos, E):
    """y[r] = E[preds[r, *pos]]*sqrt(d) + PE[*pos]; pos is a DEVICE int32 scalar (hipGraph replay)."""
    _cuda(preds, pos, E)
    R, ldp = preds.shape
    V, d = E.shape
    y = torch.empty((R, d), dtype=torch.float32, device=E.device)
    ylp = torch.empty((R, d), dtype=half_dtype(), device=E.device) if is_half() else None
    L.check(L.load().otr_decode_embed(_p(preds), ldp, _p(pos), _p(E), _p(y), _p(ylp), R, d, V, math.sqrt(d), _stream()),
            'otr_decode_embed')
    return attach_lp(y, ylp)


def decode_lookup(preds, pos, E):
    """y[r] = E[preds[r, *pos]] (plain embedding rows of the hypotheses' last tokens; pos a DEVICE int32 scalar, or None = column 0)"""
    _cuda(preds, E)
    R, ldp = preds.shape[0], preds.stride(0)
    V, d = E.shape
    y = torch.empty((R, d), dtype=torch.float32, device=E.device)
    ylp = torch.empty((R, d), dtype=half_dtype(), device=E.device) if is_half() else None
    L.check(L.load().otr_decode_lookup(_p(preds), ldp, _p(pos), _p(E), _p(y), _p(ylp), R, d, V, _stream()), 'otr_decode_lookup')
    return attach_lp(y, ylp)


def lstm_cell(gates_a, gates_b, bias_b, c_prev):
    """one torch.nn.LSTM cell update on [R, 4H] gate pre-activations (see include/otrans_hip.h: otr_lstm_cell) -> (h with its 16-bit
    twin, c); no autograd (the recurrent LM is an inference-time component here)"""
    _cuda(gates_a)
    R, H4 = gates_a.shape
    H = H4 // 4
    ga = gates_a.contiguous().float()
    gb = gates_b.contiguous().float() if gates_b is not None else None
    h = torch.empty((R, H), dtype=torch.float32, device=ga.device)
    c = torch.empty((R, H), dtype=torch.float32, device=ga.device)
    hlp = torch.empty((R, H), dtype=half_dtype(), device=ga.device) if is_half() else None
    L.check(L.load().otr_lstm_cell(_p(ga), _p(gb), _p(bias_b.contiguous().float() if bias_b is not None else None),
                                   _p(c_prev.contiguous() if c_prev is not None else None), _p(h), _p(hlp), _p(c), R, H, _stream()),
            'otr_lstm_cell')
    return attach_lp(h, hlp), c


def decode_self_attention(qkv, kcache, vcache, anc, pos, n_heads):
    """New-position query against the ancestors' cached keys/values (include/otrans_hip.h)."""
    _cuda(qkv, kcache, vcache, anc, pos)
    R, d3 = qkv.shape
    d = d3 // 3
    dk = d // n_heads
    maxlen = kcache.shape[1]
    assert qkv.is_contiguous() and kcache.dtype == qkv.dtype and kcache.shape == (R, maxlen, d) == vcache.shape
    assert anc.shape == (R, maxlen) and anc.dtype == torch.int32
    out = torch.empty((R, d), dtype=qkv.dtype, device=qkv.device)
    L.check(L.load().otr_decode_self_attention(_p(qkv), _p(kcache), _p(vcache), _p(anc), _p(pos), _p(out), _code(qkv.dtype),
                                               R, n_heads, dk, maxlen, 1.0 / math.sqrt(dk), _stream()),
            'otr_decode_self_attention')
    return out


# ---------------------------------------------------------------------------------------- conv frontend
def conv_geometry(T, F):
    T1 = (T - 3) // 2 + 1
    T2 = (T1 - 3) // 2 + 1
    F1 = (F - 1) // 2 + 1
    F2 = (F1 - 1) // 2 + 1
    return T1, F1, T2, F2


_CONV2_WIDE = True                 # 256 -> 256 channels: conv2 input gradient on csrc/conv2wide.hip
_CONV2_IMPLICIT_DGRAD = True       # tests switch it off to compare with the explicit (column matrix) path


class ConvSubsampleFn(torch.autograd.Function):
    """Two Conv2dLayers of frontend/conv.py:141-142 (3x3, stride 2, pad (0,1), ReLU).

    x [B,T,F] f32; w1 [C1,1,3,3]; w2 [C2,C1,3,3] (reference layout; regrouped to channel-last taps
    [C2,3,3,C1] here, in the compute dtype).  Returns channel-last act2 viewed as [B, T2, F2*C2]
    (column index f*C2 + c)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p_drop=0.0):
        """p_drop > 0: each Conv2dLayer is dropout(relu(conv(x))) (frontend/conv.py:63-66).  The masks are applied in place
        on act1 / act2 (which are saved MASKED: their > 0 pattern is then relu' AND mask) and, in backward, on the incoming
        gradients with the same (seed, offset) -- zeros stay zero, kept elements pick up the 1/(1-p)."""
        _cuda(x, w1, b1, w2, b2)
        B, T, F = x.shape
        C1, C2 = w1.shape[0], w2.shape[0]
        T1, F1, T2, F2 = conv_geometry(T, F)
        adt = act_dtype()
        x = x.contiguous()
        w1_param = w1
        w1 = w1.contiguous()
        w2l = weight_lp(w2)
        w2r = regrouped_lp(w2, (C2, 9, C1))
        w2r = w2r.view(C2, 3, 3, C1) if w2r is not None else (w2l if w2l is not None else w2).view(C2, C1, 3, 3).permute(0, 2, 3, 1).contiguous()
        desc = L.ConvDesc(B, T, F, C1, C2, T1, F1, T2, F2, _code(adt), _compute_code(), _code(w2r.dtype))
        act1 = torch.empty((B, T1, F1, C1), dtype=adt, device=x.device)
        act2 = torch.empty((B, T2, F2 * C2), dtype=adt, device=x.device)
        lib = L.load()
        seed, offs = None, (0, 0)
        fused = 1
        if p_drop == 0 and b2 is not None and x.dtype == torch.float32 and w1.dtype == torch.float32:
            # both layers in one launch where the shapes are served (csrc/conv2fwd.hip); 1 = not served
            fused = lib.otr_conv12_fwd(C.byref(desc), _p(x), _p(w1), _p(b1), _p(act1), _p(w2r), _p(b2), _p(act2), _stream())
            if fused != 1:
                L.check(fused, 'otr_conv12_fwd')
        if fused == 1:
            L.check(lib.otr_conv1_fwd(C.byref(desc), _p(x), _p(w1), _p(b1), _p(act1), _stream()), 'otr_conv1_fwd')
            if p_drop > 0:
                seed = rng_seed_tensor(x.device)
                offs = (_next_rng_offset(act1.numel()), _next_rng_offset(act2.numel()))
                L.check(lib.otr_dropout(_p(act1), _p(act1), _code(adt), act1.numel(), p_drop, _p(seed), offs[0], _stream()), 'otr_dropout')
            L.check(lib.otr_conv2_fwd(C.byref(desc), _p(act1), _p(w2r), _p(b2), _p(act2), _stream()), 'otr_conv2_fwd')
        if p_drop > 0:
            L.check(lib.otr_dropout(_p(act2), _p(act2), _code(adt), act2.numel(), p_drop, _p(seed), offs[1], _stream()), 'otr_dropout')
        ctx.drop = (p_drop, seed, offs)
        ctx.save_for_backward(x, w2r, act1, act2)
        ctx.desc_args = (B, T, F, C1, C2, T1, F1, T2, F2)
        ctx.refs = (w1_param, b1, b2)
        ctx.w2_ref = w2
        return act2

    @staticmethod
    def backward(ctx, dact2):
        x, w2r, act1, act2 = ctx.saved_tensors
        B, T, F, C1, C2, T1, F1, T2, F2 = ctx.desc_args
        adt = act2.dtype
        desc = L.ConvDesc(B, T, F, C1, C2, T1, F1, T2, F2, _code(adt), _compute_code(), _code(w2r.dtype))
        lib = L.load()
        p_drop, seed, offs = ctx.drop
        dact2 = dact2.contiguous()
        if p_drop > 0:
            gm = torch.empty_like(dact2)
            L.check(lib.otr_dropout(_p(dact2), _p(gm), _code(adt), dact2.numel(), p_drop, _p(seed), offs[1], _stream()), 'otr_dropout')
            dact2 = gm
        M2 = B * T2 * F2
        w1p, b1p, b2p = ctx.refs
        gw1, gb1, gb2 = grad_target(w1p), grad_target(b1p), grad_target(b2p)
        fused = relu_bwd_colsum_raw(act2.view(M2, C2), dact2.view(M2, C2))       # ReLU mask + the bias-gradient partial sums
        if fused is not None:
            g2, db2 = fused[0].view(dact2.shape), colsum_raw(fused[1], out=gb2)
        else:
            g2 = relu_bwd_raw(act2, dact2)
            db2 = colsum_raw(g2.view(M2, C2), out=gb2)
        dw2r = torch.empty((C2, 3, 3, C1), dtype=torch.float32, device=x.device)
        dact1 = torch.empty_like(act1)
        # (r06, measured and removed: this weight gradient on a side stream -- a BRANCH of the captured step beside the input gradient
        #  and conv1's weight gradient, which are independent of it -- made the step 0.065 ms SLOWER (3.99 -> 4.06 ms).  Branches do run
        #  side by side on this stack, but ONE fork anywhere in a hipGraph takes the whole graph off the runtime's batched-packet path:
        #  200 trivial nodes 318 -> 621 us, 200 streaming nodes 442 -> 611 us; tools/ubench/boundary.hip, profiles/r06_boundary_probe.txt)
        L.check(lib.otr_conv2_wgrad(C.byref(desc), _p(g2), _p(act1), _p(dw2r), _p(_workspace(x.device)), _WS_BYTES, _stream()),
                'otr_conv2_wgrad')
        rc = 1
        if _CONV2_IMPLICIT_DGRAD and _CONV2_WIDE and C1 == 256 and C2 == 256:
            rc = lib.otr_conv2_dgrad_wide(C.byref(desc), _p(g2), _p(w2r), _p(act1), _p(dact1), _p(_workspace(x.device)), _WS_BYTES, _stream())
        if rc == 1:
            rc = lib.otr_conv2_dgrad(C.byref(desc), _p(g2), _p(w2r), _p(act1), _p(dact1), _stream()) if _CONV2_IMPLICIT_DGRAD else 1
        if rc == 1:                          # operands do not qualify for the implicit kernel: column matrix + col2im
            dcol = torch.empty((M2, 9 * C1), dtype=adt, device=x.device)
            L.check(lib.otr_conv2_dgrad_cols(C.byref(desc), _p(g2), _p(w2r), _p(dcol), _stream()), 'otr_conv2_dgrad_cols')
            L.check(lib.otr_conv2_col2im(C.byref(desc), _p(dcol), _p(act1), _p(dact1), _stream()), 'otr_conv2_col2im')
        else:
            L.check(rc, 'otr_conv2_dgrad')
        if p_drop > 0:
            L.check(lib.otr_dropout(_p(dact1), _p(dact1), _code(adt), dact1.numel(), p_drop, _p(seed), offs[0], _stream()), 'otr_dropout')
        if gw1 is not None and gb1 is not None:
            dw1, db1 = gw1, gb1
        else:
            dwb = torch.zeros((C1 * 10,), dtype=torch.float32, device=x.device)
            dw1, db1 = dwb[:C1 * 9], dwb[C1 * 9:]
        nrow = lib.otr_conv1_wgrad_partial_rows()
        part = torch.empty((nrow, C1 * 10), dtype=torch.float32, device=x.device)      # per-workgroup sums: no atomics
        L.check(lib.otr_conv1_wgrad(C.byref(desc), _p(x), _p(dact1), None, None, _p(part), _stream()), 'otr_conv1_wgrad')
        inpl = gw1 is not None and gb1 is not None
        colsum_raw(part[:, :C1 * 9], out=dw1.view(-1), defer=inpl)
        colsum_raw(part[:, C1 * 9:], out=db1, defer=inpl)
        gw2 = grad_target(ctx.w2_ref)
        if gw2 is not None and gw2.is_contiguous() and gw2.dtype == torch.float32:
            regroup_add(gw2, dw2r, C2, C1, 9, False)                 # [C2, 9, C1] -> += [C2, C1, 3, 3] in one launch (it was AccumulateGrad's strided add)
            dw2 = None
        else:
            dw2 = dw2r.permute(0, 3, 1, 2)
        return (None, None if inpl else dw1.view(C1, 1, 3, 3), None if inpl else db1, dw2, None if gb2 is not None else db2, None)


def conv_subsample_with_dropout(x, w1, b1, w2, b2, p_drop):
    return ConvSubsampleFn.apply(x, w1, b1, w2, b2, float(p_drop))


# ---------------------------------------------------------------------------------------- conformer pieces
def _gemm_ptr(kind, M, N, K, x, w, y, bias=None, accumulate=0):
    """Raw GEMM on (tensor, element offset, leading dim) triples -- for head-sliced operands.
    kind: 'fwd' y[M,N] = x[M,K] w[N,K]^T; 'dgrad' x[M,K] = y[M,N] w[N,K]; 'wgrad' w[N,K] = y[M,N]^T x[M,K]."""
    (xt, xo, ldx), (wt, wo, ldw), (yt, yo, ldy) = x, w, y
    d = L.LinearDesc(M, N, K, _code(xt.dtype), _code(wt.dtype), _code(yt.dtype), _compute_code(), ldx, ldw, ldy, 0,
                     accumulate)
    ws = _workspace(xt.device)
    lib = L.load()
    if kind == 'fwd':
        L.check(lib.otr_linear_fwd(C.byref(d), _p(xt, xo), _p(wt, wo), _p(bias), _p(yt, yo), _p(ws), _WS_BYTES, _stream()),
                'otr_linear_fwd')
    elif kind == 'dgrad':
        L.check(lib.otr_linear_dgrad(C.byref(d), _p(yt, yo), _p(wt, wo), _p(xt, xo), _p(ws), _WS_BYTES, _stream()),
                'otr_linear_dgrad')
    else:
        L.check(lib.otr_linear_wgrad(C.byref(d), _p(yt, yo), _p(xt, xo), _p(wt, wo), _p(ws), _WS_BYTES, _stream()),
                'otr_linear_wgrad')


_GEMM_BATCHED = True
_BN_PART = True          # ConformerConvFn: BatchNorm batch statistics through per-workgroup sums
_DW_PART = True      # ConformerConvFn: depthwise-conv parameter gradients through per-workgroup sums
_DY16_WIDE = True           # LinearFn.backward: cast an fp32 gradient of the frontend's output layer to 16 bits once
_ADD2_COLSUM = True         # RelPosAttentionFn.backward: d(q+u) + d(q+v) and the two column sums in one pass
_CONV_MID_FUSED = True      # ConformerConvFn.backward: BatchNorm apply + depthwise conv + GLU backward in one launch
_POS_DEFER = True      # RelPosAttentionFn: the per-head dp products join the grouped weight-gradient launch


def _gemm_heads(M, N, K, x, w, y, nb, bsx, bsw, bsy):
    """y_b[M,N] = x_b[M,K] w_b[N,K]^T for b < nb on (tensor, element offset, leading dim) triples whose heads are bs* elements
    apart: ONE launch (otr_linear_fwd_batched) where the operands qualify, else nb launches of otr_linear_fwd."""
    (xt, xo, ldx), (wt, wo, ldw), (yt, yo, ldy) = x, w, y
    if _GEMM_BATCHED and nb > 1:
        d = L.LinearDesc(M, N, K, _code(xt.dtype), _code(wt.dtype), _code(yt.dtype), _compute_code(), ldx, ldw, ldy, 0, 0)
        rc = L.load().otr_linear_fwd_batched(C.byref(d), _p(xt, xo), _p(wt, wo), _p(yt, yo), nb, bsx, bsw, bsy, _stream())
        if rc == 0:
            return
        if rc < 0:
            L.check(rc, 'otr_linear_fwd_batched')
    for b in range(nb):
        _gemm_ptr('fwd', M, N, K, (xt, xo + b * bsx, ldx), (wt, wo + b * bsw, ldw), (yt, yo + b * bsy, ldy))


class ResidualAddFn(torch.autograd.Function):
    """y = x + scale * dropout(a): the pre-norm residual branches of encoder/conformer.py:50-73."""

    @staticmethod
    def forward(ctx, x, a, scale, p_drop, link=None):
        _cuda(x, a)
        # PreNormLink: x's gradient through the skip connection is handed to the LayerNorm at the head of the branch
        ctx.link = link if (link is not None and link.armed and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]) else None
        x2 = x.contiguous()
        a2 = a.contiguous()
        y = torch.empty_like(x2)
        seed = rng_seed_tensor(x.device) if p_drop > 0 else None
        off = _next_rng_offset(x2.numel()) if p_drop > 0 else 0
        L.check(L.load().otr_residual_add_fwd(_p(x2), _p(a2), _code(a2.dtype), _p(y), x2.numel(), scale, p_drop, _p(seed),
                                              off, _stream()), 'otr_residual_add_fwd')
        ctx.cfg = (scale, p_drop, off, a2.dtype, a.shape)
        ctx.save_for_backward(seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        scale, p_drop, off, adt, ashape = ctx.cfg
        (seed,) = ctx.saved_tensors
        dy = dy.contiguous()
        da = torch.empty(ashape, dtype=adt, device=dy.device)
        L.check(L.load().otr_residual_add_bwd(_p(dy), _p(da), _code(adt), dy.numel(), scale, p_drop, _p(seed), off,
                                              _stream()), 'otr_residual_add_bwd')
        if ctx.link is not None:
            ctx.link.buf = dy.view(-1, dy.shape[-1])
            _park(ctx.link)
            return None, da, None, None, None
        return dy, da, None, None, None


def residual_add(x, a, scale=1.0, p_drop=0.0, link=None):
    return ResidualAddFn.apply(x, a, float(scale), float(p_drop), link)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) on the HIP path (otr_dropout): the mask is regenerated from (seed, offset) in the backward pass"""

    @staticmethod
    def forward(ctx, x, p_drop):
        _cuda(x)
        x2 = x.contiguous()
        y = torch.empty_like(x2)
        seed = rng_seed_tensor(x.device)
        off = _next_rng_offset(x2.numel())
        L.check(L.load().otr_dropout(_p(x2), _p(y), _code(x2.dtype), x2.numel(), p_drop, _p(seed), off, _stream()), 'otr_dropout')
        ctx.save_for_backward(seed)
        ctx.cfg = (p_drop, off)
        return y

    @staticmethod
    def backward(ctx, dy):
        (seed,) = ctx.saved_tensors
        p_drop, off = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.check(L.load().otr_dropout(_p(dy), _p(dx), _code(dy.dtype), dy.numel(), p_drop, _p(seed), off, _stream()), 'otr_dropout')
        return dx, None


def dropout(x, p, training=True):
    """F.dropout(x, p, training) through the library's counter RNG; identity for p == 0 or eval"""
    if p <= 0.0 or not training or x.numel() == 0:
        return x
    if x.numel() % 4:
        raise L.OtransHipError('dropout: tensor size must be a multiple of 4')
    return DropoutFn.apply(x, float(p))


_PADDED_ROWS = {}


def _zero_padded_rows(t2, rows):
    """t2 [r, c] -> [rows, c] with zero rows appended; cached by storage (constant tables only)"""
    if t2.shape[0] == rows:
        return t2
    key = (t2.data_ptr(), t2._version, tuple(t2.shape), rows, t2.dtype, str(t2.device))
    hit = _PADDED_ROWS.get(key)
    if hit is None:
        if len(_PADDED_ROWS) > 16:
            _PADDED_ROWS.clear()
        out = torch.zeros((rows, t2.shape[1]), dtype=t2.dtype, device=t2.device)
        out[:t2.shape[0]].copy_(t2)
        hit = _PADDED_ROWS[key] = (out, t2)              # (keeps the source alive: the key is its address)
    return hit[0]


def _dp_from_pool(Pp, d, device):
    """a zeroed fp32 [Pp, d] accumulator for RelPosAttentionFn's deferred dp products: slices of one buffer zeroed ONCE per backward pass
    (16 blocks' worth; the pool belongs to the pass -- flush_weight_grads drops it -- and a second one is made if it runs out)"""
    task = torch._C._current_graph_task_id()
    pool = _wq.get('dp_pool')
    if pool is None or pool[0] != task or pool[2] >= pool[1].shape[0] or tuple(pool[1].shape[1:]) != (Pp, d) or pool[1].device != device:
        pool = _wq['dp_pool'] = [task, torch.zeros((16, Pp, d), dtype=torch.float32, device=device), 0]
    i = pool[2]
    pool[2] += 1
    return pool[1][i]


_DBD_PERSIST = True        # tests flip it to compare with a fresh tensor per backward pass


_DBD16 = True              # RelPosAttentionFn: the score term's gradient tensor in the 16-bit type (16-bit modes)


def _persistent_dbd(owner, like, dtype=None):
    """The gradient tensor of the relative-position score term, [B, T, H, Pp] fp32 (64 MB per Conformer block at the bench batch).  Only its
    band (column j - i + T - 1 of row i) is ever non-zero, and the attention backward rewrites EVERY in-range band entry (masked pairs
    with 0), so the tensor is zeroed ONCE per (layer, shape) and kept: the per-step zero fill was 10.5 us x 12 blocks.
    The buffer lives ON the layer's pos_proj weight (`owner._otr_dbd`), so it dies with the model; only the LAST shape is kept (a
    ragged batch with a new T replaces it: no stale 64 MB buffers pinned per layer); and a second request for the same layer inside
    ONE backward pass (a weight-shared layer applied twice: the deferred dp products of the first application still read the
    buffer at flush time) gets a fresh tensor instead of the shared one."""
    dtype = dtype or like.dtype
    if not _DBD_PERSIST:
        return torch.zeros_like(like, dtype=dtype)
    task = torch._C._current_graph_task_id()
    st = getattr(owner, '_otr_dbd', None)
    if st is not None and st['buf'].shape == like.shape and st['buf'].device == like.device and st['buf'].dtype == dtype:
        if task != -1 and st['task'] == task:
            return torch.zeros_like(like, dtype=dtype)
        st['task'] = task
        return st['buf']
    buf = torch.zeros_like(like, dtype=dtype)
    owner._otr_dbd = {'buf': buf, 'task': task}
    return buf


_PE16 = {}


def relpos_tables(pos_emb, weights):
    """p_i = pos_proj_i(sinusoid) and its transpose for EVERY layer of a Conformer stack (module/attention.py:217-221 computes them layer by
    layer: a [2T-1, d] x [d, d] GEMM and a transposing copy in each of 12 blocks, ~15 us of launches per block for a table that depends on
    nothing but the weights) in TWO batched launches: p_i = pe W_i^T and pt_i = W_i pe^T (the transpose is the same product with the
    operands swapped).  The layers' 16-bit weight shadows must sit at one constant stride (FlatDataParallel lays a stack's blocks out
    that way); anything else returns None and every layer computes its own.  Returns [(pe padded, p_i [Pp, d], pt_i [d, Pp])]."""
    if not is_half() or len(weights) < 2 or not pos_emb.is_cuda:
        return None
    wl = [weight_lp(w) for w in weights]
    if any(t is None or not t.is_contiguous() or t.shape != wl[0].shape or t.dtype != half_dtype() for t in wl):
        return None
    d = wl[0].shape[1]
    if wl[0].shape[0] != d:
        return None
    es = wl[0].element_size()
    step = wl[1].data_ptr() - wl[0].data_ptr()
    if step <= 0 or step % 16 != 0 or any(wl[i + 1].data_ptr() - wl[i].data_ptr() != step for i in range(len(wl) - 1)):
        return None
    P = pos_emb.numel() // d
    Pp = (P + 7) // 8 * 8
    pe = _zero_padded_rows(pos_emb.reshape(P, d).contiguous(), Pp)          # fp32 constant table, padded once per (T, d)
    key = (pe.data_ptr(), pe._version, half_dtype())
    hit = _PE16.get(key)
    if hit is None:
        if len(_PE16) > 8:
            _PE16.clear()
        hit = _PE16[key] = (pe.to(half_dtype()), pe)                       # its 16-bit twin (what the GEMM's loader makes of it), once
    pe16 = hit[0]
    n = len(wl)
    p_all = torch.empty((n, Pp, d), dtype=half_dtype(), device=pe.device)
    pt_all = torch.empty((n, d, Pp), dtype=half_dtype(), device=pe.device)
    _gemm_heads(Pp, d, d, (pe, 0, d), (wl[0], 0, d), (p_all, 0, d), n, 0, step // es, Pp * d)        # p_i = pe . W_i^T
    _gemm_heads(d, Pp, d, (wl[0], 0, d), (pe16, 0, d), (pt_all, 0, Pp), n, step // es, 0, d * Pp)    # pt_i = W_i . pe^T
    return [(pe, p_all[i], pt_all[i]) for i in range(n)]


class RelPosAttentionFn(torch.autograd.Function):
    """MultiHeadedSelfAttentionWithRelPos.forward after the qvk projection (module/attention.py:217-253):
    softmax(((q+u) k^T + shift((q+v) p^T)) / sqrt(dk)) v with p = pos_proj(sinusoid[-(T-1)..T-1]).
    The shifted [B,h,T,T] matrix is never built: the attention kernels read the un-shifted
    (q+v) p^T term at column j - i + T - 1."""

    @staticmethod
    def forward(ctx, qkv, pos_emb, pos_w, posu, posv, key_mask_u8, n_heads, tables=None):
        """tables: (pe padded [Pp, d], p [Pp, d], pt [d, Pp]) of this layer, precomputed for all layers of the stack in two batched launches
        (relpos_tables) -- or None: computed here"""
        _cuda(qkv, pos_emb, pos_w, posu, posv)
        B, T, d3 = qkv.shape
        d = d3 // 3
        H, dk, P, M = n_heads, d // n_heads, 2 * T - 1, B * T
        adt = qkv.dtype
        qkv = qkv.contiguous()
        pe = pos_emb.reshape(P, d).contiguous()
        wl = weight_lp(pos_w)
        # the relative-position axis (2T-1, odd) is padded to a multiple of 8 everywhere: every per-head GEMM then has
        # 16-byte aligned operands / outputs and whole contraction chunks, i.e. takes the fast kernels (the unpadded
        # layout ran the generic bounds-checked ones: 96 launches of 33-42 us per step)
        Pp = (P + 7) // 8 * 8
        if tables is not None and tables[1].dtype == adt and tuple(tables[1].shape) == (Pp, d):
            pe, p, pt = tables
        else:
            pe = _zero_padded_rows(pe, Pp)               # the sinusoid table is a constant: padded once per (T, d)
            p = linear_fwd_raw(pe, wl if wl is not None else pos_w, None, adt)              # [Pp, d], rows >= P are zero
            pt = p.t().contiguous()                                                         # [d, Pp]: dgrad as a forward GEMM
        u, v = posu.reshape(d).contiguous(), posv.reshape(d).contiguous()
        quv = torch.empty((B, T, 2 * d), dtype=adt, device=qkv.device)
        lib = L.load()
        L.check(lib.otr_head_bias_add(_p(qkv), d3, _p(u), _p(v), _p(quv), _code(adt), M, d, _stream()), 'otr_head_bias_add')
        bd = torch.empty((B, T, H, Pp), dtype=torch.float32, device=qkv.device)
        _gemm_heads(M, Pp, dk, (quv, d, 2 * d), (p, 0, d), (bd, 0, H * Pp), H, dk, dk, Pp)     # bd_h = (q+v)_h p_h^T
        out = torch.empty((B, T, d), dtype=adt, device=qkv.device)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
        desc = _attn_desc(B, H, T, T, dk, adt, (T * 2 * d, 2 * d), (T * d3, d3), (T * d3, d3), (T * d, d), False)
        L.check(lib.otr_attention_bias_fwd(C.byref(desc), _p(quv), _p(qkv, d), _p(qkv, 2 * d), _p(key_mask_u8), _p(bd),
                                           T * H * Pp, Pp, H * Pp, 1, _p(out), _p(lse), _stream()), 'otr_attention_bias_fwd')
        ctx.save_for_backward(qkv, quv, pt, pe, bd, out, lse, key_mask_u8, pos_w)
        ctx.H = H
        ctx.uv_refs = (posu, posv)                     # in-place / deferred parameter gradients (grad_target)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, quv, pt, pe, bd, out, lse, km, pos_w = ctx.saved_tensors
        H = ctx.H
        B, T, d3 = qkv.shape
        d = d3 // 3
        dk, P, M = d // H, 2 * T - 1, B * T
        Pp = pt.shape[1]
        adt = qkv.dtype
        lib = L.load()
        dout = dout.contiguous()
        # r06: the score term's gradient travels 16-bit in the 16-bit modes (it is written once by the attention backward and read by two
        # GEMMs: 64 MB per block in fp32); the forward tensor bd stays fp32 (it is part of the logits' arithmetic)
        dbd = _persistent_dbd(pos_w, bd, half_dtype() if (_DBD16 and adt == half_dtype()) else torch.float32)
        dquv = torch.empty_like(quv)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        desc = _attn_desc(B, H, T, T, dk, adt, (T * 2 * d, 2 * d), (T * d3, d3), (T * d3, d3), (T * d, d), False)
        L.check(lib.otr_attention_bias_bwd(C.byref(desc), _p(quv), _p(qkv, d), _p(qkv, 2 * d), _p(km), _p(bd), _p(dbd), _code(dbd.dtype),
                                           T * H * Pp, Pp, H * Pp, 1, _p(out), _p(dout), _p(lse), _p(delta), _p(dquv),
                                           _p(dqkv, d), _p(dqkv, 2 * d), _stream()), 'otr_attention_bias_bwd')
        # d(q+v)_h = dbd_h . p_h  as forward-type GEMMs on the transposed p (contraction over the padded axis), the four heads in one launch
        _gemm_heads(M, dk, Pp, (dbd, 0, H * Pp), (pt, 0, Pp), (dquv, d, 2 * d), H, Pp, dk * Pp, dk)
        pu, pv = ctx.uv_refs
        gu, gv, gw = grad_target(pu), grad_target(pv), grad_target(pos_w)
        # dp_h = dbd_h^T (q+v)_h feeds nothing but pos_proj's weight gradient: with an in-place gradient buffer the four products join
        # the grouped weight-gradient launch at the end of backward (they were 4 x [split-K GEMM + reduce] = 84 us per block in the
        # chain), and dW_pos = dp^T pe follows it (flush_weight_grads: post)
        dbd2, quv2 = dbd.view(M, H * Pp), quv.view(M, 2 * d)
        n0 = len(_wq['w'])
        if gw is not None and _wq['on'] and _in_backward() and _POS_DEFER:
            dp = _dp_from_pool(Pp, d, qkv.device)       # zeros; one fill per backward pass for all blocks, not one per block
            for h in range(H):
                linear_wgrad_raw(dbd2[:, h * Pp:(h + 1) * Pp], quv2[:, d + h * dk:d + (h + 1) * dk], None, out=dp[:, h * dk:(h + 1) * dk])
        deferred = len(_wq['w']) == n0 + H
        if not deferred:
            assert len(_wq['w']) == n0
            dp = torch.empty((Pp, d), dtype=torch.float32, device=qkv.device)
            for h in range(H):
                _gemm_ptr('wgrad', M, Pp, dk, (quv, d + h * dk, 2 * d), (dp, h * dk, d), (dbd, h * Pp, H * Pp))
        dq2 = dquv.view(M, 2 * d)
        inpl_uv = gu is not None and gv is not None and gu.is_contiguous() and gv.is_contiguous()
        if inpl_uv and _ADD2_COLSUM:
            # r06: the sum and the two column sums in one pass (otr_add2_strided_colsum): per-workgroup partials [M / 32][2 d] join the
            # grouped column sums instead of the two [M, d] operands (147 MB per step re-read)
            part = torch.empty((lib.otr_add2_colsum_partial_rows(M), 2 * d), dtype=torch.float32, device=qkv.device)
            L.check(lib.otr_add2_strided_colsum(_p(dquv), 2 * d, _p(dquv, d), 2 * d, _p(dqkv), d3, _code(adt), M, d, _p(part), _stream()),
                    'otr_add2_strided_colsum')
            colsum_raw(part[:, :d], out=gu.view(-1))
            colsum_raw(part[:, d:], out=gv.view(-1))
            du = dv = None
        else:
            L.check(lib.otr_add2_strided(_p(dquv), 2 * d, _p(dquv, d), 2 * d, _p(dqkv), d3, _code(adt), M, d, _stream()),
                    'otr_add2_strided')
        if inpl_uv and _ADD2_COLSUM:
            pass
        elif inpl_uv:
            # the two column sums join the grouped launch at the end of backward (they were 2 launches + 2 gradient adds per block)
            colsum_raw(dq2[:, :d], out=gu.view(-1))
            colsum_raw(dq2[:, d:], out=gv.view(-1))
            du = dv = None
        else:
            du, dv = colsum_raw(dq2[:, :d]).view(1, 1, H, dk), colsum_raw(dq2[:, d:]).view(1, 1, H, dk)
        if deferred:
            _wq['post'].append(lambda: linear_wgrad_raw(dp, pe, pos_w, out=gw))      # queued again: one more grouped launch for all blocks
            dw = None
        else:
            dw = linear_wgrad_raw(dp, pe, pos_w, out=gw)     # contraction over the padded axis (zero rows): the fast grouped kernel takes it
        return dqkv, None, None if gw is not None else dw, du, dv, None, None, None


class ConformerConvFn(torch.autograd.Function):
    """ConformerConvolutionModule.forward (module/conformer.py:36-57): Linear C->2C, GLU, zero padded
    frames, depthwise Conv1d, BatchNorm1d (batch statistics incl. padded frames), swish, Linear C->C,
    zero padded frames."""

    @staticmethod
    def forward(ctx, x, mask_u8, w1, b1, wdw, bdw, gamma, beta, run_mean, run_var, w2, b2, training, eps, momentum, mask_out=True):
        _cuda(x, w1, wdw, gamma, beta, w2)
        B, T, Cc = x.shape
        M = B * T
        adt = act_dtype()
        lib = L.load()
        xc = lp_of(x)
        x2 = _rows(xc if xc is not None else x)
        w1l, w2l = weight_lp(w1), weight_lp(w2)
        ctx.refs = (w1, b1, w2, b2)                    # in-place / deferred weight gradients (grad_target)
        ctx.bn_refs = (gamma, beta)
        ctx.dw_refs = (wdw, bdw)
        ctx.w1t, ctx.w2t = weight_lpt(w1), weight_lpt(w2)
        w1c = w1l if w1l is not None else w1
        w2c = w2l if w2l is not None else w2
        h = linear_fwd_raw(x2, w1c, b1, adt)                                            # [M, 2C]
        g = torch.empty((M, Cc), dtype=adt, device=x.device)
        L.check(lib.otr_glu_fwd(_p(h), _p(g), _code(adt), M, Cc, _p(mask_u8), _stream()), 'otr_glu_fwd')
        k = wdw.shape[-1]
        wk = wdw.reshape(Cc, k).contiguous()
        y = torch.empty((M, Cc), dtype=torch.float32, device=x.device)
        saved = torch.empty((2 * Cc,), dtype=torch.float32, device=x.device)
        s = torch.empty((M, Cc), dtype=adt, device=x.device)
        if training and _BN_PART:
            # batch statistics through per-workgroup sums: no zeroing launch, no atomics (otr_dwconv_fwd_part / otr_bn_swish_fwd_part)
            nblk = lib.otr_dwconv_fwd_partial_rows(M)
            spart = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=x.device)
            L.check(lib.otr_dwconv_fwd_part(_p(g), _code(adt), _p(wk), _p(bdw), _p(y), _p(spart), B, T, Cc, k, (k - 1) // 2, _stream()),
                    'otr_dwconv_fwd_part')
            L.check(lib.otr_bn_swish_fwd_part(_p(y), _p(spart), nblk, _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(saved), _p(s),
                                              _code(adt), M, Cc, eps, momentum, _stream()), 'otr_bn_swish_fwd_part')
        else:
            stats = torch.empty((2 * Cc,), dtype=torch.float32, device=x.device) if training else None
            L.check(lib.otr_dwconv_fwd(_p(g), _code(adt), _p(wk), _p(bdw), _p(y), _p(stats), B, T, Cc, k, (k - 1) // 2, _stream()),
                    'otr_dwconv_fwd')
            L.check(lib.otr_bn_swish_fwd(_p(y), _p(stats), _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(saved), _p(s),
                                         _code(adt), M, Cc, eps, momentum, int(training), _stream()), 'otr_bn_swish_fwd')
        # the branch leaves in the activation type (the residual add takes it as such): its gradient then arrives in that
        # type too, so the w_2 weight / bias gradients join the deferred 256-wide launch instead of an fp32-operand GEMM each
        o = linear_fwd_raw(s, w2c, b2, adt)
        # mask_out False: the consumer zeroes the padded frames' rows of this branch and of its gradient itself (ResidualLnFn a_mask:
        # the masked_fill of module/conformer.py:109 and its mirror in backward were a launch each)
        ctx.mask_out = bool(mask_out)
        if mask_out:
            out = torch.empty_like(o)
            L.check(lib.otr_row_mask_cast(_p(o), _code(adt), _p(mask_u8), _p(out), _code(adt), M, Cc, _stream()), 'otr_row_mask_cast')
        else:
            out = o
        ctx.save_for_backward(x2, mask_u8, w1c, wk, gamma, beta, w2c, h, g, y, saved, s)
        ctx.cfg = (B, T, Cc, k, training, x.shape, x.dtype, bdw is not None, wdw.shape)
        return out.view(B, T, Cc)

    @staticmethod
    def backward(ctx, dout):
        x2, mask_u8, w1c, wk, gamma, beta, w2c, h, g, y, saved, s = ctx.saved_tensors
        B, T, Cc, k, training, xshape, xdtype, has_dwb, wdw_shape = ctx.cfg
        M = B * T
        adt = s.dtype
        lib = L.load()
        dout = dout.contiguous()
        if not ctx.mask_out and dout.dtype == adt:
            dm = dout.view(M, Cc)                           # masked by the producer of this gradient (ResidualLnFn a_mask)
        else:
            dm = torch.empty((M, Cc), dtype=adt, device=dout.device)
            L.check(lib.otr_row_mask_cast(_p(dout), _code(dout.dtype), _p(mask_u8), _p(dm), _code(adt), M, Cc, _stream()), 'otr_row_mask_cast')
        w1p, b1p, w2p, b2p = ctx.refs
        gw1, gb1, gw2, gb2 = grad_target(w1p), grad_target(b1p), grad_target(w2p), grad_target(b2p)
        ds = linear_fwd_raw(dm, ctx.w2t, None, adt) if ctx.w2t is not None else linear_dgrad_raw(dm, w2c, adt)
        dw2 = linear_wgrad_raw(dm, s, w2c, out=gw2)
        db2 = colsum_raw(dm, out=gb2) if b2p is not None else None
        red = torch.empty((2 * Cc,), dtype=torch.float32, device=dout.device)
        bn_part = torch.empty((lib.otr_bn_swish_bwd_partial_rows(M), 2 * Cc), dtype=torch.float32, device=dout.device)
        gp, bp = ctx.bn_refs
        gg, gbt = grad_target(gp), grad_target(bp)
        bn_inplace = gg is not None and gbt is not None        # the reduction launch adds d gamma / d beta where they live
        wdwp, bdwp = ctx.dw_refs
        gwd, gbd = grad_target(wdwp), (grad_target(bdwp) if has_dwb else None)
        dw_inplace = gwd is not None and gwd.is_contiguous() and (not has_dwb or gbd is not None)   # the kernel's sums are += already
        if _CONV_MID_FUSED and dw_inplace and _DW_PART and _wq['on'] and _in_backward() and ds.dtype == adt and h.dtype == adt and g.dtype == adt:
            # r06: BatchNorm's apply step + the depthwise conv's backward + the GLU's backward in one launch (otr_conformer_conv_bwd_mid):
            # neither dy [M, C] fp32 nor dg [M, C] exists
            L.check(lib.otr_bn_swish_bwd_sums(_p(y), _p(ds), _code(adt), _p(saved), _p(gamma), _p(beta), _p(red), _p(bn_part),
                                              _p(gg) if bn_inplace else None, _p(gbt) if bn_inplace else None, M, Cc, _stream()),
                    'otr_bn_swish_bwd_sums')
            nrow = lib.otr_dwconv_bwd_partial_rows(M)
            dpart = torch.empty((nrow, Cc * k + Cc), dtype=torch.float32, device=dout.device)
            part = torch.empty((nrow, 2 * Cc), dtype=torch.float32, device=dout.device)
            dh = torch.empty_like(h)
            L.check(lib.otr_conformer_conv_bwd_mid(_p(y), _p(ds), _p(saved), _p(gamma), _p(beta), _p(red), _p(g), _p(wk), _p(h), _p(mask_u8),
                                                   _p(dh), _p(dpart), _p(part), _code(adt), int(training), B, T, Cc, k, (k - 1) // 2, _stream()),
                    'otr_conformer_conv_bwd_mid')
            colsum_raw(dpart[:, :Cc * k], out=gwd.view(-1))
            if has_dwb:
                colsum_raw(dpart[:, Cc * k:], out=gbd)
            db1 = colsum_raw(part, out=gb1) if b1p is not None else None
            dx = (linear_fwd_raw(dh, ctx.w1t, None, xdtype) if ctx.w1t is not None else linear_dgrad_raw(dh, w1c, xdtype)).view(xshape)
            dw1 = linear_wgrad_raw(dh, x2, w1c, out=gw1)
            return (dx, None, None if gw1 is not None else dw1, None if gb1 is not None else db1, None, None,
                    None if bn_inplace else red[Cc:], None if bn_inplace else red[:Cc], None, None,
                    None if gw2 is not None else dw2, None if gb2 is not None else db2, None, None, None, None)
        dy = torch.empty((M, Cc), dtype=torch.float32, device=dout.device)
        L.check(lib.otr_bn_swish_bwd(_p(y), _p(ds), _code(adt), _p(saved), _p(gamma), _p(beta), _p(red), _p(bn_part),
                                     _p(gg) if bn_inplace else None, _p(gbt) if bn_inplace else None, _p(dy), M, Cc,
                                     int(training), _stream()), 'otr_bn_swish_bwd')
        dg = torch.empty((M, Cc), dtype=adt, device=dout.device)
        dwk = None if dw_inplace else torch.zeros((Cc * k + Cc,), dtype=torch.float32, device=dout.device)
        if dw_inplace and _DW_PART and _wq['on'] and _in_backward():
            # the kernel leaves per-workgroup sums; the grouped column-sum launch at the end of backward adds them where the gradients
            # live (otr_dwconv_bwd_part: no atomics)
            dpart = torch.empty((lib.otr_dwconv_bwd_partial_rows(M), Cc * k + Cc), dtype=torch.float32, device=dout.device)
            L.check(lib.otr_dwconv_bwd_part(_p(dy), _p(g), _code(adt), _p(wk), _p(dg), _p(dpart), B, T, Cc, k, (k - 1) // 2, _stream()),
                    'otr_dwconv_bwd_part')
            colsum_raw(dpart[:, :Cc * k], out=gwd.view(-1))
            if has_dwb:
                colsum_raw(dpart[:, Cc * k:], out=gbd)
        else:
            L.check(lib.otr_dwconv_bwd(_p(dy), _p(g), _code(adt), _p(wk), _p(dg), _p(gwd) if dw_inplace else _p(dwk),
                                       (_p(gbd) if has_dwb else None) if dw_inplace else _p(dwk, Cc * k), B, T, Cc, k,
                                       (k - 1) // 2, _stream()), 'otr_dwconv_bwd')
        dh = torch.empty_like(h)
        nblk = (M + GLU_RPB - 1) // GLU_RPB
        part = torch.empty((nblk, 2 * Cc), dtype=torch.float32, device=dout.device)
        L.check(lib.otr_glu_bwd(_p(h), _p(dg), _p(dh), _p(part), _code(adt), M, Cc, _p(mask_u8), 0, _stream()), 'otr_glu_bwd')
        db1 = colsum_raw(part, out=gb1) if b1p is not None else None
        dx = (linear_fwd_raw(dh, ctx.w1t, None, xdtype) if ctx.w1t is not None else linear_dgrad_raw(dh, w1c, xdtype)).view(xshape)
        dw1 = linear_wgrad_raw(dh, x2, w1c, out=gw1)
        return (dx, None, None if gw1 is not None else dw1, None if gb1 is not None else db1,
                None if dw_inplace else dwk[:Cc * k].view(wdw_shape), dwk[Cc * k:] if (has_dwb and not dw_inplace) else None,
                None if bn_inplace else red[Cc:], None if bn_inplace else red[:Cc], None, None,
                None if gw2 is not None else dw2,
                None if gb2 is not None else db2, None, None, None, None)


class LookaheadConvFn(torch.autograd.Function):
    """The CTC head's look-ahead convolution (model/ctc.py:35-39): y[b,t,c] = sum_{j<=L} w[c,0,j] x[b,t+j,c], zeros past
    the end of the (padded) batch, no bias -- the depthwise-conv kernels with pad = 0."""

    @staticmethod
    def forward(ctx, x, w):
        _cuda(x, w)
        B, T, Cc = x.shape
        k = w.shape[-1]
        x2 = x.contiguous()
        wk = w.reshape(Cc, k).contiguous().float()
        y = torch.empty((B, T, Cc), dtype=torch.float32, device=x.device)
        L.check(L.load().otr_dwconv_fwd(_p(x2), _code(x2.dtype), _p(wk), None, _p(y), None, B, T, Cc, k, 0, _stream()),
                'otr_dwconv_fwd')
        ctx.save_for_backward(x2, wk)
        ctx.wshape = w.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, wk = ctx.saved_tensors
        B, T, Cc = x2.shape
        k = wk.shape[1]
        dy = dy.contiguous().float()
        dx = torch.empty_like(x2)
        dw = torch.zeros((Cc * k,), dtype=torch.float32, device=dy.device)
        L.check(L.load().otr_dwconv_bwd(_p(dy), _p(x2), _code(x2.dtype), _p(wk), _p(dx), _p(dw), None, B, T, Cc, k, 0,
                                        _stream()), 'otr_dwconv_bwd')
        return dx, dw.view(ctx.wshape)


# ---------------------------------------------------------------------------------------- losses
class LabelSmoothingLossFn(torch.autograd.Function):
    """LabelSmoothingLoss.forward (module/loss.py:21-48) + its gradient in the same pass."""

    @staticmethod
    def forward(ctx, logits, target, smoothing, pad_idx):
        _cuda(logits, target)
        V = logits.shape[-1]
        lg = logits.reshape(-1, V)
        # the head of a row-padded product (ops.padded_rows: rows of V8 = ceil8(V) floats) is read in place, and the gradient
        # leaves as the head of a zero-tailed [R, V8] buffer
        ld = lg.stride(0) if (lg.dim() == 2 and lg.shape[0] > 1 and lg.stride(1) == 1 and lg.dtype == torch.float32) else V
        if not (V < ld < V + 8 and ld % 8 == 0 and lg.data_ptr() % 16 == 0):
            lg, ld = lg.contiguous(), V
        R = lg.shape[0]
        tg = target.reshape(-1).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        dlogits = torch.empty((R, ld), dtype=torch.float32, device=lg.device) if ctx.needs_input_grad[0] else None
        scratch = torch.empty((R + 2,), dtype=torch.float32, device=lg.device)
        L.check(L.load().otr_label_smoothing_loss_ld(_p(lg), ld, _p(tg), R, V, smoothing, pad_idx, _p(loss), _p(dlogits), ld,
                                                     _p(scratch), _stream()), 'otr_label_smoothing_loss_ld')
        ctx.save_for_backward(dlogits)
        ctx.shape, ctx.V = logits.shape, V
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        out = torch.empty_like(dlogits)
        g = g.contiguous().float()
        L.check(L.load().otr_scale(_p(dlogits), _p(out), dlogits.numel(), _p(g), 1.0, _stream()), 'otr_scale')
        if out.shape[1] != ctx.V:
            out._otr_zero_tail = tuple(out.shape)                  # LinearFn.backward may read it at its full width (_zero_tail_of)
            out = out[:, :ctx.V]
        return out.view(ctx.shape), None, None, None


class LabelSmoothingLossFusedFn(torch.autograd.Function):
    """LabelSmoothingLoss.forward + its gradient + the scalar factor of the backward seed in ONE launch (otr_label_smoothing_loss_fused).
    Semantics: loss = LabelSmoothingLossFn(...), d loss / d logits multiplied by `gscale` (a device scalar; None = 1) -- i.e.
    ScaleGradFn o LabelSmoothingLossFn.  The kernel writes gscale x d loss / d logits; backward multiplies by the incoming gradient,
    unless that is the cached unit seed of ops.backward (then the buffer leaves as it is: no launch)."""

    @staticmethod
    def forward(ctx, logits, ld, target, ldt, Lt, V, smoothing, pad_idx, gscale, ticket):
        lg = logits.reshape(-1, V)              # a view (label_smoothing_loss checked the strides)
        R = lg.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        need = ctx.needs_input_grad[0]
        # the logits' producer takes its output gradient as a 16-bit operand (_Grad16Link: the row-padded output layer): the
        # gradient is written in that type, one rounding where every other branch gradient of the model already has one
        g16 = getattr(logits, '_otr_g16', None)
        ctx.g16 = g16 if (need and g16 is not None and g16.armed and is_half() and ld % 8 == 0) else None
        ddt = half_dtype() if ctx.g16 is not None else torch.float32
        dlogits = torch.empty((R, ld), dtype=ddt, device=lg.device) if need else None
        scratch = torch.empty((R + 2,), dtype=torch.float32, device=lg.device)
        L.check(L.load().otr_label_smoothing_loss_fused(_p(lg), ld, _p(target), ldt, Lt, R, V, smoothing, pad_idx, _p(gscale), _p(loss),
                                                        _p(dlogits), _code(ddt), ld, _p(scratch), _p(ticket), _stream()),
                'otr_label_smoothing_loss_fused')
        ctx.save_for_backward(dlogits)
        ctx.shape, ctx.V = logits.shape, V
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        seed = _state.get(('unit_grad', g.device, g.dtype))
        unit = seed is not None and g.data_ptr() == seed.data_ptr()
        if ctx.g16 is not None:
            link = ctx.g16
            if unit and link.buf is None and _in_backward():
                link.buf = dlogits                                 # [R, ld] 16-bit, zero behind column V: LinearFn.backward's operand
                _park(link)
                return (_zero_placeholder(dlogits.device, ctx.shape),) + (None,) * 9
            dlogits = dlogits.float()                              # a gradient other than the unit seed: the general (fp32) route
            unit = False if not unit else unit
        if unit:
            out = dlogits                                          # g == 1: the saved buffer IS the gradient
        else:
            out = torch.empty_like(dlogits)
            g = g.contiguous().float()
            L.check(L.load().otr_scale(_p(dlogits), _p(out), dlogits.numel(), _p(g), 1.0, _stream()), 'otr_scale')
        if out.shape[1] != ctx.V:
            out._otr_zero_tail = tuple(out.shape)                  # LinearFn.backward may read it at its full width (_zero_tail_of)
            out = out[:, :ctx.V]
        return (out.view(ctx.shape),) + (None,) * 9


_LS_FUSED = True


def _ls_ticket(device):
    """the arrival counter of otr_label_smoothing_loss_fused: one zeroed uint32 per device, allocated outside any capture"""
    t = _state.setdefault('ls_ticket', {}).get(device)
    if t is None and not torch.cuda.is_current_stream_capturing():
        t = _state['ls_ticket'][device] = torch.zeros(4, dtype=torch.int32, device=device)
    return t


def label_smoothing_loss(logits, target, smoothing, pad_idx, grad_scale=None):
    """LabelSmoothingLoss (module/loss.py:21-48) of logits [..., V] against target [...]; with grad_scale (a device scalar, the loss
    scale) the gradient that flows back into the logits carries that factor (= ops.ScaleGradFn applied to the loss).  One launch where
    the fused kernel's alignment rules hold (fp32 logits with 16-byte aligned rows -- the row-padded output layer's are -- V <= 8192,
    <= 8192 rows), the three-kernel form + ScaleGradFn otherwise."""
    V = logits.shape[-1]
    lg = logits.reshape(-1, V)
    ok = (_LS_FUSED and lg.is_cuda and lg.dtype == torch.float32 and lg.dim() == 2 and lg.stride(1) == 1 and V <= 8192 and 0 < lg.shape[0] <= 8192
          and lg.data_ptr() % 16 == 0 and target.dtype == torch.int64)
    if ok:
        ld = lg.stride(0) if lg.shape[0] > 1 else (V + 3) // 4 * 4
        ok = ld % 4 == 0 and V <= ld < V + 8
    ticket = _ls_ticket(lg.device) if ok else None
    if ok and ticket is not None:
        if target.dim() == 2 and target.stride(1) == 1 and target.stride(0) >= target.shape[1] and target.numel() == lg.shape[0]:
            tg, ldt, Lt = target, target.stride(0), target.shape[1]
        else:
            tg = target.reshape(-1).contiguous()
            ldt = Lt = tg.numel()
        if Lt <= 0x7fffffff and tg.numel() == lg.shape[0] and (lg.shape[0] == 1 or lg.shape[0] % Lt == 0):
            return LabelSmoothingLossFusedFn.apply(logits, ld, tg, ldt, Lt, V, smoothing, pad_idx, grad_scale, ticket)
    loss = LabelSmoothingLossFn.apply(logits, target, smoothing, pad_idx)
    return ScaleGradFn.apply(loss, grad_scale) if (grad_scale is not None and loss.requires_grad) else loss


class CTCLossFn(torch.autograd.Function):
    """CTCAssistor.compute_loss (model/ctc.py:50-53): log_softmax + CTC ('mean', zero_infinity)."""

    @staticmethod
    def forward(ctx, logits, targets, in_len, tgt_len, blank):
        _cuda(logits, targets, in_len, tgt_len)
        B, T, V = logits.shape
        lp = log_softmax(logits)
        tg = targets.contiguous()
        il = in_len.to(torch.int32).contiguous()
        tl = tgt_len.to(torch.int32).contiguous()
        max_tgt = tg.shape[1]
        ws = torch.empty((B, T, 2 * max_tgt + 1), dtype=torch.float32, device=logits.device)
        nll = torch.empty((B,), dtype=torch.float32, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(lp) if ctx.needs_input_grad[0] else None
        L.check(L.load().otr_ctc_loss(_p(lp), _p(tg), tg.stride(0), _p(il), _p(tl), B, T, V, max_tgt, blank, _p(ws),
                                      _p(nll), _p(loss), _p(dlogits), _stream()), 'otr_ctc_loss')
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        out = torch.empty_like(dlogits)
        g = g.contiguous().float()
        L.check(L.load().otr_scale(_p(dlogits), _p(out), dlogits.numel(), _p(g), 1.0, _stream()), 'otr_scale')
        return out, None, None, None, None


def log_softmax(x):
    """F.log_softmax(x, -1) for fp32 x (no autograd; decode / CTC inference paths)."""
    _cuda(x)
    V = x.shape[-1]
    x2 = x.reshape(-1, V).contiguous().float()
    y = torch.empty_like(x2)
    L.check(L.load().otr_log_softmax(_p(x2), _p(y), x2.shape[0], V, _stream()), 'otr_log_softmax')
    return y.view(x.shape)
