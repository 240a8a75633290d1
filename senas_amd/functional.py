"""torch.autograd.Function wrappers over the C ABI (include/senas_hip.h).

Tensors keep the reference's logical NCHW shape but live in ``torch.channels_last`` memory, i.e.
NHWC on the device -- that is what the kernels index.  torch is used for device memory, streams
and the autograd graph only; every pass over an activation tensor is a libsenas_hip kernel.

There is no CPU path: a non-CUDA tensor raises.
"""
import contextlib
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvGeom, SenasHipError
from .arena import zeros32, zeros64

CL = torch.channels_last
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t):
    if not t.is_cuda:
        raise SenasHipError('senas_amd runs on the GPU only (got a %s tensor); there is no CPU fallback' % t.device)
    if t.dtype != torch.float32:
        raise SenasHipError('senas_amd computes in float32 (got %s)' % t.dtype)
    return t


def nhwc(t):
    """float32 CUDA tensor, logical NCHW, physical NHWC, dense."""
    _dev(t)
    if t.dim() != 4:
        raise SenasHipError('expected a 4-d NCHW tensor, got shape %s' % (tuple(t.shape),))
    return t.contiguous(memory_format=CL)


def nhwc_stored(t):
    """nhwc() for a tensor that may be a bf16-STORED convolution output or its gradient (math mode 'bf16s'): float32 or
    bfloat16, CUDA, dense NHWC."""
    if t.dtype != torch.bfloat16:
        return nhwc(t)
    if not t.is_cuda:
        raise SenasHipError('senas_amd runs on the GPU only (got a %s tensor); there is no CPU fallback' % t.device)
    if t.dim() != 4:
        raise SenasHipError('expected a 4-d NCHW tensor, got shape %s' % (tuple(t.shape),))
    return t.contiguous(memory_format=CL)


def nhwc_slice(t):
    """(tensor, pixel stride in elements) for a kernel that can read a channel slice of a wider NHWC tensor in place: the
    slice itself when its layout is [n][h][w][wider c] with 16-byte alignment, else a dense NHWC copy.  A bf16 tensor (a
    bf16-stored convolution output, math mode 'bf16s') comes back dense with a NEGATIVE stride: the node kernels' convention."""
    if t.dtype == torch.bfloat16:
        t = nhwc_stored(t)
        return t, -t.shape[1]
    _dev(t)
    if t.dim() == 4:
        n, c, h, w = t.shape
        ct = t.stride(3)
        if ct > c and ct % 4 == 0 and c % 4 == 0 and t.stride() == (h * w * ct, 1, w * ct, ct) and t.data_ptr() % 16 == 0:
            return t, ct
    return nhwc(t), t.shape[1]


def new_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.float32, memory_format=CL)


def _p(t):
    return None if t is None else t.data_ptr()


# A STACKED convolution output / gradient (k edges x c channels each) travels between the autograd nodes of this package as a 5-D
# tensor [k][n][c][h][w] whose strides say which of two layouts the memory has:
#   planar      (k parts, each a dense NHWC tensor):   strides (n*h*w*c, h*w*c, 1, w*c, c)      -- senas_conv2d_fwd_planar
#   interleaved (one NHWC tensor of k*c channels):     strides (c, h*w*k*c, 1, w*k*c, k*c)      -- what every other kernel reads
def planar_empty(k, n, c, h, w, like):
    flat = torch.empty(k * n * c * h * w, device=like.device, dtype=torch.float32)
    return flat.as_strided((k, n, c, h, w), (n * h * w * c, h * w * c, 1, w * c, c))


def stacked_5d(t4, k):
    """The 5-D interleaved view of a dense NHWC tensor [n][k*c][h][w] (no copy)."""
    n, kc, h, w = t4.shape
    c = kc // k
    t4 = nhwc(t4)
    return t4.as_strided((k, n, c, h, w), (c, h * w * kc, 1, w * kc, kc), t4.storage_offset())


def stacked_4d(t5):
    """A 5-D stacked tensor as the dense NHWC tensor [n][k*c][h][w] the kernels read: a view when it is interleaved, else a copy."""
    k, n, c, h, w = t5.shape
    kc = k * c
    if t5.stride() == (c, h * w * kc, 1, w * kc, kc):
        return t5.as_strided((n, kc, h, w), (h * w * kc, 1, w * kc, kc), t5.storage_offset())
    return t5.permute(1, 0, 2, 3, 4).reshape(n, kc, h, w).contiguous(memory_format=CL)


def new_stats(n, c, like):
    return zeros64((n, c, 2), like.device)


class KernelTimer(object):
    """HIP-event timing of individual launches on the stream they are launched on (torch's current
    stream), used by bench.py for the roofline of the dominant kernel.  Off unless installed."""

    def __init__(self):
        self.records = []

    def span(self, name, flops, nbytes, tag=None):
        return _Span(self, name, flops, nbytes, tag)

    @staticmethod
    def empty_pair_ms(reps=200):
        """What a start/stop event pair reads with NOTHING between them (median): the marker-to-marker latency every
        span includes on top of its kernel."""
        pairs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in pairs)[reps // 2]

    def summary(self, overhead_ms=0.0):
        """Per kernel symbol: launches, total ms (each span less ``overhead_ms``, never below half its reading),
        algorithmic flops and bytes."""
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, e0, e1, _ in self.records:
            a = agg.setdefault(name, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            a['launches'] += 1
            raw = e0.elapsed_time(e1)
            a['ms'] += max(raw - overhead_ms, 0.5 * raw)
            a['flops'] += flops
            a['bytes'] += nbytes
        return agg


class _Span(object):
    def __init__(self, timer, name, flops, nbytes, tag=None):
        self.t, self.name, self.flops, self.nbytes, self.tag = timer, name, flops, nbytes, tag

    def __enter__(self):
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        self.e1.record()
        self.t.records.append((self.name, self.flops, self.nbytes, self.e0, self.e1, self.tag))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


TIMER = None
MAX_STACK = _lib.MAX_STACK
MAX_BNRELU = _lib.MAX_BNRELU
MAX_DSTAIL = _lib.MAX_DSTAIL
_NOSPAN = _NoSpan()

# (weight data_ptr, direction) -> packed fragment image kept fresh by the INSTALLED senas_amd.packing.WeightPacker;
# empty: every convolution call repacks its own weights
PACKED = {}
# Bumped whenever weights change behind torch's version counters (the fused SGD kernel writes them through raw
# pointers).  A packer's images / stacked buffers only count while the packer was refreshed in the current generation
# -- each packer carries its own stamp, so refreshing one packer never re-validates another one's stale images.
WEIGHT_GEN = 0


def weights_moved():
    global WEIGHT_GEN
    WEIGHT_GEN += 1


def _packed(w, direction):
    """Address of the cached fragment image of ``w``, or None.  An entry only counts while its packer is current, the
    parameter it was made for is alive and still lives at the address it is filed under (a freed model's addresses
    get reused by the next model's weights, and its images must not be) and torch has not seen it change since."""
    ent = PACKED.get((w.data_ptr(), direction))
    if ent is None or ent.packer.gen != WEIGHT_GEN:
        return None
    owner = ent.ref()
    if owner is None or owner.data_ptr() != w.data_ptr() or owner._version != ent.version:
        return None
    return ent.img.data_ptr()


# Arithmetic of the dense stride-1 convolutions that have a bf16-matrix-pipe form (csrc/conv_bf.hip): 0 = fp32 MFMA (bit-for-bit
# fp32 FMA chains, the default), else the number of bf16 products per fp32 product: 1 "bf16" (operands rounded to bf16),
# 3 "bf16x3", 6 "bf16x6" (operands split into 2 / 3 bf16 planes; fp32 accumulation throughout).  Tensors in HBM stay fp32.
MATH_TERMS = 0
MATH_NAMES = {'f32': 0, 'bf16': 1, 'bf16x3': 3, 'bf16x6': 6, 'bf16s': 1}
# 'bf16s': plain bf16 products (as 'bf16') AND the outputs of those convolutions -- and the gradients that come back for them --
# STORED as bf16 tensors: the convolution writes bf16 (statistics from its fp32 accumulators), the cell node reads bf16 terms and
# writes bf16 term gradients, the data- and weight-gradient kernels stage that gradient by a copy.  States / node outputs, their
# gradients, weights and weight gradients stay fp32.
MATH_STORED = False


def set_math(name):
    """Select the arithmetic of the dense convolutions: 'f32' | 'bf16x6' | 'bf16x3' | 'bf16' | 'bf16s'.  Returns the previous
    name.  Step drivers / packers built before the call keep the images of the mode they were built in: set it first."""
    global MATH_TERMS, MATH_STORED
    if name not in MATH_NAMES:
        raise SenasHipError('unknown math mode %r (one of %s)' % (name, sorted(MATH_NAMES)))
    prev = math_name()
    MATH_TERMS = MATH_NAMES[name]
    MATH_STORED = name == 'bf16s'
    return prev


def math_name():
    if MATH_STORED:
        return 'bf16s'
    return next(k for k, v in MATH_NAMES.items() if v == MATH_TERMS and k != 'bf16s')


def _packed_lp(w, direction):
    """The cached bf16 image of ``w`` for the current math mode (directions 2 + direction in the packer's table)."""
    return _packed(w, (MATH_TERMS, direction))


# senas_amd.gradsink.GradSink of the running step driver, or None: parameter gradients go through autograd
SINK = None


# Between GradSink.begin() and .finish(): a list that collects the second stage of two-stage weight gradients
# ((senas_sum_item, workspace kept alive)); finish() folds them all in a few launches.  None: every call sums at once.
DEFER = None


def flush_deferred(reopen=False):
    """Run the deferred weight-gradient sums (senas_wgrad_sum_batched, up to 64 per launch) and release their partials.
    ``reopen``: keep the deferral window open afterwards (a flush in the middle of a backward pass)."""
    global DEFER
    items, DEFER = DEFER, ([] if (reopen and DEFER is not None) else None)
    if not items:
        return
    L = _lib.lib()
    for i in range(0, len(items), _lib.MAX_SUMS):
        chunk = items[i:i + _lib.MAX_SUMS]
        arr = (_lib.SumItem * len(chunk))(*[it for it, _ in chunk])
        _lib.check(L.senas_wgrad_sum_batched(arr, len(chunk), _stream()), 'senas_wgrad_sum_batched')


# HIP streams that carry part of the pass in flight besides the caller's: the macro grid runs every column of up cells on a
# stream of its own (senas_amd.grid.Lanes), and autograd replays every node's backward pass on the stream its forward pass ran
# on.  Whatever reads a table that kernels of several lanes ADD into (the mixing-weight gradients, the gamma table, the flat
# gradient buffer at the end of backward) joins them first.
LANES = set()
SKIP_MAX = _lib.SKIP_MAX  # tensors one skip_stack launch takes (deeper columns fall back to blends + torch.cat)
MIX_SLOTS = 32          # rows of a d loss / d M table: one per cell of a kind within a pass (cells of different lanes never share one)


_OWN_STREAMS = {}       # (device index, name) -> torch.cuda.ExternalStream over a hipStream_t of this library


def own_stream(device, name):
    """A HIP stream that is nobody else's, by (device, name): made once per process by senas_stream_create and wrapped for torch.
    torch.cuda.Stream() hands out a pool of 32 streams round-robin -- after a few dozen step drivers a "new" stream IS one of the
    lanes, and the capture's star topology (grid.Lanes) silently turns into lane-to-lane waits."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, name)
    st = _OWN_STREAMS.get(key)
    if st is None:
        handle = C.c_void_p()
        with torch.cuda.device(index):
            _lib.check(_lib.lib().senas_stream_create(C.byref(handle)), 'senas_stream_create')
        st = _OWN_STREAMS[key] = torch.cuda.ExternalStream(handle.value, device=torch.device('cuda', index))
    return st


def join_lanes():
    """Make the current stream wait for everything launched so far on the lanes of the running pass (the queued weight
    gradients are launched first)."""
    flush_wgrads()
    if not LANES:
        return
    cur = torch.cuda.current_stream()
    capturing = torch.cuda.is_current_stream_capturing()
    for s in LANES:
        if s == cur or s.device != cur.device:
            continue
        if capturing:
            # a lane that is not part of this capture holds nothing of this pass (and waiting for it would tie the graph to
            # work outside the capture)
            with torch.cuda.stream(s):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        cur.wait_stream(s)


# A stream beside the pass for weight-gradient kernels whose results go straight into the flat gradient buffer (set by a step
# driver while its pass runs, else None).  Nothing downstream of a convolution's backward pass reads d loss / d w before the end
# of the pass, but on one stream every weight-gradient launch sits between a data gradient and its consumer: a third of the
# launches on the backward critical path of a search cell.  They are QUEUED on the host instead and launched on the lane at the
# end of the cell's backward pass (one stream wait per cell: a wait per kernel costs more than it buys -- measured).
WLANE = None
_WQ = []            # (closure, tensors it reads) of the weight-gradient launches waiting for the next flush


def run_wgrad(autograd_grads, tensors, fn):
    """Run ``fn`` (a weight-gradient launch) now, or -- if there is a weight-gradient lane and every destination is a view of
    the flat gradient buffer (``autograd_grads`` all None: autograd is handed nothing, nobody reads the result before
    GradSink.finish() has joined the lanes) -- queue it for the lane."""
    if WLANE is None or any(g is not None for g in autograd_grads):
        fn()
    else:
        _WQ.append((fn, [t for t in tensors if t is not None and t.is_cuda]))


def flush_wgrads(inline=False):
    """Launch the queued weight gradients on the lane, behind what the current stream holds so far.  ``inline``: on the
    current stream itself (whatever it launches next is then ordered behind them)."""
    if not _WQ:
        return
    q = list(_WQ)
    del _WQ[:]
    W = WLANE
    cur = torch.cuda.current_stream()
    if inline or W is None or W.device != cur.device:
        for fn, _ in q:
            fn()
        return
    W.wait_stream(cur)
    LANES.add(W)
    with torch.cuda.stream(W):
        for fn, tensors in q:
            for t in tensors:
                t.record_stream(W)           # (the caching allocator must not hand the block on while the lane reads it)
            fn()
        if DEFER:
            # the second stage of this batch's two-stage weight gradients right behind them, on the lane: at the end of the
            # pass the batched sums (0.6 ms per search step) would sit on the critical path
            flush_deferred(reopen=True)


class _CellIn(torch.autograd.Function):
    """Identity on a cell's input; on the way back it marks the end of the cell's backward pass: the weight gradients the cell
    queued are launched on the weight-gradient lane."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        flush_wgrads()
        return g


def cell_in(x):
    if WLANE is None or not (torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled()):
        return x
    y = _CellIn.apply(x)
    st = getattr(x, '_senas_stats', None)
    if st is not None:
        y._senas_stats = st
    return y


def wgrad_dest(w):
    """Where a backward kernel writes d loss / d w, and what autograd is handed for it: the parameter's view in the
    flat gradient buffer and None (first gradient of the pass under a step driver), or a fresh tensor twice."""
    if SINK is not None:
        v = SINK.dest(w)
        if v is not None:
            return v, None
        if SINK.pending(w):
            # a further gradient of a parameter within one pass (a module applied twice, e.g. the shared head under deep
            # supervision): autograd ADDS this one into the view, the first one's kernel OVERWRITES it -- so the first one must
            # have run by then.  It may still sit in the weight-gradient lane's queue (the queue is then launched on the CURRENT
            # stream: a lane that waited for the weight-gradient lane, which waits for the lanes, would close a cycle of waits
            # between two non-origin streams of a capture -- grid.Lanes) or have its second stage deferred (that sum runs now).
            if _WQ:
                flush_wgrads(inline=True)
            if DEFER:
                flush_deferred(reopen=True)
    t = torch.empty_like(w)
    return t, t


def may_defer(*autograd_grads):
    """A two-stage weight gradient may leave its sum to flush_deferred() only when every destination is a view of the
    flat gradient buffer (autograd is handed None): a fresh tensor goes to autograd at once and must be complete."""
    return DEFER is not None and all(g is None for g in autograd_grads)


def _span(kind, g, x, w, y, problems=1):
    """HIP-event span of one convolution launch (bench.py's roofline probe).  ``problems``: 2 for a pair launch -- one kernel
    instance working on two problems of this geometry: twice the flops and bytes under the same kernel symbol."""
    if TIMER is None:
        return _NOSPAN
    macs = problems * (g.n * g.hi * g.wi * g.ci * (g.co // g.groups) if g.transposed else g.n * g.ho * g.wo * g.co * (g.ci // g.groups)) * g.kh * g.kw
    which = {'conv_fwd': 0, 'conv_dgrad': 1, 'conv_wgrad': 2}[kind]
    name = b''
    if MATH_TERMS:
        name = _lib.lib().senas_conv2d_kernel_name_lp(C.byref(g), which, MATH_TERMS)
    name = (name or _lib.lib().senas_conv2d_kernel_name(C.byref(g), which)).decode()      # the symbol rocprofv3 reports
    tag = (kind,) + tuple(getattr(g, f) for f, _ in g._fields_) + (('pair',) if problems > 1 else ())
    return TIMER.span(name, 2.0 * macs, 4.0 * problems * (x.numel() + y.numel() + w.numel()), tag)


def _unspan():
    """Drop the span just recorded: the entry point inside it declined without launching."""
    if TIMER is not None and TIMER.records:
        TIMER.records.pop()


# ------------------------------------------------------------------------------------------ convolution
def conv_out_size(i, k, stride, pad, dil, transposed, out_pad):
    if transposed:
        return (i - 1) * stride - 2 * pad + dil * (k - 1) + out_pad + 1
    return (i + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _conv_wgrad(g, x, in_relu, dy, w, dest=None):
    """d loss / d w of one convolution call (two-stage kernels leave their sum to flush_deferred when they may): what autograd
    is handed for w -- None when the kernel wrote into the parameter's view of the flat gradient buffer.  ``dest``: a
    destination already taken from wgrad_dest (which must be asked once per gradient)."""
    L = _lib.lib()
    dwt, dw = wgrad_dest(w) if dest is None else dest
    nbytes, zero = C.c_int64(), C.c_int32()
    if dy.dtype == torch.bfloat16:                  # 'bf16s': the bf16-pipe form with the gradient operand staged by a copy
        _lib.check(L.senas_conv2d_bwd_weight_ws_lp(C.byref(g), 1, C.byref(nbytes)), 'senas_conv2d_bwd_weight_ws_lp')
        if not nbytes.value:
            raise SenasHipError('conv2d: a bf16-stored gradient for a geometry the bf16-pipe weight-gradient kernel does not serve')
        wsw = torch.empty(nbytes.value, device=x.device, dtype=torch.uint8)
        with _span('conv_wgrad', g, x, w, dy):
            item = _lib.SumItem() if may_defer(dw) else None
            _lib.check(L.senas_conv2d_bwd_weight_bf16s(C.byref(g), x.data_ptr(), in_relu, dy.data_ptr(), dwt.data_ptr(), wsw.data_ptr(),
                                                       C.byref(item) if item is not None else None, _stream()), 'senas_conv2d_bwd_weight_bf16s')
            if item is not None and item.kind:
                DEFER.append((item, wsw))
        return dw
    if MATH_TERMS:                                  # the bf16-pipe form, where the geometry has one
        _lib.check(L.senas_conv2d_bwd_weight_ws_lp(C.byref(g), MATH_TERMS, C.byref(nbytes)), 'senas_conv2d_bwd_weight_ws_lp')
    if MATH_TERMS and nbytes.value:
        wsw = torch.empty(nbytes.value, device=x.device, dtype=torch.uint8)
        with _span('conv_wgrad', g, x, w, dy):
            item = _lib.SumItem() if may_defer(dw) else None
            _lib.check(L.senas_conv2d_bwd_weight_lp(C.byref(g), x.data_ptr(), in_relu, dy.data_ptr(), dwt.data_ptr(), wsw.data_ptr(),
                                                    MATH_TERMS, C.byref(item) if item is not None else None, _stream()),
                       'senas_conv2d_bwd_weight_lp')
            if item is not None and item.kind:
                DEFER.append((item, wsw))
        return dw
    _lib.check(L.senas_conv2d_bwd_weight_ws(C.byref(g), C.byref(nbytes), C.byref(zero)), 'senas_conv2d_bwd_weight_ws')
    # pre-zeroed arena slice where the path accumulates with atomics (no memset launch per conv); plain
    # scratch where it writes per-block partials
    wsw = zeros32(nbytes.value // 4 + 1, x.device) if zero.value else torch.empty(nbytes.value, device=x.device, dtype=torch.uint8)
    with _span('conv_wgrad', g, x, w, dy):
        if may_defer(dw):
            item = _lib.SumItem()
            _lib.check(L.senas_conv2d_bwd_weight_deferred(C.byref(g), x.data_ptr(), in_relu, dy.data_ptr(), dwt.data_ptr(),
                                                          wsw.data_ptr(), int(zero.value), C.byref(item), _stream()),
                       'senas_conv2d_bwd_weight_deferred')
            if item.kind:
                DEFER.append((item, wsw))             # the partial images stay alive until the batched sum has run
        else:
            _lib.check(L.senas_conv2d_bwd_weight(C.byref(g), x.data_ptr(), in_relu, dy.data_ptr(), dwt.data_ptr(),
                                                 wsw.data_ptr(), int(zero.value), _stream()), 'senas_conv2d_bwd_weight')
    return dw


def _conv_wgrad_pair(ga, gb, x, in_relu, dya, dyb, wa, wb, da, db):
    """The weight gradients of the two convolutions of a pair as ONE first-stage launch (senas_conv2d_bwd_weight_pair); where
    the two share no kernel, the two single calls.  da, db: the destinations (wgrad_dest).  Returns what autograd is handed for
    wa and wb."""
    L = _lib.lib()
    sizes = []
    for g in (ga, gb):
        nbytes, zero = C.c_int64(), C.c_int32()
        _lib.check(L.senas_conv2d_bwd_weight_ws(C.byref(g), C.byref(nbytes), C.byref(zero)), 'senas_conv2d_bwd_weight_ws')
        sizes.append((nbytes.value, zero.value))
    if not MATH_TERMS and not sizes[0][1] and not sizes[1][1]:
        wsa, wsb = (torch.empty(nb, device=x.device, dtype=torch.uint8) for nb, _ in sizes)
        defer = may_defer(da[1], db[1])
        ia, ib = (_lib.SumItem(), _lib.SumItem()) if defer else (None, None)
        with _span('conv_wgrad', ga, x, wa, dya, problems=2):
            rc = L.senas_conv2d_bwd_weight_pair(C.byref(ga), C.byref(gb), x.data_ptr(), in_relu, dya.data_ptr(), dyb.data_ptr(),
                                                da[0].data_ptr(), db[0].data_ptr(), wsa.data_ptr(), wsb.data_ptr(),
                                                C.byref(ia) if defer else None, C.byref(ib) if defer else None, _stream())
        if rc == _lib.UNSUPPORTED:
            _unspan()
        else:
            _lib.check(rc, 'senas_conv2d_bwd_weight_pair')
            if defer:
                for item, ws in ((ia, wsa), (ib, wsb)):
                    if item.kind:
                        DEFER.append((item, ws))        # the partial images stay alive until the batched sum has run
            return da[1], db[1]
    return _conv_wgrad(ga, x, in_relu, dya, wa, dest=da), _conv_wgrad(gb, x, in_relu, dyb, wb, dest=db)


class _Conv2d(torch.autograd.Function):
    """y = conv(relu?(x), w) (+ producer-side batch-norm statistics of y).  ``stacked`` = k > 0: the convolution is a stacked one
    (k edges' candidates, weights stacked along c_out): y comes back 5-D [k][n][c_out/k][h][w] -- planar parts where a kernel
    has that epilogue (senas_conv2d_fwd_planar), else the 5-D view of the interleaved tensor."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, dil, transposed, out_pad, groups, in_relu, want_stats, stacked=0):
        x = nhwc(x)
        w = _dev(w).contiguous()
        n, ci, hi, wi = x.shape
        kh, kw = w.shape[2], w.shape[3]
        if transposed:
            if w.shape[0] != ci:
                raise SenasHipError('conv_transpose weight %s does not match %d input channels' % (tuple(w.shape), ci))
            co = w.shape[1] * groups
        else:
            if w.shape[1] * groups != ci:
                raise SenasHipError('conv weight %s does not match %d input channels' % (tuple(w.shape), ci))
            co = w.shape[0]
        ho = conv_out_size(hi, kh, stride, pad, dil, transposed, out_pad)
        wo = conv_out_size(wi, kw, stride, pad, dil, transposed, out_pad)
        g = ConvGeom(n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, dil, int(transposed), groups)
        L = _lib.lib()
        stats = new_stats(n, co, x) if want_stats else None
        ws = torch.empty(int(L.senas_conv2d_ws_bytes(C.byref(g))), device=x.device, dtype=torch.uint8)
        y = None
        if stacked and PLANAR and not MATH_TERMS and co == stacked * 8:
            # the 8-channel parts are read one by one by different node kernels: planar parts where a kernel has that epilogue
            # (declined -> the interleaved launch below)
            y5 = planar_empty(stacked, n, 8, ho, wo, x)
            with _span('conv_fwd', g, x, w, y5):
                rc = L.senas_conv2d_fwd_planar(C.byref(g), x.data_ptr(), w.data_ptr(), y5.data_ptr(), n * ho * wo * 8, int(in_relu),
                                               _p(stats), ws.data_ptr(), _packed(w, 0), _stream())
            if rc == _lib.UNSUPPORTED:
                _unspan()
            else:
                _lib.check(rc, 'senas_conv2d_fwd_planar')
                y = y5
                PLANAR_LAUNCHES[0] += 1
        if y is None and MATH_STORED and not stacked:
            # 'bf16s': the output as a bf16 tensor where the bf16-pipe kernel serves the shape (declined -> the launches below)
            yh = torch.empty((n, co, ho, wo), device=x.device, dtype=torch.bfloat16, memory_format=CL)
            with _span('conv_fwd', g, x, w, yh):
                rc = L.senas_conv2d_fwd_bf16s(C.byref(g), x.data_ptr(), w.data_ptr(), yh.data_ptr(), int(in_relu), _p(stats), ws.data_ptr(),
                                              _packed_lp(w, 0), _stream())
            if rc == _lib.UNSUPPORTED:
                _unspan()
            else:
                _lib.check(rc, 'senas_conv2d_fwd_bf16s')
                y = yh
        if y is None:
            y4 = new_nhwc(n, co, ho, wo, x)
            with _span('conv_fwd', g, x, w, y4):
                rc = _lib.UNSUPPORTED
                if MATH_TERMS:
                    rc = L.senas_conv2d_fwd_lp(C.byref(g), x.data_ptr(), w.data_ptr(), y4.data_ptr(), int(in_relu), _p(stats),
                                               ws.data_ptr(), _packed_lp(w, 0), MATH_TERMS, _stream())
                if rc == _lib.UNSUPPORTED:                      # (off the bf16 path: the fp32 kernels)
                    rc = L.senas_conv2d_fwd(C.byref(g), x.data_ptr(), w.data_ptr(), y4.data_ptr(), int(in_relu), _p(stats),
                                            ws.data_ptr(), _packed(w, 0), _stream())
                _lib.check(rc, 'senas_conv2d_fwd')
            y = stacked_5d(y4, stacked) if stacked else y4
        ctx.save_for_backward(x, w)
        ctx.g, ctx.in_relu = g, int(in_relu)
        ctx.set_materialize_grads(False)          # no zero tensor for the (non-differentiable) statistics output
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds):
        x, w = ctx.saved_tensors
        g, L = ctx.g, _lib.lib()
        if dy is None:
            return (None,) * 11
        dy = nhwc_stored(stacked_4d(dy) if dy.dim() == 5 else dy)
        stored = dy.dtype == torch.bfloat16              # ('bf16s': the gradient of a bf16-stored output arrives as bf16)
        dx = dw = None
        if ctx.needs_input_grad[1]:                       # (queued for the weight-gradient lane where there is one)
            dest = wgrad_dest(w)
            dw = dest[1]
            in_relu = ctx.in_relu
            run_wgrad((dw,), (x, dy), lambda: _conv_wgrad(g, x, in_relu, dy, w, dest=dest))
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=CL)
            ws = torch.empty(int(L.senas_conv2d_ws_bytes(C.byref(g))), device=x.device, dtype=torch.uint8)
            with _span('conv_dgrad', g, x, w, dy):
                rc = _lib.UNSUPPORTED
                if stored:
                    rc = L.senas_conv2d_bwd_data_bf16s(C.byref(g), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctx.in_relu, x.data_ptr(),
                                                       ws.data_ptr(), _packed(w, (1, 1)), _stream())
                    _lib.check(rc, 'senas_conv2d_bwd_data_bf16s')       # (what wrote a bf16 output serves its gradient: no fallback)
                elif MATH_TERMS:
                    rc = L.senas_conv2d_bwd_data_lp(C.byref(g), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctx.in_relu,
                                                    x.data_ptr(), ws.data_ptr(), _packed_lp(w, 1), MATH_TERMS, _stream())
                if rc == _lib.UNSUPPORTED:
                    rc = L.senas_conv2d_bwd_data(C.byref(g), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctx.in_relu,
                                                 x.data_ptr(), ws.data_ptr(), _packed(w, 1), _stream())
                _lib.check(rc, 'senas_conv2d_bwd_data')
        return dx, dw, None, None, None, None, None, None, None, None, None


# Stacked convolutions of a search cell write planar 8-channel parts where a kernel has that epilogue (class-wide switch)
PLANAR = os.environ.get('SENAS_PLANAR', '1') != '0'
PLANAR_LAUNCHES = [0]        # (tests: how many stacked convolutions took the planar epilogue)


def conv2d(x, w, stride=1, pad=0, dil=1, transposed=False, out_pad=0, groups=1, in_relu=False, want_stats=False, stacked=0):
    """Returns (y, stats) -- stats is None unless want_stats.  ``stacked`` = k: the output is a stacked convolution's (k parts
    of c_out / k channels each) and comes back 5-D [k][n][c][h][w]; functional.unstack takes it apart."""
    return _Conv2d.apply(x, w, stride, pad, dil, transposed, out_pad, groups, in_relu, want_stats, stacked)


class _Conv2dPair(torch.autograd.Function):
    """Two "same" Conv2d of ONE tensor that differ in the dilation only -- dil_3_conv_5 and dil_2_conv_5 of the same edges
    (utils/operations.py:69-72) -- as one forward and one data-gradient launch (senas_conv2d_fwd_pair / _bwd_data_pair; where
    the pair has no common kernel the entry points decline and the two single launches run).  xa, xb: two aliases of the input
    (its gradient is then ONE n-ary sum over all readers, functional.fan_out).  Outputs ya, stats_a, yb, stats_b."""

    @staticmethod
    def forward(ctx, xa, xb, wa, wb, stride, pad_a, dil_a, pad_b, dil_b, in_relu, want_stats, stacked=0):
        x = nhwc(xa)
        wa, wb = _dev(wa).contiguous(), _dev(wb).contiguous()
        n, ci, hi, wi = x.shape
        co, k = wa.shape[0], wa.shape[2]
        if wa.shape != wb.shape or wa.shape[1] != ci or wa.shape[2] != wa.shape[3]:
            raise SenasHipError('conv2d_pair: weights %s / %s on %d input channels' % (tuple(wa.shape), tuple(wb.shape), ci))
        ho, wo = conv_out_size(hi, k, stride, pad_a, dil_a, False, 0), conv_out_size(wi, k, stride, pad_a, dil_a, False, 0)
        if (ho, wo) != (conv_out_size(hi, k, stride, pad_b, dil_b, False, 0), conv_out_size(wi, k, stride, pad_b, dil_b, False, 0)):
            raise SenasHipError('conv2d_pair: the two convolutions disagree in their output size')
        ga = ConvGeom(n, hi, wi, ci, ho, wo, co, k, k, stride, pad_a, dil_a, 0, 1)
        gb = ConvGeom(n, hi, wi, ci, ho, wo, co, k, k, stride, pad_b, dil_b, 0, 1)
        L = _lib.lib()
        sa = new_stats(n, co, x) if want_stats else None
        sb = new_stats(n, co, x) if want_stats else None
        nb = int(L.senas_conv2d_ws_bytes(C.byref(ga)))
        wsa, wsb = (torch.empty(nb, device=x.device, dtype=torch.uint8) for _ in range(2))
        planar = False
        if stacked and PLANAR and not MATH_TERMS and co == stacked * 8:      # both outputs in planar 8-channel parts (see _Conv2d)
            ya, yb = planar_empty(stacked, n, 8, ho, wo, x), planar_empty(stacked, n, 8, ho, wo, x)
            with _span('conv_fwd', ga, x, wa, ya, problems=2):
                rc = L.senas_conv2d_fwd_pair_planar(C.byref(ga), C.byref(gb), x.data_ptr(), wa.data_ptr(), wb.data_ptr(), ya.data_ptr(),
                                                    yb.data_ptr(), n * ho * wo * 8, int(in_relu), _p(sa), _p(sb), wsa.data_ptr(), wsb.data_ptr(),
                                                    _packed(wa, 0), _packed(wb, 0), _stream())
            if rc == _lib.UNSUPPORTED:
                _unspan()
            else:
                _lib.check(rc, 'senas_conv2d_fwd_pair_planar')
                planar = True
                PLANAR_LAUNCHES[0] += 2
        if planar:
            ctx.save_for_backward(x, wa, wb)
            ctx.geoms, ctx.in_relu = (ga, gb), int(in_relu)
            ctx.set_materialize_grads(False)
            if want_stats:
                ctx.mark_non_differentiable(sa, sb)
            return ya, sa, yb, sb
        ya, yb = new_nhwc(n, co, ho, wo, x), new_nhwc(n, co, ho, wo, x)
        with _span('conv_fwd', ga, x, wa, ya, problems=2):
            rc = L.senas_conv2d_fwd_pair(C.byref(ga), C.byref(gb), x.data_ptr(), wa.data_ptr(), wb.data_ptr(), ya.data_ptr(), yb.data_ptr(),
                                         int(in_relu), _p(sa), _p(sb), wsa.data_ptr(), wsb.data_ptr(), _packed(wa, 0), _packed(wb, 0), _stream())
        if rc == _lib.UNSUPPORTED:
            _unspan()
            for g, w, y, st, ws in ((ga, wa, ya, sa, wsa), (gb, wb, yb, sb, wsb)):
                with _span('conv_fwd', g, x, w, y):
                    _lib.check(L.senas_conv2d_fwd(C.byref(g), x.data_ptr(), w.data_ptr(), y.data_ptr(), int(in_relu), _p(st), ws.data_ptr(),
                                                  _packed(w, 0), _stream()), 'senas_conv2d_fwd')
        else:
            _lib.check(rc, 'senas_conv2d_fwd_pair')
        ctx.save_for_backward(x, wa, wb)
        ctx.geoms, ctx.in_relu = (ga, gb), int(in_relu)
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(sa, sb)
        if stacked:
            ya, yb = stacked_5d(ya, stacked), stacked_5d(yb, stacked)
        return ya, sa, yb, sb

    @staticmethod
    def backward(ctx, dya, _sa, dyb, _sb):
        x, wa, wb = ctx.saved_tensors
        (ga, gb), L = ctx.geoms, _lib.lib()
        if dya is None and dyb is None:
            return (None,) * 12
        dya = nhwc(stacked_4d(dya) if dya.dim() == 5 else dya) if dya is not None else None
        dyb = nhwc(stacked_4d(dyb) if dyb.dim() == 5 else dyb) if dyb is not None else None
        dxa = dxb = None
        # the weight gradients: queued for the weight-gradient lane where there is one (see _Conv2d.backward)
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and dya is not None and dyb is not None and wa.data_ptr() != wb.data_ptr():
            da, db = wgrad_dest(wa), wgrad_dest(wb)
            dwa, dwb = da[1], db[1]
            in_relu = ctx.in_relu
            run_wgrad((dwa, dwb), (x, dya, dyb), lambda: _conv_wgrad_pair(ga, gb, x, in_relu, dya, dyb, wa, wb, da, db))
        else:
            dwa = dwb = None
            in_relu = ctx.in_relu
            if ctx.needs_input_grad[2] and dya is not None:
                da = wgrad_dest(wa)
                dwa = da[1]
                run_wgrad((dwa,), (x, dya), lambda: _conv_wgrad(ga, x, in_relu, dya, wa, dest=da))
            if ctx.needs_input_grad[3] and dyb is not None:
                db = wgrad_dest(wb)
                dwb = db[1]
                run_wgrad((dwb,), (x, dyb), lambda: _conv_wgrad(gb, x, in_relu, dyb, wb, dest=db))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            nb = int(L.senas_conv2d_ws_bytes(C.byref(ga)))
            rc = _lib.UNSUPPORTED
            if dya is not None and dyb is not None:
                dxa, dxb = torch.empty_like(x, memory_format=CL), torch.empty_like(x, memory_format=CL)
                wsa, wsb = (torch.empty(nb, device=x.device, dtype=torch.uint8) for _ in range(2))
                with _span('conv_dgrad', ga, x, wa, dya, problems=2):
                    rc = L.senas_conv2d_bwd_data_pair(C.byref(ga), C.byref(gb), dya.data_ptr(), dyb.data_ptr(), wa.data_ptr(), wb.data_ptr(),
                                                      dxa.data_ptr(), dxb.data_ptr(), ctx.in_relu, x.data_ptr(), wsa.data_ptr(), wsb.data_ptr(),
                                                      _packed(wa, 1), _packed(wb, 1), _stream())
                if rc == _lib.UNSUPPORTED:
                    _unspan()
            if rc == _lib.UNSUPPORTED:
                outs = []
                for g, w, dy in ((ga, wa, dya), (gb, wb, dyb)):
                    if dy is None:
                        outs.append(None)
                        continue
                    dx = torch.empty_like(x, memory_format=CL)
                    ws = torch.empty(nb, device=x.device, dtype=torch.uint8)
                    with _span('conv_dgrad', g, x, w, dy):
                        _lib.check(L.senas_conv2d_bwd_data(C.byref(g), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctx.in_relu, x.data_ptr(),
                                                           ws.data_ptr(), _packed(w, 1), _stream()), 'senas_conv2d_bwd_data')
                    outs.append(dx)
                dxa, dxb = outs
            else:
                _lib.check(rc, 'senas_conv2d_bwd_data_pair')
        return dxa, dxb, dwa, dwb, None, None, None, None, None, None, None, None


def _lp_serves(x, w, stride, pad, dil):
    """Does the bf16-pipe forward kernel take this Conv2d in the current math mode?"""
    n, ci, hi, wi = x.shape
    co, k = w.shape[0], w.shape[2]
    ho, wo = conv_out_size(hi, k, stride, pad, dil, False, 0), conv_out_size(wi, k, stride, pad, dil, False, 0)
    g = ConvGeom(n, hi, wi, ci, ho, wo, co, k, k, stride, pad, dil, 0, 1)
    return bool(_lib.lib().senas_conv2d_kernel_name_lp(C.byref(g), 0, MATH_TERMS))


def conv2d_pair(xa, xb, wa, wb, stride, pad_a, dil_a, pad_b, dil_b, in_relu=False, want_stats=False, stacked=0):
    """((ya, stats_a), (yb, stats_b)) of two convolutions of one tensor (two aliases of it) that differ in the dilation only.
    ``stacked``: as functional.conv2d."""
    if MATH_TERMS and _lp_serves(xa, wa, stride, pad_a, dil_a):
        # the bf16-pipe kernels have no pair form: two calls where they serve the shape (what they do not serve -- the search
        # cell's 8-channel edges, small maps -- keeps its fp32 pair launch in every math mode)
        return (conv2d(xa, wa, stride, pad_a, dil_a, in_relu=in_relu, want_stats=want_stats, stacked=stacked),
                conv2d(xb, wb, stride, pad_b, dil_b, in_relu=in_relu, want_stats=want_stats, stacked=stacked))
    ya, sa, yb, sb = _Conv2dPair.apply(xa, xb, wa, wb, stride, pad_a, dil_a, pad_b, dil_b, in_relu, want_stats, stacked)
    return (ya, sa), (yb, sb)


# ------------------------------------------------------------------------------------------ pooling / resampling
class _AvgPool3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, in_relu, want_stats):
        x = nhwc(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = new_nhwc(n, c, ho, wo, x)
        stats = new_stats(n, c, x) if want_stats else None
        _lib.check(_lib.lib().senas_avgpool3_fwd(n, h, w, c, stride, x.data_ptr(), int(in_relu), y.data_ptr(), _p(stats),
                                                 _stream()), 'senas_avgpool3_fwd')
        ctx.save_for_backward(x)
        ctx.meta = (stride, int(in_relu))
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds=None):
        (x,) = ctx.saved_tensors
        stride, in_relu = ctx.meta
        n, c, h, w = x.shape
        dy = nhwc(dy)
        dx = torch.empty_like(x, memory_format=CL)
        _lib.check(_lib.lib().senas_avgpool3_bwd(n, h, w, c, stride, dy.data_ptr(), in_relu, x.data_ptr(), dx.data_ptr(),
                                                 _stream()), 'senas_avgpool3_bwd')
        return dx, None, None, None


class _MaxPool3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, in_relu, want_stats):
        x = nhwc(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = new_nhwc(n, c, ho, wo, x)
        arg = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8)
        stats = new_stats(n, c, x) if want_stats else None
        _lib.check(_lib.lib().senas_maxpool3_fwd(n, h, w, c, stride, x.data_ptr(), int(in_relu), y.data_ptr(),
                                                 arg.data_ptr(), _p(stats), _stream()), 'senas_maxpool3_fwd')
        ctx.save_for_backward(x, arg)
        ctx.meta = (stride, int(in_relu))
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds=None):
        x, arg = ctx.saved_tensors
        stride, in_relu = ctx.meta
        n, c, h, w = x.shape
        dy = nhwc(dy)
        dx = torch.empty_like(x, memory_format=CL)
        _lib.check(_lib.lib().senas_maxpool3_bwd(n, h, w, c, stride, dy.data_ptr(), arg.data_ptr(), in_relu, x.data_ptr(),
                                                 dx.data_ptr(), _stream()), 'senas_maxpool3_bwd')
        return dx, None, None, None


class _Bilinear2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, want_stats):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, 2 * h, 2 * w, x)
        stats = new_stats(n, c, x) if want_stats else None
        _lib.check(_lib.lib().senas_bilinear2x_fwd(n, h, w, c, x.data_ptr(), y.data_ptr(), _p(stats), _stream()),
                   'senas_bilinear2x_fwd')
        ctx.shape = (n, c, h, w)
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds=None):
        n, c, h, w = ctx.shape
        dy = nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy)
        _lib.check(_lib.lib().senas_bilinear2x_bwd(n, h, w, c, dy.data_ptr(), dx.data_ptr(), _stream()),
                   'senas_bilinear2x_bwd')
        return dx, None


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        y = torch.empty_like(x, memory_format=CL)
        _lib.check(_lib.lib().senas_relu_fwd(x.numel(), x.data_ptr(), y.data_ptr(), _stream()), 'senas_relu_fwd')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = nhwc(dy)
        dx = torch.empty_like(y, memory_format=CL)
        _lib.check(_lib.lib().senas_relu_bwd(y.numel(), dy.data_ptr(), y.data_ptr(), dx.data_ptr(), _stream()),
                   'senas_relu_bwd')
        return dx


class _ZeroFeature(torch.autograd.Function):
    """The all-zero feature map ZeroOp feeds its adapter (x.mul(0.)): zeros forward, and -- as in the
    reference's autograd -- an exactly-zero (not absent) gradient for x."""

    @staticmethod
    def forward(ctx, x, c_out):
        n, _, h, w = x.shape
        ctx.save_for_backward(x)
        return torch.zeros((n, c_out, h, w), device=x.device, dtype=torch.float32).contiguous(memory_format=CL)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return torch.zeros_like(x), None


def zero_feature(x, c_out):
    return _ZeroFeature.apply(_dev(x), c_out)


def avg_pool3(x, stride, in_relu=False, want_stats=False):
    """y, or (y, stats) with want_stats: the producer-side per-image channel sums the next BatchNorm needs."""
    y, st = _AvgPool3.apply(x, stride, in_relu, want_stats)
    return (y, st) if want_stats else y


def max_pool3(x, stride, in_relu=False, want_stats=False):
    y, st = _MaxPool3.apply(x, stride, in_relu, want_stats)
    return (y, st) if want_stats else y


def bilinear2x(x, want_stats=False):
    y, st = _Bilinear2x.apply(x, want_stats)
    return (y, st) if want_stats else y


def relu(x):
    return _ReLU.apply(x)


class _Blend2(torch.autograd.Function):
    """g[0] * x1 + g[1] * x2 (the gamma-gated skip blend of the supernet, search/senas_search.py:98-102) in one launch;
    the backward pass produces both input gradients and d g in one more (plus the fp64 -> fp32 hand-over of d g)."""

    @staticmethod
    def forward(ctx, x1, x2, g):
        x1, x2 = nhwc(x1), nhwc(x2)
        if x1.shape != x2.shape or x1.numel() % 4 != 0 or g.numel() != 2:
            raise SenasHipError('blend2: shapes %s / %s / %s' % (tuple(x1.shape), tuple(x2.shape), tuple(g.shape)))
        gc = _dev(g).contiguous()
        y = torch.empty_like(x1, memory_format=CL)
        _lib.check(_lib.lib().senas_blend2_fwd(x1.numel(), x1.data_ptr(), x2.data_ptr(), gc.data_ptr(), y.data_ptr(), _stream()),
                   'senas_blend2_fwd')
        ctx.save_for_backward(x1, x2, gc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, gc = ctx.saved_tensors
        dy = nhwc(dy)
        need = ctx.needs_input_grad
        dx1 = torch.empty_like(x1, memory_format=CL) if need[0] else None
        dx2 = torch.empty_like(x2, memory_format=CL) if need[1] else None
        dg = zeros64((2,), x1.device)
        _lib.check(_lib.lib().senas_blend2_bwd(x1.numel(), dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), gc.data_ptr(), _p(dx1), _p(dx2),
                                               dg.data_ptr(), _stream()), 'senas_blend2_bwd')
        return dx1, dx2, (dg.float() if need[2] else None)


def blend2(x1, x2, g):
    return _Blend2.apply(x1, x2, g)


class _GammaRows(torch.autograd.Function):
    """softmax(gamma) as a table whose rows the skip blends of one forward pass read in place; they ADD their d loss / d row
    into one fp64 buffer (senas_blend2_bwd accumulates), which this backward hands on once -- instead of a select, a
    zero-fill, a copy, a cast and an add per blend."""

    @staticmethod
    def forward(ctx, table, acc):
        ctx.save_for_backward(acc)
        ctx.set_materialize_grads(False)
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g):
        (acc,) = ctx.saved_tensors
        join_lanes()                              # the blends of every lane have added into acc
        d = acc.float()
        return (d if g is None else d + g), None


class GammaRows(object):
    def __init__(self, table):
        t = _dev(table).contiguous()
        self.acc = zeros64(tuple(t.shape), t.device)
        self.table = _GammaRows.apply(t, self.acc)


class _Blend2Row(torch.autograd.Function):
    """_Blend2 with the mixing pair given as row ``idx`` of a GammaRows table."""

    @staticmethod
    def forward(ctx, x1, x2, table, idx, acc):
        x1, x2 = nhwc(x1), nhwc(x2)
        if x1.shape != x2.shape or x1.numel() % 4 != 0 or table.shape[1] != 2:
            raise SenasHipError('blend2: shapes %s / %s / %s' % (tuple(x1.shape), tuple(x2.shape), tuple(table.shape)))
        y = torch.empty_like(x1, memory_format=CL)
        gp = table.data_ptr() + 8 * idx
        _lib.check(_lib.lib().senas_blend2_fwd(x1.numel(), x1.data_ptr(), x2.data_ptr(), gp, y.data_ptr(), _stream()), 'senas_blend2_fwd')
        ctx.save_for_backward(x1, x2, table, acc)
        ctx.idx = idx
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, table, acc = ctx.saved_tensors
        dy = nhwc(dy)
        need = ctx.needs_input_grad
        dx1 = torch.empty_like(x1, memory_format=CL) if need[0] else None
        dx2 = torch.empty_like(x2, memory_format=CL) if need[1] else None
        _lib.check(_lib.lib().senas_blend2_bwd(x1.numel(), dy.data_ptr(), x1.data_ptr(), x2.data_ptr(), table.data_ptr() + 8 * ctx.idx,
                                               _p(dx1), _p(dx2), acc.data_ptr() + 16 * ctx.idx, _stream()), 'senas_blend2_bwd')
        return dx1, dx2, None, None, None


def blend2_row(x1, x2, rows, idx):
    """rows.table[idx][0] * x1 + rows.table[idx][1] * x2 (rows: GammaRows)."""
    return _Blend2Row.apply(x1, x2, rows.table, idx, rows.acc)


class _SkipStack(torch.autograd.Function):
    """in0 of a supernet up cell (search/senas_search.py:96-103) in one launch per direction: the channel concatenation of
    xs[0] and the blends rows[idx[k]][0] * xs[k-1] + rows[idx[k]][1] * xs[k], k = 1 .. m-1 (senas_skipcat_fwd / _bwd) -- every
    tensor of the column read once, the concatenation written once, and on the way back every gradient written once from
    the concatenation's own gradient (no slice copies, no blend outputs in between)."""

    @staticmethod
    def forward(ctx, table, acc, idx, *xs):
        xs = tuple(nhwc(x) for x in xs)
        m, ref = len(xs), xs[0]
        n, c, h, w = ref.shape
        if not 2 <= m <= _lib.SKIP_MAX or c % 4 != 0 or any(x.shape != ref.shape or x.dtype != torch.float32 for x in xs) \
                or table.dim() != 2 or table.shape[1] != 2 or len(idx) != m:
            raise SenasHipError('skip_stack: %d tensors of %s, table %s' % (m, [tuple(x.shape) for x in xs], tuple(table.shape)))
        y = torch.empty((n, m * c, h, w), device=ref.device, dtype=torch.float32, memory_format=CL)
        ctx.ptrs = (C.c_void_p * m)(*[x.data_ptr() for x in xs])
        ctx.idx = (C.c_int32 * m)(*[int(k) for k in idx])
        _lib.check(_lib.lib().senas_skipcat_fwd(n * h * w, c, m, ctx.ptrs, table.data_ptr(), table.shape[0], ctx.idx, y.data_ptr(),
                                                _stream()), 'senas_skipcat_fwd')
        ctx.save_for_backward(table, acc, *xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        table, acc, *xs = ctx.saved_tensors
        dy = nhwc(dy)
        m, ref = len(xs), xs[0]
        n, c, h, w = ref.shape
        need = ctx.needs_input_grad[3:]
        dxs = [torch.empty_like(ref, memory_format=CL) if need[k] else None for k in range(m)]
        outs = (C.c_void_p * m)(*[_p(d) for d in dxs])
        _lib.check(_lib.lib().senas_skipcat_bwd(n * h * w, c, m, dy.data_ptr(), ctx.ptrs, table.data_ptr(), table.shape[0], ctx.idx, outs,
                                                acc.data_ptr(), _stream()), 'senas_skipcat_bwd')
        return (None, None, None) + tuple(dxs)


def skip_stack(xs, rows, idx):
    """cat([xs[0]] + [rows.table[idx[k]][0] * xs[k-1] + rows.table[idx[k]][1] * xs[k] for k >= 1], dim=1) (rows: GammaRows;
    idx[0] is ignored)."""
    return _SkipStack.apply(rows.table, rows.acc, tuple(idx), *xs)


class _EdgeMix(torch.autograd.Function):
    """M[e][k] = beta[e] * (w_norm[e][k] if edge e is a NORM edge else w_chg[e][k]) for all edges of a cell kind at once
    (search/cell.py:33-36,100-106: every MixedOp scales its candidates by its alpha row, every node scales the edge by
    beta).  The cells of one kind share M within a forward pass; their nodes read their rows of it in place and ADD
    d loss / d M into the zero-filled ``dM`` (senas_node_bwd, dmix_accumulate: every cell into its own slot, the nodes of one
    cell in stream order), which this backward pass turns into the three gradients -- a handful of tiny launches per pass
    instead of ~10 per cell."""

    @staticmethod
    def forward(ctx, w_norm, w_chg, betas, is_norm, dM):
        W = torch.where(is_norm, w_norm, w_chg)
        M = (betas.unsqueeze(1) * W).contiguous()
        ctx.save_for_backward(W, betas, is_norm, dM)
        ctx.set_materialize_grads(False)
        return M

    @staticmethod
    def backward(ctx, gM):
        W, betas, is_norm, dM = ctx.saved_tensors
        join_lanes()                              # the nodes of every lane have added into their cell's slot of dM
        dM = dM.sum(0) if dM.dim() == 3 else dM
        g = dM if gM is None else dM + gM
        dW = g * betas.unsqueeze(1)
        zero = torch.zeros((), device=g.device, dtype=g.dtype)
        return torch.where(is_norm, dW, zero), torch.where(is_norm, zero, dW), (g * W).sum(1), None, None


class _ArchMix(torch.autograd.Function):
    """softmax(alpha) x 4, the windowed softmax(beta) x 2, softmax(gamma) and the two per-kind mixing matrices in ONE launch
    (senas_arch_mix_fwd); backward: from the accumulation tables the nodes / blends added into, straight to the seven
    parameter gradients in one more (senas_arch_mix_bwd).  Outputs: M_dn, M_up, softmax(gamma) (differentiable: they tie the
    consumers to this node) and the six softmax tables (bookkeeping only)."""

    @staticmethod
    def forward(ctx, nodes, dM_dn, dM_up, dG, a_dn, a_up, a_dn_nm, a_up_nm, b_dn, b_up, gamma):
        params = [_dev(t).contiguous() for t in (a_dn, a_up, a_dn_nm, a_up_nm, b_dn, b_up, gamma)]
        k, ops = params[0].shape
        dev_ = params[0].device
        new = lambda *shape: torch.empty(shape, device=dev_, dtype=torch.float32)
        soft = [new(k, ops) for _ in range(4)] + [new(k), new(k), new(*gamma.shape)]
        M = [new(k, ops), new(k, ops)]
        a = _lib.ArchMix()
        for i in range(4):
            a.alpha[i], a.s_alpha[i] = params[i].data_ptr(), soft[i].data_ptr()
        for i in range(2):
            a.beta[i], a.s_beta[i], a.M[i] = params[4 + i].data_ptr(), soft[4 + i].data_ptr(), M[i].data_ptr()
        a.gamma, a.s_gamma = params[6].data_ptr(), soft[6].data_ptr()
        a.k, a.ops, a.nodes, a.grows = k, ops, nodes, gamma.shape[0]
        _lib.check(_lib.lib().senas_arch_mix_fwd(C.byref(a), _stream()), 'senas_arch_mix_fwd')
        ctx.save_for_backward(dM_dn, dM_up, dG, *params, *soft, *M)
        ctx.nodes = nodes
        ctx.shared_nm = a_up_nm.data_ptr() == a_dn_nm.data_ptr()
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*soft[:6])
        return (M[0], M[1], soft[6]) + tuple(soft[:6])

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        dM_dn, dM_up, dG = saved[:3]
        params, soft, M = saved[3:10], saved[10:17], saved[17:19]
        if any(g is not None for g in grads[:3]):
            raise SenasHipError('arch_mix: gradients reach the mixing tables through autograd; they are accumulated in place')
        k, ops = params[0].shape
        dests, outs = [], []
        for i, p in enumerate(params):
            if i == 3 and ctx.shared_nm:                 # alphas_up_nm IS alphas_dn_nm: its gradient rides in that one's
                dests.append(None)
                outs.append(None)
                continue
            d, g = wgrad_dest(p)
            dests.append(d)
            outs.append(g if p.numel() else None)        # (an empty gamma -- depth 2 -- has no gradient, as in the reference)
        a = _lib.ArchMix()
        for i in range(4):
            a.alpha[i], a.s_alpha[i] = params[i].data_ptr(), soft[i].data_ptr()
            a.d_alpha[i] = dests[i].data_ptr() if dests[i] is not None else None
        for i in range(2):
            a.beta[i], a.s_beta[i], a.M[i] = params[4 + i].data_ptr(), soft[4 + i].data_ptr(), M[i].data_ptr()
            a.d_beta[i] = dests[4 + i].data_ptr()
        a.dM[0], a.dM[1], a.dG = dM_dn.data_ptr(), dM_up.data_ptr(), dG.data_ptr()
        a.gamma, a.s_gamma, a.d_gamma = params[6].data_ptr(), soft[6].data_ptr(), dests[6].data_ptr()
        a.k, a.ops, a.nodes, a.grows = k, ops, ctx.nodes, params[6].shape[0]
        a.slots = dM_dn.shape[0]
        join_lanes()                              # nodes and blends of every lane have added into dM / dG
        _lib.check(_lib.lib().senas_arch_mix_bwd(C.byref(a), _stream()), 'senas_arch_mix_bwd')
        return (None, None, None, None) + tuple(outs)


class ArchTables(object):
    """What NAS.forward hands the network in place of ~20 softmax / select / multiply / concatenate launches: the softmax
    tables (the positional arguments of SenasSearch.forward), with the mixing matrices of both cell kinds parked on the beta
    tables (Cell._node_mixes looks there) and the blend table on the gamma one (SenasSearch: functional.GammaRows)."""

    def __init__(self, nodes, a_dn, a_up, a_dn_nm, a_up_nm, b_dn, b_up, gamma):
        k, ops = a_dn.shape
        dev_ = a_dn.device
        dM = [zeros32(MIX_SLOTS * k * ops, dev_).view(MIX_SLOTS, k, ops) for _ in range(2)]
        dG = zeros64(tuple(gamma.shape), dev_)
        out = _ArchMix.apply(nodes, dM[0], dM[1], dG, a_dn, a_up, a_dn_nm, a_up_nm, b_dn, b_up, gamma)
        M_dn, M_up, s_gamma = out[:3]
        self.s_dn, self.s_up, self.s_dn_nm, self.s_up_nm, self.bs_dn, self.bs_up = out[3:]
        self.s_gamma = s_gamma
        rows = GammaRows.__new__(GammaRows)
        rows.table, rows.acc = s_gamma, dG
        s_gamma._senas_rows = rows
        for kind, (M, bs, w_norm, w_chg) in enumerate(((M_dn, self.bs_dn, self.s_dn_nm, self.s_dn), (M_up, self.bs_up, self.s_up_nm, self.s_up))):
            kinds = [(j >= 2) if kind == 0 else (j != 1) for i in range(nodes) for j in range(2 + i)]
            bs._senas_mix = {(id(w_norm), id(w_chg), tuple(kinds)): MixSlots(M, dM[kind], [2 + i for i in range(nodes)])}

    def args(self):
        """(alpha_dn_nm, alpha_up_nm, alpha_dn, alpha_up, beta_dn, beta_up, gamma) as SenasSearch.forward takes them."""
        return self.s_dn_nm, self.s_up_nm, self.s_dn, self.s_up, self.bs_dn, self.bs_up, self.s_gamma


class SharedMix(object):
    """Rows [off, off + rows) of an _EdgeMix matrix as the mixing weights of one node: ``M`` (for autograd ordering and
    the kernel's read), the flat offset / length of the node's weights in it, and the accumulation buffer ``dM`` (the [k][ops]
    slot of the node's cell)."""
    __slots__ = ('M', 'off', 'count', 'dM')

    def __init__(self, M, off, count, dM):
        self.M, self.off, self.count, self.dM = M, off, count, dM


class MixSlots(object):
    """The mixing matrix of a cell kind for one pass and its gradient table ``dM`` [MIX_SLOTS][k][ops].  Every cell forward
    takes the next slot (host order, so the assignment is the same on every pass and rank): the nodes of a cell add their
    d loss / d mix into the cell's own rows, cells on different lanes never write the same address, and the fold over the
    slots runs in slot order (bitwise reproducible)."""

    def __init__(self, M, dM, counts):
        self.M, self.dM, self.counts, self.taken = M, dM, counts, 0

    def take(self):
        if self.taken >= self.dM.shape[0]:
            raise SenasHipError('more than %d cells of one kind in a pass (functional.MIX_SLOTS)' % self.dM.shape[0])
        row = self.dM[self.taken]
        self.taken += 1
        ops = self.dM.shape[2]
        mixes, offset = [], 0
        for cnt in self.counts:
            mixes.append(SharedMix(self.M, offset * ops, cnt * ops, row))
            offset += cnt
        return mixes


class _FanOut(torch.autograd.Function):
    """n aliases of x for n consumers; the backward pass adds the incoming gradients in ONE launch (senas_sum_n)
    instead of leaving n-1 binary accumulations to autograd."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        ref = next((g for g in gs if g.dim() == 4 and (g.is_contiguous(memory_format=CL) or g.is_contiguous())), gs[0])

        def stride_of(g):
            """Pixel stride of g if it is (a channel slice of) an NHWC tensor laid out like ``ref``, else None."""
            if not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 4 and g.shape == ref.shape and g.data_ptr() % 16 == 0):
                return None
            n, c, h, w = g.shape
            ct = g.stride(3) if c > 1 else 0
            if c % 4 == 0 and ct >= c and ct % 4 == 0 and g.stride() == (h * w * ct, 1, w * ct, ct):
                return ct
            return None

        same = [g for g in gs if g.is_cuda and g.dtype == torch.float32 and g.stride() == ref.stride() and g.shape == ref.shape and
                (g.is_contiguous(memory_format=CL) or g.is_contiguous()) and g.data_ptr() % 16 == 0]
        strides = [stride_of(g) for g in gs]
        out = None
        if len(same) < len(gs) and ref.dim() == 4 and ref.is_contiguous(memory_format=CL) and all(st is not None for st in strides) \
                and len(gs) <= _lib.MAX_TERMS:
            # some gradients are channel slices (the torch.cat consumer's): one strided n-ary sum instead of binary adds
            n, c, h, w = ref.shape
            dst = torch.empty((n, c, h, w), device=ref.device, dtype=torch.float32, memory_format=CL)
            ptrs = (C.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
            st = (C.c_int32 * len(gs))(*strides)
            _lib.check(_lib.lib().senas_sum_n_strided(len(gs), n * h * w, c, ptrs, st, dst.data_ptr(), _stream()), 'senas_sum_n_strided')
            return dst, None
        dense = same
        rest = [g for g in gs if not any(g is d for d in dense)]
        while dense:
            take = _lib.MAX_TERMS - (1 if out is not None else 0)
            chunk, dense = ([out] if out is not None else []) + dense[:take], dense[take:]
            dst = torch.empty_like(ref)
            ptrs = (C.c_void_p * len(chunk))(*[g.data_ptr() for g in chunk])
            _lib.check(_lib.lib().senas_sum_n(len(chunk), ref.numel(), ptrs, dst.data_ptr(), _stream()), 'senas_sum_n')
            out = dst
        for g in rest:                                    # odd layouts: let torch add them
            out = g if out is None else out + g
        return out, None


# tools/lane_timeline.py: a StampRecorder while a timeline is being taken, else None (no stamp is launched)
STAMPS = None


class StampRecorder(object):
    """Device time stamps in stream order at named points of a pass (senas_stamp), forward and backward: every ``stamp`` call
    takes the next pair of slots; under HIP-graph replay the captured stamps are rewritten by every replay."""

    def __init__(self, device, capacity=16384):
        self.buf = torch.zeros(capacity, device=device, dtype=torch.int64)
        self.names = []

    def slots(self, name):
        k = len(self.names)
        if 2 * k + 1 >= self.buf.numel():
            raise SenasHipError('stamp recorder full')
        self.names.append(name)
        return self.buf.data_ptr() + 16 * k, self.buf.data_ptr() + 16 * k + 8

    def read(self):
        """[(name, forward ticks, backward ticks)] of the slots written since the buffer was last zeroed (100 MHz ticks)."""
        v = self.buf.cpu().tolist()
        return [(n, v[2 * k], v[2 * k + 1]) for k, n in enumerate(self.names) if v[2 * k] or v[2 * k + 1]]


class _Stamp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd_slot, bwd_slot):
        _lib.check(_lib.lib().senas_stamp(fwd_slot, _stream()), 'senas_stamp')
        ctx.slot = bwd_slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        _lib.check(_lib.lib().senas_stamp(ctx.slot, _stream()), 'senas_stamp')
        return g, None, None


def stamp(x, name):
    """x, with a time stamp launched on the current stream now and another where its gradient passes on the way back."""
    if STAMPS is None:
        return x
    f, b = STAMPS.slots(name)
    y = _Stamp.apply(x, f, b)
    st = getattr(x, '_senas_stats', None)
    if st is not None:
        y._senas_stats = st
    return y


# Marker kinds (empty kernels the lane scheduler reads and contracts out of the captured graph: csrc/sched.hip)
MARK_RELAY, MARK_PRODUCER, MARK_CONSUMER = 0, 1, 2


def marker(kind):
    """An empty kernel of the given kind on the current stream while a pass is being captured (nothing outside a capture)."""
    if torch.cuda.is_current_stream_capturing():
        _lib.check(_lib.lib().senas_marker(int(kind), _stream()), 'senas_marker')


def relay_marker():
    """A RELAY marker on the current (origin) stream: it gives the chain of hand-overs the capture records on that stream nodes
    the lane scheduler recognises.  A reader that sits behind a CONSUMER marker gets the relay's PRODUCER-marked parents as its
    dependencies and nothing else of the origin stream's history; the origin stream's own next node keeps all of them
    (csrc/sched.hip, note at relay_marker_kernel)."""
    marker(MARK_RELAY)


def _ride(x, y):
    """Producer-side statistics riding on x ride on its alias y too."""
    st = getattr(x, '_senas_stats', None)
    if st is not None:
        y._senas_stats = st
    return y


class _Hop(torch.autograd.Function):
    """An alias of x behind an autograd node of the CURRENT stream.  Autograd replays a node on the stream its forward pass
    ran on and makes that stream wait for the producer of every gradient it receives: a tensor that goes from one lane of
    the macro grid to another passes a hop made on the caller's stream, so that on the way back, too, lanes only ever wait
    for that stream and only that stream waits for lanes (senas_amd.grid.Lanes: the star topology the HIP runtime's
    capture bookkeeping needs)."""

    @staticmethod
    def forward(ctx, x):
        relay_marker()
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        relay_marker()
        return g


def hop(x):
    if not (torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled()):
        if torch.is_tensor(x) and x.is_cuda:
            relay_marker()
        return x
    return _ride(x, _Hop.apply(x))


class _LaneOut(torch.autograd.Function):
    """The producer's end of a hand-over, made on the PRODUCER's lane (grid.Lanes.hand): nothing forward (the PRODUCER marker
    sits in front of the event the reader waits for, grid.Lanes.mark); on the way back the gradient arrives here from the origin
    stream: a CONSUMER marker."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        marker(MARK_CONSUMER)
        return g


class _LaneIn(torch.autograd.Function):
    """The reader's end of a hand-over, made on the READER's lane: a CONSUMER marker forward; on the way back the gradient leaves
    this lane for the origin stream: a PRODUCER marker."""

    @staticmethod
    def forward(ctx, x):
        marker(MARK_CONSUMER)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        marker(MARK_PRODUCER)
        return g


def lane_out(x):
    if not (torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled()):
        return x
    return _ride(x, _LaneOut.apply(x))


def lane_in(x):
    if not (torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled()):
        if torch.is_tensor(x) and x.is_cuda:
            marker(MARK_CONSUMER)
        return x
    return _ride(x, _LaneIn.apply(x))


def fan_out(x, n):
    """n aliases of x (n > 1), or [x]; producer-side statistics riding on x (``_senas_stats``) ride on the aliases too."""
    if n <= 1:
        return [x]
    out = list(_FanOut.apply(x, n))
    st = getattr(x, '_senas_stats', None)
    if st is not None:
        for a in out:
            a._senas_stats = st
    return out


class _DwMulti(torch.autograd.Function):
    """k depthwise convolutions of ONE input (senas_dwconv_pair_*): ka of geometry ga followed by kb of geometry gb (the 3x3
    and the 5x5 DepSepConv candidates of the same edges; kb may be 0).  Outputs z_1..z_k (+ their statistics); the
    backward pass produces dx = sum_p dgrad_p in one launch and the k weight gradients in one (+ their deferred sums)."""

    @staticmethod
    def forward(ctx, x, ga, ka, gb, kb, want_stats, *ws):
        k = ka + kb
        x = nhwc(x)
        ws = [_dev(w).contiguous() for w in ws]
        g = ga
        ys = [new_nhwc(g.n, g.co, g.ho, g.wo, x) for _ in range(k)]
        stats = [new_stats(g.n, g.co, x) for _ in range(k)] if want_stats else []
        wp = (C.c_void_p * k)(*[w.data_ptr() for w in ws])
        yp = (C.c_void_p * k)(*[y.data_ptr() for y in ys])
        sp = (C.c_void_p * k)(*[s.data_ptr() for s in stats]) if want_stats else None
        gbp = C.byref(gb) if kb else None
        _lib.check(_lib.lib().senas_dwconv_pair_fwd(C.byref(ga), ka, gbp, kb, x.data_ptr(), wp, yp, sp, _stream()), 'senas_dwconv_pair_fwd')
        ctx.geoms, ctx.k = (ga, ka, gb, kb), k
        ctx.save_for_backward(x, *ws)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*stats)
        return tuple(ys) + tuple(stats)

    @staticmethod
    def backward(ctx, *grads):
        (ga, ka, gb, kb), k, L = ctx.geoms, ctx.k, _lib.lib()
        g = ga
        gbp = C.byref(gb) if kb else None
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dys = list(grads[:k])
        if all(d is None for d in dys):
            return (None,) * (6 + k)
        dys = [nhwc(d) if d is not None else torch.zeros((g.n, g.co, g.ho, g.wo), device=x.device).contiguous(memory_format=CL) for d in dys]
        dyp = (C.c_void_p * k)(*[d.data_ptr() for d in dys])
        dx = None
        dws = [None] * k
        if any(ctx.needs_input_grad[6:]):                 # (queued for the weight-gradient lane where there is one)
            dwt, dws = zip(*[wgrad_dest(w) for w in ws])
            defer = may_defer(*dws)

            def launch():
                scratch = torch.empty(int(L.senas_dwconv_pair_ws_bytes(C.byref(ga), ka, gbp, kb)), device=x.device, dtype=torch.uint8)
                dwp = (C.c_void_p * k)(*[d.data_ptr() for d in dwt])
                items = (_lib.SumItem * k)() if defer else None
                _lib.check(L.senas_dwconv_pair_bwd_weight(C.byref(ga), ka, gbp, kb, x.data_ptr(), dyp, dwp, scratch.data_ptr(), items, _stream()),
                           'senas_dwconv_pair_bwd_weight')
                if items is not None:
                    for t in range(k):
                        one = _lib.SumItem()
                        C.memmove(C.byref(one), C.byref(items[t]), C.sizeof(one))
                        DEFER.append((one, scratch))
            run_wgrad(dws, [x] + dys, launch)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=CL)
            wp = (C.c_void_p * k)(*[w.data_ptr() for w in ws])
            _lib.check(L.senas_dwconv_pair_bwd_data(C.byref(ga), ka, gbp, kb, dyp, wp, dx.data_ptr(), _stream()), 'senas_dwconv_pair_bwd_data')
        return (dx, None, None, None, None, None) + tuple(dws)


class _DwMulti2(torch.autograd.Function):
    """The same for problems that read one of TWO inputs (senas_dwconv_pair_fwd_xs): the DepSepConv candidates of both input
    states of a search cell.  src[p] in {0, 1} names problem p's input (kernel order: the ka 3x3 problems, then the kb 5x5
    ones).  One launch forward, one for all weight gradients; the data gradients stay one launch per input."""

    @staticmethod
    def forward(ctx, x0, x1, ga, ka, gb, kb, src, want_stats, *ws):
        k = ka + kb
        xs = [nhwc(x0), nhwc(x1)]
        ws = [_dev(w).contiguous() for w in ws]
        g = ga
        ys = [new_nhwc(g.n, g.co, g.ho, g.wo, xs[0]) for _ in range(k)]
        stats = [new_stats(g.n, g.co, xs[0]) for _ in range(k)] if want_stats else []
        xp = (C.c_void_p * k)(*[xs[src[p]].data_ptr() for p in range(k)])
        wp = (C.c_void_p * k)(*[w.data_ptr() for w in ws])
        yp = (C.c_void_p * k)(*[y.data_ptr() for y in ys])
        sp = (C.c_void_p * k)(*[s.data_ptr() for s in stats]) if want_stats else None
        _lib.check(_lib.lib().senas_dwconv_pair_fwd_xs(C.byref(ga), ka, C.byref(gb) if kb else None, kb, None, xp, wp, yp, sp, _stream()),
                   'senas_dwconv_pair_fwd_xs')
        ctx.geoms, ctx.k, ctx.src = (ga, ka, gb, kb), k, tuple(src)
        ctx.save_for_backward(xs[0], xs[1], *ws)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*stats)
        return tuple(ys) + tuple(stats)

    @staticmethod
    def backward(ctx, *grads):
        (ga, ka, gb, kb), k, src, L = ctx.geoms, ctx.k, ctx.src, _lib.lib()
        g = ga
        xs, ws = ctx.saved_tensors[:2], ctx.saved_tensors[2:]
        dys = list(grads[:k])
        if all(d is None for d in dys):
            return (None,) * (8 + k)
        dev_ = xs[0].device
        dys = [nhwc(d) if d is not None else torch.zeros((g.n, g.co, g.ho, g.wo), device=dev_).contiguous(memory_format=CL) for d in dys]
        dws = [None] * k
        if any(ctx.needs_input_grad[8:]):                 # (queued for the weight-gradient lane where there is one)
            dwt, dws = zip(*[wgrad_dest(w) for w in ws])
            defer = may_defer(*dws)

            def launch():
                gbp = C.byref(gb) if kb else None
                scratch = torch.empty(int(L.senas_dwconv_pair_ws_bytes(C.byref(ga), ka, gbp, kb)), device=dev_, dtype=torch.uint8)
                xp = (C.c_void_p * k)(*[xs[src[p]].data_ptr() for p in range(k)])
                dyq = (C.c_void_p * k)(*[d.data_ptr() for d in dys])
                dwp = (C.c_void_p * k)(*[d.data_ptr() for d in dwt])
                items = (_lib.SumItem * k)() if defer else None
                _lib.check(L.senas_dwconv_pair_bwd_weight_xs(C.byref(ga), ka, gbp, kb, None, xp, dyq, dwp, scratch.data_ptr(), items, _stream()),
                           'senas_dwconv_pair_bwd_weight_xs')
                if items is not None:
                    for t in range(k):
                        one = _lib.SumItem()
                        C.memmove(C.byref(one), C.byref(items[t]), C.sizeof(one))
                        DEFER.append((one, scratch))
            run_wgrad(dws, list(xs) + dys, launch)
        dxs = [None, None]
        for i in range(2):
            if not ctx.needs_input_grad[i]:
                continue
            pa = [p for p in range(ka) if src[p] == i]
            pb = [p for p in range(ka, k) if src[p] == i]
            if not pa and not pb:
                continue
            ps = pa + pb
            dyp = (C.c_void_p * len(ps))(*[dys[p].data_ptr() for p in ps])
            wp = (C.c_void_p * len(ps))(*[ws[p].data_ptr() for p in ps])
            dx = torch.empty_like(xs[i], memory_format=CL)
            if pa:
                rc = L.senas_dwconv_pair_bwd_data(C.byref(ga), len(pa), C.byref(gb) if pb else None, len(pb), dyp, wp, dx.data_ptr(), _stream())
            else:
                rc = L.senas_dwconv_pair_bwd_data(C.byref(gb), len(pb), None, 0, dyp, wp, dx.data_ptr(), _stream())
            _lib.check(rc, 'senas_dwconv_pair_bwd_data')
            dxs[i] = dx
        return (dxs[0], dxs[1], None, None, None, None, None, None) + tuple(dws)


def _dw_geom(x, c0):
    tr = isinstance(c0, torch.nn.ConvTranspose2d)
    n, ci, hi, wi = x.shape
    kk, s, p, d = c0.kernel_size[0], c0.stride[0], c0.padding[0], c0.dilation[0]
    op = c0.output_padding[0] if tr else 0
    ho, wo = conv_out_size(hi, kk, s, p, d, tr, op), conv_out_size(wi, kk, s, p, d, tr, op)
    return ConvGeom(n, hi, wi, ci, ho, wo, ci, kk, kk, s, p, d, int(tr), ci)


def dwconv_multi(x, convs, want_stats):
    """[(z_p, stats_p)] (in the order of ``convs``) of k depthwise convolutions of one tensor that agree in everything but --
    at most two -- kernel sizes, or None when the shape is off the batched path (the caller then runs them one by one)."""
    k = len(convs)
    if not 2 <= k <= _lib.MAX_DWMULTI:
        return None
    sizes = sorted(set(c.kernel_size[0] for c in convs))
    if len(sizes) > 2 or (len(sizes) == 2 and sizes != [3, 5]):
        return None
    order = sorted(range(k), key=lambda i: convs[i].kernel_size[0])             # 3x3 problems first
    groups = [[i for i in order if convs[i].kernel_size[0] == sz] for sz in sizes]
    for grp in groups:
        c0 = convs[grp[0]]
        if c0.groups != c0.in_channels or c0.in_channels != c0.out_channels or c0.in_channels != x.shape[1]:
            return None
        for i in grp[1:]:
            c = convs[i]
            if (type(c), c.weight.shape, c.stride, c.padding, c.dilation, c.groups) != (type(c0), c0.weight.shape, c0.stride, c0.padding,
                                                                                     c0.dilation, c0.groups):
                return None
    ga = _dw_geom(x, convs[groups[0][0]])
    gb = _dw_geom(x, convs[groups[1][0]]) if len(groups) == 2 else None
    ka, kb = len(groups[0]), len(groups[1]) if gb is not None else 0
    if _lib.lib().senas_dwconv_pair_ws_bytes(C.byref(ga), ka, C.byref(gb) if kb else None, kb) == 0:
        return None
    out = _DwMulti.apply(x, ga, ka, gb, kb, bool(want_stats), *[convs[i].weight for i in order])
    res = [None] * k
    for pos, i in enumerate(order):
        res[i] = (out[pos], out[k + pos] if want_stats else None)
    return res


def dwconv_multi2(x0, convs0, x1, convs1, want_stats):
    """dwconv_multi for the candidates of TWO tensors of one shape (the input states of a search cell) in ONE forward and one
    weight-gradient launch: ([(z, stats)] for convs0, [(z, stats)] for convs1), or None off that path."""
    convs = list(convs0) + list(convs1)
    k = len(convs)
    if not (convs0 and convs1) or k > _lib.MAX_DWMULTI or x0.shape != x1.shape:
        return None
    sizes = sorted(set(c.kernel_size[0] for c in convs))
    if len(sizes) > 2 or (len(sizes) == 2 and sizes != [3, 5]):
        return None
    c0 = convs[0]
    for c in convs:
        if c.groups != c.in_channels or c.in_channels != c.out_channels or c.in_channels != x0.shape[1]:
            return None
        if (type(c), c.stride, c.dilation, c.groups) != (type(c0), c0.stride, c0.dilation, c0.groups) or c.padding[0] != c.kernel_size[0] // 2:
            return None
    order = sorted(range(k), key=lambda i: (convs[i].kernel_size[0], i))         # 3x3 problems first, each input's together
    ka = sum(1 for c in convs if c.kernel_size[0] == sizes[0])
    kb = k - ka
    ga = _dw_geom(x0, convs[order[0]])
    gb = _dw_geom(x0, convs[order[-1]]) if kb else None
    if _lib.lib().senas_dwconv_pair_ws_bytes(C.byref(ga), ka, C.byref(gb) if kb else None, kb) == 0:
        return None
    src = tuple(0 if i < len(convs0) else 1 for i in order)
    out = _DwMulti2.apply(x0, x1, ga, ka, gb, kb, src, bool(want_stats), *[convs[i].weight for i in order])
    res = [None] * k
    for pos, i in enumerate(order):
        res[i] = (out[pos], out[k + pos] if want_stats else None)
    return res[:len(convs0)], res[len(convs0):]


class _PwMulti(torch.autograd.Function):
    """k independent 1x1 convolutions of one shape (senas_pw_multi_*).  flat = [x_1..x_k, w_1..w_k]; outputs
    z_1..z_k (+ their statistics)."""

    @staticmethod
    def forward(ctx, k, want_stats, *flat):
        xs = [nhwc(x) for x in flat[:k]]
        ws = [_dev(w).contiguous() for w in flat[k:2 * k]]
        n, cin, h, w_ = xs[0].shape
        cout = ws[0].shape[0]
        ys = [new_nhwc(n, cout, h, w_, xs[0]) for _ in range(k)]
        stats = [new_stats(n, cout, xs[0]) for _ in range(k)] if want_stats else []
        arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])
        _lib.check(_lib.lib().senas_pw_multi_fwd(k, n, h * w_, cin, cout, arr(xs), arr(ws), arr(ys), arr(stats) if want_stats else None,
                                                 _stream()), 'senas_pw_multi_fwd')
        ctx.k, ctx.dims = k, (n, cin, cout, h, w_)
        ctx.save_for_backward(*xs, *ws)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*stats)
        return tuple(ys) + tuple(stats)

    @staticmethod
    def backward(ctx, *grads):
        k, L = ctx.k, _lib.lib()
        n, cin, cout, h, w_ = ctx.dims
        xs, ws = ctx.saved_tensors[:k], ctx.saved_tensors[k:]
        dev = xs[0].device
        dys = [nhwc(d) if d is not None else torch.zeros((n, cout, h, w_), device=dev).contiguous(memory_format=CL) for d in grads[:k]]
        arr = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
        need = ctx.needs_input_grad
        dxs = [torch.empty_like(xs[t], memory_format=CL) if need[2 + t] else None for t in range(k)]
        if any(d is not None for d in dxs):
            _lib.check(L.senas_pw_multi_bwd_data(k, n, h * w_, cin, cout, arr(dys), arr(ws), arr(dxs), _stream()), 'senas_pw_multi_bwd_data')
        dws = [None] * k
        if any(need[2 + k:]):
            dwt, dws = zip(*[wgrad_dest(w) for w in ws])
            scratch = torch.empty(int(L.senas_pw_multi_ws_bytes(k, n, h * w_, cin, cout)), device=dev, dtype=torch.uint8)
            _lib.check(L.senas_pw_multi_bwd_weight(k, n, h * w_, cin, cout, arr(xs), arr(dys), arr(dwt), scratch.data_ptr(), _stream()),
                       'senas_pw_multi_bwd_weight')
        return (None, None) + tuple(dxs) + tuple(dws)


def pw_multi(xs, convs, want_stats):
    """[(z_p, stats_p)] of k same-shape 1x1 convolutions with their own inputs, or None off the batched path."""
    k = len(convs)
    c0 = convs[0]
    if not 2 <= k <= _lib.MAX_PWMULTI or isinstance(c0, torch.nn.ConvTranspose2d):
        return None
    for c in convs:
        if (c.kernel_size, c.stride, c.padding, c.groups, c.weight.shape) != ((1, 1), (1, 1), (0, 0), 1, c0.weight.shape) or c.bias is not None:
            return None
    n, cin, h, w_ = xs[0].shape
    if any(tuple(x.shape) != (n, cin, h, w_) for x in xs):
        return None
    cout = c0.weight.shape[0]
    if _lib.lib().senas_pw_multi_ws_bytes(k, n, h * w_, cin, cout) == 0:
        return None
    out = _PwMulti.apply(k, bool(want_stats), *xs, *[c.weight for c in convs])
    return [(out[i], out[k + i] if want_stats else None) for i in range(k)]


class _BnReluMulti(torch.autograd.Function):
    """relu(BatchNorm2d_t(z_t)) for k independent tensors of one shape in ONE launch (backward: two) --
    senas_bnrelu_multi_fwd / _bwd.  flat = [z_1..z_k, gamma_1..gamma_k, beta_1..beta_k]."""

    @staticmethod
    def forward(ctx, meta, *flat):
        k = meta['k']
        zs = [nhwc(z) for z in flat[:k]]
        gammas = [_dev(g).contiguous() for g in flat[k:2 * k]]
        betas = [_dev(b).contiguous() for b in flat[2 * k:3 * k]]
        n, c, h, w = zs[0].shape
        dev = zs[0].device
        tracked = any(ctx.needs_input_grad)
        ys = [new_nhwc(n, c, h, w, zs[0]) for _ in range(k)]
        masks = [torch.empty(n * h * w * (c // 4), device=dev, dtype=torch.uint8) if tracked else None for _ in range(k)]
        saved = torch.empty((k, 2, c), device=dev, dtype=torch.float32)
        items = (_lib.BnReluItem * k)()
        for t in range(k):
            if tuple(zs[t].shape) != (n, c, h, w):
                raise SenasHipError('bnrelu_multi: tensors disagree in shape')
            rm, rv, nbt = meta['buffers'][t]
            it = items[t]
            it.z, it.y, it.mask8 = zs[t].data_ptr(), ys[t].data_ptr(), _p(masks[t])
            it.stats = _p(meta['stats'][t])
            it.gamma, it.beta = gammas[t].data_ptr(), betas[t].data_ptr()
            it.running_mean, it.running_var, it.num_batches_tracked = _p(rm), _p(rv), _p(nbt)
            it.mean_invstd = saved[t].data_ptr()
        _lib.check(_lib.lib().senas_bnrelu_multi_fwd(items, k, n, h * w, c, int(meta['training']), BN_MOMENTUM, BN_EPS, _stream()),
                   'senas_bnrelu_multi_fwd')
        ctx.k, ctx.shape = k, (n, c, h, w)
        ctx.masks = masks
        ctx.save_for_backward(saved, *zs, *gammas, *betas)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        k = ctx.k
        n, c, h, w = ctx.shape
        saved = ctx.saved_tensors[0]
        zs, gammas, betas = ctx.saved_tensors[1:1 + k], ctx.saved_tensors[1 + k:1 + 2 * k], ctx.saved_tensors[1 + 2 * k:1 + 3 * k]
        dev = saved.device
        live = [t for t in range(k) if dys[t] is not None]
        if not live:
            return (None,) * (1 + 3 * k)
        need = ctx.needs_input_grad
        dzs = [torch.empty_like(zs[t], memory_format=CL) if need[1 + t] else None for t in range(k)]
        # (a tensor that receives no gradient in this pass keeps a scratch destination and hands autograd nothing)
        dgt, dgs = zip(*[wgrad_dest(gammas[t]) if t in live else (None, None) for t in range(k)])
        dbt, dbs = zip(*[wgrad_dest(betas[t]) if t in live else (None, None) for t in range(k)])
        dgs, dbs = list(dgs), list(dbs)
        sums = zeros64((k, n, c, 2), dev)
        items = (_lib.BnReluItem * len(live))()
        keep = []
        for i, t in enumerate(live):
            dy = _dev(dys[t])
            ct = dy.stride(3) if dy.dim() == 4 else 0
            if (tuple(dy.shape) == (n, c, h, w) and ct > c and ct % 4 == 0 and dy.stride() == (h * w * ct, 1, w * ct, ct) and
                    dy.data_ptr() % 16 == 0):
                stride = ct                                   # a channel slice of a wider NHWC tensor, read in place
            else:
                dy, stride = nhwc(dy), c
            keep.append(dy)
            it = items[i]
            it.z, it.mask8, it.gamma = zs[t].data_ptr(), ctx.masks[t].data_ptr(), gammas[t].data_ptr()
            it.mean_invstd = saved[t].data_ptr()
            it.dy, it.dy_pixel_stride = dy.data_ptr(), stride
            it.dz, it.dgamma, it.dbeta = _p(dzs[t]), dgt[t].data_ptr(), dbt[t].data_ptr()
            it.sums = sums[t].data_ptr()
        _lib.check(_lib.lib().senas_bnrelu_multi_bwd(items, len(live), n, h * w, c, _stream()), 'senas_bnrelu_multi_bwd')
        for t in range(k):
            if t not in live:
                dzs[t] = None
        return (None,) + tuple(dzs) + tuple(dgs) + tuple(dbs)


def bnrelu_multi_ok(zs, bns):
    """Can these k tensors share the batched BatchNorm + ReLU launches?  (Same shape, 4..64 channels in quads, k <= 8;
    eval-mode modules only without autograd -- the kernel pair has no eval-mode backward.)"""
    if not 1 <= len(zs) <= _lib.MAX_BNRELU:
        return False
    n, c, h, w = zs[0].shape
    if c % 4 != 0 or c > 64 or 256 % (c // 4) != 0:
        return False
    training = bns[0].training
    if any(bn.training != training for bn in bns) or any(tuple(z.shape) != (n, c, h, w) for z in zs):
        return False
    return training or not (torch.is_grad_enabled() and any(z.requires_grad for z in zs))


def bnrelu_multi(zs, bns, stats):
    """[relu(bn_t(z_t))] for the k (tensor, BatchNorm2d, producer statistics) triples, one launch."""
    k = len(zs)
    training = bns[0].training
    st = list(stats)
    for t in range(k):
        if training and st[t] is None:
            st[t] = chan_stats(zs[t])
    meta = {'k': k, 'training': training, 'stats': st,
            'buffers': [(bn.running_mean, bn.running_var, bn.num_batches_tracked) for bn in bns]}
    flat = list(zs) + [bn.weight for bn in bns] + [bn.bias for bn in bns]
    return list(_BnReluMulti.apply(meta, *flat))


class _DsTail(torch.autograd.Function):
    """z2_t = W_t relu(BatchNorm2d_t(z1_t)) for k DepSepConv candidates in ONE launch, two backward (senas_dstail_*): the
    batch-norm + ReLU of the depthwise half is applied on load by the pointwise half, forward and backward, so the
    activated tensor and its gradient never exist in memory.  flat = [z1.., gamma1.., beta1.., w..]; outputs z2.. (+ the
    producer-side statistics of every z2 for the BatchNorm2d that follows)."""

    @staticmethod
    def forward(ctx, meta, *flat):
        k = meta['k']
        z1s = [nhwc(z) for z in flat[:k]]
        gammas = [_dev(g).contiguous() for g in flat[k:2 * k]]
        betas = [_dev(b).contiguous() for b in flat[2 * k:3 * k]]
        ws = [_dev(w).contiguous() for w in flat[3 * k:4 * k]]
        n, cin, h, w_ = z1s[0].shape
        cout = ws[0].shape[0]
        dev = z1s[0].device
        z2s = [new_nhwc(n, cout, h, w_, z1s[0]) for _ in range(k)]
        stats2 = [new_stats(n, cout, z1s[0]) for _ in range(k)] if meta['want_stats'] else []
        saved = torch.empty((k, 2, cin), device=dev, dtype=torch.float32)
        items = (_lib.DsTailItem * k)()
        for t in range(k):
            if tuple(z1s[t].shape) != (n, cin, h, w_) or tuple(ws[t].shape[:2]) != (cout, cin):
                raise SenasHipError('dstail: problems disagree in shape')
            rm, rv, nbt = meta['buffers'][t]
            it = items[t]
            it.z1, it.stats1 = z1s[t].data_ptr(), _p(meta['stats'][t])
            it.gamma1, it.beta1 = gammas[t].data_ptr(), betas[t].data_ptr()
            it.running_mean1, it.running_var1, it.num_batches_tracked1 = _p(rm), _p(rv), _p(nbt)
            it.mean_invstd = saved[t].data_ptr()
            it.w, it.z2 = ws[t].data_ptr(), z2s[t].data_ptr()
            it.stats2 = stats2[t].data_ptr() if stats2 else None
        _lib.check(_lib.lib().senas_dstail_fwd(items, k, n, h * w_, cin, cout, int(meta['training']), BN_MOMENTUM, BN_EPS, _stream()),
                   'senas_dstail_fwd')
        ctx.k, ctx.dims = k, (n, cin, cout, h, w_)
        ctx.save_for_backward(saved, *z1s, *gammas, *betas, *ws)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*stats2)
        return tuple(z2s) + tuple(stats2)

    @staticmethod
    def backward(ctx, *grads):
        k, L = ctx.k, _lib.lib()
        n, cin, cout, h, w_ = ctx.dims
        saved = ctx.saved_tensors[0]
        z1s, gammas, betas, ws = (ctx.saved_tensors[1 + i * k:1 + (i + 1) * k] for i in range(4))
        dev = saved.device
        if all(g is None for g in grads[:k]):
            return (None,) * (1 + 4 * k)
        need = ctx.needs_input_grad
        want_w = any(need[1 + 3 * k:1 + 4 * k])
        dz1s = [torch.empty_like(z1s[t], memory_format=CL) if need[1 + t] else None for t in range(k)]
        dgt, dgs = zip(*[wgrad_dest(g) for g in gammas])
        dbt, dbs = zip(*[wgrad_dest(b) for b in betas])
        dwt, dws = zip(*[wgrad_dest(w) for w in ws]) if want_w else ((None,) * k, (None,) * k)
        sums = zeros64((k, n, cin, 2), dev)
        # fp64 accumulators of the k weight gradients (senas_dstail_ws_bytes per problem: one image of cout x cin per batch image)
        dw_acc = zeros64((k, int(L.senas_dstail_ws_bytes(k, n, h * w_, cin, cout)) // 8), dev) if want_w else None
        items = (_lib.DsTailItem * k)()
        keep = []
        for t in range(k):
            dy = grads[t]
            if dy is None:
                dy = torch.zeros((n, cout, h, w_), device=dev).contiguous(memory_format=CL)
            _dev(dy)
            ct = dy.stride(3) if dy.dim() == 4 else 0
            if tuple(dy.shape) == (n, cout, h, w_) and ct > cout and dy.stride() == (h * w_ * ct, 1, w_ * ct, ct):
                stride = ct                                   # a channel slice of a wider NHWC tensor, read in place
            else:
                dy, stride = nhwc(dy), cout
            keep.append(dy)
            it = items[t]
            it.z1, it.gamma1, it.beta1 = z1s[t].data_ptr(), gammas[t].data_ptr(), betas[t].data_ptr()
            it.mean_invstd, it.w = saved[t].data_ptr(), ws[t].data_ptr()
            it.dz2, it.dz2_pixel_stride, it.sums = dy.data_ptr(), stride, sums[t].data_ptr()
            it.dz1, it.dgamma1, it.dbeta1 = _p(dz1s[t]), dgt[t].data_ptr(), dbt[t].data_ptr()
            if want_w:
                it.dw, it.dw_acc = dwt[t].data_ptr(), dw_acc[t].data_ptr()
        _lib.check(L.senas_dstail_bwd(items, k, n, h * w_, cin, cout, _stream()), 'senas_dstail_bwd')
        return (None,) + tuple(dz1s) + tuple(dgs) + tuple(dbs) + tuple(dws)


def dstail(z1s, bns, stats, convs, want_stats):
    """[(z2_t, stats2_t)] for k (depthwise output, its BatchNorm2d, its producer statistics, the 1x1 convolution that
    follows the ReLU) quadruples, or None when the shapes are off the fused path (the caller then runs batch-norm + ReLU and
    the pointwise convolutions as separate launches)."""
    k = len(z1s)
    if not 1 <= k <= _lib.MAX_DSTAIL:
        return None
    n, cin, h, w_ = z1s[0].shape
    c0 = convs[0]
    cout = c0.weight.shape[0]
    if isinstance(c0, torch.nn.ConvTranspose2d) or any(tuple(z.shape) != (n, cin, h, w_) for z in z1s):
        return None
    for c in convs:
        if (c.kernel_size, c.stride, c.padding, c.groups, tuple(c.weight.shape)) != ((1, 1), (1, 1), (0, 0), 1, (cout, cin, 1, 1)) or c.bias is not None:
            return None
    training = bns[0].training
    if any(bn.training != training for bn in bns):
        return None
    if not training and torch.is_grad_enabled() and any(z.requires_grad for z in z1s):
        return None                                          # (the kernel pair has no eval-mode backward)
    if _lib.lib().senas_dstail_ws_bytes(k, n, h * w_, cin, cout) == 0:
        return None
    st = list(stats)
    for t in range(k):
        if training and st[t] is None:
            st[t] = chan_stats(z1s[t])
    meta = {'k': k, 'training': training, 'stats': st, 'want_stats': bool(want_stats),
            'buffers': [(bn.running_mean, bn.running_var, bn.num_batches_tracked) for bn in bns]}
    flat = list(z1s) + [bn.weight for bn in bns] + [bn.bias for bn in bns] + [c.weight for c in convs]
    out = _DsTail.apply(meta, *flat)
    return [(out[t], out[k + t] if want_stats else None) for t in range(k)]


class _StackFn(torch.autograd.Function):
    """The stacked weight buffer as a differentiable function of the per-edge parameters it is assembled from: the
    forward pass hands out the (already filled) buffer, the backward pass hands every parameter its slice of d buffer."""

    @staticmethod
    def forward(ctx, buf, dim, *params):
        ctx.dim, ctx.sizes = dim, [p.shape[dim] for p in params]
        ctx.set_materialize_grads(False)
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, dw):
        if dw is None:              # written into the stack's gradient buffer of the flat-gradient sink (gradsink.py)
            return (None,) * (2 + len(ctx.sizes))
        grads, off = [], 0
        for size in ctx.sizes:
            grads.append(dw.narrow(ctx.dim, off, size))
            off += size
        return (None, None) + tuple(grads)


class StackedWeight(object):
    """Weights of k same-shaped convolutions concatenated along the output-channel dimension in a persistent buffer.
    Unmanaged (no step driver): refilled on every use, one ``cat`` launch.  Managed by a ``WeightPacker``: the packer
    refills every stacked buffer of the model (one multi-tensor copy) and repacks its MFMA image together with all
    other weights at the start of a pass, so using it costs no launch at all."""

    def __init__(self, params, dim, pad_parts=0, pad_to=None):
        self.params, self.dim, self.pad_parts = list(params), dim, pad_parts
        self.pad_to = pad_to                   # total size along dim (zero-filled beyond the parameters), overrides pad_parts
        self.buf = None
        self.managed = False
        self.packer = None                     # the WeightPacker that keeps the buffer filled while ``managed``
        self.filled = None                     # the parameters' version counters when a packer last filled the buffer
        self.eager_fill = None                 # (version counters, WEIGHT_GEN) of the last unmanaged fill by tensor()

    def __deepcopy__(self, memo):
        # a copied model gets an unmanaged stack over ITS parameters, not a copy of the packer / gradient buffers behind this one
        import copy
        return StackedWeight([copy.deepcopy(p, memo) for p in self.params], self.dim, self.pad_parts, self.pad_to)

    def mark_filled(self):
        self.filled = tuple(p._version for p in self.params)

    def current(self):
        return (self.managed and self.packer is not None and self.packer.gen == WEIGHT_GEN and
                self.filled == tuple(p._version for p in self.params))

    def buffer(self):
        p0 = self.params[0]
        if self.buf is None or self.buf.device != p0.device:
            shape = list(p0.shape)
            shape[self.dim] = sum(p.shape[self.dim] for p in self.params) + self.pad_parts * p0.shape[self.dim]
            if self.pad_to is not None:
                shape[self.dim] = max(shape[self.dim], self.pad_to)
            # zero-filled: the padding parts (zero weights that bring a stack of three to a full 32-channel tile) stay zero
            self.buf = torch.zeros(shape, device=p0.device, dtype=p0.dtype)
            self.managed = False
            self.eager_fill = None
        return self.buf

    def slices(self):
        buf, out, off = self.buffer(), [], 0
        for p in self.params:
            out.append(buf.narrow(self.dim, off, p.shape[self.dim]))
            off += p.shape[self.dim]
        return out

    def tensor(self):
        buf = self.buffer()
        if not self.current():
            # unmanaged: refill -- but only when a parameter moved since the last fill.  A module applied more than once per
            # pass (the shared head under deep supervision) must not rewrite the buffer between its applications: the
            # earlier application's convolution saved a view of it for its backward pass
            stamp = (tuple((p._version, p.data_ptr()) for p in self.params), WEIGHT_GEN)
            if self.eager_fill != stamp:
                with torch.no_grad():
                    for dst, p in zip(self.slices(), self.params):
                        dst.copy_(p)
                self.eager_fill = stamp
        return _StackFn.apply(buf, self.dim, *self.params)


class GradLanding(object):
    """The stacked gradient buffer [n, k*c, h, w] of an un-stacked tensor, allocated by the first consumer that writes its
    part during a backward pass; when every part was written in place the un-stacking's backward pass hands the buffer
    on as it is (no concatenation launch)."""

    def __init__(self, k, shape, used=None, persistent=False):
        self.k, self.shape = k, shape                      # shape of ONE part (n, c, h, w)
        self.used = k if used is None else used            # parts beyond `used` are zero-weight padding: their gradient is zero
        # persistent: the buffer lives across passes (owned by the stack it serves), so its padding parts are zeroed once,
        # when it is allocated, and never again -- every pass rewrites exactly the used parts
        self.persistent = persistent
        self.buf = None

    def part(self, e, like):
        n, c, h, w = self.shape
        if self.buf is None or self.buf.device != like.device:
            make = torch.zeros if (self.persistent and self.used < self.k) else torch.empty
            self.buf = make((n, self.k * c, h, w), device=like.device, dtype=torch.float32).contiguous(memory_format=CL)
        return self.buf[:, e * c:(e + 1) * c]

    def take(self, grads):
        """The buffer if ``grads`` are exactly its parts, else None; either way the buffer is released."""
        buf = self.buf
        if not self.persistent:
            self.buf = None
        if buf is None:
            return None
        n, c, h, w = self.shape
        if len(grads) != self.used:
            return None
        for e, g in enumerate(grads):
            if g is None or g.data_ptr() != buf.data_ptr() + 4 * e * c or tuple(g.shape) != self.shape or g.stride() != buf[:, :c].stride():
                return None
        if self.used < self.k and not self.persistent:
            buf[:, self.used * c:].zero_()
        return buf


class _Unstack(torch.autograd.Function):
    """[n, k*c, h, w] -> k tensors [n, c, h, w] (+ their producer-side batch-norm statistics); the backward pass is the
    channel concatenation of the k gradients."""

    @staticmethod
    def forward(ctx, z, k, want_stats, landing, used):
        z = nhwc(z)
        n, kc, h, w = z.shape
        if kc % k != 0 or not 1 <= used <= k <= _lib.MAX_STACK:
            raise SenasHipError('unstack: %d channels into %d parts' % (kc, k))
        c = kc // k
        parts = [new_nhwc(n, c, h, w, z) for _ in range(used)]
        stats = [new_stats(n, c, z) for _ in range(used)] if want_stats else []
        pad = [None] * (k - used)
        dp = (C.c_void_p * k)(*([t.data_ptr() for t in parts] + pad))
        sp = (C.c_void_p * k)(*([t.data_ptr() for t in stats] + pad)) if want_stats else None
        _lib.check(_lib.lib().senas_unstack_fwd(n, h * w, c, k, z.data_ptr(), dp, sp, _stream()), 'senas_unstack_fwd')
        ctx.k, ctx.shape = used, (n, c, h, w)
        ctx.k_total = k
        ctx.landing = landing
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*stats)
        return tuple(parts) + tuple(stats)

    @staticmethod
    def backward(ctx, *grads):
        gs = list(grads[:ctx.k])
        if all(g is None for g in gs):
            return None, None, None, None, None
        if ctx.landing is not None:
            whole = ctx.landing.take(gs)
            if whole is not None:                          # every consumer wrote its part in place
                return whole, None, None, None, None
        ref = next(g for g in gs if g is not None)
        gs = [g if g is not None else torch.zeros(ctx.shape, device=ref.device, dtype=ref.dtype).contiguous(memory_format=CL) for g in gs]
        gs += [torch.zeros_like(ref) for _ in range(ctx.k_total - ctx.k)]           # zero-weight padding parts
        return torch.cat(gs, dim=1).contiguous(memory_format=CL), None, None, None, None


class _UnstackView(torch.autograd.Function):
    """[n, k*c, h, w] -> k channel slices [n, c, h, w] that ALIAS the stacked tensor: no kernel runs -- the consumers (the
    cell node kernels) read a slice in place through its pixel stride, and the matching slice of the stacked convolution's
    own producer-side statistics.  Backward as _Unstack: the consumers write their gradients straight into the stacked
    gradient buffer (GradLanding)."""

    @staticmethod
    def forward(ctx, z, k, landing, used):
        ctx.five = z.dim() == 5
        parts = []
        if ctx.five:
            # a stacked convolution's 5-D output [k][n][c][h][w], planar or interleaved: part e is z[e] with z's own strides
            _dev(z)
            if z.shape[0] != k:
                raise SenasHipError('unstack: a %d-part stacked tensor into %d parts' % (z.shape[0], k))
            n, c, h, w = z.shape[1:]
            for e in range(used):
                parts.append(torch.empty(0, device=z.device, dtype=z.dtype).set_(z.untyped_storage(), z.storage_offset() + e * z.stride(0),
                                                                                 (n, c, h, w), tuple(z.stride()[1:])))
        else:
            z = nhwc(z)
            n, kc, h, w = z.shape
            c = kc // k
            for e in range(used):
                # the same storage, not a view in autograd's eyes (this Function owns the backward pass)
                parts.append(torch.empty(0, device=z.device, dtype=z.dtype).set_(z.untyped_storage(), z.storage_offset() + e * c,
                                                                                 (n, c, h, w), (h * w * kc, 1, w * kc, kc)))
        ctx.k, ctx.shape, ctx.k_total, ctx.landing = used, (n, c, h, w), k, landing
        ctx.set_materialize_grads(False)
        return tuple(parts)

    @staticmethod
    def backward(ctx, *grads):
        out = _Unstack.backward(ctx, *grads)[:4]
        if ctx.five and out[0] is not None:                  # (the gradient of a 5-D stacked tensor: the 5-D view of the interleaved one)
            out = (stacked_5d(out[0], ctx.k_total),) + tuple(out[1:])
        return out


def unstack(z, k, want_stats=True, used=None, owner=None, stats=None):
    """The per-edge parts of a stacked convolution output: [(z_e, stats_e or None, grad_slot_e)] for the first ``used`` of
    its k parts (the rest is zero-weight padding).  ``owner``: an object (the stack's StackedWeight) that keeps the
    gradient landing buffer of this shape from pass to pass.  ``stats``: the stacked convolution's own statistics
    (double [n, k*c, 2]) -- with them (or when none are wanted) the parts are aliases of ``z`` and nothing is launched."""
    used = k if used is None else used
    if z.dim() == 5:
        n, kc, h, w = z.shape[1], z.shape[0] * z.shape[2], z.shape[3], z.shape[4]
    else:
        n, kc, h, w = z.shape
    landing = None
    if (kc // k) % 4 == 0:
        shape = (n, kc // k, h, w)
        if owner is not None:
            cache = owner.__dict__.setdefault('landings', {})
            landing = cache.get((k, used, shape))
            if landing is None:
                landing = cache[(k, used, shape)] = GradLanding(k, shape, used, persistent=True)
        else:
            landing = GradLanding(k, shape, used)
    c = kc // k
    if c % 4 == 0 and (stats is not None or not want_stats):
        out = _UnstackView.apply(z, k, landing, used)
        return [(out[e], stats[:, e * c:(e + 1) * c] if stats is not None else None, (landing, e) if landing is not None else None)
                for e in range(used)]
    out = _Unstack.apply(stacked_4d(z) if z.dim() == 5 else z, k, want_stats, landing, used)
    return [(out[e], out[used + e] if want_stats else None, (landing, e) if landing is not None else None) for e in range(used)]


def chan_stats(z):
    """Per-image per-channel (sum, sum of squares) of an NHWC tensor, fp64 [n][c][2]."""
    z = nhwc(z)
    n, c, h, w = z.shape
    st = new_stats(n, c, z)
    _lib.check(_lib.lib().senas_chan_stats(n, h * w, c, z.data_ptr(), st.data_ptr(), _stream()), 'senas_chan_stats')
    return st


# ------------------------------------------------------------------------------------------ node terms
class Term(object):
    """One addend of a node: a raw tensor ``z`` (or None for the all-zero input of the 'none'
    op) that still has to go through its own BatchNorm2d ``bn`` and, for se_conv_3, its SE gate.
    ``stats`` are producer-side statistics of ``z`` when the producer kernel already has them.
    ``passengers`` are parameters that must receive an exactly-zero gradient (the 1x1 adapter
    conv behind a zero input), as autograd gives them in the reference."""

    __slots__ = ('z', 'bn', 'se', 'stats', 'passengers', 'grad_slot')

    def __init__(self, z, bn, se=None, stats=None, passengers=(), grad_slot=None):
        self.z, self.bn, self.se, self.stats, self.passengers = z, bn, se, stats, tuple(passengers)
        # (GradLanding, part): where the consumer should WRITE d loss / d z -- a channel slice of the stacked gradient
        # buffer the producer's backward pass needs anyway (functional.unstack); None: a tensor of its own
        self.grad_slot = grad_slot


class _CatSlices(torch.autograd.Function):
    """The concatenation buffer of a cell as a function of the node outputs that were written into its channel slices by
    their own kernels (node.bn_combine(cat=...)): no copy forward; backward hands every node its slice of d buffer."""

    @staticmethod
    def forward(ctx, buf, width, *ys):
        ctx.width, ctx.k = width, len(ys)
        out = torch.empty(0, device=buf.device, dtype=buf.dtype).set_(buf.untyped_storage(), buf.storage_offset(), buf.shape, buf.stride())
        return out

    @staticmethod
    def backward(ctx, dbuf):
        c = ctx.width
        return (None, None) + tuple(dbuf[:, i * c:(i + 1) * c] for i in range(ctx.k))


def cat_slices(buf, width, ys):
    return _CatSlices.apply(buf, width, *ys)


def bn_combine(terms, mix=None, residual=None, relu=False, out_stats=False, cat=None):
    """See senas_amd.node.bn_combine (the fused cell node)."""
    from .node import bn_combine as _impl
    return _impl(terms, mix=mix, residual=residual, relu=relu, out_stats=out_stats, cat=cat)
