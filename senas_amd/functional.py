"""torch.autograd.Function wrappers over the C ABI (include/senas_hip.h).

Tensors keep the reference's logical NCHW shape but live in ``torch.channels_last`` memory, i.e.
NHWC on the device -- that is what the kernels index.  torch is used for device memory, streams
and the autograd graph only; every pass over an activation tensor is a libsenas_hip kernel.

There is no CPU path: a non-CUDA tensor raises.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvGeom, SenasHipError

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


def new_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.float32, memory_format=CL)


def _p(t):
    return None if t is None else t.data_ptr()


def new_stats(n, c, like):
    return torch.zeros((n, c, 2), device=like.device, dtype=torch.float64)


class KernelTimer(object):
    """HIP-event timing of individual launches on the stream they are launched on (torch's current
    stream), used by bench.py for the roofline of the dominant kernel.  Off unless installed."""

    def __init__(self):
        self.records = []

    def span(self, name, flops, nbytes):
        return _Span(self, name, flops, nbytes)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            a['launches'] += 1
            a['ms'] += e0.elapsed_time(e1)
            a['flops'] += flops
            a['bytes'] += nbytes
        return agg


class _Span(object):
    def __init__(self, timer, name, flops, nbytes):
        self.t, self.name, self.flops, self.nbytes = timer, name, flops, nbytes

    def __enter__(self):
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        self.e1.record()
        self.t.records.append((self.name, self.flops, self.nbytes, self.e0, self.e1))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


TIMER = None
_NOSPAN = _NoSpan()


def _span(kind, g, x, w, y):
    if TIMER is None:
        return _NOSPAN
    dense = g.groups == 1
    macs = (g.n * g.hi * g.wi * g.ci * (g.co // g.groups) if g.transposed else g.n * g.ho * g.wo * g.co * (g.ci // g.groups)) * g.kh * g.kw
    shape = '%s%dx%d d%d s%d %s%d->%d' % ('T' if g.transposed else '', g.kh, g.kw, g.dil, g.stride, '' if dense else 'dw ', g.ci, g.co)
    return TIMER.span('%s[%s]' % (kind, shape), 2.0 * macs, 4.0 * (x.numel() + y.numel() + w.numel()))


# ------------------------------------------------------------------------------------------ convolution
def conv_out_size(i, k, stride, pad, dil, transposed, out_pad):
    if transposed:
        return (i - 1) * stride - 2 * pad + dil * (k - 1) + out_pad + 1
    return (i + 2 * pad - dil * (k - 1) - 1) // stride + 1


class _Conv2d(torch.autograd.Function):
    """y = conv(relu?(x), w) (+ producer-side batch-norm statistics of y)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, dil, transposed, out_pad, groups, in_relu, want_stats):
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
        y = new_nhwc(n, co, ho, wo, x)
        stats = new_stats(n, co, x) if want_stats else None
        ws = torch.empty(int(L.senas_conv2d_ws_bytes(C.byref(g))), device=x.device, dtype=torch.uint8)
        with _span('conv_fwd', g, x, w, y):
            _lib.check(L.senas_conv2d_fwd(C.byref(g), x.data_ptr(), w.data_ptr(), y.data_ptr(), int(in_relu), _p(stats),
                                          ws.data_ptr(), _stream()), 'senas_conv2d_fwd')
        ctx.save_for_backward(x, w)
        ctx.g, ctx.in_relu = g, int(in_relu)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds):
        x, w = ctx.saved_tensors
        g, L = ctx.g, _lib.lib()
        dy = nhwc(dy)
        ws = torch.empty(int(L.senas_conv2d_ws_bytes(C.byref(g))), device=x.device, dtype=torch.uint8)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=CL)
            with _span('conv_dgrad', g, x, w, dy):
                _lib.check(L.senas_conv2d_bwd_data(C.byref(g), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctx.in_relu,
                                                   x.data_ptr(), ws.data_ptr(), _stream()), 'senas_conv2d_bwd_data')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with _span('conv_wgrad', g, x, w, dy):
                _lib.check(L.senas_conv2d_bwd_weight(C.byref(g), x.data_ptr(), ctx.in_relu, dy.data_ptr(), dw.data_ptr(),
                                                     ws.data_ptr(), _stream()), 'senas_conv2d_bwd_weight')
        return dx, dw, None, None, None, None, None, None, None, None


def conv2d(x, w, stride=1, pad=0, dil=1, transposed=False, out_pad=0, groups=1, in_relu=False, want_stats=False):
    """Returns (y, stats) -- stats is None unless want_stats."""
    return _Conv2d.apply(x, w, stride, pad, dil, transposed, out_pad, groups, in_relu, want_stats)


# ------------------------------------------------------------------------------------------ pooling / resampling
class _AvgPool3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, in_relu):
        x = nhwc(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = new_nhwc(n, c, ho, wo, x)
        _lib.check(_lib.lib().senas_avgpool3_fwd(n, h, w, c, stride, x.data_ptr(), int(in_relu), y.data_ptr(), None,
                                                 _stream()), 'senas_avgpool3_fwd')
        ctx.save_for_backward(x)
        ctx.meta = (stride, int(in_relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        stride, in_relu = ctx.meta
        n, c, h, w = x.shape
        dy = nhwc(dy)
        dx = torch.empty_like(x, memory_format=CL)
        _lib.check(_lib.lib().senas_avgpool3_bwd(n, h, w, c, stride, dy.data_ptr(), in_relu, x.data_ptr(), dx.data_ptr(),
                                                 _stream()), 'senas_avgpool3_bwd')
        return dx, None, None


class _MaxPool3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, in_relu):
        x = nhwc(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = new_nhwc(n, c, ho, wo, x)
        arg = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().senas_maxpool3_fwd(n, h, w, c, stride, x.data_ptr(), int(in_relu), y.data_ptr(),
                                                 arg.data_ptr(), None, _stream()), 'senas_maxpool3_fwd')
        ctx.save_for_backward(x, arg)
        ctx.meta = (stride, int(in_relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, arg = ctx.saved_tensors
        stride, in_relu = ctx.meta
        n, c, h, w = x.shape
        dy = nhwc(dy)
        dx = torch.empty_like(x, memory_format=CL)
        _lib.check(_lib.lib().senas_maxpool3_bwd(n, h, w, c, stride, dy.data_ptr(), arg.data_ptr(), in_relu, x.data_ptr(),
                                                 dx.data_ptr(), _stream()), 'senas_maxpool3_bwd')
        return dx, None, None


class _Bilinear2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, 2 * h, 2 * w, x)
        _lib.check(_lib.lib().senas_bilinear2x_fwd(n, h, w, c, x.data_ptr(), y.data_ptr(), None, _stream()),
                   'senas_bilinear2x_fwd')
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy)
        _lib.check(_lib.lib().senas_bilinear2x_bwd(n, h, w, c, dy.data_ptr(), dx.data_ptr(), _stream()),
                   'senas_bilinear2x_bwd')
        return dx


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


def avg_pool3(x, stride, in_relu=False):
    return _AvgPool3.apply(x, stride, in_relu)


def max_pool3(x, stride, in_relu=False):
    return _MaxPool3.apply(x, stride, in_relu)


def bilinear2x(x):
    return _Bilinear2x.apply(x)


def relu(x):
    return _ReLU.apply(x)


def chan_stats(z):
    """Per-image per-channel (sum, sum of squares) of an NHWC tensor, fp64 [n][c][2]."""
    z = nhwc(z)
    n, c, h, w = z.shape
    st = new_stats(n, c, z)
    _lib.check(_lib.lib().senas_chan_stats(n, h * w, c, z.data_ptr(), st.data_ptr(), _stream()), 'senas_chan_stats')
    return st


# ------------------------------------------------------------------------------------------ normalise + mix + activate
class Term(object):
    """One addend of a node: a raw tensor ``z`` (or None for the all-zero input of the 'none'
    op) that still has to go through its own BatchNorm2d ``bn`` and, for se_conv_3, its SE gate.
    ``stats`` are producer-side statistics of ``z`` when the producer kernel already has them.
    ``passengers`` are parameters that must receive an exactly-zero gradient (the 1x1 adapter
    conv behind a zero input), as autograd gives them in the reference."""

    __slots__ = ('z', 'bn', 'se', 'stats', 'passengers')

    def __init__(self, z, bn, se=None, stats=None, passengers=()):
        self.z, self.bn, self.se, self.stats, self.passengers = z, bn, se, stats, tuple(passengers)


class _BNCombine(torch.autograd.Function):
    """y = act( sum_t mix_t * gate_t * BN_t(z_t) + residual ), one pass over every z_t.

    flat = [z (real terms)..., gamma (all terms)..., beta (all terms)..., se_w1 (se terms)..., se_w2 ..., passengers...]
    """

    @staticmethod
    def forward(ctx, meta, mix, residual, *flat):
        L = _lib.lib()
        T, real, se_ids = meta['T'], meta['real'], meta['se']
        nr, ns = len(real), len(se_ids)
        zs = [nhwc(z) for z in flat[:nr]]
        gammas, betas = flat[nr:nr + T], flat[nr + T:nr + 2 * T]
        w1s, w2s = flat[nr + 2 * T:nr + 2 * T + ns], flat[nr + 2 * T + ns:nr + 2 * T + 2 * ns]
        n, c, h, w = meta['shape']
        hw = h * w
        training = meta['training']
        dev = gammas[0].device
        for z in zs:
            if tuple(z.shape) != (n, c, h, w):
                raise SenasHipError('node terms disagree in shape: %s vs %s' % (tuple(z.shape), (n, c, h, w)))
        # batch statistics (producer-side where available)
        stats = torch.zeros((T, n, c, 2), device=dev, dtype=torch.float64)
        for k, t in enumerate(real):
            st = meta['stats'][t]
            if st is None and (training or t in se_ids):
                st = chan_stats(zs[k])
            if st is not None:
                stats[t].copy_(st)
        coefs = torch.empty((T, 4, c), device=dev, dtype=torch.float32)      # mean, invstd, scale, shift
        for t in range(T):
            rm, rv, nbt = meta['buffers'][t]
            _lib.check(L.senas_bn_finalize(n, hw, c, stats[t].data_ptr(), gammas[t].data_ptr(), betas[t].data_ptr(),
                                           _p(rm), _p(rv), _p(nbt), BN_MOMENTUM, BN_EPS, int(training),
                                           coefs[t, 0].data_ptr(), coefs[t, 1].data_ptr(), coefs[t, 2].data_ptr(),
                                           coefs[t, 3].data_ptr(), _stream()), 'senas_bn_finalize')
        scale, shift = coefs[:, 2], coefs[:, 3]
        gate = torch.ones((T, n, c), device=dev, dtype=torch.float32)
        se_saved = []
        for k, t in enumerate(se_ids):
            m = (scale[t].double() * (stats[t, :, :, 0] / hw) + shift[t].double()).float()          # [n, c]
            a1 = m @ w1s[k].t()
            hdn = torch.relu(a1)
            gt = torch.sigmoid(hdn @ w2s[k].t())
            gate[t] = gt
            se_saved.append((m, a1, hdn, gt))
        wmix = mix.detach().float() if mix is not None else torch.ones(T, device=dev)
        wg = wmix.view(T, 1, 1) * gate                                       # [T, n, c]
        coef = (wg * scale.view(T, 1, c)).contiguous()
        bias = (wg * shift.view(T, 1, c)).sum(0).contiguous()
        y = torch.empty((n, c, h, w), device=dev, dtype=torch.float32, memory_format=CL)
        res = nhwc(residual) if residual is not None else None
        zp = _lib.ptr_array([z.data_ptr() for z in zs])
        coef_real = coef[real].contiguous() if nr != T else coef
        _lib.check(L.senas_combine_fwd(n, hw, c, nr, zp, coef_real.data_ptr(), bias.data_ptr(), _p(res),
                                       int(meta['relu']), y.data_ptr(), _stream()), 'senas_combine_fwd')
        ctx.meta = meta
        ctx.has_mix, ctx.has_res = mix is not None, residual is not None
        ctx.nflat = len(flat)
        ctx.se_saved = se_saved
        ctx.save_for_backward(y, stats, coefs, gate, wmix, *zs, *w1s, *w2s)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        meta = ctx.meta
        T, real, se_ids = meta['T'], meta['real'], meta['se']
        nr, ns = len(real), len(se_ids)
        n, c, h, w = meta['shape']
        hw, M = h * w, float(n * h * w)
        saved = ctx.saved_tensors
        y, stats, coefs, gate, wmix = saved[:5]
        zs = saved[5:5 + nr]
        w1s, w2s = saved[5 + nr:5 + nr + ns], saved[5 + nr + ns:5 + nr + 2 * ns]
        dev = y.device
        dy = nhwc(dy)
        relu = int(meta['relu'])
        zp = _lib.ptr_array([z.data_ptr() for z in zs])
        p1 = torch.zeros((n, c), device=dev, dtype=torch.float64)
        p2r = torch.zeros((max(nr, 1), n, c), device=dev, dtype=torch.float64)
        _lib.check(L.senas_combine_bwd_reduce(n, hw, c, nr, zp, dy.data_ptr(), y.data_ptr(), relu, p1.data_ptr(),
                                              p2r.data_ptr(), _stream()), 'senas_combine_bwd_reduce')
        p2 = torch.zeros((T, n, c), device=dev, dtype=torch.float64)
        if nr:
            p2[real] = p2r[:nr]
        mean, invstd, scale, shift = (coefs[:, i].double() for i in range(4))       # [T, c]
        g64, w64 = gate.double(), wmix.double()
        Z = stats[..., 0]                                                            # [T, n, c]
        # d loss / d (mix_t * gate_t): full-tensor dot product of ds with BN_t(z_t), per (n, c)
        dot = scale.view(T, 1, c) * p2 + shift.view(T, 1, c) * p1.view(1, n, c)      # [T, n, c]
        dmix = (g64 * dot).sum((1, 2)) if ctx.has_mix else None
        e = torch.zeros((T, n, c), device=dev, dtype=torch.float64)
        dw1s, dw2s = [], []
        for k, t in enumerate(se_ids):
            m, a1, hdn, gt = ctx.se_saved[k]
            dg = (w64[t] * dot[t]).float()
            da2 = dg * gt * (1 - gt)
            dw2s.append(da2.t() @ hdn)
            dh = da2 @ w2s[k]
            da1 = dh * (a1 > 0).float()
            dw1s.append(da1.t() @ m)
            e[t] = (da1 @ w1s[k]).double() / hw
        u1 = w64.view(T, 1, 1) * g64                                                 # [T, n, c]
        s1 = (u1 * p1.view(1, n, c) + hw * e).sum(1)                                 # [T, c]
        s2 = (u1 * p2 + e * Z).sum(1)
        dbeta = s1
        kk = s2 - mean * s1
        dgamma = invstd * kk
        A = scale.view(T, 1, c) * u1
        if meta['training']:
            B = (-scale * invstd * invstd * kk / M).view(T, 1, c).expand(T, n, c)
            Cc = scale.view(T, 1, c) * e + (-scale * s1 / M + scale * invstd * invstd * mean * kk / M).view(T, 1, c)
        else:
            B = torch.zeros((T, n, c), device=dev, dtype=torch.float64)
            Cc = scale.view(T, 1, c) * e
        need = ctx.needs_input_grad
        dzs = [None] * nr
        ds_out = None
        want_dz = [need[3 + k] for k in range(nr)]
        if any(want_dz) or (ctx.has_res and need[2]):
            for k in range(nr):
                if want_dz[k]:
                    dzs[k] = torch.empty_like(zs[k], memory_format=CL)
            if ctx.has_res and need[2]:
                ds_out = torch.empty_like(y, memory_format=CL)
            Af = A[real].float().contiguous() if nr else A.float()
            Bf = B[real].float().contiguous() if nr else Af
            Cf = Cc[real].float().contiguous() if nr else Af
            dzp = _lib.ptr_array([_p(d) for d in dzs])
            _lib.check(L.senas_combine_bwd_apply(n, hw, c, nr, zp, dy.data_ptr(), y.data_ptr(), relu, Af.data_ptr(),
                                                 Bf.data_ptr(), Cf.data_ptr(), dzp, _p(ds_out), _stream()),
                       'senas_combine_bwd_apply')
        grads = list(dzs)
        grads += [dgamma[t].float() for t in range(T)]
        grads += [dbeta[t].float() for t in range(T)]
        grads += dw1s + dw2s
        grads += [torch.zeros_like(p) for p in meta['passengers']]
        assert len(grads) == ctx.nflat
        return (None, dmix.float() if dmix is not None else None, ds_out) + tuple(grads)


def bn_combine(terms, mix=None, residual=None, relu=False):
    """Normalise every term with its own BatchNorm2d (train or eval mode as the module says), apply
    SE gates, mix with ``mix`` (1-d tensor, one weight per term; None = all ones), add ``residual``
    and optionally ReLU -- one read of every term, one write."""
    T = len(terms)
    if T == 0 or T > _lib.MAX_TERMS:
        raise SenasHipError('bn_combine: %d terms (supported: 1..%d)' % (T, _lib.MAX_TERMS))
    real = [t for t, tm in enumerate(terms) if tm.z is not None]
    se_ids = [t for t, tm in enumerate(terms) if tm.se is not None]
    ref = next((tm.z for tm in terms if tm.z is not None), residual)
    if ref is None:
        raise SenasHipError('bn_combine: needs at least one tensor term or a residual to fix the shape')
    training = terms[0].bn.training
    passengers = [p for tm in terms for p in tm.passengers]
    meta = {
        'T': T, 'real': real, 'se': se_ids, 'shape': tuple(ref.shape), 'training': training, 'relu': bool(relu),
        'stats': [tm.stats for tm in terms],
        'buffers': [(tm.bn.running_mean, tm.bn.running_var, tm.bn.num_batches_tracked) for tm in terms],
        'passengers': passengers,
    }
    flat = [terms[t].z for t in real]
    flat += [tm.bn.weight for tm in terms] + [tm.bn.bias for tm in terms]
    flat += [terms[t].se.excitation[0].weight for t in se_ids] + [terms[t].se.excitation[2].weight for t in se_ids]
    flat += passengers
    return _BNCombine.apply(meta, mix, residual, *flat)
