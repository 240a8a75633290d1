"""The fused cell node as one autograd Function over ``senas_node_fwd`` / ``senas_node_bwd``.

y = act( sum_t mix_t * gate_t * BN_t(z_t) + residual )

Replaces, per node: every candidate's BatchNorm2d (training statistics, running-stat update), the SE
gate of se_conv_3, the alpha- and beta-weighted sums of MixedOp / Cell (search/cell.py:34-36,100-107),
the two-op sum of BuildCell (models/senas_model.py:55-63), BasicBlock's residual add and the ReLU.
Forward = 2 launches (prepare + one pass over the z_t), backward = 3 (+1 per 8 terms beyond 8).
"""
import ctypes as C

import torch

from . import _lib
from . import functional as F
from ._lib import NodeDesc, SenasHipError
from .arena import zeros64

CL = torch.channels_last
SE_MID_MAX = 16


def _arr(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


def _desc(meta, gammas, betas, w1s, w2s, stats, mix):
    T = meta['T']
    n, c, h, w = meta['shape']
    d = NodeDesc()
    d.nterms, d.n, d.c, d.training, d.relu = T, n, c, int(meta['training']), int(meta['relu'])
    d.hw, d.eps, d.momentum = h * w, F.BN_EPS, F.BN_MOMENTUM
    se_pos = {t: k for k, t in enumerate(meta['se'])}
    for t in range(T):
        rm, rv, nbt = meta['buffers'][t]
        d.stats[t] = stats[t].data_ptr() if stats[t] is not None else None
        # (a slice of a stacked convolution's statistics: images further apart than 2c doubles)
        d.stats_image_stride[t] = stats[t].stride(0) if (stats[t] is not None and stats[t].dim() == 3) else 0
        d.gamma[t], d.beta[t] = gammas[t].data_ptr(), betas[t].data_ptr()
        d.running_mean[t] = rm.data_ptr() if rm is not None else None
        d.running_var[t] = rv.data_ptr() if rv is not None else None
        d.num_batches_tracked[t] = nbt.data_ptr() if nbt is not None else None
        if t in se_pos:
            k = se_pos[t]
            d.se_w1[t], d.se_w2[t], d.se_mid[t] = w1s[k].data_ptr(), w2s[k].data_ptr(), w1s[k].shape[0]
    d.mix = mix.data_ptr() if mix is not None else None
    return d


class _Node(torch.autograd.Function):
    """flat = [z (real terms)..., gamma (all terms)..., beta (all terms)..., se_w1 (se terms)..., se_w2 ..., passengers...]"""

    @staticmethod
    def forward(ctx, meta, mix, residual, *flat):
        L = _lib.lib()
        T, real, se_ids = meta['T'], meta['real'], meta['se']
        nr, ns = len(real), len(se_ids)
        zs, zstr = zip(*[F.nhwc_slice(z) for z in flat[:nr]]) if nr else ((), ())
        gammas = [F._dev(g).contiguous() for g in flat[nr:nr + T]]
        betas = [F._dev(b).contiguous() for b in flat[nr + T:nr + 2 * T]]
        w1s = [w.contiguous() for w in flat[nr + 2 * T:nr + 2 * T + ns]]
        w2s = [w.contiguous() for w in flat[nr + 2 * T + ns:nr + 2 * T + 2 * ns]]
        n, c, h, w = meta['shape']
        training = meta['training']
        dev = gammas[0].device
        for wt in w1s:
            if wt.shape[0] > SE_MID_MAX:
                raise SenasHipError('SE hidden width %d > %d' % (wt.shape[0], SE_MID_MAX))
        zfull, stats = [None] * T, [None] * T
        zstrides = (C.c_int32 * T)()
        for k, t in enumerate(real):
            z = zs[k]
            if tuple(z.shape) != (n, c, h, w):
                raise SenasHipError('node terms disagree in shape: %s vs %s' % (tuple(z.shape), (n, c, h, w)))
            zfull[t] = z
            zstrides[t] = zstr[k]
            st = meta['stats'][t]
            if st is None and (training or t in se_ids):
                st = F.chan_stats(z)
            stats[t] = st
        mixc = mix.detach().float().contiguous() if mix is not None else None
        if mixc is not None and meta.get('mix_off') is not None:        # rows of a shared _EdgeMix matrix, read in place
            mixc = mixc.view(-1)[meta['mix_off']:meta['mix_off'] + T]
        d = _desc(meta, gammas, betas, w1s, w2s, stats, mixc)
        coefs = torch.empty((T, 4, c), device=dev, dtype=torch.float32)
        gate = torch.empty((T, n, c), device=dev, dtype=torch.float32)
        scratch = torch.empty((2, T, n, c), device=dev, dtype=torch.float32)
        se_m = torch.empty((T, n, c), device=dev, dtype=torch.float32) if ns else None
        se_a1 = torch.empty((T, n, SE_MID_MAX), device=dev, dtype=torch.float32) if ns else None
        # the node's output: a dense tensor, and / or its channel slice of the cell's concatenation buffer (meta['cat'])
        cat = meta.get('cat')
        y2, y2s, y2pad = None, 0, 0
        if cat is not None:
            buf, off, dense = cat[:3]
            y2pad = cat[3] if len(cat) > 3 else 0        # zero channels this node writes behind its slice
            y2s = buf.shape[1]
            y2 = torch.empty(0, device=dev, dtype=torch.float32).set_(buf.untyped_storage(), buf.storage_offset() + off, (n, c, h, w),
                                                                     (h * w * y2s, 1, w * y2s, y2s))
        y = torch.empty((n, c, h, w), device=dev, dtype=torch.float32, memory_format=CL) if (cat is None or cat[2]) else None
        # ReLU mask for the backward pass: one byte per 16-byte piece of y instead of y itself
        # (not kept when nothing will be differentiated: inference)
        tracked = any(ctx.needs_input_grad)
        mask8 = torch.empty(n * h * w * (c // 4), device=dev, dtype=torch.uint8) if (meta['relu'] and c % 4 == 0 and tracked) else None
        res = F.nhwc(residual) if residual is not None else None
        zp = _arr([z.data_ptr() if z is not None else None for z in zfull])
        _lib.check(L.senas_node_fwd(C.byref(d), zp, zstrides, F._p(res), F._p(y), coefs.data_ptr(), gate.data_ptr(),
                                    scratch[0].data_ptr(), scratch[1].data_ptr(), F._p(se_m), F._p(se_a1), F._p(mask8),
                                    F._p(meta.get('out_stats')), F._p(y2), y2s, y2pad, F._stream()),
                   'senas_node_fwd')
        ctx.meta = meta
        ctx.has_mix, ctx.has_res, ctx.nflat = mix is not None, residual is not None, len(flat)
        ctx.stats = stats
        ctx.se_buf = (se_m, se_a1)
        ctx.mask8 = mask8
        ctx.out_shape = (n, c, h, w)
        if y is None:
            y = y2                                   # nothing but the concatenation reads this node: its slice IS the output
        # y itself is only needed for the mask when there is no byte map
        ysave = y.contiguous(memory_format=CL) if (meta['relu'] and mask8 is None) else coefs
        ctx.save_for_backward(ysave, coefs, gate, mixc if mixc is not None else coefs, *zs, *gammas, *betas, *w1s, *w2s)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        meta = ctx.meta
        T, real, se_ids = meta['T'], meta['real'], meta['se']
        nr, ns = len(real), len(se_ids)
        n, c, h, w = meta['shape']
        saved = ctx.saved_tensors
        y, coefs, gate, mixc = saved[:4]
        zs = saved[4:4 + nr]
        gammas, betas = saved[4 + nr:4 + nr + T], saved[4 + nr + T:4 + nr + 2 * T]
        w1s = saved[4 + nr + 2 * T:4 + nr + 2 * T + ns]
        w2s = saved[4 + nr + 2 * T + ns:4 + nr + 2 * T + 2 * ns]
        dev = coefs.device
        # a channel slice of a wider NHWC tensor (the gradient of torch.cat along channels) is read in place
        F._dev(dy)
        ct = dy.stride(3) if dy.dim() == 4 else 0
        if (dy.dim() == 4 and tuple(dy.shape) == (n, c, h, w) and ct > c and ct % 4 == 0 and c % 4 == 0 and
                dy.stride() == (h * w * ct, 1, w * ct, ct) and dy.data_ptr() % 16 == 0):
            dy_stride = ct
        else:
            dy, dy_stride = F.nhwc(dy), c
        mask8 = ctx.mask8
        yptr = y.data_ptr() if (meta['relu'] and mask8 is None) else None
        mix = mixc if ctx.has_mix else None
        d = _desc(meta, gammas, betas, w1s, w2s, ctx.stats, mix)
        zfull = [None] * T
        zstrides = (C.c_int32 * T)()
        for k, t in enumerate(real):
            zfull[t] = zs[k]
            zstrides[t] = zs[k].stride(3) if zs[k].stride(3) > c else c       # a slice saved by the forward pass stays a slice
            if zs[k].dtype == torch.bfloat16:                                 # (a bf16-stored term: the kernels' negative-stride convention)
                zstrides[t] = -c
        need = ctx.needs_input_grad
        dzs, dz_strides = [None] * T, (C.c_int32 * T)()
        slots = meta.get('slots') or [None] * T
        for k, t in enumerate(real):
            if need[3 + k]:
                if slots[t] is not None and c % 4 == 0:
                    landing, part = slots[t]               # write straight into the producer's stacked gradient buffer
                    dzs[t] = landing.part(part, zs[k])
                    dz_strides[t] = dzs[t].stride(3)
                else:
                    dzs[t] = torch.empty_like(zs[k], memory_format=CL)       # (bf16 for a bf16-stored term: written as such)
                    if dzs[t].dtype == torch.bfloat16:
                        dz_strides[t] = -c
        ds_out = torch.empty((n, c, h, w), device=dev, dtype=torch.float32, memory_format=CL) if (ctx.has_res and need[2]) else None
        p = zeros64((T + 1, n, c), dev)
        # destinations: the parameter's view in the flat gradient buffer under a step driver (autograd then gets None),
        # else own tensors that autograd adopts as .grad
        dgt, dgs = zip(*[F.wgrad_dest(g) for g in gammas])
        dbt, dbs = zip(*[F.wgrad_dest(b) for b in betas])
        shared = meta.get('dmix_buf') if ctx.has_mix else None
        if shared is not None:                               # the cells of a kind add into one buffer (functional._EdgeMix)
            dmix_dst, dmix, dmix_acc = shared.view(-1)[meta['mix_off']:meta['mix_off'] + T], None, 1
        else:
            dmix = torch.empty(T, device=dev, dtype=torch.float32) if ctx.has_mix else None
            dmix_dst, dmix_acc = dmix, 0
        abk = torch.empty((3, T, n, c), device=dev, dtype=torch.float32)
        dw1t, dw1s = zip(*[F.wgrad_dest(wt) for wt in w1s]) if ns else ((), ())
        dw2t, dw2s = zip(*[F.wgrad_dest(wt) for wt in w2s]) if ns else ((), ())
        se_pos = {t: k for k, t in enumerate(se_ids)}
        dw1p = _arr([dw1t[se_pos[t]].data_ptr() if t in se_pos else None for t in range(T)])
        dw2p = _arr([dw2t[se_pos[t]].data_ptr() if t in se_pos else None for t in range(T)])
        zp = _arr([z.data_ptr() if z is not None else None for z in zfull])
        dzp = _arr([z.data_ptr() if z is not None else None for z in dzs])
        se_m, se_a1 = ctx.se_buf
        _lib.check(L.senas_node_bwd(C.byref(d), zp, zstrides, dy.data_ptr(), dy_stride, yptr, F._p(mask8), coefs.data_ptr(), gate.data_ptr(),
                                    F._p(se_m), F._p(se_a1), p[0].data_ptr(), p[1:].data_ptr(),
                                    _arr([t_.data_ptr() for t_ in dgt]), _arr([t_.data_ptr() for t_ in dbt]),
                                    F._p(dmix_dst), dmix_acc, dw1p, dw2p, abk.data_ptr(), dzp, dz_strides, F._p(ds_out),
                                    F._stream()), 'senas_node_bwd')
        grads = [dzs[t] for t in real]
        grads += list(dgs) + list(dbs)
        grads += list(dw1s) + list(dw2s)
        # exactly-zero gradients: the zeroed view of the flat buffer already is one
        grads += [None if (not q.requires_grad or (F.SINK is not None and F.SINK.dest(q) is not None)) else torch.zeros_like(q)
                  for q in meta['passengers']]
        assert len(grads) == ctx.nflat
        return (None, dmix, ds_out) + tuple(grads)


def max_terms(c):
    """Addends one node launch takes at ``c`` channels: SENAS_MAX_TERMS, or fewer where the combine kernel's LDS stage of
    coefficients and shifts (2 * T * c + c floats, csrc/node.hip senas_node_fwd) would pass 64 KiB (c = 256: 31)."""
    return max(1, min(_lib.MAX_TERMS, (16384 - c) // (2 * c)))


def bn_combine(terms, mix=None, residual=None, relu=False, out_stats=False, cat=None):
    """Normalise every term with its own BatchNorm2d (train or eval mode as the module says), apply
    SE gates, mix with ``mix`` (1-d tensor, one weight per term; None = all ones), add ``residual``
    and optionally ReLU -- one read of every term, one write.  ``out_stats``: also leave the per-image channel
    sums of the result on it (``y._senas_stats``), for a consumer that batch-normalises it directly.
    ``cat = (buffer [n, C, h, w] NHWC, channel offset, dense[, zero_pad])``: the result is (also) written into that channel
    slice of the cell's concatenation buffer (followed by ``zero_pad`` zero channels); with ``dense`` False nothing else is
    written and the slice itself is returned."""
    T = len(terms)
    if T == 0:
        raise SenasHipError('bn_combine: no terms')
    width = next((tm.z.shape[1] for tm in terms if tm.z is not None), residual.shape[1] if residual is not None else 1)
    cap = max_terms(width)
    if T > cap:
        # more addends than one launch describes (a node with six inputs of a ``--meta_node_num 5`` search cell has 36,
        # experiments/search_arc.py:38-44): the first ``cap`` as a partial sum without activation, the rest on top of it
        # as their residual -- the same sum, one more pass over the partial result
        head, tail = terms[:cap], terms[cap:]
        if isinstance(mix, F.SharedMix):
            if mix.count != T:
                raise SenasHipError('bn_combine: %d shared mixing weights for %d terms' % (mix.count, T))
            mix_h, mix_t = F.SharedMix(mix.M, mix.off, len(head), mix.dM), F.SharedMix(mix.M, mix.off + len(head), len(tail), mix.dM)
        elif mix is not None:
            mix_h, mix_t = mix[:len(head)], mix[len(head):]
        else:
            mix_h = mix_t = None
        part = bn_combine(head, mix=mix_h, residual=residual, relu=False)
        return bn_combine(tail, mix=mix_t, residual=part, relu=relu, out_stats=out_stats, cat=cat)
    real = [t for t, tm in enumerate(terms) if tm.z is not None]
    se_ids = [t for t, tm in enumerate(terms) if tm.se is not None]
    ref = next((tm.z for tm in terms if tm.z is not None), residual)
    if ref is None:
        raise SenasHipError('bn_combine: needs at least one tensor term or a residual to fix the shape')
    if ref.shape[1] > 256:
        raise SenasHipError('bn_combine: more than 256 channels is not on the SENAS path')
    passengers = [p for tm in terms for p in tm.passengers]
    shared = mix if isinstance(mix, F.SharedMix) else None
    if shared is not None:
        if shared.count != T:
            raise SenasHipError('bn_combine: %d shared mixing weights for %d terms' % (shared.count, T))
        mix = shared.M
    meta = {
        'mix_off': shared.off if shared is not None else None, 'dmix_buf': shared.dM if shared is not None else None,
        'T': T, 'real': real, 'se': se_ids, 'shape': tuple(ref.shape), 'training': terms[0].bn.training, 'relu': bool(relu),
        'stats': [tm.stats for tm in terms],
        'buffers': [(tm.bn.running_mean, tm.bn.running_var, tm.bn.num_batches_tracked) for tm in terms],
        'passengers': passengers,
        'slots': [tm.grad_slot for tm in terms],
    }
    flat = [terms[t].z for t in real]
    flat += [tm.bn.weight for tm in terms] + [tm.bn.bias for tm in terms]
    flat += [terms[t].se.excitation[0].weight for t in se_ids] + [terms[t].se.excitation[2].weight for t in se_ids]
    flat += passengers
    c = ref.shape[1]
    if cat is not None:
        meta['cat'] = cat
    if out_stats and c % 4 == 0 and (c // 4) & (c // 4 - 1) == 0 and c <= 256:
        meta['out_stats'] = F.new_stats(ref.shape[0], c, ref)
    y = _Node.apply(meta, mix, residual, *flat)
    if meta.get('out_stats') is not None:
        y._senas_stats = meta['out_stats']
    return y
