"""Inference / validation pass: what ``experiments/testing_model.py:150-190`` and the
validation loops of the drivers (``train_model.py`` / ``search_arc.py`` ``infer``) do per batch --
``model.eval()``, ``torch.no_grad()``, forward, criterion on ``predicts[-1]``, ``metric.update(target, predicts[-1])``,
``argmax`` masks -- as one stream-ordered, HIP-graph-replayed launch list on static buffers with no host round trip
per batch (the reference syncs for ``loss.item()`` and three times inside the metric, every step).

Eval mode changes what the kernels do: BatchNorm uses the running statistics, so no producer kernel accumulates
channel sums (SE blocks excepted: their squeeze still needs the per-image mean), and no node keeps a ReLU mask.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from . import functional as F
from .arena import reset_arena
from .metrics import SegmentationMetric
from .operations import AdapterBlock, BasicBlock, ConvBn, ConvBnSe, DepSepConv, RectifyBlock, ShrinkBlock, ZeroOp, _Rectify
from .packing import WeightPacker


class Evaluator(object):
    """``ev = Evaluator(model, nclass, x_like, y_like, criterion)``; per batch ``ev(x, y)``; at the end
    ``ev.result() -> (mean loss, pixAcc, mIoU, dice)`` (the tuple ``testing_model.py`` logs).

    ``ev.logits`` (N x nclass x H x W) and ``ev.mask`` (N x H x W int64 arg-max, what ``save_mask`` writes as PNG)
    are static tensors refreshed by every call."""

    def __init__(self, model, nclass, x, y=None, criterion=None, use_graph=True, warmup=2):
        self.model = model.eval()
        self.criterion = criterion
        self.metric = SegmentationMetric(nclass)
        self.x = x.clone()
        self.y = y.clone() if y is not None else None
        if criterion is not None and y is None:
            raise ValueError('a criterion needs a target buffer')
        self.loss_sum = torch.zeros((), device=x.device, dtype=torch.float64)
        self.batches = 0
        self.logits = self.mask = None
        self.graph = None
        self.packer = WeightPacker(model)          # weights are constant between refresh() calls: every replay reuses the images
        self.packer.install()
        self.packer.refresh()
        if use_graph:
            self._capture(warmup)

    def refresh(self):
        """Re-derive everything cached from the weights (packed images, stacked buffers).  ``__call__`` does it by itself
        when an optimizer moved the weights since the last call, so an evaluator can be reused for per-epoch validation
        as the drivers' ``infer()`` loops are (experiments/train_model.py:306-340)."""
        self.packer.refresh()

    def _forward(self, x):
        return self.model(x)

    @torch.no_grad()
    def _eager(self):
        logits = self._forward(self.x)[-1]
        if self.y is not None:
            if self.criterion is not None:
                self.loss_sum += self.criterion([logits], self.y).double()
            self.metric.update(self.y, logits)
        return logits, torch.argmax(logits, 1)

    def _capture(self, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # a forward pass that runs on several lanes (grid.Lanes: the model's own forward, not the folded launch list) is captured
        # but replayed by the lane scheduler -- the runtime's executor is never given a multi-branch graph (csrc/sched.hip)
        from .grid import Lanes, MacroGrid
        from .lanesched import LaneSchedule
        from ._lib import SenasHipError
        grid = next((m for m in self.model.modules() if isinstance(m, MacroGrid)), None)
        lanes = (type(self)._forward is Evaluator._forward and Lanes.enabled and grid is not None and grid.lanes and grid._depth > 2)
        self.sched = None
        for attempt in (0, 1):
            reset_arena()
            graph = torch.cuda.CUDAGraph(keep_graph=True) if lanes else torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'), Lanes.scheduled():   # (other threads -- the RCCL watchdog -- may touch the runtime)
                self.logits, self.mask = self._eager()
            if not lanes:
                break
            try:
                self.sched = LaneSchedule(graph)
                break
            except SenasHipError:
                if attempt:
                    raise
                grid.lanes, lanes = False, False         # (this model keeps the serial schedule)
        reset_arena()
        self.graph = graph
        self.reset()

    def reset(self):
        self.metric.reset_counts() if self.graph is not None else self.metric.reset()
        self.loss_sum.zero_()
        self.batches = 0

    def __call__(self, x, y=None):
        self.x.copy_(x, non_blocking=True)
        if self.y is not None:
            if y is None:
                raise ValueError('this evaluator was built with targets')
            self.y.copy_(y, non_blocking=True)
        if self.packer.stale():                    # the captured launches read the images by address: keep them current
            self.refresh()
        if self.graph is None:
            self.logits, self.mask = self._eager()
        else:
            if getattr(self, 'sched', None) is not None:
                self.sched.launch()
            else:
                self.graph.replay()
            if self.y is not None:
                self.metric._acc_n += 1             # the replay ran the captured update launch
        self.batches += 1
        return self.logits, self.mask

    def result(self):
        """(mean loss, pixAcc, mIoU, dice) -- the only host synchronisation of the pass."""
        if self.y is None or self.batches == 0:
            raise ValueError('no labelled batch was evaluated')
        pix, miou, dice = self.metric.get()
        mean_loss = float(self.loss_sum.item()) / self.batches if self.criterion is not None else None
        return mean_loss, pix, miou, dice


# ------------------------------------------------------------------------------------------------ folded forward
# Eval-mode BatchNorm2d is a per-channel affine (scale = gamma / sqrt(running_var + eps), shift = beta - running_mean *
# scale), so it needs no pass of its own: it rides in the epilogue of the convolution that produces the tensor
# (senas_conv2d_fwd_epilogue).  A cell node ReLU(op_a(s_i) + op_b(s_j)) becomes: op_a's raw output z_a (its affine
# still pending), then op_b's convolution with the epilogue  ReLU(scale_b * acc + shift_b + shift_a + scale_a * z_a).
# Terms that have no epilogue kernel (pooling, bilinear, transposed / strided / thin convolutions) stay raw and are
# either the addend of the other term's convolution or, when neither term has one, go through senas_combine_fwd.

class _Raw(object):
    """A tensor whose per-(image, channel) affine is still pending: value = scale * z + shift."""
    __slots__ = ('z', 'scale', 'shift')

    def __init__(self, z, scale, shift):
        self.z, self.scale, self.shift = z, scale, shift


class _Lazy(object):
    """A convolution not launched yet (it may become the fused producer of its node): value = scale * conv(x) + shift."""
    __slots__ = ('conv', 'x', 'in_relu', 'scale', 'shift')

    def __init__(self, conv, x, in_relu, scale, shift):
        self.conv, self.x, self.in_relu, self.scale, self.shift = conv, x, in_relu, scale, shift


class FoldedForward(object):
    """Eval-mode forward of a ``SenasModel`` with batch-norm folded into the producers (module docstring above).
    ``refresh()`` re-derives the affines after the weights / running statistics changed (in place: a captured graph
    keeps reading the same buffers)."""

    def __init__(self, model, batch):
        self.model, self.n = model, batch
        self.affine = {}
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                dev = m.weight.device
                self.affine[id(m)] = (m, torch.empty((batch, m.num_features), device=dev), torch.empty((batch, m.num_features), device=dev))
        self.consts = {}
        self.const_ids = set()                 # tensors that only change in refresh(): sums of them are cached
        for _, scale, shift in self.affine.values():
            self.const_ids.update((id(scale), id(shift)))
        self.derived = {}                      # key -> (result, recipe) recomputed in place by refresh()
        self.fused_launches = self.fallback_launches = 0
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        for m, scale, shift in self.affine.values():
            s = m.weight.double() / torch.sqrt(m.running_var.double() + m.eps)
            scale.copy_(s.float().expand_as(scale))
            shift.copy_((m.bias.double() - m.running_mean.double() * s).float().expand_as(shift))
        for out, recipe in self.derived.values():
            out.copy_(recipe())

    def _cached(self, kind, parts, recipe):
        """recipe() of constant tensors, computed once and kept current by refresh(); of anything else, computed now."""
        if not all(id(p) in self.const_ids for p in parts):
            return recipe()
        key = (kind,) + tuple(id(p) for p in parts)
        if key not in self.derived:
            out = recipe().contiguous()
            self.derived[key] = (out, recipe)
            self.const_ids.add(id(out))
        return self.derived[key][0]

    def _sum(self, parts):
        if len(parts) == 1:
            return parts[0]
        return self._cached('sum', parts, lambda: torch.stack(list(parts)).sum(0))

    def _aff(self, bn):
        _, scale, shift = self.affine[id(bn)]
        return scale, shift

    # ------------------------------------------------------------------ launches
    @staticmethod
    def _geom(conv, x):
        w = conv.weight
        tr = isinstance(conv, nn.ConvTranspose2d)
        n, ci, hi, wi = x.shape
        k, s, p, d, g = w.shape[2], conv.stride[0], conv.padding[0], conv.dilation[0], conv.groups
        co = w.shape[1] * g if tr else w.shape[0]
        op = conv.output_padding[0] if tr else 0
        ho, wo = F.conv_out_size(hi, k, s, p, d, tr, op), F.conv_out_size(wi, w.shape[3], s, p, d, tr, op)
        return F.ConvGeom(n, hi, wi, ci, ho, wo, co, k, w.shape[3], s, p, d, int(tr), g)

    def _conv_plain(self, conv, x, in_relu, stats=None):
        g, L = self._geom(conv, x), _lib.lib()
        y = F.new_nhwc(g.n, g.co, g.ho, g.wo, x)
        ws = torch.empty(int(L.senas_conv2d_ws_bytes(C.byref(g))), device=x.device, dtype=torch.uint8)
        _lib.check(L.senas_conv2d_fwd(C.byref(g), x.data_ptr(), conv.weight.data_ptr(), y.data_ptr(), int(in_relu), F._p(stats),
                                      ws.data_ptr(), F._packed(conv.weight, 0), F._stream()), 'senas_conv2d_fwd')
        return y

    def _conv_epilogue(self, conv, x, in_relu, scale, bias, addend=None, add_scale=None, relu=False):
        """The fused launch, or None when this geometry has no epilogue kernel."""
        g, L = self._geom(conv, x), _lib.lib()
        y = F.new_nhwc(g.n, g.co, g.ho, g.wo, x)
        e = _lib.ConvEpilogue(scale.data_ptr(), bias.data_ptr(), F._p(addend), F._p(add_scale), int(relu))
        packed = F._packed(conv.weight, 0)
        ws = None if (packed is not None or conv.groups != 1) else torch.empty(int(L.senas_conv2d_ws_bytes(C.byref(g))), device=x.device,
                                                                                dtype=torch.uint8)
        code = L.senas_conv2d_fwd_epilogue(C.byref(g), x.data_ptr(), conv.weight.data_ptr(), y.data_ptr(), int(in_relu), C.byref(e),
                                           F._p(ws), packed, F._stream())
        if code == _lib.UNSUPPORTED:
            return None
        _lib.check(code, 'senas_conv2d_fwd_epilogue')
        self.fused_launches += 1
        return y

    def _combine(self, raws, relu, residual=None):
        """act(sum_t scale_t * z_t + shift_t (+ residual)) over raw tensors: one senas_combine_fwd pass."""
        zs = [F.nhwc(r.z) for r in raws]
        n, c, h, w = zs[0].shape
        scales = [r.scale for r in raws]
        coef = scales[0] if len(raws) == 1 else self._cached('stack', scales, lambda: torch.stack(scales))
        bias = self._sum([r.shift for r in raws])
        y = F.new_nhwc(n, c, h, w, zs[0])
        zp = (C.c_void_p * len(zs))(*[z.data_ptr() for z in zs])
        _lib.check(_lib.lib().senas_combine_fwd(n, h * w, c, len(zs), zp, coef.data_ptr(), bias.data_ptr(), F._p(residual),
                                                int(relu), y.data_ptr(), F._stream()), 'senas_combine_fwd')
        self.fallback_launches += 1
        return y

    def _finish(self, terms, relu, residual=None):
        """Materialise act(sum of terms (+ residual)).  At most one _Lazy is kept for the fused launch."""
        lazies = [t for t in terms if isinstance(t, _Lazy)]
        raws = [t for t in terms if isinstance(t, _Raw) and t.z is not None]
        extra = [t.shift for t in terms if isinstance(t, _Raw) and t.z is None]      # 'none' ops: only the shift survives
        lazies.sort(key=lambda t: self._has_epilogue(t.conv, t.x))                    # a fusable one goes last
        if lazies:
            last = lazies.pop()
            pend = raws + [self._run_lazy(t) for t in lazies]      # everything else is a tensor before the fused launch
            if len(pend) + (1 if residual is not None else 0) <= 1:
                bias = self._sum([last.shift] + extra + [p.shift for p in pend])
                addend = pend[0].z if pend else residual
                add_scale = pend[0].scale if pend else None
                y = self._conv_epilogue(last.conv, last.x, last.in_relu, last.scale, bias, addend, add_scale, relu)
                if y is not None:
                    return y
            raws = pend + [self._run_lazy(last, force_plain=True)]
        if not raws:
            raise _lib.SenasHipError('a node whose ops are all \'none\' has no tensor to shape its output')
        if extra:
            raws = [_Raw(raws[0].z, raws[0].scale, self._sum([raws[0].shift] + extra))] + raws[1:]
        return self._combine(raws, relu, residual)

    def _run_lazy(self, t, force_plain=False):
        """A lazy convolution as a tensor: with its affine applied by its own epilogue when there is one."""
        if not force_plain:
            y = self._conv_epilogue(t.conv, t.x, t.in_relu, t.scale, t.shift)
            if y is not None:
                return _Raw(y, self._ones(t.scale), self._zeros(t.shift))
        return _Raw(self._conv_plain(t.conv, t.x, t.in_relu), t.scale, t.shift)

    def _ones(self, like):
        if like.shape[1] not in self.consts:
            self.consts[like.shape[1]] = (torch.ones_like(like), torch.zeros_like(like))
            self.const_ids.update(id(t) for t in self.consts[like.shape[1]])
        return self.consts[like.shape[1]][0]

    def _zeros(self, like):
        self._ones(like)
        return self.consts[like.shape[1]][1]

    @staticmethod
    def _has_epilogue(conv, x):
        """Host-side guess of senas_conv2d_fwd_epilogue's answer, used to pick WHICH convolution of a node is launched
        last; a wrong guess costs a launch, not correctness (the entry point itself decides)."""
        if isinstance(conv, nn.ConvTranspose2d) or conv.groups != 1 or conv.stride[0] != 1:
            return False
        k, d, ci, co = conv.kernel_size[0], conv.dilation[0], conv.in_channels, conv.out_channels
        return ci % 16 == 0 and co > 4 and conv.padding[0] == d * (k // 2) and x.shape[3] >= 8 and x.shape[2] >= 4

    # ------------------------------------------------------------------ candidate ops -> terms
    def term(self, op, x):
        if isinstance(op, ConvBnSe):
            # the gate needs the per-image mean of the normalised output: raw convolution with channel sums
            conv, bn, se = op.conv, op.norm, op.se          # (eval mode: a Dropout2d in front of the convolution is the identity)
            g = self._geom(conv, x)
            stats = F.new_stats(g.n, g.co, x)
            z = self._conv_plain(conv, x, False, stats)
            scale, shift = self._aff(bn)
            mean = (stats[:, :, 0] / float(g.ho * g.wo)).float() * scale + shift
            hid = torch.relu(mean @ se.excitation[0].weight.t())
            gate = torch.sigmoid(hid @ se.excitation[2].weight.t())
            return _Raw(z, gate * scale, gate * shift)
        if isinstance(op, ConvBn):
            return _Lazy(op.conv, x, False, *self._aff(op.norm))
        if isinstance(op, DepSepConv):
            s1, b1 = self._aff(op.norm1)
            mid = self._conv_epilogue(op.dw, x, False, s1, b1, relu=True)
            if mid is None:
                mid = self._combine([_Raw(self._conv_plain(op.dw, x, False), s1, b1)], relu=True)
            return _Lazy(op.pw, mid, False, *self._aff(op.norm2))
        if isinstance(op, AdapterBlock):
            scale, shift = self._aff(op.norm)
            if isinstance(op.module, ZeroOp):
                return _Raw(None, scale, shift)
            y = op._resample(x)
            if op.c_in != op.c_ot:
                return _Lazy(op.conv, y, False, scale, shift)
            return _Raw(y, scale, shift)
        raise NotImplementedError('folded inference of %s' % type(op).__name__)

    # ------------------------------------------------------------------ blocks
    def block(self, m, x):
        if isinstance(m, ShrinkBlock):
            return self._finish([_Lazy(m.conv, x, True, *self._aff(m.norm))], relu=False)
        if isinstance(m, RectifyBlock):
            return self._finish([_Lazy(m.conv, x, False, *self._aff(m.norm))], relu=False)
        if isinstance(m, _Rectify):
            op, aff = m[1], self._aff(m[2])
            if isinstance(op, (nn.Conv2d, nn.ConvTranspose2d)):
                return self._finish([_Lazy(op, x, True, *aff)], relu=False)
            if isinstance(op, nn.AvgPool2d):
                return self._combine([_Raw(F._AvgPool3.apply(x, op.stride, True, False)[0], *aff)], relu=False)
            return self._combine([_Raw(F._Bilinear2x.apply(F.relu(x), False)[0], *aff)], relu=False)
        if isinstance(m, BasicBlock):
            a = self._finish([_Lazy(m.conv1, x, False, *self._aff(m.bn1))], relu=True)
            res = x if m.downsample is None else m.downsample(x)
            return self._finish([_Lazy(m.conv2, a, False, *self._aff(m.bn2))], relu=False, residual=res)
        raise NotImplementedError('folded inference of %s' % type(m).__name__)

    def cell(self, cell, in0, in1):
        states = [self.block(cell.preprocess0, in0), F.relu(in1)]
        for i in range(cell._num_meta_node):
            terms = []
            for e in (2 * i, 2 * i + 1):
                terms.append(self.term(cell._ops[e], states[cell._indices[e]]))
            states.append(self._finish(terms, relu=True))
        return self.block(cell.post_process, torch.cat([states[i] for i in cell._concat], dim=1))

    def __call__(self, x):
        net = self.model
        if self.n != x.shape[0]:
            raise ValueError('built for batch %d, got %d' % (self.n, x.shape[0]))
        stem0 = net.stem0
        s0 = self._finish([_Lazy(stem0[0], F.nhwc(x), False, *self._aff(stem0[1]))], relu=False)
        outs = [self.block(net.stem1[2], F.max_pool3(s0, 2, in_relu=True))]
        depth = net._depth
        for j in range(1, depth):
            outs.append(self.cell(net.blocks[0][j], s0 if j == 1 else outs[-2], outs[-1]))
        for j in reversed(range(depth - 1)):
            for i in range(1, depth - j):
                cell = net.blocks[i][j]
                if cell is None:
                    outs[i + j] = None
                    continue
                skips = [outs[t] for t in range(j, i + j) if outs[t] is not None]
                outs[i + j] = self.cell(cell, skips[0] if len(skips) == 1 else torch.cat(skips, dim=1), outs[i + j])
        head = net.head_block[-1]
        tails = outs if net._supervision else outs[-1:]
        return [head.segmentation_head(self.cell(head.up_cell, s0, o)) for o in tails]


class FoldedEvaluator(Evaluator):
    """``Evaluator`` whose forward is the batch-norm-folded launch list (``FoldedForward``)."""

    def __init__(self, model, nclass, x, y=None, criterion=None, use_graph=True, warmup=2):
        self.folded = FoldedForward(model.eval(), x.shape[0])
        super().__init__(model, nclass, x, y, criterion, use_graph, warmup)

    def _forward(self, x):
        return self.folded(x)

    def refresh(self):
        """After the weights changed (validation between epochs): repack and re-derive the affines, in place."""
        self.packer.refresh()
        self.folded.refresh()
