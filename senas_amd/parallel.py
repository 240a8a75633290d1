"""Data parallelism for the search / train step: one process per GPU, weight gradients all-reduced
with RCCL over xGMI (``torch.distributed`` backend "nccl" on ROCm).

The reference only has ``nn.DataParallel`` (train_model.py:135-137; the search-phase branch is
broken).  Semantics kept: the batch is split across replicas, batch-norm
statistics stay per replica, gradients are averaged.  Architecture gradients (246 scalars) ride in
the same flat buffer -- both optimizers step them, so they must not diverge between ranks.

Two modes:
  * ``overlap=False`` (default; what a HIP-graph-replayed backward needs): after backward the
    gradients are gathered into one flat fp32 buffer by a multi-tensor copy, all-reduced in ONE call
    (8 MB: latency-bound over xGMI, so one message), averaged, and ``p.grad`` is pointed at the
    buffer's views.  On one GPU nothing is copied at all.
  * ``overlap=True`` (eager backward): ``p.grad`` are views of the flat buffer from the start, cut
    into a few buckets in reverse registration order (the order backward produces gradients); a
    post-accumulate hook launches a bucket's all-reduce on a side stream as soon as its last gradient
    has landed, while backward keeps running on the compute stream.
"""
import torch
import torch.distributed as dist


def _unique(params):
    seen, out = set(), []
    for p in params:
        if p.requires_grad and id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


class GradAllReducer(object):
    def __init__(self, params, world_size=None, num_buckets=2, process_group=None, overlap=False):
        self.params = _unique(params)
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.overlap = bool(overlap) and self.world > 1
        self.enabled = True
        dev = self.params[0].device
        self.cuda = dev.type == 'cuda'
        self.flat, self.views, self.buckets = None, None, []
        if self.world > 1:
            order = list(reversed(self.params))          # gradients arrive roughly in reverse registration order
            total = sum(p.numel() for p in order)
            self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
            self._order, self._offsets, off = order, {}, 0
            for p in order:
                self._offsets[id(p)] = off
                off += p.numel()
            self.views = [self.flat[self._offsets[id(p)]:self._offsets[id(p)] + p.numel()].view_as(p) for p in order]
        if self.overlap:
            per = (self.flat.numel() + num_buckets - 1) // max(1, num_buckets)
            self._bucket_of, cur, start = {}, [], 0
            for p in self._order:
                cur.append(p)
                end = self._offsets[id(p)] + p.numel()
                if end - start >= per or p is self._order[-1]:
                    for q in cur:
                        self._bucket_of[id(q)] = len(self.buckets)
                    self.buckets.append({'lo': start, 'hi': end, 'count': len(cur), 'ready': 0, 'work': None})
                    cur, start = [], end
            self.side = torch.cuda.Stream(device=dev) if self.cuda else None
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
            self._attach()

    # ------------------------------------------------------------------ common
    def after_backward(self):
        """Hook of the step drivers, called right after backward (inside the captured graph): nothing to do here."""

    def zero_grad(self):
        """Replaces optimizer.zero_grad()."""
        if self.overlap:
            self.flat.zero_()
            self._attach()
            for b in self.buckets:
                b['ready'], b['work'] = 0, None
        else:
            for p in self.params:
                p.grad = None

    def finish(self):
        """Call after backward; leaves the averaged gradients in ``p.grad``."""
        if self.world == 1:
            return
        if self.overlap:
            self._finish_overlapped()
        else:
            self.reduce_all()

    # ------------------------------------------------------------------ one-shot mode
    def reduce_all(self):
        if self.world == 1:
            return
        have = [(v, p.grad) for v, p in zip(self.views, self._order) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for v, p in zip(self.views, self._order) if p.grad is None]
        if have:
            self._gather(have)
        # A parameter without a gradient on this rank contributes zeros and comes back with the (possibly non-zero)
        # mean of the other ranks' gradients -- or an exact zero, which the optimizer then treats like any gradient
        # (weight decay, momentum), where torch on one GPU would skip a parameter whose .grad is None.  The step
        # drivers (gradsink.GradSink) behave the same way on any number of ranks: every registered parameter always
        # has a gradient, exactly zero if nothing reached it -- as the reference's 'none' candidates get from autograd.
        for v in missing:
            v.zero_()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / self.world)
        for v, p in zip(self.views, self._order):
            p.grad = v

    def _gather(self, have):
        """Gradients -> flat buffer.  On the GPU: ONE launch (senas_copy_rows_batched) over a device table that is rebuilt
        only when a gradient moved (under HIP-graph replay they never do); torch's multi-tensor copy issues one
        device-to-device copy per tensor -- ~800 launches per step for the derived network."""
        if not self.cuda or any(not g.is_contiguous() or g.dtype != torch.float32 for _, g in have):
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            return
        from . import _lib
        from . import functional as F
        from .packing import _CopyItem
        key = tuple((g.data_ptr(), v.data_ptr(), g.numel()) for v, g in have)       # sources, destinations and sizes
        if getattr(self, '_gather_key', None) != key:
            items = [_CopyItem(g.data_ptr(), v.data_ptr(), 1, g.numel(), g.numel(), g.numel(), 0) for v, g in have]
            raw = bytes((_CopyItem * len(items))(*items))
            self._gather_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(have[0][0].device)
            self._gather_key, self._gather_n = key, len(items)
            self._gather_max = max(it.row_len for it in items)
        _lib.check(_lib.lib().senas_copy_rows_batched(self._gather_table.data_ptr(), self._gather_n, self._gather_max, F._stream()),
                   'senas_copy_rows_batched')

    # ------------------------------------------------------------------ overlapped mode
    def _attach(self):
        for v, p in zip(self.views, self._order):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def _on_grad(self, p):
        if not self.enabled:
            return
        b = self.buckets[self._bucket_of[id(p)]]
        b['ready'] += 1
        if b['ready'] == b['count']:
            self._launch(b)

    def _launch(self, b):
        chunk = self.flat[b['lo']:b['hi']]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                b['work'] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b['work'] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _finish_overlapped(self):
        for b in self.buckets:
            if b['work'] is None:                         # never filled (parameters without a gradient this step)
                self._launch(b)
        for b in self.buckets:
            if self.cuda:
                with torch.cuda.stream(self.side):
                    b['work'].wait()
            else:
                b['work'].wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)
        self.flat.mul_(1.0 / self.world)


class SinkReducer(object):
    """The same contract over segments first..last of a ``gradsink.GradSink``: ``p.grad`` are persistent views of the
    flat buffer the backward kernels write into, so there is nothing to gather -- ``finish()`` is one all-reduce of
    the span and one scale launch, and on one GPU nothing at all."""

    overlap = False

    def __init__(self, sink, first=0, last=None, world_size=None, process_group=None):
        self.sink, self.first = sink, first
        self.last = len(sink.segments) - 1 if last is None else last
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.params = [p for i in range(self.first, self.last + 1) for p in sink.segments[i]]

    def zero_grad(self):
        self.sink.begin(self.first, self.last)

    def after_backward(self):
        self.sink.finish()

    def finish(self):
        if self.world == 1:
            return
        span = self.sink.span(self.first, self.last)
        dist.all_reduce(span, op=dist.ReduceOp.SUM, group=self.group)
        span.mul_(1.0 / self.world)

    reduce_all = finish


def broadcast_parameters(module, src=0, process_group=None):
    """Start every replica from rank ``src``'s parameters and buffers (one flat broadcast per dtype)."""
    seen, uniq = set(), []
    for t in list(module.parameters()) + list(module.buffers()):
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append(t)
    for dtype in sorted({t.dtype for t in uniq}, key=str):
        group = [t for t in uniq if t.dtype == dtype]
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        dist.broadcast(flat, src=src, group=process_group)
        off = 0
        with torch.no_grad():
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
