"""Data parallelism for the search / train step: one process per GPU, weight gradients all-reduced
with RCCL over xGMI (``torch.distributed`` backend "nccl" on ROCm), overlapped with backward.

The reference only has ``nn.DataParallel`` (train_model.py:135-137; the search-phase branch is
broken).  Semantics kept: the batch is split across replicas, batch-norm
statistics stay per replica, gradients are averaged.  Architecture gradients (246 scalars) ride in
the same buckets -- both optimizers step them, so they must not diverge between ranks.

Mechanics: every parameter's ``.grad`` is a view into one flat fp32 buffer, cut into a few
contiguous buckets in reverse registration order (the order backward produces gradients).  A
post-accumulate hook counts the bucket's gradients; when the last one lands the bucket is
all-reduced asynchronously on a side stream while backward keeps running on the compute stream.
Message size is ~8 MB (2.2 M floats) in total: latency-bound over xGMI, so few large buckets.
"""
import torch
import torch.distributed as dist


class GradAllReducer(object):
    def __init__(self, params, world_size=None, num_buckets=2, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        # gradients arrive roughly in reverse registration order: bucket 0 = last parameters
        order = list(reversed(self.params))
        per = (total + num_buckets - 1) // max(1, num_buckets)
        self.buckets, self._bucket_of = [], {}
        off, cur, cur_start = 0, [], 0
        self._offsets = {}
        for p in order:
            n = p.numel()
            self._offsets[id(p)] = off
            p.grad = self.flat[off:off + n].view_as(p)
            cur.append(p)
            off += n
            if off - cur_start >= per or p is order[-1]:
                b = {'lo': cur_start, 'hi': off, 'count': len(cur), 'ready': 0, 'work': None}
                for q in cur:
                    self._bucket_of[id(q)] = len(self.buckets)
                self.buckets.append(b)
                cur, cur_start = [], off
        self.cuda = dev.type == 'cuda'
        self.side = torch.cuda.Stream(device=dev) if self.cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.enabled = True

    # --------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces optimizer.zero_grad(): one memset of the flat buffer; .grad views stay attached."""
        self.flat.zero_()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self._view_ptr(p):
                self._reattach(p)
        for b in self.buckets:
            b['ready'], b['work'] = 0, None

    def _view_ptr(self, p):
        return self.flat.data_ptr() + self._offset(p) * 4

    def _offset(self, p):
        return self._offsets[id(p)]

    def _reattach(self, p):
        off = self._offset(p)
        p.grad = self.flat[off:off + p.numel()].view_as(p)

    def _on_grad(self, p):
        if not self.enabled or self.world == 1:
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

    def finish(self):
        """Call after backward: waits for the in-flight buckets (launching any that never filled,
        e.g. parameters without a gradient this step) and turns sums into means."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b['work'] is None:
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


    def reduce_all(self):
        """Non-overlapped form (gradients produced by a replayed HIP graph): one all-reduce of the whole
        flat buffer on the current stream, then the mean."""
        if self.world == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / self.world)


def broadcast_parameters(module, src=0, process_group=None):
    """Start every replica from rank ``src``'s parameters and buffers (one flat broadcast each)."""
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    seen, uniq = set(), []
    for t in tensors:
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append(t)
    for dtype in {t.dtype for t in uniq}:
        group = [t for t in uniq if t.dtype == dtype]
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        dist.broadcast(flat, src=src, group=process_group)
        off = 0
        with torch.no_grad():
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
