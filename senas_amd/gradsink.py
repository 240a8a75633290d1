"""Flat gradient buffer that the backward kernels write into directly.

The reference lets autograd hand every parameter its gradient (``loss.backward()`` at
experiments/train_model.py:283, search_arc.py:279): one ``AccumulateGrad`` per parameter, which on this path
meant ~1 000 device-to-device copies, ~600 memsets and a few hundred tiny adds per supernet step (slices of stacked
weight gradients are views, so autograd clones them; absent gradients are materialised as zeros) -- 12 % of the
step's kernel time (profiles/r1_i_search_step_by_grid.txt).

With a ``GradSink`` installed, ``p.grad`` of every registered parameter is a persistent view into ONE flat fp32
buffer, laid out in *segments* that can be zeroed / all-reduced on their own (weights | architecture; later-in-backward
| earlier-in-backward for the overlapped all-reduce).  A backward kernel asks ``functional.wgrad_dest(w)`` where to put
d loss / d w: the first gradient a parameter receives after ``begin()`` is written in place by the kernel (and autograd
is told "no gradient"); any further one in the same pass (a module applied twice, e.g. the shared head under deep
supervision) goes through autograd, which accumulates into the same view.  Gradients of stacked weights
(``functional.StackedWeight``) land in the stack's own buffer and ``finish()`` adds every slice to its parameter's
view in one launch.  The flat buffer is also the RCCL all-reduce buffer and the fused optimizer's gradient table, so
nothing is gathered or re-pointed between backward, all-reduce and the optimizer step.
"""
import torch

ALIGN = 4          # floats: every view starts on a 16-byte boundary (vector stores of the gradient kernels)


def _unique(params):
    seen, out = set(), []
    for p in params:
        if id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


class GradSink(object):
    def __init__(self, segments, stacks=()):
        """segments: list of parameter lists (a parameter is placed in the first segment that names it);
        stacks: the model's ``functional.StackedWeight`` objects."""
        seen = set()
        self.segments = []
        for seg in segments:
            seg = [p for p in _unique(seg) if id(p) not in seen]
            seen.update(id(p) for p in seg)
            self.segments.append(seg)
        self.params = [p for seg in self.segments for p in seg]
        if not self.params:
            raise ValueError('GradSink: no parameters')
        dev = self.params[0].device
        off, self.bounds, offsets = 0, [], {}
        for seg in self.segments:
            lo = off
            for p in seg:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                    raise ValueError('GradSink: parameters must be contiguous fp32 tensors on one device')
                offsets[id(p)] = off
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.bounds.append((lo, off))
        self.flat = torch.zeros(max(off, 1), device=dev, dtype=torch.float32)
        self.views = {id(p): self.flat[offsets[id(p)]:offsets[id(p)] + p.numel()].view_as(p) for p in self.params}
        self._by_ptr = {p.data_ptr(): p for p in self.params}
        self._seg_of = {id(p): i for i, seg in enumerate(self.segments) for p in seg}
        self.written = set()
        # stacked weights: a persistent gradient buffer each + one table that adds every slice to its parameter's view
        self.stacks = [sw for sw in stacks if any(id(p) in self.views for p in sw.params)]
        self._stack_by_ptr = {}
        self._stacks_written = []
        self._scatter = {}
        for sw in self.stacks:
            buf = sw.buffer()
            sw.grad_buf = torch.zeros_like(buf)
            self._stack_by_ptr[buf.data_ptr()] = sw

    # ------------------------------------------------------------------ layout
    def span(self, first=0, last=None):
        """The flat slice that holds segments first..last (inclusive)."""
        last = first if last is None else last
        return self.flat[self.bounds[first][0]:self.bounds[last][1]]

    def attach(self):
        for p in self.params:
            v = self.views[id(p)]
            if p.grad is not v:
                p.grad = v

    # ------------------------------------------------------------------ one backward pass
    def begin(self, first=0, last=None):
        """Zero segments first..last and make their parameters eligible for an in-place gradient write again.
        Replaces optimizer.zero_grad()."""
        last = len(self.segments) - 1 if last is None else last
        self.span(first, last).zero_()
        self.attach()
        if first == 0 and last == len(self.segments) - 1:
            self.written.clear()
        else:
            live = set(p.data_ptr() for i in range(first, last + 1) for p in self.segments[i])
            live |= set(k for k, sw in self._stack_by_ptr.items() if any(p.data_ptr() in live for p in sw.params))
            self.written -= live
        self._stacks_written = []
        self.resume()

    def resume(self):
        """(Re)open the window in which two-stage weight gradients leave their sums to ``finish()`` (functional.DEFER);
        ``begin()`` does it, and so does the second half of a backward pass that is cut in two."""
        if self.flat.is_cuda:
            from . import functional as F
            F.DEFER = []

    def dest(self, w):
        """Tensor the kernel should write d loss / d w into, or None: not registered, frozen, detached from its view, or
        already written in this pass (autograd accumulates the later ones)."""
        key = w.data_ptr()
        if key in self.written:
            return None
        p = self._by_ptr.get(key)
        if p is not None:
            v = self.views[id(p)]
            if not p.requires_grad or p.shape != w.shape or p.grad is None or p.grad.data_ptr() != v.data_ptr():
                return None
            self.written.add(key)
            return v
        sw = self._stack_by_ptr.get(key)
        if sw is not None and w.shape == sw.grad_buf.shape and \
                all(q.requires_grad and q.grad is not None and q.grad.data_ptr() == self.views[id(q)].data_ptr() for q in sw.params):
            self.written.add(key)
            self._stacks_written.append(key)
            return sw.grad_buf
        return None

    def pending(self, w):
        """Was ``w`` (a registered parameter or stacked buffer) already handed its in-place destination in this pass?"""
        return w.data_ptr() in self.written

    def finish(self):
        """After backward: fold the deferred weight-gradient sums (a few launches for all convolutions of the pass), then
        add the slices of the stacked weight gradients written in this pass to their parameters' views (one launch; the
        table is built once per set of stacks -- under graph replay, once)."""
        from . import functional as F
        if self.flat.is_cuda:
            F.join_lanes()               # the backward kernels of the macro grid's columns ran on their own streams (grid.Lanes)
            F.flush_deferred()
        if not self._stacks_written:
            return
        from . import _lib
        from .packing import _CopyItem, copy_table
        key = tuple(self._stacks_written)
        if key not in self._scatter:
            items = []
            for k in key:
                sw = self._stack_by_ptr[k]
                gb, off = sw.grad_buf, 0
                for p in sw.params:
                    src = gb.narrow(sw.dim, off, p.shape[sw.dim])
                    off += p.shape[sw.dim]
                    dst = self.views[id(p)]
                    if sw.dim == 0:
                        items.append(_CopyItem(src.data_ptr(), dst.data_ptr(), 1, p.numel(), p.numel(), p.numel(), 1))
                    else:
                        row = p.numel() // p.shape[0]
                        items.append(_CopyItem(src.data_ptr(), dst.data_ptr(), p.shape[0], row, gb.stride(0), row, 1))
            self._scatter[key] = copy_table(items, self.flat.device)
        table, n, mx = self._scatter[key]
        _lib.check(_lib.lib().senas_copy_rows_batched(table.data_ptr(), n, mx, F._stream()), 'senas_copy_rows_batched')
        self._stacks_written = []

    # ------------------------------------------------------------------ installation
    def install(self):
        from . import functional as F
        F.SINK = self
        self.attach()
        return self

    def uninstall(self):
        from . import functional as F
        if F.SINK is self:
            F.SINK = None
            F.DEFER = None
