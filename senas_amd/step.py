"""Stream-ordered train / search steps (the step loop of
experiments/train_model.py:264-305 and experiments/search_arc.py:252-299, without the per-step host
syncs), with the launch-bound part -- forward, loss, backward: ~10^3 short kernels -- captured once in
a HIP graph and replayed.

What is inside the graph: forward, criterion, backward (gradients are written to tensors owned by the
graph's memory pool).  What stays eager: the gradient all-reduce (RCCL is never called inside a
capture), clip_grad_norm_ and the optimizer step -- a dozen launches.
"""
import torch

from . import functional as F
from . import optim
from .arena import reset_arena
from .packing import WeightPacker
from .parallel import GradAllReducer


class GraphedForwardBackward(object):
    """Captures ``loss = criterion(model(x), y); loss.backward()`` on static input buffers.

    ``frozen``: parameters whose gradients this pass does not need (``requires_grad`` is off while the pass is built
    or run eagerly, so their weight-gradient kernels are never launched).  ``repoint``: parameters whose ``.grad``
    must be re-pointed at this graph's gradient tensors after a replay (needed when another graph writes gradients
    of the same parameters elsewhere)."""

    def __init__(self, model, criterion, x, y, reducer, warmup=2, use_graph=True, packer=None, frozen=(), repoint=()):
        self.model, self.criterion, self.reducer = model, criterion, reducer
        self.x, self.y = x, y                      # static buffers; refill with .copy_() between steps
        self.loss = None
        self.graph = None
        self.graph_grads = None
        self.frozen = [p for p in frozen if p.requires_grad]
        self.repoint_ids = set(id(p) for p in repoint)
        if packer is None:
            packer = WeightPacker(model)           # one launch per step refreshes every conv's weight image
            packer.install()
        self.packer = packer
        if use_graph:
            self._capture(warmup)

    def _eager(self):
        self.reducer.zero_grad()
        self.packer.refresh()
        for p in self.frozen:
            p.requires_grad_(False)
        try:
            loss = self.criterion(self.model(self.x), self.y)
            loss.backward()
        finally:
            for p in self.frozen:
                p.requires_grad_(True)
        return loss.detach()

    def _capture(self, warmup):
        if self.reducer.overlap:
            raise ValueError('a graph-replayed backward cannot drive gradient hooks; use overlap=False')
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        reset_arena()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):   # other threads (the RCCL watchdog) may touch the runtime
            self.loss = self._eager()
        reset_arena()
        self.graph = graph
        self.graph_grads = [p.grad for p in self.reducer.params]     # written in place by every replay
        self.repoint = [(p, g) for p, g in zip(self.reducer.params, self.graph_grads) if id(p) in self.repoint_ids]

    def __call__(self):
        if self.graph is None:
            self.loss = self._eager()
        else:
            self.graph.replay()
            F.PACKED_VALID = True                  # the replay starts with the packer's refresh launches
            if self.reducer.world > 1:             # reduce_all() re-points p.grad at the flat buffer
                for p, g in zip(self.reducer.params, self.graph_grads):
                    p.grad = g
            else:
                for p, g in self.repoint:
                    p.grad = g
        return self.loss


class SearchStep(object):
    """One search step as experiments/search_arc.py:252-299 runs it after ``alpha_begin``:
    ``Architecture.step`` on a validation batch (first-order: forward/backward, Adam on alpha/beta/gamma),
    then the weight step on a training batch (SGD over ALL parameters -- architecture included -- after
    clip_grad_norm_).

    The two passes are captured separately.  The architecture pass only needs d loss / d (alpha, beta, gamma): the
    weight gradients it would also produce are thrown away by the ``model_optimizer.zero_grad()`` that follows
    (search_arc.py:271), so that pass is built with the weights frozen -- no weight-gradient kernel runs, and its
    all-reduce carries 246 floats instead of 7.9 MB.  The results are identical to the reference's order of operations."""

    def __init__(self, model, criterion, weight_optimizer, arch_optimizer, x, y, world_size=1, grad_clip=5.0,
                 use_graph=True):
        self.params = [p for p in model.parameters()]
        arch = [p for g in arch_optimizer.param_groups for p in g['params']]
        arch_ids = set(id(p) for p in arch)
        weights = [p for p in self.params if id(p) not in arch_ids]
        self.reducer = GradAllReducer(self.params, world_size=world_size)
        self.arch_reducer = GradAllReducer(arch, world_size=world_size)
        self.opt_w, self.opt_a, self.grad_clip = weight_optimizer, arch_optimizer, grad_clip
        self.fb_arch = GraphedForwardBackward(model, criterion, x, y, self.arch_reducer, use_graph=use_graph, frozen=weights,
                                              repoint=arch)
        self.fb = GraphedForwardBackward(model, criterion, x, y, self.reducer, use_graph=use_graph, packer=self.fb_arch.packer,
                                         repoint=arch)
        self.graphed = self.fb.graph is not None
        # static gradient addresses (graph replays) -> clip + SGD in two launches instead of ~110
        self.fused = optim.FusedClipSGD(weight_optimizer, grad_clip) if (self.graphed and optim.supported(weight_optimizer)) else None

    def __call__(self, x_train, y_train, x_valid=None, y_valid=None):
        fb = self.fb
        if x_valid is not None:                    # architecture step (skipped before alpha_begin)
            fb.x.copy_(x_valid, non_blocking=True)
            fb.y.copy_(y_valid, non_blocking=True)
            self.fb_arch()
            self.arch_reducer.finish()
            self.opt_a.step()
        fb.x.copy_(x_train, non_blocking=True)
        fb.y.copy_(y_train, non_blocking=True)
        loss = fb()
        self.reducer.finish()
        if self.fused is not None:
            self.fused.step()
        else:
            if self.grad_clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
            self.opt_w.step()
        F.PACKED_VALID = False                     # the weights moved: the packed / stacked images are stale until the next pass
        return loss


class TrainStep(object):
    """One optimisation step of the derived network: graph(fwd+loss+bwd) -> all-reduce -> clip -> SGD."""

    def __init__(self, model, criterion, optimizer, x, y, world_size=1, grad_clip=5.0, use_graph=True, overlap=False):
        self.params = [p for p in model.parameters()]
        self.reducer = GradAllReducer(self.params, world_size=world_size, overlap=overlap and not use_graph)
        self.optimizer, self.grad_clip, self.world = optimizer, grad_clip, world_size
        self.fb = GraphedForwardBackward(model, criterion, x, y, self.reducer, use_graph=use_graph)
        self.graphed = self.fb.graph is not None
        self.fused = optim.FusedClipSGD(optimizer, grad_clip) if (self.graphed and optim.supported(optimizer)) else None

    def __call__(self):
        loss = self.fb()
        self.reducer.finish()
        if self.fused is not None:
            self.fused.step()
        else:
            if self.grad_clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
            self.optimizer.step()
        F.PACKED_VALID = False                     # the weights moved: the packed images are stale until the next pass
        return loss
