"""Stream-ordered train / search steps (the step loop of
experiments/train_model.py:264-305 and experiments/search_arc.py:252-299, without the per-step host
syncs), with the launch-bound part -- forward, loss, backward: ~10^3 short kernels -- captured once in
a HIP graph and replayed.

What is inside the graph: forward, criterion, backward.  Parameter gradients do not travel through autograd:
the backward kernels write them into one flat buffer (``gradsink.GradSink``; ``p.grad`` are persistent views
of it), which is also the all-reduce buffer and the fused optimizer's gradient table.  What stays eager: the
gradient all-reduce (RCCL is never called inside a capture), clip_grad_norm_ and the optimizer step -- a few launches.

With more than one rank the backward pass is captured as TWO graphs cut at the outputs of the down path
(``grid.MacroGrid.cut``): the up-path / head gradients -- the first ~70 % of the flat buffer, complete once the first
graph has run -- are all-reduced while the second graph (down cells, stems) still runs; only the second, smaller
all-reduce is exposed.
"""
import os

import torch
import torch.distributed as dist

from . import functional as F
from . import optim
from .arena import reset_arena
from .gradsink import GradSink
from .grid import Lanes, MacroGrid
from .lanesched import LaneSchedule
from .packing import WeightPacker
from .parallel import SinkReducer


LANES_FOR_WORLD = int(os.environ.get('SENAS_LANES_FOR_WORLD', 3))      # scheduler streams when world_size > 1


def _lib_error():
    from ._lib import SenasHipError
    return SenasHipError


def _macro_grid(model):
    for m in model.modules():
        if isinstance(m, MacroGrid):
            return m
    return None


def _segments(model, exclude=()):
    """[later-in-backward weights (up cells, head), earlier-in-backward weights (stems, down cells)] of a model built
    on the macro grid, else [all parameters]."""
    skip = set(id(p) for p in exclude)
    grid = _macro_grid(model)
    if grid is None:
        return [[p for p in model.parameters() if id(p) not in skip]]
    down = [p for p in grid.down_parameters() if id(p) not in skip]
    down_ids = set(id(p) for p in down)
    up = [p for p in model.parameters() if id(p) not in skip and id(p) not in down_ids]
    return [up, down]


class GraphedForwardBackward(object):
    """Captures ``loss = criterion(model(x), y); loss.backward()`` on static input buffers.

    ``reducer``: the ``SinkReducer`` of the segments this pass produces gradients for (it zeroes them first).
    ``frozen``: parameters whose gradients this pass does not need (``requires_grad`` is off while the pass is built
    or run eagerly, so their weight-gradient kernels are never launched).
    ``early``: a ``SinkReducer`` over the leading segments whose gradients are complete once the part of backward above
    the macro grid's cut has run; given (and more than one rank), backward is split there and that all-reduce overlaps
    the rest of backward."""

    def __init__(self, model, criterion, x, y, reducer, warmup=2, use_graph=True, packer=None, frozen=(), early=None,
                 count_nodes=False, refresh=True, max_lanes=None):
        self.model, self.criterion, self.reducer = model, criterion, reducer
        # refresh=False: the caller guarantees that the packed / stacked weight images are current when the pass starts
        # (the weight pass of a search step right after the architecture pass: the weights have not moved in between)
        self.refresh = refresh
        self.count_nodes, self.nodes = count_nodes, None
        self.x, self.y = x, y                      # static buffers; refill with .copy_() between steps
        self.loss = None
        self.graph = self.graph_tail = None
        self.sched = self.sched_tail = None        # lane schedulers of the captured passes (None: the runtime's own graph replay)
        self.wlane = None                          # the weight-gradient lane of this driver's passes (functional.WLANE), or None
        self.max_lanes = max_lanes                 # streams of the lane scheduler (None: lanesched.MAX_LANES)
        if reducer.world > 1:
            # the collective's own stream is busy while the second backward graph runs: with it, three scheduler streams beat four
            # on both steps (rehearsed with a stand-in kernel stream on one GPU: profiles/r5_rccl_standin.txt)
            self.max_lanes = min(int(max_lanes or LANES_FOR_WORLD), LANES_FOR_WORLD)
        self.frozen = [p for p in frozen if p.requires_grad]
        self.grid = _macro_grid(model)
        if (self.grid is not None and Lanes.enabled and self.grid._depth > 2 and not getattr(self.grid, '_supervision', False)
                and next(model.parameters()).is_cuda):
            # weight-gradient kernels beside the pass (functional.wgrad_lane).  Not under deep supervision: the shared head
            # receives several gradients per pass, and the later ones are accumulated by autograd into the view the first one's
            # kernel writes
            self.wlane = F.own_stream(next(model.parameters()).device, 'wgrad')
        if self.grid is not None and any(isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.p > 0 for m in model.modules()):
            # torch's graph replay advances the Philox offsets of the captured dropout draws; the lane scheduler replays the
            # captured launches as they are -- a network with dropout keeps the serial schedule and torch's own replay
            self.grid.lanes = False
        self.early = early if (early is not None and early.world > 1 and self.grid is not None) else None
        self._work = None
        self._collectives = True
        if packer is None:
            packer = WeightPacker(model)           # one launch per step refreshes every conv's weight image
            packer.install()
        self.packer = packer
        if use_graph:
            self._capture(warmup)

    # ------------------------------------------------------------------ the pass, whole or in two parts
    def _head(self):
        """zero_grad, forward, loss and -- without a cut -- all of backward; with a cut, backward down to the cut."""
        self.reducer.zero_grad()
        if self.refresh:
            self.packer.refresh()
        for p in self.frozen:
            p.requires_grad_(False)
        cut = self.early is not None
        if cut:
            self._cut_src, self._cut_leaf = [], []
            self.grid.cut = self._cut
        F.WLANE = self.wlane if (self.grid is not None and self.grid.lanes and Lanes.enabled) else None
        del F._WQ[:]                      # (weight gradients a failed pass may have left queued belong to tensors that are gone)
        try:
            loss = self.criterion(self.model(self.x), self.y)
            loss.backward()
            F.join_lanes()                # the macro grid's columns ran on their own streams (grid.Lanes): the flat gradient buffer is complete after this
        finally:
            F.WLANE = None
            if cut:
                self.grid.cut = None
            if not cut:
                for p in self.frozen:
                    p.requires_grad_(True)
        self.reducer.after_backward()
        return loss.detach()

    def _cut(self, t):
        """grid.MacroGrid.cut: one tensor of the down path (the stem output, a down cell's output) -> the leaf the up path reads."""
        leaf = t.detach().requires_grad_(t.requires_grad)
        st = getattr(t, '_senas_stats', None)
        if st is not None:
            leaf._senas_stats = st
        if t.requires_grad:
            self._cut_src.append(t)
        self._cut_leaf.append(leaf)
        return leaf

    def _tail(self):
        """The rest of backward below the cut."""
        self.reducer.sink.resume()
        F.WLANE = self.wlane if (self.grid is not None and self.grid.lanes and Lanes.enabled) else None
        try:
            leaves = [l for l in self._cut_leaf if l.requires_grad]
            pairs = [(s, l.grad) for s, l in zip(self._cut_src, leaves) if l.grad is not None]
            if pairs:
                torch.autograd.backward([s for s, _ in pairs], [g for _, g in pairs])
            F.join_lanes()
        finally:
            F.WLANE = None
            self._cut_src = self._cut_leaf = None
            for p in self.frozen:
                p.requires_grad_(True)
        self.reducer.after_backward()

    def _eager(self):
        loss = self._head()
        if self.early is not None:
            self._launch_early()
            self._tail()
        return loss

    def _launch_early(self):
        if not self._collectives:
            return
        e = self.early
        self._work = dist.all_reduce(e.sink.span(e.first, e.last), op=dist.ReduceOp.SUM, group=e.group, async_op=True)

    # ------------------------------------------------------------------ capture / replay
    def _capture(self, warmup):
        # the eager warm-up passes run the model in training mode: batch-norm running statistics (and their step
        # counters) would advance on whatever the static buffers hold -- put them back afterwards
        buffers = [b for b in self.model.buffers()]
        kept = [b.detach().clone() for b in buffers]
        self._collectives = False                  # the warm-up passes run cut in two like the captured ones, without the all-reduce
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._collectives = True
        try:
            graph = self._capture_graphs()
        except _lib_error() as err:
            # the lane scheduler could not rebuild the captured pass (a node type it does not re-issue, a runtime error): the
            # pass is captured again on ONE stream and replayed by the runtime -- never a multi-branch graph on its executor
            if self.grid is None or not self.grid.lanes:
                raise
            import sys
            sys.stderr.write('[senas_amd.step] lane scheduler unavailable (%s): this pass keeps the serial schedule\n' % err)
            self.grid.lanes = False
            self.sched = self.sched_tail = self.graph_tail = None
            graph = self._capture_graphs()
        reset_arena()
        self.graph = graph
        with torch.no_grad():
            for b, k in zip(buffers, kept):
                b.copy_(k)

    def _capture_graphs(self):
        reset_arena()
        # a pass that runs on several lanes (grid.Lanes) is captured but never handed to the runtime's graph executor: the
        # captured graph is replayed by the lane scheduler (lanesched.LaneSchedule, csrc/sched.hip)
        if os.environ.get('SENAS_NO_SCHED') and self.grid is not None:
            # without the lane scheduler the pass is captured on ONE stream: a multi-branch capture is never handed to the
            # runtime's own graph executor (SIGSEGV in hip::Graph::UpdateStreams: profiles/r4_graph_executor.txt)
            self.grid.lanes = False
        lanes = Lanes.enabled and self.grid is not None and self.grid.lanes and self.grid._depth > 2
        keep = lanes or self.count_nodes
        graph = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'), Lanes.scheduled():   # (other threads -- the RCCL watchdog -- may touch the runtime)
            self.loss = self._head()
        if self.count_nodes:
            self.nodes = _graph_nodes(graph)
        if lanes:
            self.sched = LaneSchedule(graph, self.max_lanes)
        elif keep:
            graph.instantiate()              # (a single-stream capture kept for its node count)
        if self.early is not None:
            tail = torch.cuda.CUDAGraph(keep_graph=True) if lanes else torch.cuda.CUDAGraph()
            with torch.cuda.graph(tail, pool=graph.pool(), capture_error_mode='thread_local'), Lanes.scheduled():
                self._tail()
            self.graph_tail = tail
            if lanes:
                self.sched_tail = LaneSchedule(tail, self.max_lanes)
        return graph

    def __call__(self):
        if self.graph is None:
            self.loss = self._eager()
        else:
            if self.sched is not None:
                self.sched.launch()
            else:
                self.graph.replay()
            if self.refresh:
                self.packer.mark_refreshed()       # the replay starts with the packer's refresh launches
            if self.graph_tail is not None:
                self._launch_early()
                if self.sched_tail is not None:
                    self.sched_tail.launch()
                else:
                    self.graph_tail.replay()
        return self.loss

    def finish(self):
        """After the pass: the (remaining) all-reduce; gradients are averaged over the ranks when this returns
        (stream-ordered)."""
        r, e = self.reducer, self.early
        if r.world == 1:
            return
        if e is None:
            r.finish()
            return
        rest = r.sink.span(e.last + 1, r.last)
        dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=r.group)
        if self._work is not None:
            self._work.wait()
            self._work = None
        else:                                      # the early all-reduce was not launched (collectives off during the pass)
            dist.all_reduce(e.sink.span(e.first, e.last), op=dist.ReduceOp.SUM, group=e.group)
        r.sink.span(r.first, r.last).mul_(1.0 / r.world)


def _graph_nodes(graph):
    """Number of nodes (kernel launches, memsets, copies) of a captured graph that was kept (``keep_graph=True``), or
    None when the runtime does not say."""
    try:
        import ctypes as C
        hip = C.CDLL('libamdhip64.so')
        n = C.c_size_t(0)
        err = hip.hipGraphGetNodes(C.c_void_p(graph.raw_cuda_graph()), None, C.byref(n))
        return int(n.value) if err == 0 else None
    except (OSError, AttributeError, RuntimeError):
        return None


# "critical" policy (round 5): the number of streams the lane scheduler deals a pass's segments to (at most the four hardware queues).
# "chain" policy (round 4): the chains it covers a pass of the search step with -- 5 and 6 within the run-to-run spread, 4 and 7 lose
# 0.7 - 1.0 ms (profiles/r4_wlane_modes.txt).
SEARCH_LANES = int(os.environ.get('SENAS_SEARCH_LANES', 6))


def _model_stacks(model):
    return [sw for m in model.modules() if hasattr(m, 'stacked_weights') for sw in m.stacked_weights()]


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
                 use_graph=True, process_group=None, count_nodes=False):
        self.params = [p for p in model.parameters()]
        arch = [p for g in arch_optimizer.param_groups for p in g['params']]
        arch_ids = set(id(p) for p in arch)
        weights = [p for p in self.params if id(p) not in arch_ids]
        self.opt_w, self.opt_a, self.grad_clip = weight_optimizer, arch_optimizer, grad_clip
        packer = WeightPacker(model)
        packer.install()
        # flat gradient buffer: [up-path weights | down-path weights | architecture]
        segs = _segments(model, exclude=arch) + [arch]
        self.sink = GradSink(segs, _model_stacks(model)).install()
        last = len(segs) - 1
        self.reducer = SinkReducer(self.sink, 0, last, world_size, process_group)
        self.arch_reducer = SinkReducer(self.sink, last, last, world_size, process_group)
        early = SinkReducer(self.sink, 0, 0, world_size, process_group) if (world_size > 1 and last >= 2) else None
        self.fb_arch = GraphedForwardBackward(model, criterion, x, y, self.arch_reducer, use_graph=use_graph, packer=packer,
                                              frozen=weights, count_nodes=count_nodes, max_lanes=SEARCH_LANES)
        # the weight pass does not repack: either the architecture pass just did, or __call__ does it (before alpha_begin)
        packer.refresh()
        self.packer = packer
        self.fb = GraphedForwardBackward(model, criterion, x, y, self.reducer, use_graph=use_graph, packer=packer, early=early,
                                         count_nodes=count_nodes and early is None, refresh=False, max_lanes=SEARCH_LANES)
        self.graphed = self.fb.graph is not None
        # static gradient addresses -> clip + SGD in two launches instead of ~110
        self.fused = optim.FusedClipSGD(weight_optimizer, grad_clip) if optim.supported(weight_optimizer) else None

    def close(self):
        """Uninstall the packed-weight cache and the gradient sink (``p.grad`` stay views of the flat buffer)."""
        _close(self)

    def graph_nodes(self):
        """Launches per step inside the two captured graphs (needs ``count_nodes=True``), or None."""
        a, b = self.fb_arch.nodes, self.fb.nodes
        return a + b if (a is not None and b is not None) else None

    def __call__(self, x_train, y_train, x_valid=None, y_valid=None):
        fb = self.fb
        if x_valid is not None:                    # architecture step (skipped before alpha_begin)
            fb.x.copy_(x_valid, non_blocking=True)
            fb.y.copy_(y_valid, non_blocking=True)
            self.fb_arch()
            self.fb_arch.finish()
            self.opt_a.step()
        elif self.packer.stale():                  # no architecture pass in front (before alpha_begin): repack here
            self.packer.refresh()
        fb.x.copy_(x_train, non_blocking=True)
        fb.y.copy_(y_train, non_blocking=True)
        loss = fb()
        fb.finish()
        if self.fused is not None:
            self.fused.step()
        else:
            if self.grad_clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
            self.opt_w.step()
        F.weights_moved()                          # the packed / stacked images are stale until the next pass refreshes them
        return loss


def _close(step):
    step.sink.uninstall()
    step.fb.packer.uninstall()


class TrainStep(object):
    """One optimisation step of the derived network: graph(fwd+loss+bwd) -> all-reduce -> clip -> SGD."""

    def __init__(self, model, criterion, optimizer, x, y, world_size=1, grad_clip=5.0, use_graph=True, process_group=None):
        self.params = [p for p in model.parameters()]
        self.optimizer, self.grad_clip, self.world = optimizer, grad_clip, world_size
        packer = WeightPacker(model)
        packer.install()
        segs = _segments(model)
        self.sink = GradSink(segs, _model_stacks(model)).install()
        self.reducer = SinkReducer(self.sink, 0, len(segs) - 1, world_size, process_group)
        early = SinkReducer(self.sink, 0, 0, world_size, process_group) if (world_size > 1 and len(segs) > 1) else None
        self.fb = GraphedForwardBackward(model, criterion, x, y, self.reducer, use_graph=use_graph, packer=packer, early=early)
        self.graphed = self.fb.graph is not None
        self.fused = optim.FusedClipSGD(optimizer, grad_clip) if optim.supported(optimizer) else None

    def close(self):
        _close(self)

    def __call__(self):
        loss = self.fb()
        self.fb.finish()
        if self.fused is not None:
            self.fused.step()
        else:
            if self.grad_clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
            self.optimizer.step()
        F.weights_moved()                          # the packed images are stale until the next pass refreshes them
        return loss
