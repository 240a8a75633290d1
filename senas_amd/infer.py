"""Inference / validation pass: what ``experiments/testing_model.py:150-190`` and the
validation loops of the drivers (``train_model.py`` / ``search_arc.py`` ``infer``) do per batch --
``model.eval()``, ``torch.no_grad()``, forward, criterion on ``predicts[-1]``, ``metric.update(target, predicts[-1])``,
``argmax`` masks -- as one stream-ordered, HIP-graph-replayed launch list on static buffers with no host round trip
per batch (the reference syncs for ``loss.item()`` and three times inside the metric, every step).

Eval mode changes what the kernels do: BatchNorm uses the running statistics, so no producer kernel accumulates
channel sums (SE blocks excepted: their squeeze still needs the per-image mean), and no node keeps a ReLU mask.
"""
import torch

from .arena import reset_arena
from .metrics import SegmentationMetric
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
        self.packer = WeightPacker(model)          # weights are constant here: one refresh, then every replay reuses it
        self.packer.install()
        self.packer.refresh()
        if use_graph:
            self._capture(warmup)

    @torch.no_grad()
    def _eager(self):
        logits = self.model(self.x)[-1]
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
        reset_arena()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.logits, self.mask = self._eager()
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
        if self.graph is None:
            self.logits, self.mask = self._eager()
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
