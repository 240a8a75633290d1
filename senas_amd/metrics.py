"""pixAcc / mIoU / Dice bookkeeping (reference: utils/metrics.py:11-71,127-173).

Same ``SegmentationMetric(nclass).update(labels, preds) / .get() / .reset()`` surface.  ``update`` is one
device kernel pair (``senas_seg_metric_update``): counts are accumulated on the device as
int64 and only brought to the host in ``get()`` -- the reference syncs three times per step
(metrics.py:159-161).  No CPU path.
"""
import numpy as np
import torch

from . import _lib
from . import functional as F
from .arena import zeros64

SMOOTH = np.spacing(1)


class SegmentationMetric(object):
    def __init__(self, nclass):
        if not 2 <= nclass <= 8:
            raise _lib.SenasHipError('SegmentationMetric: %d classes (supported: 2..8)' % nclass)
        self.nclass = nclass
        self.reset()

    def reset(self):
        self._acc_sum, self._acc_n, self._counts = None, 0, None

    def reset_counts(self):
        """Zero the device accumulators in place (a HIP graph that captured ``update`` keeps their addresses)."""
        if self._counts is None:
            return
        self._counts.zero_()
        self._acc_sum.zero_()
        self._acc_n = 0

    @torch.no_grad()
    def update(self, labels, preds):
        """One launch pair (senas_seg_metric_update): arg-max, per-image pixel accuracy, per-class tp/fp/fn -- all
        accumulated on the device; nothing reaches the host before get()."""
        x = F.nhwc(preds)
        n, c, h, w = x.shape
        if c != self.nclass or not labels.is_cuda or tuple(labels.shape) != (n, h, w):
            raise _lib.SenasHipError('SegmentationMetric.update: labels %s / preds %s do not match %d classes' %
                                     (tuple(labels.shape), tuple(preds.shape), self.nclass))
        if self._counts is None:
            self._counts = torch.zeros((c - 1, 3), device=x.device, dtype=torch.int64)
            self._acc_sum = torch.zeros(1, device=x.device, dtype=torch.float64)
        t = labels.long().contiguous()
        part = zeros64((2 * n + 3 * (c - 1),), x.device)          # 8-byte zeroed slots, used as uint64 counters
        _lib.check(_lib.lib().senas_seg_metric_update(n, h * w, c, x.data_ptr(), t.data_ptr(), float(SMOOTH), part.data_ptr(),
                                                      self._counts.data_ptr(), self._acc_sum.data_ptr(), F._stream()),
                   'senas_seg_metric_update')
        self._acc_n += 1

    def counts(self):
        c = self._counts.cpu().numpy().astype(np.float32)
        return c[:, 0], c[:, 1], c[:, 2]

    def get(self):
        tp, fp, fn = self.counts()
        pix = round(100.0 * float(self._acc_sum.item() / self._acc_n), 3)
        miou = round(100.0 * float(np.mean((tp + SMOOTH) / (tp + fp + fn + SMOOTH))), 3)
        dice = round(100.0 * float(np.mean((2 * tp + SMOOTH) / (2 * tp + fp + fn + SMOOTH))), 3)
        return pix, miou, dice
