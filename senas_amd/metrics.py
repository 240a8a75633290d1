"""pixAcc / mIoU / Dice bookkeeping (reference: utils/metrics.py:11-71,127-173).

Same ``SegmentationMetric(nclass).update(labels, preds) / .get() / .reset()`` surface.  Counts
are accumulated on the device as int64 and only brought to the host in ``get()`` -- the reference
syncs three times per step (metrics.py:159-161).
"""
import numpy as np
import torch

SMOOTH = np.spacing(1)


class SegmentationMetric(object):
    def __init__(self, nclass):
        self.nclass = nclass
        self.reset()

    def reset(self):
        self._acc_sum, self._acc_n, self._counts = None, 0, None

    @torch.no_grad()
    def update(self, labels, preds):
        seg = preds.argmax(1)                              # argmax of softmax == argmax of logits
        fg = labels > 0
        # mean_pix_accuracy (metrics.py:127-142): bitwise AND of the arg-max with (target > 0)
        labeled = fg.float().sum((1, 2))
        correct = (seg & fg).float().sum((1, 2))
        acc = ((correct + SMOOTH) / (labeled + SMOOTH)).mean()
        self._acc_sum = acc if self._acc_sum is None else self._acc_sum + acc
        self._acc_n += 1
        rows = []
        for c in range(1, self.nclass):
            p, t = seg == c, labels == c
            rows.append(torch.stack([(p & t).sum(), (p & ~t).sum(), (~p & t).sum()]))
        cnt = torch.stack(rows)                            # [nclass-1, 3] int64 on the device
        self._counts = cnt if self._counts is None else self._counts + cnt

    def counts(self):
        c = self._counts.cpu().numpy().astype(np.float32)
        return c[:, 0], c[:, 1], c[:, 2]

    def get(self):
        tp, fp, fn = self.counts()
        pix = round(100.0 * float(self._acc_sum.item() / self._acc_n), 3)
        miou = round(100.0 * float(np.mean((tp + SMOOTH) / (tp + fp + fn + SMOOTH))), 3)
        dice = round(100.0 * float(np.mean((2 * tp + SMOOTH) / (2 * tp + fp + fn + SMOOTH))), 3)
        return pix, miou, dice
