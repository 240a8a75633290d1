"""Segmentation losses used by the search / train steps (reference: utils/loss/loss.py).

``SegmentationLosses(name)(outputs, target)`` keeps the reference call shape: ``outputs`` is the
list the models return and the loss is taken on ``outputs[-1]`` (loss.py:26-27).
dice_ce = SoftDiceLoss (soft TP/FP/FN over batch+space, background class dropped, smooth 1e-5,
denominator + 1e-8; loss.py:45-70,173-228) + nn.CrossEntropyLoss (loss.py:124-159).

dice_ce / dice_loss run as ``senas_dice_ce_fwd`` / ``senas_dice_ce_bwd``: one pass over
the logits per direction, no one-hot tensor, no host round trip (the reference builds the one-hot on the CPU
and copies it over, loss.py:199-203).  Like every op of this package there is no CPU path.
"""

import torch
import torch.nn as nn

from . import _lib
from . import functional as F
from .arena import zeros64

MAX_CLASSES = 8


class _DiceCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, w_ce, w_dice, smooth, do_bg):
        x = F.nhwc(logits)
        n, c, h, w = x.shape
        if c > MAX_CLASSES:
            raise _lib.SenasHipError('dice_ce: %d classes (supported: 1..%d)' % (c, MAX_CLASSES))
        if not target.is_cuda or tuple(target.shape) != (n, h, w):
            raise _lib.SenasHipError('dice_ce: target must be a CUDA tensor of shape %s' % ((n, h, w),))
        t = target.long().contiguous()
        acc = zeros64((1 + 3 * c,), x.device)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        coef = torch.empty(3 * c + 1, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().senas_dice_ce_fwd(n * h * w, c, x.data_ptr(), t.data_ptr(), w_ce, w_dice, smooth, int(do_bg),
                                                acc.data_ptr(), loss.data_ptr(), coef.data_ptr(), F._stream()), 'senas_dice_ce_fwd')
        ctx.save_for_backward(x, t, coef)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        x, t, coef = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x, memory_format=F.CL)
        dl = dloss.reshape(1).float().contiguous()
        _lib.check(_lib.lib().senas_dice_ce_bwd(n * h * w, c, x.data_ptr(), t.data_ptr(), coef.data_ptr(), dl.data_ptr(),
                                                dx.data_ptr(), F._stream()), 'senas_dice_ce_bwd')
        return dx, None, None, None, None, None


def dice_ce_loss(logits, target, weight_ce=1.0, weight_dice=1.0, smooth=1e-5, do_bg=False):
    return _DiceCE.apply(logits, target, float(weight_ce), float(weight_dice), float(smooth), bool(do_bg))


def soft_dice_loss(logits, target, smooth=1e-5, do_bg=False):
    return dice_ce_loss(logits, target, 0.0, 1.0, smooth, do_bg)


class DiceCrossEntropyLoss(nn.Module):
    def __init__(self, weight_ce=1, weight_dice=1):
        super().__init__()
        self.weight_ce, self.weight_dice = weight_ce, weight_dice

    def forward(self, net_output, target):
        return dice_ce_loss(net_output, target, self.weight_ce, self.weight_dice)


class SegmentationLosses(nn.Module):
    def __init__(self, name='dice_ce'):
        super().__init__()
        if name == 'cross_entropy':
            self.loss = nn.CrossEntropyLoss()
        elif name == 'dice_ce':
            self.loss = DiceCrossEntropyLoss()
        elif name == 'dice_loss':
            self.loss = soft_dice_loss
        else:
            raise NotImplementedError(name)

    def forward(self, outputs, target):
        return self.loss(outputs[-1], target)


class MultiSegmentationLosses(nn.Module):
    """Deep-supervision loss (reference: utils/loss/loss.py:30-43, chosen at experiments/search_arc.py:107 and
    train_model.py:111 when ``deep_supervision: True``): ``sum_i w_i * loss([outputs[i]], target) / len(outputs)`` over the
    list a model built with ``supervision=True`` returns.  Every addend is the fused dice_ce kernel pair; the weighted
    sum is host arithmetic on 0-d tensors (``depth`` of them)."""

    def __init__(self, name, depth, weight_factors=None):
        super().__init__()
        self.loss = SegmentationLosses(name)
        factors = [1] * depth if weight_factors is None else list(weight_factors)
        if len(factors) != depth:
            raise ValueError('MultiSegmentationLosses: %d weight factors for depth %d' % (len(factors), depth))
        self.weight_factors = factors

    def forward(self, outputs, target):
        total = None
        for factor, logits in zip(self.weight_factors, outputs):      # (zip: outputs beyond ``depth`` are ignored, as there)
            term = self.loss([logits], target) * factor
            total = term if total is None else total + term
        return total / len(outputs)
