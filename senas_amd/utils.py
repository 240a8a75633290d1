"""Small host-side helpers of the hot path."""
import torch.nn as nn


def _init_conv(m):
    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


def _init_norm(m):
    nn.init.ones_(m.weight)
    if m.bias is not None:
        nn.init.zeros_(m.bias)


def _init_linear(m):
    nn.init.xavier_normal_(m.weight)
    if m.bias is not None:
        nn.init.zeros_(m.bias)


_INITIALISERS = ((nn.Conv2d, _init_conv), (nn.ConvTranspose2d, _init_conv), (nn.BatchNorm2d, _init_norm), (nn.Linear, _init_linear))


def weights_init(m):
    """``model.apply(weights_init)``: the initialisation scheme of the reference (utils/utils.py:240-250) -- He-normal
    (fan-out, ReLU gain) convolution kernels, unit batch-norm scale / zero shift, Glorot-normal linear layers."""
    for kind, init in _INITIALISERS:
        if isinstance(m, kind):
            init(m)
            return


def channel_shuffle(x, groups):
    """Interleave ``groups`` channel groups of an [N, C, H, W] tensor (PC-DARTS partial channels; dead on the SENAS
    path because MixedOp.k == 1)."""
    n, c, h, w = x.shape
    return x.reshape(n, groups, c // groups, h, w).transpose(1, 2).reshape(n, c, h, w)
