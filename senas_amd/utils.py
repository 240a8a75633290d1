"""Small host-side helpers of the hot path (reference: utils/utils.py:21-40,240-250)."""
import torch


def weights_init(m):
    """kaiming-normal (fan_out, relu) for conv / transposed conv, BN weight 1 / bias 0,
    xavier-normal Linear -- utils/utils.py:240-250."""
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)
    elif isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        torch.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
    elif isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.constant_(m.weight, 1)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)


def channel_shuffle(x, groups):
    """[N,C,H,W] -> interleave ``groups`` channel groups (dead on the SENAS path: MixedOp.k == 1)."""
    n, c, h, w = x.size()
    return x.view(n, groups, c // groups, h, w).transpose(1, 2).contiguous().view(n, -1, h, w)
