#!/usr/bin/env python3
"""Architecture gradients of the fused arch-mix path against the plain torch-softmax path of the same model on the GPU."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.senas_search import NAS  # noqa: E402


def run(net, x, y, plain):
    net._plain_arch = plain
    net.zero_grad(set_to_none=True)
    SegmentationLosses('dice_ce')(net(x), y).backward()
    return {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters() if 'net.' not in k}


def main():
    dev = torch.device('cuda:0')
    for share, depth in ((False, 4), (True, 4), (False, 2)):
        torch.manual_seed(3)
        net = NAS(1, 8, 2, depth, meta_node_num=3, use_sharing=share, double_down_channel=False, device=dev).to(dev).train()
        with torch.no_grad():
            for p in net.arch_parameters():
                p.copy_(torch.randn_like(p) * 0.7)
        x = torch.randn(2, 1, 64, 64, device=dev)
        y = torch.randint(0, 2, (2, 64, 64), device=dev)
        buf = {k: v.detach().clone() for k, v in net.state_dict().items()}
        a = run(net, x, y, True)
        net.load_state_dict(buf)
        b = run(net, x, y, False)
        net.load_state_dict(buf)
        c = run(net, x, y, True)
        for k in a:
            if a[k] is None or b[k] is None:
                print('share=%s depth=%d %-14s plain %s fused %s' % (share, depth, k, None if a[k] is None else tuple(a[k].shape), None if b[k] is None else tuple(b[k].shape)))
                continue
            if a[k].numel() == 0:
                continue
            sc = float(a[k].abs().max()) + 1e-30
            print('share=%s depth=%d %-14s fused-vs-plain %.2e   plain-vs-plain rerun %.2e' % (share, depth, k, float((a[k] - b[k]).abs().max()) / sc,
                                                                                              float((a[k] - c[k]).abs().max()) / sc))


if __name__ == '__main__':
    main()
