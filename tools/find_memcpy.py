#!/usr/bin/env python3
"""Which host-side torch ops issue device-to-device copies (rocclr copyBuffer) in one eager supernet pass."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.senas_search import NAS  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev).train()
    crit = SegmentationLosses('dice_ce')
    x = torch.randn(4, 1, 64, 64, device=dev)
    y = torch.randint(0, 2, (4, 64, 64), device=dev)
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        crit(net(x), y).backward()
    torch.cuda.synchronize()
    counts = collections.Counter()
    real = torch.Tensor.clone

    from torch.profiler import ProfilerActivity, profile
    net.zero_grad(set_to_none=True)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
        crit(net(x), y).backward()
        torch.cuda.synchronize()
    for ev in prof.key_averages(group_by_input_shape=True):
        if 'copy' in ev.key.lower() or 'clone' in ev.key.lower() or 'Memcpy' in ev.key or 'contiguous' in ev.key:
            print('%-50s count %6d  shapes %s' % (ev.key[:50], ev.count, str(ev.input_shapes)[:90]))


if __name__ == '__main__':
    main()
