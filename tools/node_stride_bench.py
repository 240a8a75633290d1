#!/usr/bin/env python3
"""What do the channel-slice reads of stacked convolution outputs cost the cell node?  One search-cell node (24 terms of 8
channels, mixing weights, ReLU, training mode) forward + backward on 4 x 8 x H x W terms that are (a) channel slices of
stacked [4, 32, H, W] tensors (pixel stride 32: what the state-major search cell feeds its nodes), (b) dense tensors;
`zstrided` / `dzstrided`: only the terms / only their gradients in the stacked layout.

Run under the profiler, one layout per process (eager launches; the kernel table gives the times):

    rocprofv3 --kernel-trace --output-format csv -d out/strided -- python3 tools/node_stride_bench.py strided 128 256
    rocprofv3 --kernel-trace --output-format csv -d out/dense -- python3 tools/node_stride_bench.py dense 128 256
    python3 tools/trace_by_grid.py out/strided ; python3 tools/trace_by_grid.py out/dense
"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from senas_amd import functional as F  # noqa: E402


def main():
    dev = torch.device('cuda')
    layout = sys.argv[1]
    sizes = [int(a) for a in sys.argv[2:]] or [128, 256]
    nterms, c, n = 24, 8, 4
    bns = [nn.BatchNorm2d(c).to(dev).train() for _ in range(nterms)]
    for h in sizes:
        stacked = [torch.randn(n, 32, h, h, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nterms // 4)]
        zs, sts, slots = [], [], []
        for zst in stacked:
            st = F.chan_stats(zst)
            landing = F.GradLanding(4, (n, c, h, h), 4, persistent=True)      # the stacked gradient buffer the nodes write into
            for e in range(4):
                if layout in ('strided', 'zstrided'):          # zstrided: slices forward, dense gradients
                    zs.append(zst[:, e * c:(e + 1) * c])
                    sts.append(st[:, e * c:(e + 1) * c])
                    slots.append((landing, e) if layout == 'strided' else None)
                elif layout == 'dzstrided':                    # dense terms, gradients written into the stacked buffer
                    z = zst[:, e * c:(e + 1) * c].contiguous(memory_format=torch.channels_last)
                    zs.append(z)
                    sts.append(F.chan_stats(z))
                    slots.append((landing, e))
                else:
                    z = zst[:, e * c:(e + 1) * c].contiguous(memory_format=torch.channels_last)
                    zs.append(z)
                    sts.append(F.chan_stats(z))
                    slots.append(None)
        zs = [z.requires_grad_(True) for z in zs]
        mix = torch.softmax(torch.randn(nterms, device=dev), 0).requires_grad_(True)
        dy = torch.randn(n, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        for _ in range(10):
            y = F.bn_combine([F.Term(z, bn, stats=st, grad_slot=sl) for z, bn, st, sl in zip(zs, bns, sts, slots)], mix=mix, relu=True)
            g = torch.autograd.grad(y, zs + [mix], dy)
            del y, g
        torch.cuda.synchronize()
        print(layout, h, 'done', flush=True)


if __name__ == '__main__':
    main()
