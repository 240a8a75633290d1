#!/usr/bin/env python3
"""Does a matrix-pipe-bound kernel overlap with an HBM-bound one on this chip?  The dense 5x5 convolution of the supernet's head
cell (4 x 32 x 256 x 256, conv_lds) on one stream, streaming kernels on the same map (ReLU: one read + one write per element) on
another: each alone, both at once.  Full overlap: max(a, b); none: a + b.

    python tools/mix_overlap.py
"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd import functional as F  # noqa: E402
from senas_amd.operations import run_conv  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cl = torch.channels_last
    x = torch.randn(4, 32, 256, 256, device=dev).contiguous(memory_format=cl)
    z = torch.randn(4, 32, 256, 256, device=dev).contiguous(memory_format=cl)
    conv = nn.Conv2d(32, 32, 5, padding=2, bias=False).to(dev)
    sa, sb = F.own_stream(dev, 'mixA'), F.own_stream(dev, 'mixB')
    main_s = torch.cuda.current_stream()

    def convs(k):
        for _ in range(k):
            run_conv(conv, x, in_relu=False, want_stats=False)

    def streams(k):
        for _ in range(k):
            F.relu(z)

    def timed(fa, fb, reps=20):
        best = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(main_s)
            sa.wait_stream(main_s)
            sb.wait_stream(main_s)
            if fa is not None:
                with torch.cuda.stream(sa):
                    fa()
            if fb is not None:
                with torch.cuda.stream(sb):
                    fb()
            main_s.wait_stream(sa)
            main_s.wait_stream(sb)
            e1.record(main_s)
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3)
        best.sort()
        return best[len(best) // 2]

    with torch.no_grad():
        for kc, kr in ((4, 16), (8, 32), (8, 16), (4, 32)):
            a = timed(lambda: convs(kc), None)
            b = timed(None, lambda: streams(kr))
            c = timed(lambda: convs(kc), lambda: streams(kr))
            print('%d convolutions alone %7.1f us | %2d streaming kernels alone %7.1f us | both at once %7.1f us  (sum %7.1f, max %7.1f: %.0f %% of the shorter one hidden)'
                  % (kc, a, kr, b, c, a + b, max(a, b), 100.0 * (a + b - c) / min(a, b)))


if __name__ == '__main__':
    main()
