#!/usr/bin/env python3
"""Forward / backward kernel time (HIP-graph timed) of the dense convolutions on the derived net's layer shapes, per math mode."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r + '/tools')
from senas_amd import functional as F
import bf_conv_check as B
for (n, ci, co, hw, k, dil) in ((8, 32, 32, 256, 5, 3), (8, 32, 32, 128, 5, 2), (8, 32, 32, 128, 5, 3), (8, 64, 32, 128, 3, 1), (8, 32, 32, 64, 5, 2), (8, 32, 32, 64, 5, 3), (8, 96, 32, 64, 3, 1), (8, 32, 32, 32, 5, 2)):
    x = torch.randn(n, ci, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(co, ci, k, k, device='cuda') * 0.05).requires_grad_(True)
    dy = torch.randn(n, co, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    row = []
    for mode in ('f32', 'bf16x6', 'bf16x3', 'bf16'):
        F.set_math(mode)
        def fwd():
            with torch.no_grad():
                F.conv2d(x, wt, 1, dil * (k // 2), dil, want_stats=True)
        def both():
            y, _ = F.conv2d(x, wt, 1, dil * (k // 2), dil, want_stats=True)
            torch.autograd.grad(y, (x, wt), dy)
        tf = B._graph_time(fwd)
        row.append('%%s %%.0f/%%.0f' %% (mode, tf, B._graph_time(both) - tf))
    print('thr %%s  n%%d %%d->%%d %%d^2 k%%d d%%d fwd/bwd us: %%s' %% (sys.argv[1], n, ci, co, hw, k, dil, '  '.join(row)), flush=True)
''' % (ROOT, ROOT)

subprocess.run([sys.executable, '-c', CHILD, '-'], check=False)
