#!/usr/bin/env python3
"""conv_bf.hip against torch-CPU fp64 on a few shapes: max error / tensor scale per math mode, fwd and dgrad, + timing of
the BASELINE layer shapes.   python tools/bf_conv_check.py [--time]"""
import os
import sys
import time

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from senas_amd import functional as F  # noqa: E402


def one(n, ci, co, h, w, k, dil, relu, modes):
    g = torch.Generator().manual_seed(n * 1000 + ci + h + k + dil)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    dy = torch.randn(n, co, h, w, generator=g)
    x64 = x.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    y64 = TF.conv2d(torch.relu(x64) if relu else x64, w64, padding=dil * (k // 2), dilation=dil)
    y64.backward(dy.double())
    out = []
    for mode in modes:
        F.set_math(mode)
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wg = wt.cuda().requires_grad_(True)
        y, st = F.conv2d(xg, wg, 1, dil * (k // 2), dil, in_relu=relu, want_stats=True)
        y.backward(dy.cuda().contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        ey = float((y.detach().cpu().double() - y64.detach()).abs().max() / y64.detach().abs().max())
        ex = float((xg.grad.cpu().double() - x64.grad).abs().max() / x64.grad.abs().max())
        ew = float((wg.grad.cpu().double() - w64.grad).abs().max() / w64.grad.abs().max())
        s_ref = y64.detach().sum((2, 3))
        es = float((st[:, :, 0].cpu() - s_ref).abs().max() / s_ref.abs().max())
        out.append('%s y %.1e dx %.1e dw %.1e st %.1e' % (mode, ey, ex, ew, es))
    F.set_math('f32')
    print('n%d %d->%d %dx%d k%d d%d%s: %s' % (n, ci, co, h, w, k, dil, ' relu' if relu else '', ' | '.join(out)), flush=True)


def _graph_time(fn, reps=10, replays=5):
    """Kernel time per call of ``fn`` (us): ``reps`` calls captured in a HIP graph, replayed -- no host launch cost in it."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    from senas_amd.arena import reset_arena
    reset_arena()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    reset_arena()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * replays)


def timing(modes):
    for (n, ci, co, hw, k, dil) in ((8, 32, 32, 256, 5, 3), (8, 32, 32, 256, 5, 2), (8, 32, 32, 128, 5, 2), (8, 128, 32, 256, 3, 1),
                                    (8, 64, 32, 128, 3, 1), (8, 32, 32, 64, 5, 2), (4, 32, 32, 256, 5, 3)):
        x = torch.randn(n, ci, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wt = (torch.randn(co, ci, k, k, device='cuda') * 0.05).requires_grad_(True)
        dy = torch.randn(n, co, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
        flop = 2.0 * n * hw * hw * ci * co * k * k
        row = []
        for mode in modes:
            F.set_math(mode)

            def fwd():
                with torch.no_grad():
                    F.conv2d(x, wt, 1, dil * (k // 2), dil, want_stats=True)

            def both():
                y, _ = F.conv2d(x, wt, 1, dil * (k // 2), dil, want_stats=True)
                torch.autograd.grad(y, (x, wt), dy)

            tf = _graph_time(fwd)
            tb = _graph_time(both) - tf
            row.append('%s fwd %.0f us (%.0f TF) bwd %.0f us (%.0f TF)' % (mode, tf, flop / tf / 1e6, tb, 2 * flop / tb / 1e6))
        F.set_math('f32')
        print('time n%d %d->%d %d^2 k%d d%d: %s' % (n, ci, co, hw, k, dil, ' | '.join(row)), flush=True)


if __name__ == '__main__':
    modes = ['f32', 'bf16x6', 'bf16x3', 'bf16']
    for case in ((2, 32, 32, 64, 64, 5, 3, False), (2, 32, 32, 40, 72, 5, 2, True), (1, 64, 32, 33, 47, 3, 1, False),
                 (2, 128, 64, 32, 32, 3, 1, True), (3, 32, 32, 16, 32, 5, 2, False), (1, 32, 32, 128, 128, 5, 3, False)):
        one(*case, modes)
    if '--time' in sys.argv:
        timing(modes)
