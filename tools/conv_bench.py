#!/usr/bin/env python3
"""Micro-benchmark of single convolution launches through the C ABI (forward, dgrad, wgrad) on the
shapes of the derived network -- for kernel tuning; bench.py stays the contract benchmark.

    python tools/conv_bench.py                 # table of shapes
    python tools/conv_bench.py --only fwd --shape 8,32,32,256,256,5,1,3 --iters 50
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd import functional as F  # noqa: E402

SHAPES = [  # n, ci, co, h, w, k, stride, dil
    (8, 32, 32, 256, 256, 5, 1, 3),
    (8, 32, 32, 256, 256, 5, 1, 2),
    (8, 32, 32, 128, 128, 5, 1, 3),
    (8, 32, 32, 128, 128, 5, 1, 2),
    (8, 32, 32, 64, 64, 5, 1, 3),
    (8, 32, 32, 32, 32, 5, 1, 3),
    (8, 128, 32, 256, 256, 3, 1, 1),
    (8, 128, 32, 128, 128, 3, 1, 1),
    (8, 32, 32, 128, 128, 3, 1, 1),
    (8, 32, 32, 128, 128, 1, 1, 1),
    (8, 1, 32, 256, 256, 7, 1, 1),
    (8, 32, 2, 256, 256, 3, 1, 1),
]


def time_it(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--shape', default='')
    ap.add_argument('--zeros', action='store_true', help='all-zero operands: the clock the chip holds without data-dependent switching')
    args = ap.parse_args()
    shapes = [tuple(int(v) for v in args.shape.split(','))] if args.shape else SHAPES
    dev = torch.device('cuda:0')
    print('%-34s %10s %10s %10s   (ms | TFLOP/s)' % ('shape n,ci,co,h,w,k,s,d', 'fwd', 'dgrad', 'wgrad'))
    for n, ci, co, h, w, k, s, d in shapes:
        g = torch.Generator(device='cuda').manual_seed(0)
        x = torch.randn(n, ci, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wt = (torch.randn(co, ci, k, k, device=dev, generator=g) * 0.05).requires_grad_(True)
        if args.zeros:
            x = torch.zeros_like(x).requires_grad_(True)
            wt = torch.zeros_like(wt).requires_grad_(True)
        pad = (k // 2) * d
        y, _ = F.conv2d(x, wt, stride=s, pad=pad, dil=d, want_stats=True)
        gy = torch.randn(y.shape, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        flops = 2.0 * n * y.shape[2] * y.shape[3] * co * ci * k * k
        res = {}
        if args.only in ('', 'fwd'):
            res['fwd'] = time_it(lambda: F.conv2d(x.detach(), wt.detach(), stride=s, pad=pad, dil=d, want_stats=True), args.iters)
        if args.only in ('', 'dgrad'):
            xx = x.detach().requires_grad_(True)
            yy, _ = F.conv2d(xx, wt.detach(), stride=s, pad=pad, dil=d)
            res['dgrad'] = time_it(lambda: torch.autograd.grad(yy, xx, gy, retain_graph=True), args.iters)
        if args.only in ('', 'wgrad'):
            ww = wt.detach().requires_grad_(True)
            yy, _ = F.conv2d(x.detach(), ww, stride=s, pad=pad, dil=d)
            res['wgrad'] = time_it(lambda: torch.autograd.grad(yy, ww, gy, retain_graph=True), args.iters)
        cells = ['%6.3f|%5.1f' % (res[kk], flops / (res[kk] * 1e-3) / 1e12) if kk in res else '     -     ' for kk in ('fwd', 'dgrad', 'wgrad')]
        print('%-34s %s' % ('%d,%d,%d,%d,%d,%d,%d,%d' % (n, ci, co, h, w, k, s, d), '  '.join(cells)))


if __name__ == '__main__':
    main()
