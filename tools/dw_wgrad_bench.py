#!/usr/bin/env python3
"""The batched depthwise weight gradient (senas_dwconv_pair_bwd_weight: ka 3x3 + kb 5x5 problems on one input) on its own:
launch time by HIP events, algorithmic rate, and the result against an fp64 sum of shifted products.

    python tools/dw_wgrad_bench.py 4,32,256,256 3 3          # n,c,h,w  ka kb
    SENAS_DW_WGRAD_X4=0 python tools/dw_wgrad_bench.py ...   # one pixel per trip (the form before round 5)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd import _lib  # noqa: E402


def main():
    n, c, h, w = (int(v) for v in sys.argv[1].split(','))
    ka, kb = int(sys.argv[2]), int(sys.argv[3])
    L = _lib.lib()
    dev = torch.device('cuda:0')
    CL = torch.channels_last
    x = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL)
    dys = [torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL) for _ in range(ka + kb)]
    ks = [3] * ka + [5] * kb
    dws = [torch.empty(c, 1, k, k, device=dev) for k in ks]
    ga = _lib.ConvGeom(n, h, w, c, h, w, c, 3, 3, 1, 1, 1, 0, c)
    gb = _lib.ConvGeom(n, h, w, c, h, w, c, 5, 5, 1, 2, 1, 0, c)
    if ka == 0:
        ga, ka, kb, gbp = gb, kb, 0, None
    else:
        gbp = C.byref(gb) if kb else None
    nbytes = int(L.senas_dwconv_pair_ws_bytes(C.byref(ga), ka, gbp, kb))
    assert nbytes > 0, 'off the batched path'
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    k = len(ks)
    dyp = (C.c_void_p * k)(*[t.data_ptr() for t in dys])
    dwp = (C.c_void_p * k)(*[t.data_ptr() for t in dws])
    st = torch.cuda.current_stream().cuda_stream

    def once():
        _lib.check(L.senas_dwconv_pair_bwd_weight(C.byref(ga), ka, gbp, kb, x.data_ptr(), dyp, dwp, ws.data_ptr(), None, st),
                   'senas_dwconv_pair_bwd_weight')

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    algo = (k + 1) * n * c * h * w * 4
    print('n=%d c=%d %dx%d, %d x 3x3 + %d x 5x5: %.1f us per call (partials + sums), %.1f MB algorithmic -> %.2f TB/s'
          % (n, c, h, w, ks.count(3), ks.count(5), us, algo / 1e6, algo / us / 1e6))
    worst = 0.0
    xd = x.double()
    for t, kk in enumerate(ks):
        p = kk // 2
        xp = torch.nn.functional.pad(xd, (p, p, p, p))
        dy = dys[t].double()
        ref = torch.stack([torch.stack([(xp[:, :, ky:ky + h, kx:kx + w] * dy).sum((0, 2, 3)) for kx in range(kk)], -1) for ky in range(kk)], -2)
        err = ((dws[t][:, 0].double() - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, err)
    print('  worst error against the fp64 sums, relative to the largest element: %.2e' % worst)
    assert worst < 1e-4


if __name__ == '__main__':
    main()
