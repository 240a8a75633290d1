#!/usr/bin/env python3
"""Phase timestamps inside one convolution launch (block 0, thread 0; 100 MHz wall clock) from the debug
library built by `make -C senas_amd/csrc phases`.  Tuning aid only; the package never loads that library.

    python tools/phase_probe.py 8,32,32,64,64,5,1,3          # n,ci,co,h,w,k,stride,dil
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ConvGeom(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('n', 'hi', 'wi', 'ci', 'ho', 'wo', 'co', 'kh', 'kw', 'stride', 'pad', 'dil', 'transposed', 'groups')]


def main():
    n, ci, co, h, w, k, s, d = (int(v) for v in sys.argv[1].split(','))
    lib = C.CDLL(os.path.join(ROOT, 'senas_amd', 'libsenas_hip_phases.so'))
    lib.senas_conv2d_ws_bytes.restype = C.c_int64
    lib.senas_conv2d_fwd.argtypes = [C.POINTER(ConvGeom)] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
    lib.senas_last_error.restype = C.c_char_p
    pad = (k // 2) * d
    g = ConvGeom(n, h, w, ci, h, w, co, k, k, s, pad, d, 0, 1)
    dev = torch.device('cuda:0')
    x = torch.randn(n, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(co, ci, k, k, device=dev) * 0.05
    y = torch.empty(n, co, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    stats = torch.zeros(n, co, 2, device=dev, dtype=torch.float64)
    ws = torch.empty(int(lib.senas_conv2d_ws_bytes(C.byref(g))), device=dev, dtype=torch.uint8)
    buf = (C.c_ulonglong * 64)()
    for it in range(3):
        rc = lib.senas_conv2d_fwd(C.byref(g), x.data_ptr(), wt.data_ptr(), y.data_ptr(), 1, stats.data_ptr(), ws.data_ptr(), None, None)
        assert rc == 0, lib.senas_last_error()
        torch.cuda.synchronize()
    assert lib.senas_debug_read_phases_conv_lds(buf) == 0
    t = [int(v) for v in buf]
    t0 = t[0]
    names = {0: 'start', 40: 'passes done', 41: 'fold done', 42: 'stores done', 43: 'stats done'}
    for p in range(8):
        names.update({1 + 4 * p: 'pass %d: entered' % p, 2 + 4 * p: 'pass %d: window in LDS' % p, 3 + 4 * p: 'pass %d: barrier' % p,
                      4 + 4 * p: 'pass %d: taps done' % p})
    prev = t0
    for k_ in sorted(names):
        if t[k_] >= t0 and t[k_] != 0 and (k_ < 40 and (k_ - 1) // 4 < ci // 16 or k_ >= 40 or k_ == 0):
            print('%-26s +%7.2f us  (%6.2f)' % (names[k_], (t[k_] - t0) / 100.0, (t[k_] - prev) / 100.0))
            prev = t[k_]


if __name__ == '__main__':
    main()
