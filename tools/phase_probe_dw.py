#!/usr/bin/env python3
"""Phase timestamps inside the depthwise weight-gradient kernel (debug library, see tools/phase_probe.py).

    python tools/phase_probe_dw.py 4,32,128,128,5        # n,c,h,w,k
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ConvGeom(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('n', 'hi', 'wi', 'ci', 'ho', 'wo', 'co', 'kh', 'kw', 'stride', 'pad', 'dil', 'transposed', 'groups')]


def main():
    n, c, h, w, k = (int(v) for v in sys.argv[1].split(','))
    lib = C.CDLL(os.path.join(ROOT, 'senas_amd', 'libsenas_hip_phases.so'))
    lib.senas_conv2d_bwd_weight.argtypes = [C.POINTER(ConvGeom), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.senas_conv2d_bwd_weight_ws.argtypes = [C.POINTER(ConvGeom), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.senas_last_error.restype = C.c_char_p
    g = ConvGeom(n, h, w, c, h, w, c, k, k, 1, k // 2, 1, 0, c)
    dev = torch.device('cuda:0')
    x = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    dw = torch.empty(c, 1, k, k, device=dev)
    nbytes, zero = C.c_int64(), C.c_int32()
    assert lib.senas_conv2d_bwd_weight_ws(C.byref(g), C.byref(nbytes), C.byref(zero)) == 0
    ws = torch.zeros(nbytes.value, device=dev, dtype=torch.uint8)
    buf = (C.c_ulonglong * 64)()
    for _ in range(3):
        rc = lib.senas_conv2d_bwd_weight(C.byref(g), x.data_ptr(), 0, dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 1, None)
        assert rc == 0, lib.senas_last_error()
        torch.cuda.synchronize()
    assert lib.senas_debug_read_phases_conv(buf) == 0
    t = [int(v) for v in buf]
    names = ['start', 'pixel loop done', 'shuffles done', 'LDS write + barrier', 'partials written']
    for i, nm in enumerate(names):
        print('%-22s +%7.2f us' % (nm, (t[i] - t[0]) / 100.0))


if __name__ == '__main__':
    main()
