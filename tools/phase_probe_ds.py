#!/usr/bin/env python3
"""The DepSepConv tail's backward pair (dstail.hip: reduce + apply) on its own: launch time by HIP events and, with the debug
library (`make -C senas_amd/csrc phases`), the phase stamps of block 0 of the reduce launch.

    python tools/phase_probe_ds.py 12,4,16384,8,8 [dz2 pixel stride [nowg]]      # k,n,hw,cin,cout
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = C.c_void_p


class Item(C.Structure):
    _fields_ = [('z1', P), ('stats1', P), ('gamma1', P), ('beta1', P), ('running_mean1', P), ('running_var1', P),
                ('num_batches_tracked1', P), ('mean_invstd', P), ('w', P), ('z2', P), ('stats2', P), ('dz2', P),
                ('dz2_pixel_stride', C.c_int64), ('sums', P), ('dz1', P), ('dgamma1', P), ('dbeta1', P), ('dw', P), ('dw_acc', P)]


def main():
    k, n, hw, cin, cout = (int(v) for v in sys.argv[1].split(','))
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else cout
    nowg = len(sys.argv) > 3 and sys.argv[3] == 'nowg'          # the architecture pass: no weight gradient of the 1x1
    name = os.environ.get('SENAS_PROBE_LIB') or (
        'libsenas_hip_phases.so' if os.path.exists(os.path.join(ROOT, 'senas_amd', 'libsenas_hip_phases.so')) else 'libsenas_hip.so')
    lib = C.CDLL(os.path.join(ROOT, 'senas_amd', name))
    lib.senas_dstail_bwd.argtypes = [C.POINTER(Item), C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, P]
    lib.senas_last_error.restype = C.c_char_p
    dev = torch.device('cuda:0')
    keep = []

    def t(*shape, dtype=torch.float32, fill=None):
        v = torch.randn(*shape, device=dev, dtype=dtype) if fill is None else torch.full(shape, fill, device=dev, dtype=dtype)
        keep.append(v)
        return v

    items = (Item * k)()
    dz2_all = t(n, hw, stride if stride > cout else cout * 1) if stride > cout else None
    for i in range(k):
        it = items[i]
        it.z1 = t(n, hw, cin).data_ptr()
        it.gamma1, it.beta1 = t(cin).data_ptr(), t(cin).data_ptr()
        mi = t(2 * cin)
        mi[cin:] = mi[cin:].abs() + 0.5
        it.mean_invstd = mi.data_ptr()
        it.w = t(cout, cin).data_ptr()
        if stride > cout:
            it.dz2 = dz2_all.data_ptr() + 4 * cout * (i % (stride // cout))
        else:
            it.dz2 = t(n, hw, cout).data_ptr()
        it.dz2_pixel_stride = stride
        it.sums = t(n, cin, 2, dtype=torch.float64, fill=0.0).data_ptr()
        it.dz1 = t(n, hw, cin).data_ptr()
        it.dgamma1, it.dbeta1 = t(cin).data_ptr(), t(cin).data_ptr()
        if not nowg:
            it.dw = t(cout, cin).data_ptr()
            it.dw_acc = t(n, cout, cin, dtype=torch.float64, fill=0.0).data_ptr()       # (senas_dstail_ws_bytes)
        st1 = t(n, cin, 2, dtype=torch.float64, fill=0.0)
        st1[..., 1] = float(hw)
        it.stats1 = st1.data_ptr()
        it.z2 = t(n, hw, cout).data_ptr()
        it.stats2 = t(n, cout, 2, dtype=torch.float64, fill=0.0).data_ptr()
    st = torch.cuda.current_stream().cuda_stream

    def once():
        rc = lib.senas_dstail_bwd(items, k, n, hw, cin, cout, st)
        assert rc == 0, lib.senas_last_error()

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
    algo = k * n * hw * 4 * (2 * cin + 2 * cout + cin)           # z1 and dz2 twice, dz1 once
    print('k=%d n=%d hw=%d cin=%d cout=%d stride=%d: reduce + apply %.1f us per pair, %.1f MB algorithmic -> %.2f TB/s'
          % (k, n, hw, cin, cout, stride, us, algo / 1e6, algo / us / 1e6))
    lib.senas_dstail_fwd.argtypes = [C.POINTER(Item), C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, P]

    def fwd():
        rc = lib.senas_dstail_fwd(items, k, n, hw, cin, cout, 1, 0.1, 1e-5, st)
        assert rc == 0, lib.senas_last_error()

    for _ in range(3):
        fwd()
    e0.record()
    for _ in range(reps):
        fwd()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    algo = k * n * hw * 4 * (cin + cout)
    print('  forward %.1f us, %.1f MB algorithmic -> %.2f TB/s' % (us, algo / 1e6, algo / us / 1e6))
    if hasattr(lib, 'senas_debug_read_phases_dstail'):
        buf = (C.c_ulonglong * 64)()
        assert lib.senas_debug_read_phases_dstail(buf) == 0
        v = [int(x) for x in buf]
        for i, nm in enumerate(['start', 'coefficients + weights in LDS', 'pixel loop done', 'S1/S2 folded', 'dW folded']):
            print('  %-32s +%7.2f us' % (nm, (v[i] - v[0]) / 100.0))


if __name__ == '__main__':
    main()
