#!/usr/bin/env python3
"""Where conv_bf's time goes: forward launches per math mode under SENAS_BF_PROBE masks (1 no staging, 2 no taps, 4 no
epilogue, 8 no LDS fragment reads inside the tap loop, 16 no weight-fragment loads; masks as arguments, default: a sweep) -- one process per mask (the library reads the variable once).  Needs the tuning build of the library
(`make -C senas_amd/csrc probe` -> senas_amd/libsenas_hip_probe.so); the shipped library has no such switch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r + '/tools')
from senas_amd import _lib
_lib.LIB_PATH = %r + '/senas_amd/libsenas_hip_probe.so'
from senas_amd import functional as F
import bf_conv_check as B
for (n, ci, co, hw, k, dil) in ((8, 32, 32, 256, 5, 3), (8, 128, 32, 256, 3, 1), (8, 32, 32, 128, 5, 2)):
    x = torch.randn(n, ci, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    wt = torch.randn(co, ci, k, k, device='cuda') * 0.05
    row = []
    for mode in ('bf16x6', 'bf16x3', 'bf16'):
        F.set_math(mode)
        def fwd():
            with torch.no_grad():
                F.conv2d(x, wt, 1, dil * (k // 2), dil, want_stats=True)
        row.append('%%s %%.0f' %% (mode, B._graph_time(fwd)))
    print('probe %%s  n%%d %%d->%%d %%d^2 k%%d d%%d: %%s us' %% (sys.argv[1], n, ci, co, hw, k, dil, '  '.join(row)), flush=True)
''' % (ROOT, ROOT, ROOT)

for mask in [int(m) for m in sys.argv[1:]] or (0, 1, 2, 4, 3, 5, 6, 7, 13, 21, 29):
    env = dict(os.environ, SENAS_BF_PROBE=str(mask))
    subprocess.run([sys.executable, '-c', CHILD, str(mask)], env=env, check=False)
