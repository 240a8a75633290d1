#!/usr/bin/env python3
"""Phase stamps of the LAST node_prepare_bwd launch of one eager search step -- run with the debug library in the
shipped library's place (both built here: `make -C senas_amd/csrc phases`):

    cp senas_amd/libsenas_hip_phases.so senas_amd/libsenas_hip.so && python tools/phase_probe_node.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from senas_amd import _lib  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    bench.bench_search(dev, 1, 0, 1, use_graph=False)
    torch.cuda.synchronize()
    L = _lib.lib()
    buf = (C.c_ulonglong * 64)()
    assert L.senas_debug_read_phases_node(buf) == 0
    v = [int(x) for x in buf]
    names = ['start', 'operands requested, dmix partial', 'wave sum + barrier', 'image sums', 'LDS + barrier', 'coefficients stored', 'last barrier']
    for i, nm in enumerate(names):
        print('%-36s +%7.2f us' % (nm, (v[i] - v[0]) / 100.0))


if __name__ == '__main__':
    main()
