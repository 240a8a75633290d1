#!/usr/bin/env python3
"""Phase stamps of the LAST node_prepare_bwd (or dstail_bwd_reduce) launch of one eager search step -- run with the debug library in the
shipped library's place (both built here: `make -C senas_amd/csrc phases`):

    cp senas_amd/libsenas_hip_phases.so senas_amd/libsenas_hip.so && python tools/phase_probe_node.py [node|wide|dstail|dwfwd]
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
    which = sys.argv[1] if len(sys.argv) > 1 else 'node'
    names = {'node': ['start', 'operands requested, dmix partial', 'wave sum + barrier', 'image sums', 'LDS + barrier', 'coefficients stored',
                      'last barrier'],
             'wide': [None] * 8 + ['start', 'operands requested', 'prologue items done', 'barrier 1', 'SE hidden layer + barrier 2',
                                   'coefficients + bias + barriers', 'element written', 'statistics flushed'],
             'dwfwd': [None] * 16 + ['start', 'weights staged + barrier', 'taps done (last chunk)', 'outputs stored, statistics accumulated', 'statistics flushed'],
             'dstail': ['start', 'coefficients + weights in LDS', 'pixel loop done', 'S atomics issued', 'dW atomics issued', 'S rows folded',
                        'dW rows folded', 'barrier passed']}[which]
    buf = (C.c_ulonglong * 64)()
    assert getattr(L, 'senas_debug_read_phases_' + {'wide': 'node', 'dwfwd': 'conv'}.get(which, which))(buf) == 0
    v = [int(x) for x in buf]
    base = min(v[i] for i, nm in enumerate(names) if nm is not None)
    for i, nm in sorted(((i, nm) for i, nm in enumerate(names) if nm is not None), key=lambda e: v[e[0]]):
        print('%-36s +%7.2f us' % (nm, (v[i] - base) / 100.0))


if __name__ == '__main__':
    main()
