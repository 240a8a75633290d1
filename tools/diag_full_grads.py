#!/usr/bin/env python3
"""Per-tensor gradient errors of the c = 32 fixtures (tests/golden/nets_full.npz) on the GPU: the ten worst tensors with
their own magnitude, the net's largest gradient and the absolute error.   python tools/diag_full_grads.py [tag]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import golden_io as gio  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def main():
    from senas_amd.loss import SegmentationLosses
    z = gio.load('nets_full')
    for tag in gio.index('nets_full'):
        if len(sys.argv) > 1 and sys.argv[1] not in tag:
            continue
        net, kw = T._build_net(z, tag)
        x = torch.from_numpy(z[tag + '/x']).cuda()
        tgt = torch.from_numpy(z[tag + '/target']).cuda()
        loss = SegmentationLosses('dice_ce')(net(x), tgt)
        loss.backward()
        got = T.grads_of(net)
        exp = gio.unpack(z, tag + '/grad64/')
        top = float(z[tag + '/grad_top'])
        rows = []
        for k, e in exp.items():
            own = float(np.abs(e).max())
            err = float(np.abs(got[k] - e).max())
            rows.append((err / max(own, 1e-3 * top), k, own, err))
        rows.sort(reverse=True)
        print(tag, 'top gradient %.3e' % top)
        for r, k, own, err in rows[:14]:
            print('  %-60s rel %.2e  own max %.3e  abs err %.3e' % (k, r, own, err))


if __name__ == '__main__':
    main()
