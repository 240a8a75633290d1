#!/usr/bin/env python3
"""tests/golden/nets_spread.npz: for every whole-net fixture of nets.npz / nets2.npz, how far the ORACLE's own fp32 gradients of
the fully-stored tensors move under 1e-6 relative perturbations of the input and of every weight (3 trials, worst case) --
the conditioning of the fixture, which the GPU test's per-tensor bound has to know (tests/test_gpu_parity.py::test_whole_net).
The oracle is pinned to the reference bit for bit on these very fixtures (tests/test_oracle_golden.py), so this is the
reference's conditioning.  Run in the build container:  python tests/golden/make_spread.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import golden_io as gio  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def main():
    torch.set_num_threads(8)
    out = {}
    for fixture, tag in T.NET_CASES:
        z = gio.load(fixture)
        kw = json.loads(str(z[tag + '/kw']))
        names = list(gio.sub(z, tag + '/gradfull64/'))
        sp = T._oracle_gradient_spread(z, tag, kw, names, trials=3)
        key = fixture + '/' + tag
        out[key + '/names'] = np.array(json.dumps(list(sp)))
        out[key + '/spread'] = np.array([sp[k] for k in sp], dtype=np.float64)
        print('%-40s worst %.1e  median %.1e' % (key, max(sp.values()), float(np.median(list(sp.values())))), flush=True)
    np.savez_compressed(os.path.join(HERE, 'nets_spread.npz'), **out)


if __name__ == '__main__':
    main()
