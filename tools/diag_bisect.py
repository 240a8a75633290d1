#!/usr/bin/env python3
"""Given the failing random genotype of a seed, replace one op at a time by a plain dilated conv and report the worst
gradient error -- finds the op (position) responsible."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import test_gpu_parity as T  # noqa: E402
from diag_ops import run  # noqa: E402
from senas_amd.genotype import Genotype  # noqa: E402


def main():
    seed = int(sys.argv[1])
    rng = np.random.RandomState(seed)
    nodes = int(rng.choice([3, 4]))
    down, up = T._random_genotype(rng, nodes)
    gamma = [int(v) for v in rng.randint(0, 2, 3)]
    if gamma[1] == 1 and gamma[2] == 0:
        gamma[2] = 1

    def make(d, u):
        return Genotype(down=d, down_concat=range(2, 2 + nodes), up=u, up_concat=range(2, 2 + nodes), gamma=gamma)
    print('original', run(make(down, up), seed=seed))
    for which, lst in (('down', down), ('up', up)):
        for i in range(len(lst)):
            alt = list(lst)
            alt[i] = ('dil_3_conv_5', lst[i][1])
            g = make(alt, up) if which == 'down' else make(down, alt)
            lerr, (gerr, k) = run(g, seed=seed)
            print('%s[%d] %-16s -> dil_3_conv_5 : worst grad %.2e (%s)' % (which, i, lst[i][0], gerr, k), flush=True)


if __name__ == '__main__':
    main()
