#!/usr/bin/env python3
"""Longer run of tests/test_gpu_parity.py::test_node_sweep_vs_torch's generator: N random fused-node configurations
(terms, channels, SE, mix, residual, ReLU, train / eval) against the float64 torch formulation.

    python tools/node_fuzz.py [count] [seed]
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    bad = 0
    for i, cfg in enumerate(T._random_node_cases(count, seed)):
        try:
            T.test_node_sweep_vs_torch(cfg)
        except Exception:
            bad += 1
            print('FAIL', cfg)
            traceback.print_exc(limit=1)
        if i % 50 == 49:
            print('%d cases, %d failures' % (i + 1, bad), flush=True)
    print('done: %d cases, %d failures' % (count, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
