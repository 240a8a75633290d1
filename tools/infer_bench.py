#!/usr/bin/env python3
"""Inference throughput of the derived network: eval forward + Dice/CE + metric + arg-max per batch,
HIP-graph replayed.

    python tools/infer_bench.py [--batch 8 --size 256 --steps 30] [--folded]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd.geno_searched import senas_node_4  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.senas_model import SenasModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--folded', action='store_true', help='the compiled plan with batch-norm folded into the convolutions')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4).to(dev)
    x = torch.randn(args.batch, 1, args.size, args.size, device=dev)
    y = torch.randint(0, 2, (args.batch, args.size, args.size), device=dev)
    if args.folded:
        from senas_amd.infer import FoldedEvaluator as Ev
    else:
        from senas_amd.infer import Evaluator as Ev
    ev = Ev(net, 2, x, y, SegmentationLosses('dice_ce'), use_graph=not args.no_graph)
    for _ in range(3):
        ev(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print({'workload': 'derived net eval pass %dx1x%dx%d' % (args.batch, args.size, args.size), 'ms_per_batch': round(dt * 1e3, 3),
           'images_per_sec': round(args.batch / dt, 1), 'folded': args.folded, 'hip_graph': ev.graph is not None, 'result': ev.result()})


if __name__ == '__main__':
    main()
