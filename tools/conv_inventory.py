#!/usr/bin/env python3
"""Per-geometry time of every convolution launch in one eager train step of the derived network
(HIP events around each launch): which shapes the step's convolution time is spent on, and at what
rate -- the table kernel tuning is steered by.

    python tools/conv_inventory.py [--c 32 --depth 5 --batch 8 --size 256]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd import functional as F  # noqa: E402
from senas_amd.geno_searched import senas_node_4  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.senas_model import SenasModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--c', type=int, default=32)
    ap.add_argument('--depth', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--search', action='store_true', help='the NAS supernet (batch 4 by default) instead of the derived net')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    if args.search:
        from senas_amd.senas_search import NAS
        net = NAS(1, args.c, 2, args.depth, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev)
    else:
        net = SenasModel(2, 1, c=args.c, depth=args.depth, genotype=senas_node_4).to(dev)
    crit = SegmentationLosses('dice_ce')
    x = torch.randn(args.batch, 1, args.size, args.size, device=dev)
    y = torch.randint(0, 2, (args.batch, args.size, args.size), device=dev)
    for it in range(3):
        if it == 2:
            F.TIMER = F.KernelTimer()
        net.zero_grad(set_to_none=True)
        crit(net(x), y).backward()
    timer, F.TIMER = F.TIMER, None
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, flops, nbytes, e0, e1, tag in timer.records:
        a = agg.setdefault((name, tag), [0, 0.0, flops, nbytes])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    print('total conv ms/step %.3f over %d launches' % (total, sum(v[0] for _, v in rows)))
    print('%-7s %-34s %-46s %4s %8s %8s %7s %7s' % ('kind', 'kernel', 'n,hi,wi,ci,ho,wo,co,kh,kw,s,p,d,T,g', 'cnt', 'ms_tot', 'us_each', 'TF/s', 'GB/s'))
    for (name, tag), (cnt, ms, flops, nbytes) in rows:
        each = ms / cnt
        print('%-7s %-34s %-46s %4d %8.3f %8.1f %7.1f %7.0f' % (tag[0][5:], name, ','.join(str(v) for v in tag[1:]), cnt, ms, each * 1e3,
                                                           flops / each / 1e9, nbytes / each / 1e6))


if __name__ == '__main__':
    main()
