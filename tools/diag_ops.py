#!/usr/bin/env python3
"""For each candidate op: a derived net (c=8, depth 4, 2x1x32x32) that uses this op wherever it is legal, max
per-parameter gradient error against the float64 oracle -- localises a faulty backward kernel."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import golden_io as gio  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import senas_ref as R  # noqa: E402
from senas_amd.genotype import Genotype  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.operations import DownOps, NormOps, UpOps  # noqa: E402
from senas_amd.senas_model import SenasModel  # noqa: E402


def run(geno, seed=3, c=8, size=32):
    net = SenasModel(2, 1, c=c, depth=4, genotype=geno)
    T._randomize(net, seed)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sd64 = {k: (v.double().requires_grad_(True) if (v.is_floating_point() and 'running' not in k) else v.double() if v.is_floating_point() else v.clone())
            for k, v in sd.items()}
    gio.share_stem(sd64)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1, size, size, generator=gen)
    y = torch.randint(0, 2, (2, size, size), generator=gen)
    ref = R.derived_forward(sd64, x.double(), R.Genotype(*geno), depth=4)[-1]
    R.dice_ce_loss(ref, y).backward()
    dev = torch.device('cuda:0')
    net = net.to(dev).train()
    out = net(x.to(dev))
    SegmentationLosses('dice_ce')(out, y.to(dev)).backward()
    worst = (0.0, '')
    for k, p in net.named_parameters():
        if k not in sd64 or sd64[k].grad is None:
            continue
        e = sd64[k].grad.numpy()
        nrm = np.sqrt((e ** 2).sum())
        if nrm < 1e-12:
            continue
        err = float(np.sqrt(((p.grad.detach().cpu().double().numpy() - e) ** 2).sum()) / nrm)
        if err > worst[0]:
            worst = (err, k)
    lerr = float((out[-1].detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
    return lerr, worst


def main():
    base = 'dil_3_conv_5'
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for op in sorted(set(DownOps + UpOps + NormOps)):
        def pick(ops):
            return op if op in ops else base
        # one op of every node is the probed op, the other stays a plain dilated conv (two 'none' ops on a node never occur)
        down = [(pick(DownOps), 0), (base, 1), (pick(NormOps), 2), (base, 0), (pick(NormOps), 3), (pick(DownOps), 1)]
        up = [(pick(NormOps), 0), (base, 1), (base, 2), (pick(UpOps), 1), (pick(NormOps), 3), (base, 0)]
        geno = Genotype(down=down, down_concat=range(2, 5), up=up, up_concat=range(2, 5), gamma=[1, 1, 1])
        lerr, (gerr, k) = run(geno, c=c)
        print('%-16s logits %.1e   worst grad %.2e  (%s)' % (op, lerr, gerr, k), flush=True)


if __name__ == '__main__':
    main()
