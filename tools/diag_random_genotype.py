#!/usr/bin/env python3
"""Per-parameter gradient error of a random-genotype derived net (tests' generator) against the float64 oracle."""
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
from senas_amd.senas_model import SenasModel  # noqa: E402


def main():
    seed = int(sys.argv[1])
    rng = np.random.RandomState(seed)
    nodes = int(rng.choice([3, 4]))
    down, up = T._random_genotype(rng, nodes)
    gamma = [int(v) for v in rng.randint(0, 2, 3)]
    if gamma[1] == 1 and gamma[2] == 0:
        gamma[2] = 1
    geno = Genotype(down=down, down_concat=range(2, 2 + nodes), up=up, up_concat=range(2, 2 + nodes), gamma=gamma)
    print(geno)
    net = SenasModel(2, 1, c=8, depth=4, genotype=geno)
    T._randomize(net, seed)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sd64 = {k: (v.double().requires_grad_(True) if (v.is_floating_point() and 'running' not in k) else v.double() if v.is_floating_point() else v.clone())
            for k, v in sd.items()}
    gio.share_stem(sd64)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1, 32, 32, generator=gen)
    y = torch.randint(0, 2, (2, 32, 32), generator=gen)
    ref = R.derived_forward(sd64, x.double(), R.Genotype(*geno), depth=4)[-1]
    R.dice_ce_loss(ref, y).backward()
    dev = torch.device('cuda:0')
    net = net.to(dev).train()
    out = net(x.to(dev))
    SegmentationLosses('dice_ce')(out, y.to(dev)).backward()
    print('logits rel err %.2e' % float((out[-1].detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max()))
    rows = []
    for k, p in net.named_parameters():
        kk = k
        if kk not in sd64 or sd64[kk].grad is None:
            continue
        e = sd64[kk].grad.numpy()
        gt = p.grad.detach().cpu().double().numpy()
        nrm = np.sqrt((e ** 2).sum())
        rows.append((float(np.sqrt(((gt - e) ** 2).sum()) / max(nrm, 1e-30)), k, float(nrm)))
    rows.sort(reverse=True)
    for err, k, nrm in rows[:25]:
        print('%.2e  |g| %.2e  %s' % (err, nrm, k))


if __name__ == '__main__':
    main()
