#!/usr/bin/env python3
"""Folded inference vs module eval forward, cell by cell, on the random genotype of a seed (tests' generator)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T  # noqa: E402
from senas_amd.genotype import Genotype  # noqa: E402
from senas_amd.infer import FoldedForward  # noqa: E402
from senas_amd.senas_model import BuildCell, SenasModel  # noqa: E402


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
    net = SenasModel(2, 1, c=16, depth=4, genotype=geno)
    T._randomize(net, seed)
    gen = torch.Generator().manual_seed(seed)
    for k, v in net.state_dict().items():
        if k.endswith('running_mean'):
            v.copy_(0.2 * torch.randn(v.shape, generator=gen))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=gen))
    x = torch.randn(2, 1, 64, 64, generator=gen)
    dev = torch.device('cuda:0')
    net = net.to(dev).eval()
    ff = FoldedForward(net, 2)
    import golden_io as gio
    from oracle import senas_ref as R
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    gio.share_stem(sd)
    with torch.no_grad():
        ref = R.derived_forward(sd, x, R.Genotype(*geno), depth=4, training=False)[-1]
        plain = net(x.to(dev))[-1].cpu()
        got = ff(x.to(dev))[-1].cpu()
    sc = float(ref.abs().max())
    print('module vs oracle %.2e   folded vs oracle %.2e   folded vs module %.2e' % (float((plain - ref).abs().max()) / sc,
          float((got - ref).abs().max()) / sc, float((got - plain).abs().max()) / sc))
    rec = {}
    for name, m in net.named_modules():
        if isinstance(m, BuildCell):
            m.register_forward_hook(lambda mod, inp, out, name=name: rec.__setitem__(name, (inp, out)))
    with torch.no_grad():
        net(x.to(dev))
        for name, (inp, out) in rec.items():
            cell = dict(net.named_modules())[name]
            got = ff.cell(cell, *inp)
            print('%-24s in0 %-18s out %-18s err %.2e' % (name, tuple(inp[0].shape), tuple(out.shape),
                                                        float((got - out).abs().max() / out.abs().max())))
            if float((got - out).abs().max() / out.abs().max()) > 1e-3:
                # node by node
                states = [ff.block(cell.preprocess0, inp[0]), torch.relu(inp[1])]
                ref_states = [cell.preprocess0(inp[0]), torch.relu(inp[1])]
                print('   preprocess0 err %.2e' % float((states[0] - ref_states[0]).abs().max() / ref_states[0].abs().max()))
                from senas_amd import functional as F
                for i in range(cell._num_meta_node):
                    es = (2 * i, 2 * i + 1)
                    ref = F.bn_combine([cell._ops[e].raw(ref_states[cell._indices[e]]) for e in es], relu=True)
                    got_n = ff._finish([ff.term(cell._ops[e], ref_states[cell._indices[e]]) for e in es], relu=True)
                    print('   node %d %s err %.2e' % (i, [type(cell._ops[e]).__name__ + str(cell._indices[e]) for e in es],
                                                    float((got_n - ref).abs().max() / ref.abs().max())))
                    ref_states.append(ref)
                break


if __name__ == '__main__':
    main()
