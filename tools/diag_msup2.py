#!/usr/bin/env python3
"""Bisect the deep-supervision gradient: per-output losses alone (GPU vs oracle fp64), their sum, and the multi loss."""
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


def oracle(z, tag, kw, which):
    sd = gio.add_missing_counters(gio.torch_sd(gio.unpack(z, tag + '/sd0/')))
    sd = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in sd.items()}
    gio.share_stem(sd, '')
    x = torch.from_numpy(z[tag + '/x']).double()
    tgt = torch.from_numpy(z[tag + '/target'])
    outs = R.derived_forward(sd, x, gio.geno_from_json(z[tag + '/genotype'], R.Genotype), depth=kw['depth'], supervision=True)
    loss = sum(R.dice_ce_loss(outs[i], tgt) for i in which)
    loss.backward()
    g = {k: v.grad.detach().numpy() for k, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
    return gio.alias_shared_stem(g, '')


def gpu(z, tag, which, retain_all=False):
    from senas_amd.loss import dice_ce_loss
    net, kw = T._build_net(z, tag)
    x = torch.from_numpy(z[tag + '/x']).cuda()
    tgt = torch.from_numpy(z[tag + '/target']).cuda()
    outs = net(x)
    loss = None
    for i in which:
        l = dice_ce_loss(outs[i], tgt)
        loss = l if loss is None else loss + l
    loss.backward()
    return T.grads_of(net), kw


def report(name, got, exp):
    top = max(float(np.abs(v).max()) for v in exp.values())
    rows = sorted(((float(np.abs(got[k] - e).max()) / max(float(np.abs(e).max()), 1e-3 * top), k) for k, e in exp.items() if k in got), reverse=True)
    print('%-16s worst: %s' % (name, ', '.join('%s %.1e' % (k, v) for v, k in rows[:4])))


def main():
    tag = 'derived.node2.c8.msup'
    z = gio.load('nets3')
    for which in ([0], [1], [2], [0, 1], [1, 2], [0, 2], [0, 1, 2]):
        got, kw = gpu(z, tag, which)
        report(str(which), got, oracle(z, tag, kw, which))


if __name__ == '__main__':
    main()
