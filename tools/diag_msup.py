#!/usr/bin/env python3
"""Deep-supervision fixtures (tests/golden/nets3.npz): per-tensor gradient error of the GPU path against the ORACLE run in
fp64 on the same weights, under the multi-output loss and under the last-output loss, next to the oracle's own fp32-vs-fp64
difference and its spread under a 1e-6 input perturbation -- is an error conditioning, or wiring?
    python tools/diag_msup.py [tag-substring]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import golden_io as gio  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import senas_ref as R  # noqa: E402  (checker)


def oracle_grads(z, tag, kw, multi, dtype, eps=0.0):
    sd = gio.add_missing_counters(gio.torch_sd(gio.unpack(z, tag + '/sd0/')))
    sd = {k: (v.detach().to(dtype).requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in sd.items()}
    gio.share_stem(sd, 'net.' if 'nas' in tag.split('.') else '')
    x = torch.from_numpy(z[tag + '/x']).to(dtype)
    if eps:
        x = x * (1 + eps * torch.randn(x.shape, generator=torch.Generator().manual_seed(1)).to(dtype))
    tgt = torch.from_numpy(z[tag + '/target'])
    if 'nas' in tag.split('.'):
        outs = R.nas_forward(sd, x, depth=kw['depth'], nodes=kw['meta_node_num'], supervision=True)
    else:
        outs = R.derived_forward(sd, x, gio.geno_from_json(z[tag + '/genotype'], R.Genotype), depth=kw['depth'], supervision=True)
    loss = R.multi_dice_ce_loss(outs, tgt, kw['depth']) if multi else R.dice_ce_loss(outs[-1], tgt)
    loss.backward()
    g = {k: v.grad.detach().double().numpy() for k, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}
    return gio.alias_shared_stem(g, 'net.' if tag.startswith('nas') else '')


def main():
    from senas_amd.loss import MultiSegmentationLosses, SegmentationLosses
    z = gio.load('nets3')
    for tag in gio.index('nets3'):
        if len(sys.argv) > 1 and sys.argv[1] not in tag:
            continue
        for multi in (True, False):
            net, kw = T._build_net(z, tag)
            x = torch.from_numpy(z[tag + '/x']).cuda()
            tgt = torch.from_numpy(z[tag + '/target']).cuda()
            crit = MultiSegmentationLosses('dice_ce', kw['depth']) if multi else SegmentationLosses('dice_ce')
            crit(net(x), tgt).backward()
            got = T.grads_of(net)
            e64 = oracle_grads(z, tag, kw, multi, torch.float64)
            e32 = oracle_grads(z, tag, kw, multi, torch.float32)
            p32 = oracle_grads(z, tag, kw, multi, torch.float32, eps=1e-6)
            top = max(float(np.abs(v).max()) for v in e64.values())
            rows = []
            for k, e in e64.items():
                if k not in got:
                    continue
                scale = max(float(np.abs(e).max()), 1e-3 * top)
                rows.append((float(np.abs(got[k] - e).max()) / scale, float(np.abs(e32[k] - e).max()) / scale,
                             float(np.abs(p32[k] - e32[k]).max()) / scale, k))
            rows.sort(reverse=True)
            print('%s  loss=%s  tensors=%d' % (tag, 'multi' if multi else 'last', len(rows)))
            for g, r, p, k in rows[:8]:
                print('   %-56s gpu-vs-64 %.2e   ref32-vs-64 %.2e   ref32 under 1e-6 input noise %.2e' % (k, g, r, p))


if __name__ == '__main__':
    main()
