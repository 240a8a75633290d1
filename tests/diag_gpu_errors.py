"""Diagnostic (not a test): prints the actual relative errors of the HIP path against the golden
vectors, per primitive / block / net, so tolerances are set from measurements."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import golden_io as gio  # noqa: E402
from test_gpu_parity import load_into, grads_of, _block, _build_net, dev  # noqa: E402


def rel(got, exp):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    return float(np.abs(got - exp).max() / max(np.abs(exp).max(), 1e-12))


def grad_errs(z, tag, mod):
    out = {}
    got = grads_of(mod)
    for k, e in gio.sub(z, tag + '/grad/').items():
        if k.endswith('#head'):
            out[k] = rel(got[k[:-5]].reshape(-1)[:e.size], e)
        elif not k.endswith('#sum'):
            out[k] = rel(got[k], e)
    return out


def main():
    from senas_amd.operations import OPS, OpType
    kinds = {'up': OpType.UP, 'down': OpType.DOWN, 'norm': OpType.NORM}
    z = gio.load('prims')
    for tag in gio.index('prims'):
        kind, name, ci, co = tag.split('.')
        mod = load_into(OPS[name](int(ci), int(co), kinds[kind], 0), gio.sub(z, tag + '/sd0/')).train()
        x = torch.from_numpy(z[tag + '/x']).to(dev()).requires_grad_(True)
        y = mod(x)
        y.backward(torch.from_numpy(z[tag + '/gy']).to(dev()))
        ge = grad_errs(z, tag, mod)
        worst = max(ge.items(), key=lambda kv: kv[1]) if ge else ('-', 0.0)
        print('%-28s y %.1e dx %.1e worst-grad %.1e (%s)' % (tag, rel(y, z[tag + '/y']), rel(x.grad, z[tag + '/dx']), worst[1], worst[0]))
    z = gio.load('blocks')
    for tag in gio.index('blocks'):
        mod = load_into(_block(tag), gio.sub(z, tag + '/sd0/')).train()
        x = torch.from_numpy(z[tag + '/x']).to(dev()).requires_grad_(True)
        y = mod(x)
        y.backward(torch.from_numpy(z[tag + '/gy']).to(dev()))
        ge = grad_errs(z, tag, mod)
        worst = max(ge.items(), key=lambda kv: kv[1]) if ge else ('-', 0.0)
        print('%-28s y %.1e dx %.1e worst-grad %.1e (%s)' % (tag, rel(y, z[tag + '/y']), rel(x.grad, z[tag + '/dx']), worst[1], worst[0]))
    from senas_amd.loss import SegmentationLosses
    z = gio.load('nets')
    for tag in gio.index('nets'):
        net, kw = _build_net(z, tag)
        x = torch.from_numpy(z[tag + '/x']).to(dev())
        tgt = torch.from_numpy(z[tag + '/target']).to(dev())
        outs = net(x)
        loss = SegmentationLosses('dice_ce')(outs, tgt)
        loss.backward()
        got = grads_of(net)
        errs = sorted(((rel(got[k], e), k, float(np.abs(e).max())) for k, e in gio.sub(z, tag + '/gradfull/').items()), reverse=True)
        print('%-24s logits %.1e loss %.1e' % (tag, rel(outs[-1], z[tag + '/logits%d' % (len(outs) - 1)]),
                                                 abs(float(loss) - float(z[tag + '/loss'])) / abs(float(z[tag + '/loss']))))
        for e, k, s in errs[:6]:
            print('      %.1e  %-50s scale %.1e' % (e, k, s))
        dg = gio.digest(z, tag + '/grad/')
        l2 = sorted(((abs(np.sqrt((got[k].astype(np.float64) ** 2).sum()) - v[1]) / max(v[1], 1e-30), k, v[1]) for k, v in dg.items()), reverse=True)
        for e, k, s in l2[:5]:
            print('   l2 %.1e  %-50s l2 %.1e' % (e, k, s))


if __name__ == '__main__':
    main()
