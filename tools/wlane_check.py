#!/usr/bin/env python3
"""Is the weight-gradient lane (functional.WLANE) race-free?  The same captured search / train step from the same state with the
lane off and on (and on again): per tensor, the largest difference of the flat gradient buffer after ONE pass on the scale of the
tensor.  Atomics alone give ~1e-6; a race gives anything."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from senas_amd import step as S  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402


def grads(kind, wlane, c, size):
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    crit = SegmentationLosses('dice_ce')
    x, y = bench.synthetic(2, 1, 2, size, 5, dev)
    if kind == 'search':
        net = NAS(1, c, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
        ow = torch.optim.SGD(net.parameters(), lr=0.0)
        oa = torch.optim.SGD(net.arch_parameters(), lr=0.0)
        drv = S.SearchStep(net, crit, ow, oa, x.clone(), y.clone(), grad_clip=0.0)
        fb = drv.fb
    else:
        net = SenasModel(2, 1, c=c, depth=5, genotype=senas_node_4).to(dev).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.0)
        drv = S.TrainStep(net, crit, opt, x, y, grad_clip=0.0)
        fb = drv.fb
    if not wlane:
        assert fb.wlane is not None
    out = []
    for rep in range(3):
        fb()
        torch.cuda.synchronize()
        out.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    drv.close()
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'search'
    c, size = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 64)
    keep = S.GraphedForwardBackward.__init__

    def no_wlane(self, *a, **k):
        keep(self, *a, **k)
    runs = {}
    for wl in (False, True):
        if not wl:
            # capture without the lane: the driver's flag is read when the pass is captured
            orig = torch.cuda.Stream
            S.GraphedForwardBackward._capture_orig = S.GraphedForwardBackward._capture

            def cap(self, warmup):
                lane, self.wlane = self.wlane, None
                try:
                    S.GraphedForwardBackward._capture_orig(self, warmup)
                finally:
                    self.wlane = lane
            S.GraphedForwardBackward._capture = cap
        else:
            S.GraphedForwardBackward._capture = S.GraphedForwardBackward._capture_orig
        runs[wl] = grads(kind, wl, c, size)
    base = runs[False][0]

    def worst(a, b):
        w, name = 0.0, None
        for k in a:
            e = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30))
            if e > w:
                w, name = e, k
        return w, name
    print('%s c=%d %dx%d' % (kind, c, size, size))
    print('lane off, replay 2 vs 1:', worst(base, runs[False][1]))
    print('lane on  vs off        :', worst(base, runs[True][0]))
    print('lane on, replay 2 vs 1 :', worst(runs[True][0], runs[True][1]))
    print('lane on, replay 3 vs 1 :', worst(runs[True][0], runs[True][2]))


if __name__ == '__main__':
    main()
