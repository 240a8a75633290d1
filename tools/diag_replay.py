import os, sys
sys.path.insert(0, '/root/repo')
import torch, bench
from senas_amd import step as S, grid
from senas_amd.loss import SegmentationLosses
from senas_amd.senas_search import NAS
lanes = '--serial' not in sys.argv
grid.Lanes.enabled = lanes
dev = torch.device('cuda:0')
torch.manual_seed(1)
crit = SegmentationLosses('dice_ce')
x, y = bench.synthetic(2, 1, 2, 64, 5, dev)
net = NAS(1, 8, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
ow = torch.optim.SGD(net.parameters(), lr=0.0)
oa = torch.optim.SGD(net.arch_parameters(), lr=0.0)
drv = S.SearchStep(net, crit, ow, oa, x.clone(), y.clone(), grad_clip=0.0, use_graph='--eager' not in sys.argv)
outs = []
for rep in range(4):
    drv.fb()
    torch.cuda.synchronize()
    outs.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
bad = []
for k in outs[0]:
    for r in (1, 2, 3):
        d = float((outs[0][k] - outs[r][k]).abs().max())
        s = float(outs[0][k].abs().max())
        if d > 1e-4 * max(s, 1e-6):
            bad.append((k, r, d, s, float(outs[r][k].abs().max())))
print('lanes', lanes, 'tensors that differ between replays:', len(bad))
for b in bad[:12]:
    print(b)
