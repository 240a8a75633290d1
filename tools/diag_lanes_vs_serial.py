import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from senas_amd import step as S, grid
from senas_amd.loss import SegmentationLosses
from senas_amd.senas_search import NAS
dev = torch.device('cuda:0')
crit = SegmentationLosses('dice_ce')
x, y = bench.synthetic(2, 1, 2, 64, 5, dev)
res = {}
C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, lanes, graphed in (('serial-eager', False, False), ('serial-eager-again', False, False), ('serial-graph', False, True), ('lanes-eager', True, False), ('lanes-graph', True, True)):
    grid.Lanes.enabled = lanes
    torch.manual_seed(1)
    net = NAS(1, C_, 2, 4, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
    ow = torch.optim.SGD(net.parameters(), lr=0.0)
    oa = torch.optim.SGD(net.arch_parameters(), lr=0.0)
    drv = S.SearchStep(net, crit, ow, oa, x.clone(), y.clone(), grad_clip=0.0, use_graph=graphed)
    for _ in range(2):
        drv.fb()
    torch.cuda.synchronize()
    res[name] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    drv.close()
base = res['serial-eager']
top = max(float(v.abs().max()) for v in base.values())
for name in ('serial-eager-again', 'serial-graph', 'lanes-eager', 'lanes-graph'):
    errs = sorted(((float((res[name][k] - base[k]).abs().max()) / max(float(base[k].abs().max()), 1e-2 * top), k) for k in base), reverse=True)
    print(name, 'worst', ['%.1e %s' % e for e in errs[:3]], 'tensors > 1e-4:', sum(1 for e in errs if e[0] > 1e-4), 'of', len(errs))
