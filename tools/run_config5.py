#!/usr/bin/env python3
"""BASELINE.json config 5 (synthetic stress variant): derived net, 3 input channels, 4 classes, 2x3x512x512 per GPU --
graph-replayed train step time plus one metric update; checks that the 512x512 / RGB / 4-class shapes run."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd.geno_searched import senas_node_4
from senas_amd.loss import SegmentationLosses
from senas_amd.metrics import SegmentationMetric
from senas_amd.senas_model import SenasModel
from senas_amd.step import TrainStep
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = SenasModel(4, 3, c=32, depth=5, genotype=senas_node_4).to(dev).train()
crit = SegmentationLosses('dice_ce')
x = torch.randn(2, 3, 512, 512, device=dev); y = torch.randint(0, 4, (2, 512, 512), device=dev)
opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
step = TrainStep(net, crit, opt, x, y, world_size=1, grad_clip=5.0)
for _ in range(3): loss = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
m = SegmentationMetric(4); net.eval()
with torch.no_grad(): m.update(y, net(x)[-1])
print('config5 2x3x512x512 4-class: %.2f ms/step, %.1f img/s, loss %.4f, metric %s, finite %s' % (dt * 1e3, 2 / dt, float(loss), m.get(), bool(torch.isfinite(loss))))
