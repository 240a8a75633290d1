#!/usr/bin/env python3
"""Host time of one search / train step call (enqueue only) against its GPU time: does the host keep ahead of the device?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from senas_amd.loss import SegmentationLosses
from senas_amd.senas_search import NAS
from senas_amd.step import SearchStep, TrainStep
dev = torch.device('cuda:0')
crit = SegmentationLosses('dice_ce')
torch.manual_seed(0)
net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
ow = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
oa = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
xt, yt = bench.synthetic(4, 1, 2, 256, 1, dev); xv, yv = bench.synthetic(4, 1, 2, 256, 101, dev)
drv = SearchStep(net, crit, ow, oa, xt.clone(), yt.clone())
for _ in range(5): drv(xt, yt, xv, yv)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(30):
    t0 = time.perf_counter(); drv(xt, yt, xv, yv); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print('search: host enqueue %.2f ms per step (max %.2f), wall %.2f ms per step' % (1e3 * sum(host) / 30, 1e3 * max(host), 1e3 * t_all / 30))
drv.close()
net = bench.build_derived(dev)
opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
x, y = bench.synthetic(8, 1, 2, 256, 1, dev)
drv = TrainStep(net, crit, opt, x, y)
for _ in range(5): drv()
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(30):
    t0 = time.perf_counter(); drv(); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print('train: host enqueue %.2f ms per step (max %.2f), wall %.2f ms per step' % (1e3 * sum(host) / 30, 1e3 * max(host), 1e3 * t_all / 30))
