#!/usr/bin/env python3
"""Which torch (aten) ops still launch kernels inside one eager search step driven by SearchStep (everything that is not
a libsenas_hip launch): op name, input shapes, count.   python tools/glue_ops.py [derived]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile
    from senas_amd.loss import SegmentationLosses
    dev = torch.device('cuda:0')
    crit = SegmentationLosses('dice_ce')
    if len(sys.argv) > 1 and sys.argv[1] == 'derived':
        from senas_amd.step import TrainStep
        net = bench.build_derived(dev)
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        x, y = bench.synthetic(8, 1, 2, 256, 1, dev)
        step = TrainStep(net, crit, opt, x, y, use_graph=False)
        run = step
    else:
        from senas_amd.senas_search import NAS
        from senas_amd.step import SearchStep
        torch.manual_seed(0)
        net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev).train()
        opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
        opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
        x, y = bench.synthetic(4, 1, 2, 256, 1, dev)
        step = SearchStep(net, crit, opt_w, opt_a, x.clone(), y.clone(), use_graph=False)

        def run():
            return step(x, y, x, y)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        run()
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True):
        if ev.device_time_total > 0 and ev.key.startswith('aten::'):
            rows.append((ev.count, ev.key, str(ev.input_shapes)[:100], ev.device_time_total))
    rows.sort(reverse=True)
    for cnt, key, shapes, t in rows[:45]:
        print('%5d  %-34s %8.1f us  %s' % (cnt, key, t, shapes))


if __name__ == '__main__':
    main()
