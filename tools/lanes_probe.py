#!/usr/bin/env python3
"""Macro-grid lanes (senas_amd.grid.Lanes): the columns of up cells on their own HIP streams.

    python tools/lanes_probe.py check            serial schedule vs lanes on small nets, eager: logits and every gradient
    python tools/lanes_probe.py time [steps]     search step and derived train step under HIP-graph replay, lanes off / on

Prints one JSON object per line.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from senas_amd import grid  # noqa: E402


def _grads(net, crit, x, y):
    for p in net.parameters():
        p.grad = None
    out = net(x)
    loss = crit(out, y)
    loss.backward()
    torch.cuda.synchronize()
    return [o.detach().clone() for o in out], {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def check():
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    dev = torch.device('cuda:0')
    crit = SegmentationLosses('dice_ce')
    cases = [('nas.c8.d5', lambda: NAS(1, 8, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False)),
             ('nas.c32.d4.sup', lambda: NAS(1, 32, 2, 4, meta_node_num=3, use_sharing=True, double_down_channel=False, supervision=True)),
             ('derived.c32.d5', lambda: SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4))]
    for name, make in cases:
        torch.manual_seed(0)
        net = make().to(dev).train()
        x, y = bench.synthetic(2, 1, 2, 64, 3, dev)
        res = {}
        for lanes in (False, True, True):
            grid.Lanes.enabled = lanes
            state = {k: v.clone() for k, v in net.state_dict().items()}
            outs, grads = _grads(net, crit, x, y)
            net.load_state_dict(state)
            res.setdefault(lanes, []).append((outs, grads))
        (o0, g0), (o1, g1), (o2, g2) = res[False][0], res[True][0], res[True][1]
        worst = lambda ga, gb: max(float((ga[k] - gb[k]).abs().max() / (ga[k].abs().max() + 1e-30)) for k in ga)
        print(json.dumps({'case': name, 'grads': len(g0),
                          'logits_serial_vs_lanes': max(float((a - b).abs().max()) for a, b in zip(o0, o1)),
                          'grad_rel_serial_vs_lanes': worst(g0, g1), 'grad_rel_lanes_vs_lanes': worst(g1, g2),
                          'bit_identical': all(torch.equal(g0[k], g1[k]) for k in g0)}), flush=True)


def timing(steps):
    dev = torch.device('cuda:0')
    args = bench.parse_args(['--steps', str(steps), '--search-steps', str(steps), '--no-cpu-baseline', '--lp-steps', '0'])
    only = os.environ.get('LANES_ONLY')
    modes = (True,) if only else (False, True)

    def search_leg():
        for lanes in modes:
            grid.Lanes.enabled = lanes
            s = bench.bench_search(dev, steps, 0, 1)
            print(json.dumps({'lanes': lanes, 'search_ms': s['ms_per_step'], 'nodes': s['roofline'].get('graph_nodes_per_step')}), flush=True)

    if not os.environ.get('TRAIN_FIRST'):
        search_leg()
    from senas_amd.loss import SegmentationLosses
    from senas_amd.step import TrainStep
    for lanes in modes:
        grid.Lanes.enabled = lanes
        net = bench.build_derived(dev)
        crit = SegmentationLosses('dice_ce')
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        x, y = bench.synthetic(args.batch, 1, 2, args.size, 1, dev)
        step = TrainStep(net, crit, opt, x, y, world_size=1, grad_clip=5.0)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        step.close()
        print(json.dumps({'lanes': lanes, 'train_ms': round(ms, 3), 'loss': float(loss)}), flush=True)
    if os.environ.get('TRAIN_FIRST'):
        search_leg()


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'check'
    if mode == 'check':
        check()
    else:
        timing(int(sys.argv[2]) if len(sys.argv) > 2 else 30)
