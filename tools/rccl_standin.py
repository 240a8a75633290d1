#!/usr/bin/env python3
"""What the step will look like with N > 1 ranks, rehearsed on ONE GPU: the step drivers built with ``world_size=2`` -- backward
captured in two graphs cut at the down path, the up-path gradients "all-reduced" while the second graph runs -- with
``torch.distributed.all_reduce`` replaced by a STAND-IN that does what RCCL's kernel does to the device: a kernel of its own on a
stream of its own, busy for the time a ring all-reduce of the span takes over xGMI (bytes / 100 GB/s + 30 us of latency; SURVEY
section 5: 7.9 MB supernet, 8.7 MB derived), touching the span once.  No second process: what is measured is the schedule -- the
lane scheduler's streams + one more busy stream -- not the wire.

    python tools/rccl_standin.py [steps]         prints one JSON line per (workload, lanes) pair

RCCL itself has never run in this repository (one-GPU boxes); profiles/r5_rccl_standin.txt records which number of scheduler
streams wins with the fifth stream busy, and senas_amd/step.py picks it for world_size > 1.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

SIDE = None
CALLS = [0]


class _Work(object):
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)
        return True


def standin_all_reduce(tensor, op=None, group=None, async_op=False):
    """A kernel stream of its own, ordered behind the caller's stream like RCCL's: busy for the ring time of the span."""
    global SIDE
    if SIDE is None:
        SIDE = torch.cuda.Stream()
    CALLS[0] += 1
    cur = torch.cuda.current_stream()
    SIDE.wait_stream(cur)
    us = 30.0 + tensor.numel() * 4 / 100e9 * 1e6
    with torch.cuda.stream(SIDE):
        tensor.add_(0.0)                                            # (the span read and written once)
        torch.cuda._sleep(int(us * 2100))                           # ~2.1 GHz shader clock: cycles of the spin kernel
    ev = torch.cuda.Event()
    ev.record(SIDE)
    if async_op:
        return _Work(ev)
    cur.wait_event(ev)
    return None


def run(kind, lanes, steps):
    import torch.distributed as dist
    from senas_amd import step as S
    from senas_amd.loss import SegmentationLosses
    dev = torch.device('cuda:0')
    crit = SegmentationLosses('dice_ce')
    dist.all_reduce = standin_all_reduce
    os.environ['SENAS_MAX_LANES'] = str(lanes)
    import senas_amd.lanesched as LS
    LS.MAX_LANES = lanes
    S.SEARCH_LANES = lanes
    CALLS[0] = 0
    if kind == 'search':
        from senas_amd.senas_search import NAS
        torch.manual_seed(0)
        net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
        ow = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
        oa = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
        xt, yt = bench.synthetic(4, 1, 2, 256, 1, dev)
        xv, yv = bench.synthetic(4, 1, 2, 256, 101, dev)
        drv = S.SearchStep(net, crit, ow, oa, xt.clone(), yt.clone(), world_size=2)
        step = lambda: drv(xt, yt, xv, yv)
        two_part = drv.fb.graph_tail is not None
    else:
        net = bench.build_derived(dev)
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        x, y = bench.synthetic(8, 1, 2, 256, 1, dev)
        drv = S.TrainStep(net, crit, opt, x, y, world_size=2)
        step = drv
        two_part = drv.fb.graph_tail is not None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    calls0 = CALLS[0]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print(json.dumps({'workload': kind, 'scheduler_streams': lanes, 'ms_per_step': round(ms, 3), 'backward_in_two_graphs': bool(two_part),
                      'standin_all_reduces_per_step': (CALLS[0] - calls0) / steps}), flush=True)
    drv.close()


if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for kind in ('search', 'train'):
        for lanes in (4, 3, 2):
            if only and only != '%s%d' % (kind, lanes):
                continue
            run(kind, lanes, steps)
