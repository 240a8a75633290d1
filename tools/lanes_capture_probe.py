#!/usr/bin/env python3
"""Where does a multi-lane capture of the supernet die?  Stages, each announced before it runs (faulthandler on):
   python -X faulthandler tools/lanes_capture_probe.py [c] [depth] [size] [--fwd-only] [--count]"""
import faulthandler
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import bench  # noqa: E402
from senas_amd import functional as F  # noqa: E402
from senas_amd import grid  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.senas_search import NAS  # noqa: E402


def say(msg):
    sys.stderr.write('[probe] %s\n' % msg)
    sys.stderr.flush()


def main():
    nums = [int(a) for a in sys.argv[1:] if not a.startswith('--')]
    c, depth, size = (nums + [8, 5, 64])[:3] if len(nums) < 3 else nums[:3]
    fwd_only, count = '--fwd-only' in sys.argv, '--count' in sys.argv
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = NAS(1, c, 2, depth, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
    crit = SegmentationLosses('dice_ce')
    x, y = bench.synthetic(2, 1, 2, size, 3, dev)

    def body():
        out = net(x)
        if fwd_only:
            return out[-1].detach()
        loss = crit(out, y)
        loss.backward()
        F.join_lanes()
        return loss.detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            with torch.set_grad_enabled(not fwd_only):
                body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    say('warm-up done; capture begins (lanes=%s)' % grid.Lanes.enabled)
    from senas_amd.arena import reset_arena
    reset_arena()
    g = torch.cuda.CUDAGraph(keep_graph=True) if count else torch.cuda.CUDAGraph()
    with torch.set_grad_enabled(not fwd_only):
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            res = body()
            say('body captured; capture ends')
    say('capture ended')
    if count:
        from senas_amd.step import _graph_nodes
        say('nodes: %s' % _graph_nodes(g))
        g.instantiate()
        say('instantiated')
    reset_arena()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        say('replay %d ok: %s' % (i, float(res.float().sum())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    say('replay: %.3f ms' % (e0.elapsed_time(e1) / 20))


if __name__ == '__main__':
    if '--serial' in sys.argv:
        grid.Lanes.enabled = False
    main()
