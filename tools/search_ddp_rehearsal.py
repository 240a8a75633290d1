#!/usr/bin/env python3
"""Two-rank rehearsal of the data-parallel SEARCH step on one GPU (gloo; RCCL wants one GPU per rank): both ranks run
SearchStep (two HIP graphs, architecture all-reduce of 246 floats, weight all-reduce) on different batches and must hold
bit-identical parameters afterwards.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 tools/search_ddp_rehearsal.py
"""
import hashlib
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.parallel import broadcast_parameters  # noqa: E402
from senas_amd.senas_search import NAS  # noqa: E402
from senas_amd.step import SearchStep  # noqa: E402


def digest(net):
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        if 'running' not in k and 'num_batches' not in k:          # batch-norm buffers are per replica by design
            h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    torch.manual_seed(3 + rank)                                   # different initial weights: the broadcast must fix that
    net = NAS(1, 16, 2, 4, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev).train()
    broadcast_parameters(net)
    crit = SegmentationLosses('dice_ce')
    opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
    opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-3)
    gen = torch.Generator().manual_seed(100 + rank)               # every rank its own shard
    xs = torch.randn(4, 2, 1, 64, 64, generator=gen).to(dev)
    ys = torch.randint(0, 2, (4, 2, 64, 64), generator=gen).to(dev)
    step = SearchStep(net, crit, opt_w, opt_a, xs[0].clone(), ys[0].clone(), world_size=world, grad_clip=5.0, use_graph=True)
    before = digest(net)
    losses = [float(step(xs[2 * k + 1], ys[2 * k + 1], xs[2 * k], ys[2 * k])) for k in range(2)]
    mine = digest(net)
    got = [None] * world
    dist.all_gather_object(got, (mine, losses))
    if rank == 0:
        same = all(g[0] == got[0][0] for g in got)
        print({'ranks': world, 'replicas_identical': same, 'moved': before != mine, 'losses': [g[1] for g in got]})
        assert same and before != mine
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
