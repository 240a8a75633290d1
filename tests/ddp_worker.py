#!/usr/bin/env python3
"""One rank of the two-rank checks of the PRODUCT step drivers (tests/test_gpu_steps.py starts it through
torch.distributed.run; not collected by pytest).

    ddp_worker.py <train|search> <backend> <one-device: 0|1> <out.json>

Both ranks build the same network from different seeds (the broadcast must fix that), run the graphed step on their own
shard of the global batch and check
  * after the first backward + all-reduce: every gradient equals the MEAN over the ranks of the CPU oracle's gradients
    for the ranks' shards (the oracle runs each shard on its own, as a replica with per-replica batch-norm statistics does);
  * after two optimizer steps: the replicas hold bit-identical parameters, and they moved.
Data parallelism replaces nn.DataParallel of experiments/train_model.py:135-137.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def digest(net):
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        if 'running' not in k and 'num_batches' not in k:          # batch-norm buffers are per replica by design
            h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def oracle_grads(kind, net, x, y, geno):
    """Gradients of the CPU oracle for this rank's shard, in named_parameters() order of ``net``."""
    from oracle import senas_ref as R            # checker only
    import golden_io as gio
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    if kind == 'train':
        gio.share_stem(sd, '')
        out = R.derived_forward(sd, x.cpu(), R.Genotype(*geno), depth=net._depth)[-1]
    else:
        gio.share_stem(sd, 'net.')
        out = R.nas_forward(sd, x.cpu(), depth=net._depth, nodes=net._meta_node_num)[-1]
    R.dice_ce_loss(out, y.cpu()).backward()
    grads = []
    for name, _ in net.named_parameters():
        key = name
        if key not in sd or sd[key].grad is None:
            key = name.replace('blocks.0.0.', 'stem1.')
        grads.append(sd[key].grad.reshape(-1))
    return torch.cat(grads)


def main():
    kind, backend, one_device, out_path = sys.argv[1], sys.argv[2], sys.argv[3] == '1', sys.argv[4]
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    index = 0 if one_device else int(os.environ.get('LOCAL_RANK', rank))
    dev = torch.device('cuda', index)
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group(backend)
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.parallel import broadcast_parameters
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep, TrainStep
    torch.manual_seed(3 + rank)                                   # different initial weights: the broadcast must fix that
    gen = torch.Generator().manual_seed(100 + rank)               # every rank its own shard
    crit = SegmentationLosses('dice_ce')
    if kind == 'train':
        net = SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4).to(dev).train()
    else:
        net = NAS(1, 8, 2, 4, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev).train()
    broadcast_parameters(net)
    xs = torch.randn(4, 2, 1, 64, 64, generator=gen).to(dev)
    ys = torch.randint(0, 2, (4, 2, 64, 64), generator=gen).to(dev)
    if kind == 'train':
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        step = TrainStep(net, crit, opt, xs[1].clone(), ys[1].clone(), world_size=world, grad_clip=5.0, use_graph=True)
        fb, run = step.fb, (lambda k: step())
    else:
        opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
        opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-3)
        step = SearchStep(net, crit, opt_w, opt_a, xs[1].clone(), ys[1].clone(), world_size=world, grad_clip=5.0, use_graph=True)
        fb, run = step.fb, (lambda k: step(xs[1], ys[1], xs[0], ys[0]))
    assert fb.graph is not None and (fb.graph_tail is not None) == (world > 1), 'the overlapped two-graph backward must be on'
    # ---- the averaged gradient of the weight pass vs the mean of the per-shard oracle gradients
    mine = oracle_grads(kind, net, xs[1], ys[1], senas_node_4)
    fb()
    fb.finish()
    torch.cuda.synchronize()
    got = torch.cat([p.grad.detach().reshape(-1) for p in net.parameters()]).cpu()
    mine = mine.to(dev) if backend == 'nccl' else mine
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    want = torch.stack([p.cpu() for p in parts]).mean(0)
    err = float((got - want).abs().max() / want.abs().max())
    # ---- two full steps: replicas identical, parameters moved
    before = digest(net)
    losses = [float(run(k)) for k in range(2)]
    torch.cuda.synchronize()
    after = digest(net)
    seen = [None] * world
    dist.all_gather_object(seen, (after, losses, err))
    if rank == 0:
        json.dump({'world': world, 'kind': kind, 'backend': backend, 'replicas_identical': all(s[0] == seen[0][0] for s in seen),
                   'moved': before != after, 'losses': [s[1] for s in seen], 'grad_vs_oracle_mean': [s[2] for s in seen],
                   'two_graph_backward': fb.graph_tail is not None}, open(out_path, 'w'))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
