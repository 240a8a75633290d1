"""The N>1 path on CPU: two processes over gloo drive senas_amd.parallel exactly as bench.py /
senas_amd.step do on GPUs (same code, no streams).  Checks: replicas start identical, every rank
ends with the mean of the per-rank gradients (architecture scalars included), both reduction modes
agree, and optimizer steps keep the replicas in lock-step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Toy(nn.Module):
    """Stands in for NAS: weights + a few 'architecture' scalars stepped by the same optimizer."""

    def __init__(self):
        super().__init__()
        self.alphas = nn.Parameter(torch.randn(9, 6) * 1e-3)
        self.betas = nn.Parameter(torch.randn(9) * 1e-3)
        self.body = nn.Sequential(nn.Conv2d(1, 4, 3, padding=1, bias=False), nn.BatchNorm2d(4), nn.ReLU(),
                                  nn.Conv2d(4, 2, 3, padding=1, bias=False))
        self.unused = nn.Parameter(torch.ones(3))          # never receives a gradient

    def forward(self, x):
        w = torch.softmax(self.alphas, -1).sum() * torch.softmax(self.betas, -1).sum()
        return self.body(x) * w


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, overlap, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from senas_amd.parallel import GradAllReducer, broadcast_parameters
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        net = Toy()
        broadcast_parameters(net, src=0)
        first = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
        red = GradAllReducer(net.parameters(), world_size=world, overlap=overlap)
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
        g = torch.Generator().manual_seed(1 + rank)        # per-rank shard of the global batch
        x = torch.randn(4, 1, 8, 8, generator=g)
        y = torch.randn(4, 2, 8, 8, generator=g)
        local = None
        for step in range(2):
            red.zero_grad()
            loss = ((net(x) - y) ** 2).mean()
            if step == 0:
                local = torch.autograd.grad(loss, [p for p in net.parameters() if p is not net.unused], retain_graph=True)
                local = torch.cat([t.reshape(-1) for t in local]).clone()
            loss.backward()
            red.finish()
            if step == 0:
                avg = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p is not net.unused]).clone()
                unused_grad = None if net.unused.grad is None else net.unused.grad.clone()
            torch.nn.utils.clip_grad_norm_([p for p in net.parameters()], 5)
            opt.step()
        final = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        out[rank] = {'first': first, 'local': local, 'avg': avg, 'final': final, 'unused': unused_grad}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_gradient_mean(overlap):
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, overlap, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert torch.equal(r0['first'], r1['first'])                        # broadcast made the replicas identical
    mean = (r0['local'] + r1['local']) / 2
    assert torch.allclose(r0['avg'], mean, rtol=1e-6, atol=1e-8)        # all-reduced gradient = mean of the shards'
    assert torch.equal(r0['avg'], r1['avg'])
    assert not torch.allclose(r0['local'], r1['local'])                 # the shards really differed
    assert torch.equal(r0['final'], r1['final'])                        # two optimizer steps later: still in lock-step
    for r in (r0, r1):
        assert r['unused'] is None or float(r['unused'].abs().max()) == 0.0


def test_single_process_is_a_no_op():
    from senas_amd.parallel import GradAllReducer
    net = Toy()
    red = GradAllReducer(net.parameters(), world_size=1)
    red.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    ((net(torch.randn(2, 1, 8, 8))) ** 2).mean().backward()
    grads = [p.grad for p in net.parameters()]
    red.finish()
    assert all(a is b for a, b in zip(grads, [p.grad for p in net.parameters()]))   # untouched, no flat copy
    assert red.flat is None
