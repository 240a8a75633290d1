"""GPU checks of the step drivers' plumbing (senas_amd/step.py, gradsink.py, parallel.py): the flat gradient sink
against plain autograd, the two-rank data-parallel step against the oracle, RCCL when two devices exist, the YAML
entry points, and BASELINE configs[4] at its stated size."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device('cuda:0')


def _nets(kind):
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    torch.manual_seed(4)
    if kind == 'derived':
        return SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4).to(dev()).train()
    if kind == 'derived_sup':
        return SenasModel(2, 1, c=8, depth=3, genotype=senas_node_4._replace(gamma=[1] * 6), supervision=True).to(dev()).train()
    sup = kind == 'supernet_sup'
    return NAS(1, 8, 2, 3 if sup else 4, meta_node_num=3, use_sharing=(kind == 'supernet_share'), double_down_channel=False,
               supervision=sup, device=dev()).to(dev()).train()


@pytest.mark.parametrize('kind', ['derived', 'derived_sup', 'supernet', 'supernet_share', 'supernet_sup'])
def test_grad_sink_matches_autograd(kind):
    """With a GradSink installed the backward kernels write parameter gradients into views of one flat buffer and hand
    autograd nothing; modules applied more than once per pass (the shared head under deep supervision) and stacked
    weights go through the accumulate paths.  The gradients must be the ones plain autograd delivers -- same kernels,
    same values up to the summation order of atomics and of gradients that arrive more than once."""
    from senas_amd import functional as F
    from senas_amd.gradsink import GradSink
    from senas_amd.loss import SegmentationLosses
    from senas_amd.step import _model_stacks
    net = _nets(kind)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 1, 64, 64, generator=gen).to(dev())
    y = torch.randint(0, 2, (2, 64, 64), generator=gen).to(dev())
    crit = SegmentationLosses('dice_ce')
    assert F.SINK is None
    crit(net(x), y).backward()
    want = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    buffers = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sink = GradSink([list(net.parameters())], _model_stacks(net)).install()
    try:
        for rep in range(2):                                   # the second pass checks begin() really starts over
            sink.begin()
            crit(net(x), y).backward()
            sink.finish()
            top = max(float(w.abs().max()) for w in want.values())
            for k, p in net.named_parameters():
                assert p.grad is sink.views[id(p)], k
                # same kernels, same values; a few weight-gradient kernels accumulate with atomics, and gradients that
                # arrive more than once are summed in a different order: rounding-level differences only
                scale = max(float(want[k].abs().max()), 1e-3 * top)
                assert float((p.grad - want[k]).abs().max()) <= 2e-5 * scale, (kind, k, float((p.grad - want[k]).abs().max()), scale)
        assert len(sink.written) > 0.9 * len([p for p in net.parameters() if p.dim() > 1])
    finally:
        sink.uninstall()
    assert F.SINK is None
    del buffers


def _two_ranks(kind, backend, one_device, tmp_path):
    out = str(tmp_path / ('ddp_%s_%s.json' % (kind, backend)))
    port = 29600 + (os.getpid() % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'ddp_worker.py'), kind, backend, '1' if one_device else '0', out]
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert done.returncode == 0, done.stdout.decode()[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize('kind', ['train', 'search'])
def test_two_rank_step_drivers(kind, tmp_path):
    """The PRODUCT TrainStep / SearchStep on two ranks (gloo; both ranks on cuda:0 -- RCCL wants one GPU per rank): the
    two-graph backward with the early all-reduce, averaged gradient = mean of the shards' oracle gradients, replicas
    bit-identical after two optimizer steps."""
    r = _two_ranks(kind, 'gloo', True, tmp_path)
    assert r['two_graph_backward'] and r['replicas_identical'] and r['moved'], r
    assert max(r['grad_vs_oracle_mean']) <= 1e-3, r


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank: runs on the first multi-GPU lease')
@pytest.mark.parametrize('kind', ['train', 'search'])
def test_two_rank_step_drivers_rccl(kind, tmp_path):
    """The same over RCCL (torch.distributed backend "nccl") on two devices."""
    r = _two_ranks(kind, 'nccl', False, tmp_path)
    assert r['two_graph_backward'] and r['replicas_identical'] and r['moved'], r
    assert max(r['grad_vs_oracle_mean']) <= 1e-3, r
