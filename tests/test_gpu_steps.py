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
    if kind == 'derived_sup_c32':
        # c = 32: the head's (and every cell's) weight gradients run on the two-stage kernels whose sums are DEFERRED
        return SenasModel(2, 1, c=32, depth=3, genotype=senas_node_4._replace(gamma=[1] * 6), supervision=True).to(dev()).train()
    if kind == 'supernet_sup_c32':
        return NAS(1, 32, 2, 3, meta_node_num=3, use_sharing=False, double_down_channel=False, supervision=True, device=dev()).to(dev()).train()
    sup = kind == 'supernet_sup'
    return NAS(1, 8, 2, 3 if sup else 4, meta_node_num=3, use_sharing=(kind == 'supernet_share'), double_down_channel=False,
               supervision=sup, device=dev()).to(dev()).train()


@pytest.mark.parametrize('kind', ['derived', 'derived_sup', 'derived_sup_c32', 'supernet', 'supernet_share', 'supernet_sup',
                                  'supernet_sup_c32'])
def test_grad_sink_matches_autograd(kind):
    """With a GradSink installed the backward kernels write parameter gradients into views of one flat buffer and hand
    autograd nothing; modules applied more than once per pass (the shared head under deep supervision) and stacked
    weights go through the accumulate paths.  The gradients must be the ones plain autograd delivers -- same kernels,
    same values up to the summation order of atomics and of gradients that arrive more than once."""
    from senas_amd import functional as F
    from senas_amd.gradsink import GradSink
    from senas_amd.loss import MultiSegmentationLosses, SegmentationLosses
    from senas_amd.step import _model_stacks
    net = _nets(kind)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 1, 64, 64, generator=gen).to(dev())
    y = torch.randint(0, 2, (2, 64, 64), generator=gen).to(dev())
    # *_sup: the criterion sums over ALL supervised outputs, so the shared head really receives one gradient per output
    # (SegmentationLosses looks at outputs[-1] only and would leave the other applications without a gradient)
    crit = MultiSegmentationLosses('dice_ce', 3) if '_sup' in kind else SegmentationLosses('dice_ce')
    if '_sup' in kind:
        calls = []
        heads = [m for m in net.modules() if type(m).__name__ == 'Head']
        hooks = [m.register_forward_hook(lambda mod, a, b: calls.append(1)) for m in heads]
        with torch.no_grad():
            net(x)
        for h in hooks:
            h.remove()
        assert len(heads) == 1 and len(calls) > 1, 'the head of %s is applied %d times: the case does not cover accumulation' % (kind, len(calls))
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


def test_bench_spawns_two_ranks():
    """``bench.py --gpus 2`` without a torch.distributed environment starts the ranks itself (watchdog, process-group
    timeout): rehearsed over gloo with both ranks on cuda:0 -- the RCCL twin needs two devices and runs on the first
    multi-GPU lease (what it replaces: nn.DataParallel, experiments/train_model.py:135-137)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--one-device', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
           '--search-steps', '2', '--no-cpu-baseline', '--rank-timeout', '600']
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert done.returncode == 0, done.stderr.decode()[-3000:]
    lines = [l for l in done.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, done.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['allreduce_overlapped_with_backward'] and out['config']['global_batch'] == 16
    assert out['search_step']['n_gpus'] == 2 and out['value'] > 0 and out['search_step']['value'] > 0


def test_bench_spawns_three_ranks():
    """The N-rank launcher beyond two ranks, on the one-GPU box: ``bench.py --gpus 3 --one-device --backend gloo`` -- three ranks
    rendezvous, shard the batch, run both step drivers with the two-part backward and print ONE line for the whole job.  Three,
    not eight: a GPU box lets at most six processes onto its card, and the test session and the launcher count (five ranks
    were killed by the box's process guard: seven processes on the card; four sit exactly at the limit); the driver's N = 8 run
    differs in the device indices and in RCCL (what it replaces: nn.DataParallel, experiments/train_model.py:135-137)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--one-device', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
           '--search-steps', '2', '--no-cpu-baseline', '--rank-timeout', '800']
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1000)
    assert done.returncode == 0, done.stderr.decode()[-3000:]
    lines = [l for l in done.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, done.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 3 and out['config']['allreduce_overlapped_with_backward'] and out['config']['global_batch'] == 24
    assert out['search_step']['n_gpus'] == 3 and out['value'] > 0 and out['search_step']['value'] > 0 and out['scaling'] == 'weak'


def test_bench_line_contract():
    """The default single-GPU ``bench.py`` run prints ONE JSON line that carries what the driver and the judge read: the
    metric of BASELINE.json on the derived train step in fp32 (``value`` in images/s, steps / warm-up as asked, weak
    scaling, no published baseline), the ``roofline`` block of the dominant kernel by HIP events, the ``cpu_baseline`` of the
    oracle port, the supernet search step as its own block, and the labelled bf16-pipe blocks -- never as ``value``."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--search-steps', '2', '--lp-steps', '2']
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert done.returncode == 0, done.stderr.decode()[-3000:]
    lines = [l for l in done.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), done.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['steps'] == 3 and out['warmup'] == 1 and out['higher_is_better'] is True
    assert out['scaling'] == 'weak' and out['vs_baseline'] is None and out['dtype'] == 'f32' and out['unit'] == 'images/s'
    assert out['data'].startswith('synthetic') and 'workload' in out['config'] and 'model' not in out['config']
    assert abs(out['value'] - 8 * 1e3 / out['ms_per_step']) <= 1e-2 * out['value']           # 8 images per step
    roof = out['roofline']
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and roof['peak'] == 157.3 and 'traffic' in roof
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) <= 1e-3 and 0.0 < roof['frac'] < 1.0
    cpu = out['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0 and cpu['sample'] and cpu['unit'] == 'images/s'
    search = out['search_step']
    assert search['value'] > 0 and search['roofline']['bound'] == 'hbm' and search['roofline']['peak'] == 8000.0
    assert search['cpu_baseline']['value'] > 0
    # the parity gates of the same invocation: step 0 of the timed networks against the oracle, and the captured
    # passes as the timed loop replays them against the same passes eagerly on one stream -- all at the bench's own size
    gate = out['parity_gate']
    assert gate['pass'] is True
    g0 = gate['step0_vs_oracle']
    assert g0['pass'] and g0['logits_max_rel_err'] <= 1e-3 and g0['loss_rel_err'] <= 1e-3 and g0['argmax_mask']['mismatches'] == 0
    assert g0['argmax_mask']['compared_bit_exact'] > 0.5 * g0['argmax_mask']['pixels']
    assert gate['schedule_vs_serial_eager']['pass'] and gate['schedule_vs_serial_eager']['worst_rel_err'] <= 5e-5
    sg = search['parity_gate']
    assert sg['step0_vs_oracle']['pass'] and sg['step0_vs_oracle']['argmax_mask']['mismatches'] == 0
    for k in ('architecture_pass_schedule_vs_serial_eager', 'weight_pass_schedule_vs_serial_eager'):
        assert sg[k]['pass'] and sg[k]['replayed_by'] == 'lane scheduler', (k, sg[k])
    for mode in ('bf16x6', 'bf16x3', 'bf16'):
        blk = out['train_step_' + mode]
        assert blk['math'] == mode and blk['value'] > 0 and blk['dtype'] != 'f32'


def test_bench_watchdog_kills_wedged_ranks(tmp_path):
    """The parent of an N-rank run gives up after --rank-timeout, kills the ranks' process group and exits non-zero (5):
    a wedged bootstrap must not eat the driver's limit.  (No GPU work: the ranks are stopped while they start up.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    for ranks in ('2', '8'):                               # (8 children: none of them gets as far as touching the card)
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', ranks, '--one-device', '--backend', 'gloo', '--rank-timeout', '0.5']
        done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert done.returncode == 5, (done.returncode, done.stderr.decode()[-1000:])
        assert b'timed out' in done.stderr


def test_yaml_entry_points(tmp_path):
    """``python -m senas_amd.search`` / ``senas_amd.train`` from the shipped YAML (values of the reference's
    configs/senas/senas_promise12.yml): both optimizers, ``alpha_begin``, the cosine schedule, the genotype log and a
    reference-format checkpoint that loads back -- a few steps on small synthetic slices."""
    import yaml
    from senas_amd import checkpoint
    from senas_amd.run import DEFAULT_CONFIG, load_config, search, train
    cfg = load_config(DEFAULT_CONFIG)
    cfg['searching']['alpha_begin'] = 1                   # epoch 0: weights only; epoch 1: architecture steps too
    path = str(tmp_path / 'cfg.yml')
    yaml.safe_dump(cfg, open(path, 'w'))
    common = ['--config', path, '--epochs', '2', '--steps-per-epoch', '2', '--size', '64', '--images', '8', '--batch-size', '2']
    slog = search(common + ['--save', str(tmp_path / 's')])
    assert len(slog) == 2 and all(np.isfinite(e['loss']) for e in slog) and slog[1]['lr'] < slog[0]['lr'] < 5e-3
    assert slog[-1]['genotype'].startswith('Genotype(down=[')
    ck = torch.load(str(tmp_path / 's' / checkpoint.CHECKPOINT_NAME), weights_only=False)
    assert ck['epoch'] == 2 and 'alphas_dict' in ck and 'arch_optimizer' in ck and len(ck['model_state']) == 6718
    tlog = train(common + ['--save', str(tmp_path / 't'), '--genotype', slog[-1]['genotype']])
    assert len(tlog) == 2 and all(np.isfinite(e['loss']) for e in tlog)
    ck = torch.load(str(tmp_path / 't' / checkpoint.CHECKPOINT_NAME), weights_only=False)
    assert ck['epoch'] == 2 and 'model_optimizer' in ck


def test_config5_rgb_4class_512(tmp_path):
    """BASELINE configs[4] per-GPU shard at its stated size: derived net, 3 input channels, 4 classes, 2x3x512x512, c = 32,
    depth 5 -- full-size property checks (a batch permutation permutes the logits and leaves loss and gradients in place;
    eval-mode logits of an image do not depend on its batch mate), the 4-class loss / metric kernels, and one graphed
    train step; the same net is compared with the oracle at 1x3x128x128."""
    from oracle import senas_ref as R            # checker only
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.metrics import SegmentationMetric
    from senas_amd.senas_model import SenasModel
    from senas_amd.step import TrainStep
    from senas_amd.utils import weights_init
    torch.manual_seed(5)
    net = SenasModel(4, 3, c=32, depth=5, genotype=senas_node_4)
    net.apply(weights_init)
    net = net.to(dev()).train()
    crit = SegmentationLosses('dice_ce')
    gen = torch.Generator().manual_seed(50)
    x = torch.randn(2, 3, 512, 512, generator=gen).to(dev())
    y = torch.randint(0, 4, (2, 512, 512), generator=gen).to(dev())
    runs = []
    for xx, yy in ((x, y), (x.flip(0).contiguous(), y.flip(0).contiguous())):
        net.zero_grad(set_to_none=True)
        out = net(xx)[-1]
        loss = crit([out], yy)
        loss.backward()
        runs.append((out.detach(), float(loss.detach()), {k: p.grad.detach().double() for k, p in net.named_parameters()}))
    (o0, l0, g0), (o1, l1, g1) = runs
    assert tuple(o0.shape) == (2, 4, 512, 512) and bool(torch.isfinite(o0).all())
    assert float((o1 - o0.flip(0)).abs().max()) <= 1e-4 * float(o0.abs().max())
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    for k in g0:
        norm = float(g0[k].norm())
        if norm > 1e-8:
            assert float((g0[k] - g1[k]).norm()) <= 1e-2 * norm, k
    # oracle at a size the CPU finishes: same weights, 1x3x128x128
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    xs, ys = x[:1, :, :128, :128].contiguous(), y[:1, :128, :128].contiguous()
    with torch.no_grad():
        ref = R.derived_forward(sd, xs.cpu(), R.Genotype(*senas_node_4))[-1]
        ref_loss = float(R.dice_ce_loss(ref, ys.cpu()))
    buffers = {k: v.detach().clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}
    with torch.no_grad():
        got = net(xs)[-1]
        got_loss = float(crit([got], ys))
    net.load_state_dict(buffers, strict=False)
    assert float((got.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
    assert abs(got_loss - ref_loss) <= 1e-4 * abs(ref_loss)
    # 4-class metric kernels on the full-size logits
    net.eval()
    m = SegmentationMetric(4)
    with torch.no_grad():
        full = net(x)[-1]
        one = net(x[1:2].contiguous())[-1]
        m.update(y, full)
    pix, miou, dice = m.get()
    assert 0.0 <= pix <= 100.0 and 0.0 <= miou <= 100.0 and 0.0 <= dice <= 100.0        # percentages, as the reference reports them
    assert float((one - full[1:2]).abs().max()) <= 1e-5 * float(full.abs().max())
    hard = R.hard_counts(full.cpu(), y.cpu())
    assert abs(R.miou_from_counts(*hard) - miou) < 2e-3 and abs(R.dice_from_counts(*hard) - dice) < 2e-3
    # one graphed train step at the full size
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
    step = TrainStep(net, crit, opt, x, y, world_size=1, grad_clip=5.0)
    before = net.stem0[0].weight.detach().clone()
    loss = step()
    assert bool(torch.isfinite(loss)) and abs(float(loss) - l0) <= 1e-4 * abs(l0)
    assert not torch.equal(before, net.stem0[0].weight.detach())
    step.close()


def test_packers_do_not_revalidate_each_other():
    """Several packed-weight caches coexist (a validation Evaluator built first, a TrainStep built later): a fused optimizer
    step makes ALL of them stale; refreshing one must not make another one's images count again, and an evaluator that is
    reused after training steps must see the new weights (the drivers validate between epochs, train_model.py:306-340)."""
    import copy
    from senas_amd import functional as F
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.infer import Evaluator
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.step import TrainStep
    torch.manual_seed(31)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(2, 1, 64, 64, generator=gen).to(dev())
    y = torch.randint(0, 2, (2, 64, 64), generator=gen).to(dev())
    crit = SegmentationLosses('dice_ce')
    net = SenasModel(2, 1, c=32, depth=3, genotype=senas_node_4).to(dev()).train()
    ev = Evaluator(net, 2, x, y, crit, use_graph=True)              # its packer is installed first ...
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=5e-2, momentum=0.9)
    step = TrainStep(net, crit, opt, x, y, use_graph=True)          # ... and replaced by the train step's
    for _ in range(2):
        step()                                                      # fused SGD: weights move behind torch's version counters
    assert ev.packer.stale() and step.fb.packer.stale()
    ev.packer.refresh()                                             # the evaluator's images are current now; the INSTALLED ones are not
    assert not ev.packer.stale() and step.fb.packer.stale()
    net.eval()
    with torch.no_grad():
        got = net(x)[-1]                                            # eager: must repack on its own, not trust the train packer's images
        twin = copy.deepcopy(net)
        want = twin(x)[-1]
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    net.train()
    step()
    logits, mask = ev(x, y)                                         # reused after another step: refreshes itself
    with torch.no_grad():
        twin = copy.deepcopy(net).eval()
        want = twin(x)[-1]
    assert float((logits - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert bool((mask == want.argmax(1)).all())
    step.close()
    ev.packer.uninstall()
    assert F.SINK is None


@pytest.mark.gpu
def test_train_step_with_dropout_replays_fresh_masks():
    """SenasModel(dropout_prob > 0) under the HIP-graph train step: the captured Dropout2d draws a NEW mask at every replay
    (torch's graph-safe Philox offset), so two replays on the same batch with a zero learning rate give different losses,
    and eval mode stays deterministic."""
    import torch
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.step import TrainStep
    from senas_amd.utils import weights_init
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = SenasModel(2, 1, c=8, depth=3, dropout_prob=0.3, genotype=senas_node_4).to(dev)
    net.apply(weights_init)
    net.train()
    x = torch.randn(2, 1, 32, 32, device=dev)
    y = torch.randint(0, 2, (2, 32, 32), device=dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.0)
    step = TrainStep(net, SegmentationLosses('dice_ce'), opt, x, y, use_graph=True)
    try:
        assert step.graphed
        losses = [float(step()) for _ in range(4)]
    finally:
        step.close()
    assert all(l == l and abs(l) < 1e3 for l in losses)
    assert len({round(l, 6) for l in losses}) > 1, 'every replay used the same dropout mask: %r' % (losses,)
    net.eval()
    with torch.no_grad():
        a, b = net(x)[-1], net(x)[-1]
    assert torch.equal(a, b)
