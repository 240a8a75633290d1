#!/usr/bin/env python3
"""Benchmark of the SENAS hot path on MI355X: images/sec of the derived-genotype train step
(BASELINE.json configs[1]: models/senas_model.py, README genotype, 8x1x256x256 per GPU) and -- in the same JSON
line, under ``search_step`` -- of the supernet search step (configs[2] / [3]: 4x1x256x256 per GPU).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment, this process only starts N ranks (``python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>``, one rank per GPU over RCCL) before
touching any GPU, relays rank 0's JSON line and fails unless that line says ``n_gpus == N``; started by the driver's
own torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

One train "step" = zero_grad -> forward -> Dice+CE loss -> backward -> (gradient all-reduce) -> clip_grad_norm_(5)
-> SGD step, on a synthetic batch already resident in HBM.  Weak scaling: 8 (train) / 4 (search) images per GPU.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_PEAK_TFLOPS = 157.3        # fp32 MFMA (= vector) dense peak
# SURVEY.md section 8(d) contract figures (forward hooks on the oracle; fwd + bwd = 3 x fwd)
DERIVED_GFLOP_PER_IMG = 126.6          # derived net, 1x256x256, one step
DERIVED_GB_PER_IMG = 1.233
SUPERNET_GB_PER_IMG = 5.690            # supernet S4, one fwd + bwd
SUPERNET_GFLOP_PER_IMG = 88.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed train steps (default: ~3.5 s of GPU time)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU (BASELINE configs[1]: 8)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--search-steps', type=int, default=80, help='timed supernet search steps (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='run forward+backward eagerly instead of replaying a HIP graph')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)')
    ap.add_argument('--one-device', action='store_true',
                    help='rehearsal: every rank uses cuda:0 (needs --backend gloo; RCCL wants one GPU per rank)')
    ap.add_argument('--rank-timeout', type=float, default=540.0,
                    help='N > 1: seconds the parent waits for the ranks before it kills them and exits non-zero')
    ap.add_argument('--pg-timeout', type=float, default=120.0, help='N > 1: torch.distributed rendezvous / collective timeout, seconds')
    ap.add_argument('--lp-steps', type=int, default=None,
                    help='timed train steps of each labelled bf16-pipe block (bf16x6 / bf16x3 / bf16 / bf16s; 0 = skip them).  Default: 60 on '
                         'one GPU, 0 on N > 1 -- a scaling run times the fp32 train step and the search step, nothing else')
    ap.add_argument('--profile-math', default='f32', choices=['f32', 'bf16x6', 'bf16x3', 'bf16', 'bf16s'],
                    help='PROFILING ONLY: run the primary (timed, event-probed) train step in this math mode, so that rocprofv3 sees '
                         'the bf16-pipe kernels in the steady-state window; the JSON line says so and is not a headline')
    ap.add_argument('--cpu-uncapped', action='store_true',
                    help='also time the CPU baseline with torch.set_num_threads(os.cpu_count()) (BASELINE.md section 3, literally)')
    return ap.parse_args(argv)


def log(msg):
    """Progress on stderr (the one JSON line goes to stdout)."""
    sys.stderr.write('[bench %s] %s\n' % (time.strftime('%H:%M:%S'), msg))
    sys.stderr.flush()


def spawn_ranks(args):
    """Parent of an N-rank run: no GPU call is made in this process."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('starting %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    # watchdog: a wedged RCCL bootstrap must not eat the caller's time limit and leave no line.  The ranks run in their own
    # process group (a fresh child -- this process never touched a GPU and never re-execs); on a timeout the whole group
    # is killed and the exit code is non-zero.
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=args.rank_timeout)
    except subprocess.TimeoutExpired:
        log('the %d ranks did not finish within %.1f s: killing process group %d' % (args.gpus, args.rank_timeout, proc.pid))
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            stdout, _ = proc.communicate(timeout=10)
        except (subprocess.TimeoutExpired, ValueError):
            stdout = b''
        sys.stderr.write((stdout or b'').decode(errors='replace')[-4000:])
        sys.stderr.write('\nbench: timed out after %.1f s waiting for %d ranks (rank logs above, if any)\n' % (args.rank_timeout, args.gpus))
        sys.exit(5)
    done = subprocess.CompletedProcess(cmd, proc.returncode, stdout)
    line = None
    for raw in done.stdout.decode().splitlines():
        if raw.startswith('{') and '"metric"' in raw:
            line = raw
    if done.returncode != 0 or line is None:
        sys.stderr.write(done.stdout.decode()[-4000:])
        sys.exit(done.returncode or 3)
    if json.loads(line).get('n_gpus') != args.gpus:
        sys.stderr.write('bench: asked for %d GPUs, the ranks report n_gpus=%s\n' % (args.gpus, json.loads(line).get('n_gpus')))
        sys.exit(4)
    print(line)
    sys.exit(0)


def build_derived(dev):
    import torch
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.models import get_segmentation_model
    from senas_amd.utils import weights_init
    torch.manual_seed(0)
    net = get_segmentation_model('senas', dataset='promise12', c=32, depth=5, supervision=False, genotype=senas_node_4,
                                 double_down_channel=False)
    net.apply(weights_init)
    return net.to(dev).train()


def synthetic(batch, in_ch, ncls, size, seed, dev):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, in_ch, size, size, generator=g)
    y = torch.randint(0, ncls, (batch, size, size), generator=g)
    return x.to(dev), y.to(dev)


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on, capped at the GPU box's
    per-GPU CPU share (16) -- os.cpu_count() reports the whole host and would oversubscribe."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _oracle_leaves(net, prefix=''):
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    a, b = prefix + 'blocks.0.0.', prefix + 'stem1.'
    for k in list(sd):
        if k.startswith(a):
            sd[k] = sd[b + k[len(a):]]
    params = []
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k and not k.startswith(a):
            v.requires_grad_(True)
            params.append(v)
    return sd, params


def cpu_baseline_train(batch, size, reps=3):
    """The CPU oracle (a port of the reference's torch-CPU path, pinned by the golden vectors) running the
    identical train step on the host cores -- a reported baseline, not the thing measured."""
    import torch
    from oracle import senas_ref as R
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    from senas_amd.utils import weights_init
    torch.manual_seed(0)
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4)
    net.apply(weights_init)
    sd, params = _oracle_leaves(net)
    opt = torch.optim.SGD(params, lr=6e-3, weight_decay=5e-4, momentum=0.9)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 1, size, size, generator=g)
    y = torch.randint(0, 2, (batch, size, size), generator=g)
    geno = R.Genotype(*senas_node_4)
    times = []
    step0 = None
    for i in range(reps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        logits = R.derived_forward(sd, x, geno)[-1]
        loss = R.dice_ce_loss(logits, y)
        if i == 0:
            step0 = (logits.detach().clone(), float(loss.detach()), y)        # the parity gate's reference (bench: gate_compare)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5)
        opt.step()
        log('cpu baseline (train) step %d: %.2f s' % (i, time.perf_counter() - t0))
        if i > 0:
            times.append(time.perf_counter() - t0)
    return batch / min(times), step0


def cpu_baseline_search(batch, size, reps=2):
    """The oracle running one search step (architecture pass on a validation batch + Adam, weight pass on a training
    batch + clip + SGD; experiments/search_arc.py:252-299) on the host cores."""
    import torch
    from oracle import senas_ref as R
    from senas_amd.senas_search import NAS
    torch.manual_seed(0)
    net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device='cpu')
    sd, params = _oracle_leaves(net, 'net.')
    arch = [sd[k] for k in ('alphas_dn', 'alphas_up', 'alphas_dn_nm', 'alphas_up_nm', 'betas_dn', 'betas_up', 'gamma')]
    opt_w = torch.optim.SGD(params, lr=5e-3, weight_decay=3e-4, momentum=0.9)
    opt_a = torch.optim.Adam(arch, lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    g = torch.Generator().manual_seed(1)
    xt, yt = torch.randn(batch, 1, size, size, generator=g), torch.randint(0, 2, (batch, size, size), generator=g)
    xv, yv = torch.randn(batch, 1, size, size, generator=g), torch.randint(0, 2, (batch, size, size), generator=g)
    with torch.no_grad():                                                     # the parity gate's reference: step 0's forward on the train batch
        logits0 = R.nas_forward(sd, xt)[-1]
        step0 = (logits0.clone(), float(R.dice_ce_loss(logits0, yt)), yt)
    times = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        opt_a.zero_grad()
        R.dice_ce_loss(R.nas_forward(sd, xv)[-1], yv).backward()
        opt_a.step()
        opt_w.zero_grad()
        R.dice_ce_loss(R.nas_forward(sd, xt)[-1], yt).backward()
        torch.nn.utils.clip_grad_norm_(params, 5)
        opt_w.step()
        log('cpu baseline (search) step %d: %.2f s' % (i, time.perf_counter() - t0))
        if i > 0:
            times.append(time.perf_counter() - t0)
    return batch / min(times), step0


# ---- parity gates in the same invocation ---------------------------------------
GATE_LOGITS = 1e-3          # north_star: outputs / loss within 1e-3 rel fp32 of the reference's CPU path
GATE_SCHEDULE = 5e-5        # the captured pass on lanes + lane scheduler against the same pass eagerly on one stream
_PENDING_GATES = {}         # GPU-side step-0 results waiting for the oracle's (the CPU baseline runs last)


def gate_forward(net, crit, x, y):
    """Step 0 of the TIMED network on the GPU, before any optimizer step: logits, loss, arg-max mask counts and Dice (the device
    metric kernels) of the freshly initialised network on the timed batch."""
    import torch
    from senas_amd.metrics import SegmentationMetric
    with torch.no_grad():
        logits = net(x)[-1]
        loss = float(crit([logits], y))
        metric = SegmentationMetric(logits.shape[1])
        metric.update(y, logits)
        pix, miou, dice = metric.get()
    return {'logits': logits.detach().float().cpu(), 'loss': loss, 'dice': dice, 'miou': miou}


def gate_compare(gpu, ref_logits, ref_loss, y):
    """The GPU's step 0 against the oracle's on the same seed-0 weights and seed-1 batch: logits and loss to 1e-3, the arg-max
    mask bit for bit wherever the oracle's top-2 margin exceeds the logits' own error bound (the pixels inside the margin are
    counted, not compared), Dice equal when the masks are."""
    import numpy as np
    import torch
    from oracle import senas_ref as R                 # checker only
    ref = ref_logits.detach().float()
    got = gpu['logits']
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max()) / scale
    loss_err = abs(gpu['loss'] - ref_loss) / abs(ref_loss)
    top2 = ref.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    decided = margin > 2.0 * GATE_LOGITS * scale      # (an error of 1e-3 of the scale on each of two logits cannot flip these)
    m_got, m_ref = got.argmax(1), ref.argmax(1)
    mismatch = int(((m_got != m_ref) & decided).sum())
    tp, fp, fn = R.hard_counts(ref, y.cpu())
    ref_dice = R.dice_from_counts(tp, fp, fn)
    same_mask = bool((m_got == m_ref).all())
    dice_ok = (abs(gpu['dice'] - ref_dice) <= 1e-3) if same_mask else (abs(gpu['dice'] - ref_dice) <= 0.5)
    ok = err <= GATE_LOGITS and loss_err <= GATE_LOGITS and mismatch == 0 and dice_ok
    return {'pass': bool(ok), 'logits_max_rel_err': err, 'loss': gpu['loss'], 'oracle_loss': ref_loss, 'loss_rel_err': loss_err,
            'bound': GATE_LOGITS,
            'argmax_mask': {'pixels': int(decided.numel()), 'compared_bit_exact': int(decided.sum()), 'mismatches': mismatch,
                            'inside_top2_margin_not_compared': int((~decided).sum()),
                            'disagreeing_inside_margin': int(((m_got != m_ref) & ~decided).sum())},
            'dice': {'gpu': gpu['dice'], 'oracle': ref_dice, 'masks_identical': same_mask}}


def gate_schedule(fb):
    """Every parameter gradient of ONE captured pass as the timed loop runs it (lanes + lane scheduler + weight-gradient lane,
    HIP-graph replay) against the same pass launched eagerly on one stream, at the bench's own size: 5e-5 of the tensor scale
    (what differs is the order of atomics)."""
    import torch
    from senas_amd import grid
    if fb.graph is None:
        return None
    params = [p for p in fb.reducer.sink.params if p.requires_grad]
    frozen = set(id(p) for p in fb.frozen)
    params = [p for p in params if id(p) not in frozen]
    collect, fb._collectives = fb._collectives, False
    try:
        fb()                                            # the replay
        torch.cuda.synchronize()
        got = [p.grad.detach().clone() for p in params]
        on, grid.Lanes.enabled = grid.Lanes.enabled, False
        try:
            fb._eager()
        finally:
            grid.Lanes.enabled = on
        torch.cuda.synchronize()
        want = [p.grad.detach().clone() for p in params]
    finally:
        fb._collectives = collect
    top = max(float(v.abs().max()) for v in want)
    worst = 0.0
    for g, w in zip(got, want):
        scale = max(float(w.abs().max()), 1e-3 * top)
        worst = max(worst, float((g - w).abs().max()) / scale)
    return {'pass': bool(worst <= GATE_SCHEDULE and top > 0), 'tensors': len(params), 'worst_rel_err': worst, 'bound': GATE_SCHEDULE,
            'replayed_by': 'lane scheduler' if fb.sched is not None else 'runtime graph replay (one stream)'}


BF16_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA


def lp_accuracy(dev):
    """Max error of every math mode against the CPU oracle on one small derived net (c = 32, depth 2, 2x1x64x64; fp32 oracle,
    fp64 for the gradients would not change the picture): logits / worst parameter-gradient L2.  Rank 0, N = 1 only."""
    import torch
    from oracle import senas_ref as R            # checker only
    from senas_amd import functional as F
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.utils import weights_init
    torch.manual_seed(11)
    net = SenasModel(2, 1, c=32, depth=2, genotype=senas_node_4)
    net.apply(weights_init)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 1, 64, 64, generator=g)
    y = torch.randint(0, 2, (2, 64, 64), generator=g)
    sd, params = _oracle_leaves(net)
    sd = {k: (v.double().detach().requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in sd.items()}
    a, b = 'blocks.0.0.', 'stem1.'
    for k in list(sd):
        if k.startswith(a):
            sd[k] = sd[b + k[len(a):]]
    ref = R.derived_forward(sd, x.double(), R.Genotype(*senas_node_4), depth=2)[-1]
    R.dice_ce_loss(ref, y).backward()
    net = net.to(dev).train()
    buffers = {k: v.detach().clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}
    out = {}
    prev = F.math_name()
    for mode in ('f32', 'bf16x6', 'bf16x3', 'bf16', 'bf16s'):
        F.set_math(mode)
        net.load_state_dict(buffers, strict=False)
        net.zero_grad(set_to_none=True)
        o = net(x.to(dev))[-1]
        SegmentationLosses('dice_ce')([o], y.to(dev)).backward()
        e_out = float((o.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        worst = 0.0
        for k, p in net.named_parameters():
            e = sd[k].grad
            if e is None or float(e.norm()) < 1e-9:
                continue
            worst = max(worst, float((p.grad.detach().cpu().double() - e).norm() / e.norm()))
        out[mode] = {'logits_max_rel_err_vs_oracle_fp64': e_out, 'worst_gradient_l2_rel_err_vs_oracle_fp64': worst}
    F.set_math(prev)
    net.zero_grad(set_to_none=True)
    return out


def bench_train_mode(mode, args, dev, rank, world, ref_ms):
    """A labelled block: the SAME derived train step with the dense stride-1 convolutions on the bf16 matrix pipe
    (senas_amd.functional.set_math): bf16x6 / bf16x3 split operands or plain bf16 operands, fp32 accumulation, fp32 tensors
    in HBM.  Never the headline: `value` stays the fp32-exact step."""
    import torch
    from senas_amd import functional as F
    from senas_amd.loss import SegmentationLosses
    from senas_amd.parallel import broadcast_parameters
    from senas_amd.step import TrainStep
    prev = F.set_math(mode)
    try:
        net = build_derived(dev)
        if world > 1:
            broadcast_parameters(net)
        crit = SegmentationLosses('dice_ce')
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        x, y = synthetic(args.batch, 1, 2, args.size, 1 + rank, dev)
        step = TrainStep(net, crit, opt, x, y, world_size=world, grad_clip=5.0, use_graph=not args.no_graph)
        for _ in range(max(2, args.warmup)):
            step()
        torch.cuda.synchronize()
        elapsed, loss = timed(step, args.lp_steps, world, dev)
        ms = 1e3 * elapsed / args.lp_steps
        # per-kernel HIP events of the dense convolutions in this mode (two eager passes)
        step.fb._collectives = False
        F.TIMER = F.KernelTimer()
        for _ in range(2):
            step.fb._eager()
        timer, F.TIMER = F.TIMER, None
        step.fb._collectives = True
        agg = timer.summary(F.KernelTimer.empty_pair_ms())
        step.close()
        terms = F.MATH_NAMES[mode]
        top = sorted(agg.items(), key=lambda kv: -kv[1]['ms'])[:18]
        for name, a in top:
            log('  [%s] %-44s %4d launches %8.3f ms/step  %7.2f TFLOP/s (fp32-equivalent)' % (mode, name, a['launches'] // 2, a['ms'] / 2,
                                                                                             a['flops'] / (a['ms'] * 1e-3) / 1e12))
        name, a = max(agg.items(), key=lambda kv: kv[1]['ms'])
        tf = a['flops'] / (a['ms'] * 1e-3) / 1e12
        lp_traffic, lp_source = pmc_traffic_lp(name) if mode == 'bf16x3' else (None, None)
        peak = BF16_PEAK_TFLOPS / terms                         # fp32-equivalent FLOP/s the pipe can deliver with `terms` MFMAs per product
        res = {'math': mode, 'dtype': {'bf16s': 'bf16 operands (RNE), f32 accumulate; the dense convolutions\' OUTPUTS and the gradients that come back for them '
                                                'STORED as bf16 (cell nodes read / write them as such), every other tensor f32',
                                       'bf16': 'bf16 operands (RNE), f32 accumulate, f32 storage',
                                       'bf16x3': 'f32 split into 2 bf16 planes, 3 MFMA products, f32 accumulate, f32 storage',
                                       'bf16x6': 'f32 split into 3 bf16 planes, 6 MFMA products, f32 accumulate, f32 storage'}[mode],
               'value': round(args.batch * world * args.lp_steps / elapsed, 3), 'unit': 'images/s', 'ms_per_step': round(ms, 3),
               'steps': args.lp_steps, 'speedup_vs_f32_step': round(ref_ms / ms, 3), 'loss': float(loss.detach()),
               'all_conv_ms_per_step': round(sum(v['ms'] for v in agg.values()) / 2, 2),
               'roofline': {'bound': 'mfma', 'kernel': name, 'achieved': round(tf, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s (fp32-equivalent)',
                            'frac': round(tf / peak, 4), 'traffic': lp_traffic, 'traffic_source': lp_source, 'launches': a['launches'],
                            'avg_launch_ms': round(a['ms'] / a['launches'], 4),
                            'peak_convention': '2.5 PFLOP/s dense bf16 MFMA / %d MFMA products per fp32 product' % terms}}
        return res
    finally:
        F.set_math(prev)


def _schedule_info(*passes):
    """How the captured passes are replayed: by the lane scheduler (macro-grid columns on their own streams, csrc/sched.hip) --
    lanes, segments and cross-lane dependencies per pass -- or by the runtime's graph replay on one stream."""
    info = []
    for fb in passes:
        for sched in (getattr(fb, 'sched', None), getattr(fb, 'sched_tail', None)):
            if sched is not None:
                info.append(sched.info())
    if not info:
        return {'executor': 'one stream (eager or the runtime\'s graph replay)', 'lanes': 1}
    return {'executor': 'lane scheduler: every column of up cells on a stream of its own, the captured pass replayed as '
                        'single-branch graph segments with an event per cross-lane dependency (senas_amd/grid.py, csrc/sched.hip)',
            'lanes': max(i['lanes'] for i in info), 'passes': info}


def timed(step, steps, world, dev):
    """EXACTLY ``steps`` calls bracketed by barrier + synchronize on both sides; MAX over ranks."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), out


def pmc_traffic(name):
    """HBM bytes per launch of kernel ``name`` from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
    separate runs over this very command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; aggregated by
    tools/pmc_traffic.py).  Returns (bytes or None, source)."""
    for fname in ('r5_pmc_traffic.json', 'r4_pmc_traffic.json', 'r3_pmc_traffic.json', 'r2_pmc_traffic.json', 'r1_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', fname)
        try:
            pmc = json.load(open(path))
            rec = pmc['kernels'].get(name)
            if rec is None:                          # symbol spelled with fewer defaulted template arguments
                close = [v for k, v in pmc['kernels'].items() if k.startswith(name[:-1] + ',')]
                rec = max(close, key=lambda v: v['launches']) if close else None
            if rec:
                return int((2 * rec['fetch_kb_avg'] + rec['write_kb_avg']) * 1024), \
                    'profiles/%s (rocprofv3 --pmc passes of an earlier run of this command; not measured in this run)' % fname
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def pmc_traffic_lp(name):
    """pmc_traffic for the bf16x3 mode's dominant kernel (the committed --profile-math bf16x3 counter passes)."""
    for fname in ('r5_pmc_traffic_bf16x3.json', 'r4_pmc_traffic_bf16x3.json', 'r3_pmc_traffic_bf16x3.json'):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', fname)))
            rec = pmc['kernels'].get(name)
            if rec is None:
                close = [v for k, v in pmc['kernels'].items() if k.split('<')[0] == name.split('<')[0]]
                rec = max(close, key=lambda v: v['launches']) if close else None
            if rec:
                return int((2 * rec['fetch_kb_avg'] + rec['write_kb_avg']) * 1024), \
                    'profiles/%s (rocprofv3 --pmc passes of bench.py --profile-math bf16x3; not measured in this run)' % fname
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def search_traffic():
    """HBM bytes per search step from the committed rocprofv3 --pmc passes over tools/search_profile.py (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled for gfx950; tools/pmc_traffic.py --all --steps N), with the time of the
    three heaviest kernel families from the committed steady-state table.  Returns (bytes or None, source, families)."""
    for fname in ('r5_pmc_traffic_search.json', 'r4_pmc_traffic_search.json', 'r3_pmc_traffic_search.json'):
        path = os.path.join(ROOT, 'profiles', fname)
        try:
            pmc = json.load(open(path))
            fam = list(pmc.get('families', {}).items())[:3]
            return int(pmc['hbm_bytes_per_step']), 'profiles/%s (rocprofv3 --pmc passes of tools/search_profile.py --eager, %d steps; not measured in this run)' \
                % (fname, pmc['steps']), {k: v for k, v in fam}
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def search_by_family(limit=14):
    """Per kernel family of the search step: kernel time per step in a SERIAL trace (rocprofv3 --kernel-trace of the step on one
    stream: what each family costs on its own, not what overlaps), HBM bytes per step from the counter passes, and the rate the
    two give -- so that the family furthest from the HBM roof is readable from the bench line.  From the committed tables of the
    same round (profiles/r<k>_search_steady.txt, r<k>_pmc_traffic_search.json); not measured in this run."""
    import re
    for tag in ('r5', 'r4'):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', '%s_pmc_traffic_search.json' % tag)))
            rows = {}
            for line in open(os.path.join(ROOT, 'profiles', '%s_search_steady.txt' % tag)):
                m = re.match(r'\s*family (\S+)\s+([0-9.]+) launches/step\s+([0-9.]+) ms/step', line)
                if m:
                    rows[m.group(1)] = (float(m.group(2)), float(m.group(3)))
            out = []
            for name, (launches, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:limit]:
                fam = pmc.get('families', {}).get(name)
                gb = fam['hbm_bytes_per_step'] / 1e9 if fam else None
                out.append({'family': name, 'launches_per_step': launches, 'serial_ms_per_step': round(ms, 3),
                            'hbm_gb_per_step': round(gb, 3) if gb is not None else None,
                            'tb_per_s': round(gb / ms, 2) if gb is not None and ms > 0 else None,
                            'frac_of_hbm_peak': round(gb / ms / (HBM_PEAK_GBS / 1e3), 3) if gb is not None and ms > 0 else None})
            if out:
                return {'families': out, 'source': 'profiles/%s_search_steady.txt (serial kernel time per step) and profiles/%s_pmc_traffic_search.json '
                                                   '(FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes); not measured in this run' % (tag, tag)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args)                          # never returns
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.lp_steps is None:
        args.lp_steps = 60 if world == 1 else 0
    if world != args.gpus:
        log('WORLD_SIZE=%d but --gpus %d: reporting the world size actually running' % (world, args.gpus))
    dev_index = 0 if args.one_device else local_rank
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ:
            log('HSA_ENABLE_IPC_MODE_LEGACY is not exported; setting it to 0 (dmabuf IPC) for RCCL')
            os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        torch.cuda.set_device(dev_index)
        import datetime
        pg_timeout = datetime.timedelta(seconds=args.pg_timeout)       # a rank that never shows up fails the others, not hangs them
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index), timeout=pg_timeout)
        else:
            dist.init_process_group(args.backend, timeout=pg_timeout)
    dev = torch.device('cuda', dev_index)
    torch.cuda.set_device(dev)

    from senas_amd import _lib, functional as F
    from senas_amd.loss import SegmentationLosses
    from senas_amd.parallel import broadcast_parameters
    _lib.lib()

    if args.profile_math != 'f32':
        F.set_math(args.profile_math)
        log('PROFILING RUN: primary train step in math mode %s' % args.profile_math)
    net = build_derived(dev)
    if world > 1:
        broadcast_parameters(net)
    crit = SegmentationLosses('dice_ce')
    opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)     # senas_promise12.yml training block
    x, y = synthetic(args.batch, 1, 2, args.size, 1 + rank, dev)
    from senas_amd.step import TrainStep
    want_gate = rank == 0 and world == 1 and not args.no_cpu_baseline and args.profile_math == 'f32'
    gate_gpu = gate_forward(net, crit, x, y) if want_gate else None
    log('model on %s, %d params; %s forward+backward' % (dev, sum(p.numel() for p in net.parameters()),
                                                          'eager' if args.no_graph else 'capturing HIP graph of'))
    step = TrainStep(net, crit, opt, x, y, world_size=world, grad_clip=5.0, use_graph=not args.no_graph)
    gate_sched = gate_schedule(step.fb) if args.profile_math == 'f32' else None          # (before any optimizer step)
    log('warm-up x%d' % args.warmup)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log('timing %d steps' % args.steps)
    if os.environ.get('SENAS_TRACE_MARKER'):
        torch.cuda._sleep(1000)
    elapsed, loss = timed(step, args.steps, world, dev)
    if os.environ.get('SENAS_TRACE_MARKER'):
        torch.cuda._sleep(1000)
    # per-kernel HIP-event timing: the same pass run eagerly right after the timed region (events cannot
    # be read back from inside a replayed graph; kernel durations are the same in both modes)
    probe_steps = 2
    schedule = _schedule_info(step.fb)
    step.fb._collectives = False                                # (no collective inside the probe passes)
    F.TIMER = F.KernelTimer()                                   # (a pair launch is one span: one kernel instance, two problems' flops)
    from senas_amd import grid
    lanes_on, grid.Lanes.enabled = grid.Lanes.enabled, False    # (one stream: a span must not see another lane's kernel beside its own)
    for _ in range(probe_steps):
        step.fb._eager()
    grid.Lanes.enabled = lanes_on
    timer, F.TIMER = F.TIMER, None
    step.fb._collectives = True
    images = args.batch * world * args.steps
    value = images / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (HIP events on the launch stream)
    event_overhead_ms = F.KernelTimer.empty_pair_ms()
    agg = timer.summary(event_overhead_ms)
    roof = None
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])[:14]:
        log('  %-44s %4d launches %8.3f ms/step  %7.2f TFLOP/s  %7.1f GB/s(alg)' % (
            name, a['launches'] // probe_steps, a['ms'] / probe_steps, a['flops'] / (a['ms'] * 1e-3) / 1e12,
            a['bytes'] / (a['ms'] * 1e-3) / 1e9))
    if agg:
        name, a = max(agg.items(), key=lambda kv: kv[1]['ms'])
        per_launch_ms = a['ms'] / a['launches']
        tflops = a['flops'] / (a['ms'] * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(name)
        # the same kernel symbol serves several layer shapes: per-geometry rates of its launches (n,hi,wi,ci,ho,wo,co,kh,kw,s,p,d)
        by_geo = {}
        for rname, flops, nbytes, e0, e1, tag in timer.records:
            if rname == name and tag is not None:
                ms = e0.elapsed_time(e1)
                b = by_geo.setdefault(tag[1:13] + tag[15:], [0, 0.0, 0.0])        # (+ 'pair': two problems of this geometry in one launch)
                b[0] += 1
                b[1] += max(ms - event_overhead_ms, 0.5 * ms)
                b[2] += flops
        geo_rows = [{'geometry': ','.join(str(v) for v in k), 'launches': v[0], 'avg_ms': round(v[1] / v[0], 4),
                     'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(by_geo.items(), key=lambda kv: -kv[1][1])[:4]]
        step_tflops = args.batch * DERIVED_GFLOP_PER_IMG / ms_per_step
        roof = {'bound': 'mfma', 'kernel': name, 'achieved': round(tflops, 3), 'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tflops / F32_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                'launches': a['launches'],
                'algorithmic_bytes_per_launch': int(a['bytes'] / a['launches']),
                'avg_launch_ms': round(per_launch_ms, 4),
                'algorithmic_gflop_per_launch': round(a['flops'] / a['launches'] / 1e9, 3),
                'share_of_step': round((a['ms'] / probe_steps) / ms_per_step, 3),
                'all_conv_ms_per_step': round(sum(v['ms'] for v in agg.values()) / probe_steps, 2),
                # the whole step against the same roof: SURVEY 8(d) FLOPs of one step / step time / peak
                'step_achieved': round(step_tflops, 2), 'step_frac': round(step_tflops / F32_PEAK_TFLOPS, 4),
                'step_hbm_gbs_algorithmic': round(args.batch * DERIVED_GB_PER_IMG / (ms_per_step * 1e-3), 1),
                'event_pair_overhead_us': round(1e3 * event_overhead_ms, 2),
                'by_geometry': geo_rows,
                'timing_source': 'HIP events around each launch, %d eager passes right after the timed region; every span '
                                 'less the reading of an empty event pair' % probe_steps}

    out = {
        'metric': 'images/sec at 256x256 - senas derived-genotype train step (fwd+loss+bwd+clip+SGD); supernet search step under search_step',
        'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if args.profile_math == 'f32' else 'PROFILING RUN in %s (not a headline)' % args.profile_math, 'data': 'synthetic (randn slices, randint labels, seed 1+rank), random-init weights',
        'config': {'workload': 'BASELINE configs[1]: SenasModel README genotype (senas_node_4), c=32 depth=5, '
                               '%dx1x%dx%d per GPU, fp32' % (args.batch, args.size, args.size),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'loss': float(loss.detach()),
                   'hip_graph': bool(step.graphed), 'timed_region_s': round(elapsed, 3), 'schedule': schedule,
                   'allreduce_overlapped_with_backward': bool(step.fb.graph_tail is not None)},
        'roofline': roof,
    }
    if gate_sched is not None:
        out['parity_gate'] = {'schedule_vs_serial_eager': gate_sched}
    log('train step: %.2f ms/step, %.2f images/s' % (ms_per_step, value))
    step.close()
    del step, net, opt
    if args.profile_math != 'f32':
        F.set_math('f32')

    if args.lp_steps > 0:
        for mode in ('bf16x6', 'bf16x3', 'bf16', 'bf16s'):
            out['train_step_' + mode] = bench_train_mode(mode, args, dev, rank, world, ms_per_step)
            if rank == 0:
                log('train step [%s]: %.2f ms/step, %.2f images/s' % (mode, out['train_step_' + mode]['ms_per_step'], out['train_step_' + mode]['value']))
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            acc = lp_accuracy(dev)
            out['accuracy_vs_oracle'] = acc
            for mode in ('bf16x6', 'bf16x3', 'bf16', 'bf16s'):
                out['train_step_' + mode]['max_error_vs_oracle'] = acc[mode]

    if args.search_steps > 0:
        out['search_step'] = bench_search(dev, args.search_steps, rank, world, use_graph=not args.no_graph,
                                          gate=args.profile_math == 'f32' and not args.no_cpu_baseline)
        if rank == 0:
            log('search step: %s' % json.dumps(out['search_step']))
        if args.lp_steps > 0:
            # labelled block: the same search step with the stacked 32 -> 32 candidates on the bf16 pipe (split operands)
            from senas_amd import functional as F
            prev = F.set_math('bf16x3')
            try:
                blk = bench_search(dev, max(10, args.search_steps // 2), rank, world, use_graph=not args.no_graph)
            finally:
                F.set_math(prev)
            blk['math'] = 'bf16x3'
            blk['dtype'] = 'f32 split into 2 bf16 planes, 3 MFMA products, f32 accumulate, f32 storage (dense stride-1 convolutions only)'
            blk['speedup_vs_f32_step'] = round(out['search_step']['ms_per_step'] / blk['ms_per_step'], 3)
            out['search_step_bf16x3'] = blk
            if rank == 0:
                log('search step [bf16x3]: %.2f ms/step' % blk['ms_per_step'])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = host_threads()
        torch.set_num_threads(threads)
        log('cpu baseline on %d threads (cpu_count %s)' % (torch.get_num_threads(), os.cpu_count()))
        common = {'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'cpu': cpu_model_name(),
                  'thread_cap': 'min(cores this process may run on, 16 = the GPU box\'s per-GPU CPU share); host reports %s CPUs' % os.cpu_count()}
        v, step0 = cpu_baseline_train(args.batch, args.size, reps=3)
        if gate_gpu is not None:
            g = gate_compare(gate_gpu, *step0)
            g['what'] = ('step 0 of the timed network (seed-0 weights, the timed seed-1 batch, %dx1x%dx%d) on the GPU against the '
                         'oracle\'s step 0 in cpu_baseline_train' % (args.batch, args.size, args.size))
            out.setdefault('parity_gate', {})['step0_vs_oracle'] = g
        out['cpu_baseline'] = dict(common, value=round(v, 3),
                                   sample='same train step (oracle/senas_ref.py, torch-CPU fp32) on %dx1x%dx%d, best of 3 after 1 warm-up'
                                          % (args.batch, args.size, args.size))
        if 'search_step' in out:
            v, step0 = cpu_baseline_search(4, 256, reps=2)
            gs = _PENDING_GATES.pop('search', None)
            if gs is not None:
                g = gate_compare(gs, *step0)
                g['what'] = 'step 0 forward of the timed supernet (seed-0 weights, the timed 4x1x256x256 train batch) against the oracle\'s'
                out['search_step'].setdefault('parity_gate', {})['step0_vs_oracle'] = g
            out['search_step']['cpu_baseline'] = dict(common, value=round(v, 3),
                                                      sample='same search step (arch pass + Adam, weight pass + clip + SGD) on 4+4 images '
                                                             '1x256x256, best of 2 after 1 warm-up; train images per second')
        if args.cpu_uncapped:
            # BASELINE.md section 3 literally: torch.set_num_threads(os.cpu_count()) -- on the GPU box that is the whole
            # 256-CPU host, of which this job owns a 16-CPU share, so the figure is reported beside the capped one, once
            torch.set_num_threads(os.cpu_count() or threads)
            v, _ = cpu_baseline_train(args.batch, args.size, reps=2)
            out['cpu_baseline']['uncapped'] = {'value': round(v, 3), 'threads': torch.get_num_threads(),
                                               'note': 'torch.set_num_threads(os.cpu_count()); cores this process may run on: %d'
                                                       % len(os.sched_getaffinity(0))}
    gates = [g for blk in (out.get('parity_gate', {}), out.get('search_step', {}).get('parity_gate', {})) for g in blk.values() if g]
    failed = [g for g in gates if not g['pass']]
    if 'parity_gate' in out:
        out['parity_gate']['pass'] = not failed
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if failed:
        log('PARITY GATE FAILED: %s' % json.dumps(failed))
        sys.exit(6)


def bench_search(dev, steps, rank, world, use_graph=True, gate=False):
    """Supernet search step (BASELINE configs[2]; configs[3] with N ranks): arch step on 4 validation images (Adam) +
    weight step on 4 train images (SGD, clip 5) per GPU -- experiments/search_arc.py:252-299.  images/sec counts train
    images, whole job."""
    import torch
    from senas_amd.loss import SegmentationLosses
    from senas_amd.parallel import broadcast_parameters
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep
    torch.manual_seed(0)
    net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev).train()
    if world > 1:
        broadcast_parameters(net)
    crit = SegmentationLosses('dice_ce')
    opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
    opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    xt, yt = synthetic(4, 1, 2, 256, 1 + rank, dev)
    xv, yv = synthetic(4, 1, 2, 256, 101 + rank, dev)
    gate_gpu = gate_forward(net, crit, xt, yt) if (gate and rank == 0 and world == 1) else None
    log('search: supernet built, capturing forward+backward')
    search = SearchStep(net, crit, opt_w, opt_a, xt.clone(), yt.clone(), world_size=world, grad_clip=5.0, use_graph=use_graph,
                        count_nodes=use_graph and world == 1)

    def step():
        return search(xt, yt, xv, yv)

    gates = {}
    if gate:
        # both captured passes as the timed loop replays them against the same passes eagerly on one stream (before any optimizer step)
        search.fb.x.copy_(xv)
        search.fb.y.copy_(yv)
        gates['architecture_pass_schedule_vs_serial_eager'] = gate_schedule(search.fb_arch)
        search.fb.x.copy_(xt)
        search.fb.y.copy_(yt)
        gates['weight_pass_schedule_vs_serial_eager'] = gate_schedule(search.fb)
        gates = {k: v for k, v in gates.items() if v is not None}
    log('search: warm-up steps')
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    log('search: timing %d steps' % steps)
    if os.environ.get('SENAS_TRACE_MARKER'):       # tools/trace_by_grid.py --steady: only the launches between the two markers
        torch.cuda._sleep(1000)
    elapsed, _ = timed(step, steps, world, dev)
    if os.environ.get('SENAS_TRACE_MARKER'):
        torch.cuda._sleep(1000)
    dt = elapsed / steps
    # algorithmic HBM bytes of one step per GPU (5.690 GB per image for forward + backward = 3 x forward):
    # the weight pass is a full forward + backward over 4 images; the architecture pass runs with the weights frozen --
    # forward + data gradients, no weight gradients -- and is charged 2/3 of that
    gbytes = 4 * SUPERNET_GB_PER_IMG * (1.0 + 2.0 / 3.0)
    achieved = gbytes / dt
    traffic, traffic_source, top_families = search_traffic()
    res = {'workload': 'BASELINE configs[%d]: NAS supernet c=32 depth=5 nodes=3, arch step (4 val) + weight step (4 train) per GPU, 1x256x256'
                       % (2 if world == 1 else 3),
           'value': round(4 * world / dt, 3), 'unit': 'train images/s', 'n_gpus': world, 'train_images_per_sec': round(4 * world / dt, 3),
           'ms_per_step': round(dt * 1e3, 2), 'steps': steps, 'timed_region_s': round(elapsed, 3), 'hip_graph': bool(search.graphed),
           'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                        'traffic_gbs_at_this_step_time': round(traffic / dt / 1e9, 1) if traffic else None,
                        'top_families_by_traffic': top_families,
                        'by_family': search_by_family(),
                        'algorithmic_gb_per_step': round(gbytes, 2),
                        'convention': '4 img x 5.690 GB (weight pass, fwd+bwd) + 4 img x 2/3 x 5.690 GB (architecture pass: '
                                      'weights frozen, forward + data gradients only); whole step, not one kernel -- the step is '
                                      'thousands of 3-40 us launches, no single kernel carries more than a few percent',
                        'graph_nodes_per_step': search.graph_nodes()},
           'schedule': _schedule_info(search.fb_arch, search.fb)}
    if gates:
        res['parity_gate'] = gates
    if gate_gpu is not None:
        _PENDING_GATES['search'] = gate_gpu         # (main compares it with the oracle's step 0 once the CPU baseline has run)
    search.close()
    return res


if __name__ == '__main__':
    main()
