"""GPU parity: the HIP path (senas_amd modules -> ctypes -> libsenas_hip.so) against
(a) the golden vectors the reference produced and (b) the CPU oracle on the same seeded inputs.

Tolerance: north_star asks for 1e-3 relative fp32; the checks here use 2e-4 of the tensor's
magnitude for activations / gradients (tighter), bit-exact for genotypes.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import golden_io as gio

pytestmark = pytest.mark.gpu

REL = 2e-4


def dev():
    return torch.device('cuda:0')


def close(got, exp, what, rel=REL, floor=1e-6):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    exp = np.asarray(exp)
    assert got.shape == exp.shape, '%s: shape %s vs %s' % (what, got.shape, exp.shape)
    scale = max(float(np.abs(exp).max()), floor)
    err = float(np.abs(got - exp).max())
    assert err <= rel * scale + 1e-7, '%s: max |err| %.3e vs scale %.3e (rel %.2e)' % (what, err, scale, err / scale)


def load_into(module, arrays):
    sd = {k: torch.from_numpy(np.array(v)) for k, v in arrays.items()}
    for k in list(sd):
        if k.endswith('running_mean'):
            sd.setdefault(k[:-len('running_mean')] + 'num_batches_tracked', torch.zeros((), dtype=torch.long))
    module.load_state_dict(sd, strict=True)
    return module.to(dev())


def grads_of(module, prefix=''):
    return {prefix + k: p.grad.detach().cpu().numpy() for k, p in module.named_parameters() if p.grad is not None}


def check_param_grads(z, tag, module):
    exp = gio.sub(z, tag + '/grad/')
    got = grads_of(module)
    # gradients that are analytically ~0 (a BN scale/bias whose effect the next BN cancels) are
    # compared on the scale of the module's largest gradient, not their own
    top = max(float(np.abs(v).max()) for k, v in exp.items() if not k.endswith('#sum'))
    seen = gio.check_grads(exp, got, REL, 1e-7 + REL * 1e-2 * top, tag)
    assert seen == set(got), '%s: gradient set differs: %s' % (tag, seen ^ set(got))


def check_buffers(z, tag, module):
    sd = module.state_dict()
    for k, e in gio.sub(z, tag + '/sd1/').items():
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == int(e), '%s %s' % (tag, k)
        else:
            close(sd[k], e, '%s buffer %s' % (tag, k), rel=1e-4)


def run_fwd_bwd(module, z, tag, xkey='x'):
    x = torch.from_numpy(z[tag + '/' + xkey]).to(dev()).requires_grad_(True)
    y = module(x)
    assert y.is_contiguous(memory_format=torch.channels_last) or y.shape[1] == 1
    close(y, z[tag + '/y'], tag + ' y')
    y.backward(torch.from_numpy(z[tag + '/gy']).to(dev()))
    close(x.grad, z[tag + '/dx'], tag + ' dx')
    return x


# ------------------------------------------------------------------ primitives
@pytest.mark.parametrize('tag', gio.index('prims'))
def test_primitive(tag):
    from senas_amd.operations import OPS, OpType
    z = gio.load('prims')
    kind, name, ci, co = tag.split('.')
    mod = OPS[name](int(ci), int(co), {'up': OpType.UP, 'down': OpType.DOWN, 'norm': OpType.NORM}[kind], 0)
    load_into(mod, gio.sub(z, tag + '/sd0/')).train()
    x = run_fwd_bwd(mod, z, tag)
    check_param_grads(z, tag, mod)
    check_buffers(z, tag, mod)
    if tag + '/y_eval' in z.files:
        mod.eval()
        with torch.no_grad():
            close(mod(x.detach()), z[tag + '/y_eval'], tag + ' eval')


def _block(tag):
    from senas_amd import operations as O
    return {
        'rectify_pool': lambda: O.build_rectify(32, 32, 'down'),
        'rectify_conv': lambda: O.build_rectify(16, 32, 'down'),
        'shrink64': lambda: O.ShrinkBlock(64, 32),
        'shrink32': lambda: O.ShrinkBlock(32, 32),
        'rectify24': lambda: O.RectifyBlock(24, 32),
        'rectify128': lambda: O.RectifyBlock(128, 32),
        'reluconv': lambda: O.ReLUConv(32, 2, kernel_size=3),
        'reluconv4': lambda: O.ReLUConv(32, 4, kernel_size=3),
        'stem0': lambda: O.ConvBn(1, 32, kernel_size=7),
        'stem0_rgb': lambda: O.ConvBn(3, 32, kernel_size=7),
        'stem1': lambda: O.Stem1(32, 32),
    }[tag]()


@pytest.mark.parametrize('tag', gio.index('blocks'))
def test_block(tag):
    z = gio.load('blocks')
    mod = load_into(_block(tag), gio.sub(z, tag + '/sd0/')).train()
    run_fwd_bwd(mod, z, tag)
    check_param_grads(z, tag, mod)
    check_buffers(z, tag, mod)


@pytest.mark.parametrize('tag', gio.index('mixed'))
def test_mixed_op(tag):
    from senas_amd.cell import MixedOp
    from senas_amd.operations import OpType
    z = gio.load('mixed')
    _, kind, ci = tag.split('.')
    mod = MixedOp(int(ci), 8, {'up': OpType.UP, 'down': OpType.DOWN, 'norm': OpType.NORM}[kind])
    load_into(mod, gio.sub(z, tag + '/sd0/')).train()
    x = torch.from_numpy(z[tag + '/x']).to(dev()).requires_grad_(True)
    araw = torch.from_numpy(z[tag + '/alpha_raw']).to(dev()).requires_grad_(True)
    alpha = torch.softmax(araw, -1)
    y = mod(x, alpha, alpha)
    close(y, z[tag + '/y'], tag + ' y')
    y.backward(torch.from_numpy(z[tag + '/gy']).to(dev()))
    close(x.grad, z[tag + '/dx'], tag + ' dx')
    close(araw.grad, z[tag + '/dalpha_raw'], tag + ' dalpha')
    check_param_grads(z, tag, mod)
    check_buffers(z, tag, mod)


@pytest.mark.parametrize('tag', gio.index('cells'))
def test_cell(tag):
    from senas_amd.cell import Cell
    from senas_amd.senas_model import BuildCell
    from senas_amd.geno_searched import senas_node_4
    z = gio.load('cells')
    fam, ctype = tag.split('.')
    in0 = torch.from_numpy(z[tag + '/in0']).to(dev()).requires_grad_(True)
    in1 = torch.from_numpy(z[tag + '/in1']).to(dev()).requires_grad_(True)
    if fam == 'cell':
        mod = Cell(3, 1, in0.shape[1], 32, 32, ctype)
        raws = [torch.from_numpy(z[tag + '/' + k]).to(dev()).requires_grad_(True) for k in ('wn_raw', 'wc_raw', 'beta_raw')]
        args = [torch.softmax(r, -1) for r in raws]
    else:
        mod = BuildCell(senas_node_4, 1, in0.shape[1], 16, 16, ctype)
        raws, args = [], []
    load_into(mod, gio.sub(z, tag + '/sd0/')).train()
    y = mod(in0, in1, *args)
    close(y, z[tag + '/y'], tag + ' y')
    y.backward(torch.from_numpy(z[tag + '/gy']).to(dev()))
    close(in0.grad, z[tag + '/din0'], tag + ' din0')
    close(in1.grad, z[tag + '/din1'], tag + ' din1')
    for r, k in zip(raws, ('dwn_raw', 'dwc_raw', 'dbeta_raw')):
        close(r.grad, z[tag + '/' + k], tag + ' ' + k)
    check_param_grads(z, tag, mod)
    check_buffers(z, tag, mod)


# ------------------------------------------------------------------ whole nets
def _build_net(z, tag):
    from senas_amd.genotype import Genotype
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    kw = json.loads(str(z[tag + '/kw']))
    if 'nas' in tag.split('.'):
        kw.setdefault('use_sharing', False)
        kw.setdefault('double_down_channel', False)
        net = NAS(multi_gpus=False, device=dev(), **kw)
    else:
        net = SenasModel(genotype=gio.geno_from_json(z[tag + '/genotype'], Genotype), **kw)
    return load_into(net, gio.unpack(z, tag + '/sd0/')).train(), kw


NET_CASES = [(f, t) for f in ('nets', 'nets2') for t in gio.index(f)]


def _stored_gradient_spread(fixture, tag):
    z = gio.load('nets_spread')
    key = fixture + '/' + tag
    return dict(zip(json.loads(str(z[key + '/names'])), (float(v) for v in z[key + '/spread'])))


def _oracle_gradient_spread(z, tag, kw, names, trials=3):
    """How far the ORACLE's own fp32 gradients of ``names`` move (relative to the tensor scale, floored at 1e-3 of the largest)
    when the input and every weight are perturbed by 1e-6 relative -- what a different fp32 summation order amounts to.  These
    c = 8 fixtures carry no stored spread (round 1 / 2), and a 1e-7 change of the mixing weights alone moves their architecture
    gradients by 5e-4 (tools/diag_archmix.py, round 3): the per-tensor bound has to know the fixture's conditioning."""
    from oracle import senas_ref as R

    def run(seed):
        sd = gio.add_missing_counters(gio.torch_sd(gio.unpack(z, tag + '/sd0/')))
        x = torch.from_numpy(z[tag + '/x'])
        if seed:
            g = torch.Generator().manual_seed(seed)
            sd = {k: ((v.detach() * (1 + 1e-6 * torch.randn(v.shape, generator=g))).requires_grad_(v.requires_grad)
                      if (v.is_floating_point() and 'running' not in k) else v) for k, v in sd.items()}
            x = x * (1 + 1e-6 * torch.randn(x.shape, generator=g))
        nas = 'nas' in tag.split('.')
        gio.share_stem(sd, 'net.' if nas else '')
        if kw.get('use_sharing'):
            sd['alphas_up_nm'] = sd['alphas_dn_nm']
        tgt = torch.from_numpy(z[tag + '/target'])
        if nas:
            outs = R.nas_forward(sd, x, depth=kw['depth'], nodes=kw['meta_node_num'], supervision=kw.get('supervision', False))
        else:
            outs = R.derived_forward(sd, x, gio.geno_from_json(z[tag + '/genotype'], R.Genotype), depth=kw['depth'],
                                     supervision=kw.get('supervision', False))
        R.dice_ce_loss(outs[-1], tgt).backward()
        got = gio.alias_shared_stem({k: v.grad.detach().numpy() for k, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is not None},
                                    'net.' if nas else '')
        return {k: got[k].astype(np.float64) for k in names if k in got}

    base = run(0)
    top = max(float(np.abs(v).max()) for v in base.values())
    spread = {k: 0.0 for k in base}
    for t in range(trials):
        pert = run(100 + t)
        for k in base:
            spread[k] = max(spread[k], float(np.abs(pert[k] - base[k]).max()) / max(float(np.abs(base[k]).max()), 1e-3 * top))
    return spread


# whole-net fixtures whose gradient counts are recorded, not bounded (see test_whole_net)
OPEN_WHOLE_NET = ('nas.c8.d5', 'nas.c8.d4.share_dd')


@pytest.mark.parametrize('fixture,tag', NET_CASES)
def test_whole_net(fixture, tag):
    """nets: round-1 cases; nets2: the reference's default flags -- NAS(use_sharing=True, double_down_channel=True)
    (search/senas_search.py:118,148,26,45) and SenasModel(double_down_channel=True) (models/senas_model.py:80)."""
    from senas_amd.genotype import Genotype
    from senas_amd.loss import SegmentationLosses
    z = gio.load(fixture)
    net, kw = _build_net(z, tag)
    if tag.startswith('nas'):
        assert net.genotype() == gio.geno_from_json(z[tag + '/genotype'], Genotype)       # bit-exact
    x = torch.from_numpy(z[tag + '/x']).to(dev())
    tgt = torch.from_numpy(z[tag + '/target']).to(dev())
    outs = net(x)
    for i, o in enumerate(outs):
        close(o, z[tag + '/logits%d' % i], '%s logits%d' % (tag, i), rel=1e-3)
    loss = SegmentationLosses('dice_ce')(outs, tgt)
    assert abs(float(loss) - float(z[tag + '/loss'])) <= 1e-4 * abs(float(z[tag + '/loss']))
    loss.backward()
    got = grads_of(net)
    # Gradients of a ReLU + batch-norm net are only piecewise smooth, so the fp32 reference itself sits
    # |ref32 - ref64| away from the exact (fp64) gradient.  The HIP path is compared against the fp64
    # reference and must be within 2e-4 of the tensor scale, or 10x the reference's own fp32 error.
    full32, full64 = gio.sub(z, tag + '/gradfull/'), gio.sub(z, tag + '/gradfull64/')
    top = max(float(np.abs(e).max()) for e in full64.values())
    spread = _stored_gradient_spread(fixture, tag)           # (tests/golden/nets_spread.npz, made by make_spread.py with the oracle)
    escaped = []
    for k, e64 in full64.items():
        scale = max(float(np.abs(e64).max()), 1e-3 * top)
        ref_err = float(np.abs(full32[k] - e64).max()) / scale
        gpu_err = float(np.abs(got[k] - e64).max()) / scale
        bound = max(2e-4, 10 * ref_err, 4 * spread.get(k, 0.0))
        assert gpu_err <= bound, '%s grad %s: gpu %.2e vs fp64, reference fp32 %.2e, oracle spread under 1e-6 noise %.2e' % (
            tag, k, gpu_err, ref_err, spread.get(k, 0.0))
        vs32 = float(np.abs(got[k] - full32[k]).max()) / scale
        if gpu_err > 1e-3 and vs32 > 2e-4:
            escaped.append((k, gpu_err, ref_err, vs32))
    # a tensor is held to north_star's 1e-3 against the fp64 reference, or to 2e-4 against the reference's own fp32 run
    # (the parity target proper: ill-conditioned c = 8 fixtures sit 4e-3 from fp64 in BOTH implementations); only the
    # rest passes on the strength of the reference's fp32-vs-fp64 spread -- bounded in number for every fixture but the
    # two OPEN ones below (the c = 32 fixtures of test_full_width_net_every_gradient have no escape at all)
    print('%s: %d of %d full gradients used the conditioning escape: %s' % (tag, len(escaped), len(full64), escaped[:4]))
    worst_full = max((float(np.abs(got[k] - e64).max()) / max(float(np.abs(e64).max()), 1e-3 * top), k) for k, e64 in full64.items())
    # OPEN fixtures: one ReLU flip near the loss moves EVERY tensor of these two by a few 1e-3 -- measured in round 3 when the
    # fused architecture tables changed the mixing weights by 1e-7 (tools/diag_archmix.py; the oracle moves as far under 1e-6
    # noise: nets_spread.npz) -- so their counts are recorded, and their per-tensor bound above carries the measured
    # conditioning.  Everywhere else the counts are pinned at what a run shows (0 escapes, <= 10 norm outliers).  The
    # depth-5 statements that CAN fail are tests/test_gpu_depth5.py (c = 32: every cell and every piece between the cells,
    # teacher-forced from the oracle's pass, 5e-5); no whole-net depth-5 gradient fixture exists.
    open_fixture = tag in OPEN_WHOLE_NET
    escape_allowed = None if open_fixture else 2
    if not open_fixture:
        assert len(escaped) <= escape_allowed, '%s: %d gradient tensors used the conditioning escape (%d allowed): %s' % (
            tag, len(escaped), escape_allowed, escaped[:4])
    margins = {'full_gradients': len(full64), 'used_conditioning_escape': len(escaped), 'escape_allowed': escape_allowed,
               'worst_full_gradient_vs_fp64': worst_full[0], 'worst_full_gradient': worst_full[1],
               'escape_count_is_recorded_not_bounded': open_fixture,
               'escaped': [{'tensor': k, 'gpu_vs_fp64': a, 'reference_fp32_vs_fp64': b, 'gpu_vs_reference_fp32': c} for k, a, b, c in escaped],
               'worst_oracle_spread': max(spread.values()) if spread else None,
               'bound': 'max(2e-4, 10 x |ref32 - ref64|, 4 x the oracle\'s own spread under 1e-6 perturbations) of the tensor scale per tensor; '
                        'escape = beyond 1e-3 of fp64 AND beyond 2e-4 of the reference fp32 run'}
    d32, d64 = gio.digest(z, tag + '/grad/'), gio.digest(z, tag + '/grad64/')
    assert set(d32) == set(got)
    top = max(v[1] for v in d64.values())
    outliers = []
    norm_bound = 1e-1 if open_fixture else 2e-2
    for k, (_, l2_64) in d64.items():
        scale = max(l2_64, 1e-3 * top)
        ref_err = abs(d32[k][1] - l2_64) / scale
        gpu_err = abs(float(np.sqrt((got[k].astype(np.float64) ** 2).sum())) - l2_64) / scale
        # (a lost or doubled contribution is O(1); norms of the open fixtures move with the same ReLU flips as their tensors)
        assert gpu_err <= max(norm_bound, 3 * ref_err), '%s |grad| %s: gpu %.2e, reference fp32 %.2e' % (tag, k, gpu_err, ref_err)
        if gpu_err > max(5e-4, 10 * ref_err):
            outliers.append((gpu_err, ref_err, k))
    # batch-norm scale gradients are cancellation residues (sum(ds*z) - mean*sum(ds)); a few of the tensors amplify
    # summation-order noise past the tight bound -- at most 2 % may (none beyond 2e-2, asserted above), except in the open fixtures
    outliers_allowed = None if open_fixture else max(2, len(d64) // 50)
    if not open_fixture:
        assert len(outliers) <= outliers_allowed, '%s: %d gradient norms beyond the tight bound (%d allowed): %s' % (
            tag, len(outliers), outliers_allowed, sorted(outliers)[-3:])
    margins.update({'gradient_norms': len(d64), 'norm_outliers': len(outliers), 'norm_outliers_allowed': outliers_allowed,
                    'norm_bound': norm_bound,
                    'worst_norm_outliers': [{'tensor': k, 'gpu': a, 'reference_fp32': b} for a, b, k in sorted(outliers)[-3:]]})
    from conftest import record_margin
    record_margin('test_whole_net[%s-%s]' % (fixture, tag), **margins)
    gio.check_digest(gio.digest(z, tag + '/bn1/'), {k: v.cpu().numpy() for k, v in net.state_dict().items()},
                     rtol=1e-3, atol_scale=1e-4, what=tag + ' bn')
    # arg-max masks: identical wherever the reference's own top-2 margin is above fp32 noise
    ref = torch.from_numpy(z[tag + '/logits%d' % (len(outs) - 1)])
    top2 = ref.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref.abs().max()
    assert bool((outs[-1].argmax(1).cpu() == ref.argmax(1))[sure].all())
    if tag + '/logits_eval' in z.files:
        net.eval()
        with torch.no_grad():
            close(net(x)[-1], z[tag + '/logits_eval'], tag + ' eval', rel=1e-3)


FULL_CASES = [(f, t) for f in ('nets_full', 'nets3') for t in gio.index(f)]


@pytest.mark.parametrize('fixture,tag', FULL_CASES)
def test_full_width_net_every_gradient(fixture, tag):
    """nets_full: c = 32 nets at the reference's initialisation scale (weights_init), 2x1x64x64; nets3 (tags ending in msup):
    deep supervision -- ``supervision=True`` nets under MultiSegmentationLosses (utils/loss/loss.py:30-43,
    search/senas_search.py:104-107), where the shared head receives one gradient per output.  EVERY parameter gradient against
    the reference's fp64 gradients.  Bound per tensor: north_star's 1e-3, or 4x the spread the REFERENCE's own fp32
    gradient of that tensor shows under 1e-6 relative perturbations of input and weights (stored with the fixture;
    make_golden.py gen_nets2: every seed of this net family moves its worst tensor by 2e-3 .. 6e-2 under such
    rounding-level perturbations, so no fp32 implementation can be held to 1e-3 on all of them).  Well-conditioned
    tensors -- the majority -- are held to 1e-3 with no escape; the count of the others is printed and bounded."""
    from senas_amd.loss import MultiSegmentationLosses, SegmentationLosses
    z = gio.load(fixture)
    net, kw = _build_net(z, tag)
    x = torch.from_numpy(z[tag + '/x']).to(dev())
    tgt = torch.from_numpy(z[tag + '/target']).to(dev())
    outs = net(x)
    close(outs[-1], z[tag + '/logits'], tag + ' logits', rel=2e-4)
    crit = MultiSegmentationLosses('dice_ce', kw['depth']) if tag.endswith('msup') else SegmentationLosses('dice_ce')
    loss = crit(outs, tgt)
    assert abs(float(loss) - float(z[tag + '/loss64'])) <= 1e-5 * abs(float(z[tag + '/loss64']))
    loss.backward()
    got = grads_of(net)
    exp = gio.unpack(z, tag + '/grad64/')
    top = float(z[tag + '/grad_top'])
    assert set(exp) == set(got), set(exp) ^ set(got)
    errs = {k: float(np.abs(got[k] - e).max()) / max(float(np.abs(e).max()), 1e-3 * top) for k, e in exp.items()}
    spread = dict(zip(json.loads(str(z[tag + '/spread_names'])), z[tag + '/spread']))
    loose = [k for k in errs if errs[k] > 1e-3]
    worst = max(errs, key=errs.get)
    print('%s: %d gradients, worst %s %.2e (reference spread there %.2e); %d beyond 1e-3, %d tensors have a reference spread above 2.5e-4'
          % (tag, len(errs), worst, errs[worst], spread[worst], len(loose), sum(1 for v in spread.values() if v > 2.5e-4)))
    from conftest import record_margin
    record_margin('test_full_width_net_every_gradient[%s-%s]' % (fixture, tag), gradients=len(errs), worst_tensor=worst, worst_vs_fp64=errs[worst],
                  reference_spread_at_worst=float(spread[worst]), beyond_1e3=len(loose), beyond_1e3_allowed=max(3, len(errs) // 8),
                  beyond_1e3_detail=[{'tensor': k, 'gpu_vs_fp64': errs[k], 'reference_spread': float(spread[k])} for k in loose],
                  bound='max(1e-3, 4 x the reference\'s own spread under 1e-6 perturbations) per tensor')
    for k, v in errs.items():
        assert v <= max(1e-3, 4 * spread[k]), (k, v, spread[k])
    assert len(loose) <= max(3, len(errs) // 8), loose


def test_search_step_trajectory():
    """Two full search steps on the GPU vs. the reference trajectory (arch Adam step + SGD/clip step)."""
    from senas_amd.genotype import Genotype
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS, Architecture
    z = gio.load('search_step')
    net = NAS(input_c=1, c=8, num_classes=2, depth=5, meta_node_num=3, use_sharing=False, double_down_channel=False,
              multi_gpus=False, device=dev())
    load_into(net, gio.unpack(z, 'sd0/')).train()
    crit = SegmentationLosses('dice_ce')
    opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
    opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    arch = Architecture(net, opt_a, crit)
    xs, ys = torch.from_numpy(z['x']).to(dev()), torch.from_numpy(z['y']).to(dev())
    for step in range(2):
        arch.step(xs[2 * step], ys[2 * step])
        opt_w.zero_grad()
        loss = crit(net(xs[2 * step + 1]), ys[2 * step + 1])
        loss.backward()
        nn.utils.clip_grad_norm_(net.parameters(), 5)
        opt_w.step()
        assert abs(float(loss) - float(z['loss%d' % step])) <= 2e-4 * abs(float(z['loss%d' % step]))
    got = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    for k, e in gio.sub(z, 'sd2full/').items():
        close(got[k], e, 'after-step ' + k, rel=1e-3)
    gio.check_digest(gio.digest(z, 'sd2/'), got, rtol=1e-3, atol_scale=1e-4, what='after-step')
    assert net.genotype() == gio.geno_from_json(z['genotype'], Genotype)


@pytest.mark.parametrize('graphed', [False, True])
def test_search_step_driver_trajectory(graphed):
    """The same two search steps through ``SearchStep`` (senas_amd/step.py): architecture pass with frozen weights
    (no weight-gradient kernels), weight pass, fused clip + SGD -- eager and as two replayed HIP graphs.  The weights
    and architecture tensors must land where the reference's loop (search_arc.py:252-299) puts them."""
    from senas_amd.genotype import Genotype
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep
    z = gio.load('search_step')
    net = NAS(input_c=1, c=8, num_classes=2, depth=5, meta_node_num=3, use_sharing=False, double_down_channel=False,
              multi_gpus=False, device=dev())
    load_into(net, gio.unpack(z, 'sd0/')).train()
    crit = SegmentationLosses('dice_ce')
    opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
    opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    xs, ys = torch.from_numpy(z['x']).to(dev()), torch.from_numpy(z['y']).to(dev())
    step = SearchStep(net, crit, opt_w, opt_a, xs[0].clone(), ys[0].clone(), grad_clip=5.0, use_graph=graphed)
    assert step.graphed == graphed
    for k in range(2):
        loss = step(xs[2 * k + 1], ys[2 * k + 1], xs[2 * k], ys[2 * k])
        assert abs(float(loss) - float(z['loss%d' % k])) <= 2e-4 * abs(float(z['loss%d' % k]))
        if k == 0:
            # the architecture pass left no weight gradient behind: every weight gradient present now is the weight pass's
            assert all(p.grad is not None for p in net.arch_parameters())
    got = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    # (the capture warm-up passes leave the batch-norm running statistics where they were: buffers are compared too)
    for k, e in gio.sub(z, 'sd2full/').items():
        close(got[k], e, 'after-step ' + k, rel=1e-3)
    assert net.genotype() == gio.geno_from_json(z['genotype'], Genotype)
    step.close()



# ------------------------------------------------------------------ full-size checks against the oracle
def _randomize(net, seed):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=gen) * (1.5 / max(1, p[0].numel()) ** 0.5))
            elif name.endswith('weight'):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=gen))


def test_derived_full_width_vs_oracle():
    """README genotype at the real width (c=32), 2x1x128x128, forward + backward vs the CPU oracle."""
    from oracle import senas_ref as R
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4)
    _randomize(net, 3)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    gio.share_stem(sd)
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 128, 128, generator=gen)
    y = torch.randint(0, 2, (2, 128, 128), generator=gen)
    ref = R.derived_forward(sd, x, R.Genotype(*senas_node_4))[-1]
    ref_loss = R.dice_ce_loss(ref, y)
    ref_loss.backward()
    net = net.to(dev()).train()
    out = net(x.to(dev()))
    loss = SegmentationLosses('dice_ce')(out, y.to(dev()))
    loss.backward()
    close(out[-1], ref.detach().numpy(), 'logits', rel=1e-3)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
    got = grads_of(net)
    for k in ('stem0.0.weight', 'blocks.0.1._ops.0.0.weight', 'blocks.4.0._ops.7.0.weight', 'head_block.0.segmentation_head.1.weight',
              'blocks.0.3.post_process.norm.weight', 'blocks.0.2._ops.3.0.weight'):
        # ~1e7 ReLU inputs: some sit within fp32 noise of zero, so single mask flips are certain at this
        # size (DESIGN.md "gradient parity"); L2 distance keeps them in proportion.
        e = sd[k].grad.numpy().astype(np.float64)
        err = float(np.sqrt(((got[k] - e) ** 2).sum()) / np.sqrt((e ** 2).sum()))
        assert err <= 1e-2, 'grad %s: L2 rel err %.2e' % (k, err)


@pytest.mark.parametrize('case', [(4, 8, 8, 64, 64, 5, 3, 2), (4, 8, 16, 32, 48, 5, 3, 2), (4, 32, 32, 64, 64, 5, 3, 2), (2, 32, 32, 16, 16, 5, 3, 3),
                                  (8, 32, 32, 128, 128, 5, 2, 3), (2, 32, 32, 40, 72, 3, 1, 2), (3, 16, 32, 24, 24, 5, 2, 1), (2, 64, 32, 32, 32, 3, 1, 1)])
def test_conv_pair_launches_vs_single_calls(case):
    """functional.conv2d_pair -- forward, both data gradients and both weight gradients of two convolutions of one tensor as
    one launch each (senas_conv2d_fwd_pair / _bwd_data_pair / _bwd_weight_pair; problem 2 on blockIdx.z / .y) -- against the
    two single calls: every problem of a pair launch is computed exactly as its own launch would (same kernel, same tile
    order, fixed-order second stages), so outputs and all four gradients are bit-identical; shapes off the pair
    kernels take the two single calls inside the same entry points."""
    from senas_amd import functional as F
    n, ci, co, h, w, k, da, db = case
    gen = torch.Generator().manual_seed(sum(case))
    cl = torch.channels_last
    x0 = torch.randn(n, ci, h, w, generator=gen)
    wa0, wb0 = (torch.randn(co, ci, k, k, generator=gen) * 0.1 for _ in range(2))
    ga, gb = (torch.randn(n, co, h, w, generator=gen).to(dev()).contiguous(memory_format=cl) for _ in range(2))
    res = []
    for paired in (True, False):
        x = x0.to(dev()).contiguous(memory_format=cl).requires_grad_(True)
        wa, wb = wa0.to(dev()).requires_grad_(True), wb0.to(dev()).requires_grad_(True)
        xa, xb = F.fan_out(x, 2)
        if paired:
            (ya, sa), (yb, sb) = F.conv2d_pair(xa, xb, wa, wb, 1, da * (k // 2), da, db * (k // 2), db, want_stats=True)
        else:
            ya, sa = F.conv2d(xa, wa, 1, da * (k // 2), da, want_stats=True)
            yb, sb = F.conv2d(xb, wb, 1, db * (k // 2), db, want_stats=True)
        torch.autograd.backward([ya, yb], [ga, gb])
        res.append([t.detach().cpu() for t in (ya, sa, yb, sb, x.grad, wa.grad, wb.grad)])
    for name, a, b in zip(('ya', 'stats a', 'yb', 'stats b', 'dx', 'dwa', 'dwb'), *res):
        if name.startswith('stats'):                 # fp64 sums folded by atomics: the order of the blocks is not fixed
            assert torch.allclose(a, b, rtol=1e-11, atol=1e-9), name
        elif name.startswith('dw') and ci % 32 != 0 and ci != 8:
            # off the two-stage kernels (split-K accumulated with atomics): the last bits move from run to run
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), name
        else:
            assert torch.equal(a, b), name
    ref = torch.nn.functional.conv2d(x0.double(), wb0.double(), padding=db * (k // 2), dilation=db)
    close(res[0][2], ref.numpy(), 'yb vs torch', rel=5e-5)


def test_derived_cell_pair_launches_change_nothing():
    """BuildCell runs two same-state dense candidates per launch (senas_model.BuildCell.paired: one forward launch, one for
    the two data gradients).  A pair launch computes each problem exactly as its single launch would: logits and EVERY
    gradient of the README genotype (c=32, depth 5, 2x1x64x64) are the same with the pairing on and off (logits bit for bit),
    and the pairing is really taken (12 cells: one pair per down cell, two per up cell)."""
    import copy
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import BuildCell, SenasModel
    base = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4)
    _randomize(base, 5)
    cells = [m for m in base.modules() if isinstance(m, BuildCell)]
    assert sum(len(v) for m in cells for v in m._pairs().values()) == 4 * 1 + 8 * 2
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(2, 1, 64, 64, generator=gen).to(dev())
    y = torch.randint(0, 2, (2, 64, 64), generator=gen).to(dev())
    res = []
    was = BuildCell.paired
    try:
        for paired in (True, False):
            BuildCell.paired = paired
            net = copy.deepcopy(base).to(dev()).train()
            out = net(x)
            SegmentationLosses('dice_ce')(out, y).backward()
            res.append((out[-1].detach().cpu(), grads_of(net)))
    finally:
        BuildCell.paired = was
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys()
    for k in res[0][1]:        # (some small-map weight gradients end in atomics: their last bit moves from run to run)
        a, b = res[0][1][k], res[1][1][k]
        assert float(np.abs(a - b).max()) <= 1e-6 * float(np.abs(b).max()) + 1e-12, k


def test_supernet_full_width_vs_oracle():
    """The supernet at its real width (c=32: 32 -> 8 candidates, 8 -> 8 inner edges, 1x1 adapters, 24 -> 32 cell outputs),
    depth 3, 2x1x64x64: forward + backward + architecture gradients vs the CPU oracle."""
    from oracle import senas_ref as R
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    kw = dict(input_c=1, c=32, num_classes=2, depth=3, meta_node_num=3)
    net = NAS(use_sharing=False, double_down_channel=False, multi_gpus=False, device=torch.device('cpu'), **kw)
    _randomize(net, 9)
    with torch.no_grad():
        for p in net.arch_parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.5)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    gio.share_stem(sd, 'net.')
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(2, 1, 64, 64, generator=gen)
    y = torch.randint(0, 2, (2, 64, 64), generator=gen)
    ref = R.nas_forward(sd, x, depth=3, nodes=3)[-1]
    ref_loss = R.dice_ce_loss(ref, y)
    ref_loss.backward()
    net = net.to(dev()).train()
    out = net(x.to(dev()))
    loss = SegmentationLosses('dice_ce')(out, y.to(dev()))
    loss.backward()
    close(out[-1], ref.detach().numpy(), 'logits', rel=1e-3)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
    got = grads_of(net)
    for k in ('alphas_dn', 'alphas_up', 'alphas_dn_nm', 'alphas_up_nm', 'betas_dn', 'betas_up', 'net.stem0.0.weight'):
        e = sd[k].grad.numpy().astype(np.float64)
        err = float(np.sqrt(((got[k] - e) ** 2).sum()) / np.sqrt((e ** 2).sum()))
        assert err <= 1e-2, 'grad %s: L2 rel err %.2e' % (k, err)
    ref_geno = R.derive_genotype(sd, depth=3, nodes=3)
    got_geno = net.genotype()
    assert (list(got_geno.down), list(got_geno.up), list(got_geno.gamma)) == (list(ref_geno.down), list(ref_geno.up), list(ref_geno.gamma))


def test_linearity_and_shapes_at_baseline_size():
    """Size-independent properties at BASELINE's full size (8x1x256x256, c=32): the convolution kernels
    are linear in x (conv(a*x1 + x2) == a*conv(x1) + conv(x2)) and a train-mode step leaves finite
    gradients for every parameter."""
    from senas_amd import functional as F
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    g = torch.Generator(device='cuda').manual_seed(7)
    x1 = torch.randn(8, 32, 128, 128, device=dev(), generator=g).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(8, 32, 128, 128, device=dev(), generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(32, 32, 5, 5, device=dev(), generator=g) * 0.05
    for kw in (dict(stride=1, pad=6, dil=3), dict(stride=2, pad=4, dil=2), dict(stride=2, pad=6, dil=3, transposed=True, out_pad=1)):
        lhs = F.conv2d(2.5 * x1 + x2, w, **kw)[0]
        rhs = 2.5 * F.conv2d(x1, w, **kw)[0] + F.conv2d(x2, w, **kw)[0]
        assert float((lhs - rhs).abs().max()) <= 2e-4 * float(rhs.abs().max())
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4).to(dev()).train()
    x = torch.randn(8, 1, 256, 256, device=dev(), generator=g)
    y = torch.randint(0, 2, (8, 256, 256), device=dev(), generator=g)
    out = net(x)
    assert out[-1].shape == (8, 2, 256, 256)
    SegmentationLosses('dice_ce')(out, y).backward()
    for k, p in net.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k


# ------------------------------------------------------------------ convolution kernels at sizes that reach the MFMA / LDS paths
_CONV_CASES = [
    # (n, ci, co, h, w, k, stride, dil, transposed, relu)
    (2, 32, 32, 40, 64, 5, 1, 3, False, False),     # LDS window kernel, ragged tile rows (40 = 5 x 8)
    (2, 32, 32, 36, 70, 5, 1, 2, False, True),      # ragged both ways + ReLU on load
    (1, 128, 32, 16, 32, 3, 1, 1, False, True),     # 8 channel passes (ShrinkBlock shape)
    (2, 24, 32, 24, 40, 3, 1, 1, False, False),     # c_in = 24: not a 16-channel multiple -> direct-global MFMA
    (2, 32, 8, 32, 32, 5, 1, 3, False, False),      # supernet width (8 output channels)
    (2, 32, 32, 32, 48, 5, 2, 3, False, False),     # stride-2 conv: dgrad runs the 4-phase transposed gather
    (2, 64, 32, 16, 16, 5, 2, 2, False, False),     # ... 32 blocks: 8 waves split the K steps of a sub-tile; even dilation (3 empty phases)
    (4, 32, 64, 48, 48, 3, 2, 1, False, False),     # ... 288 blocks: 4 waves per sub-tile
    (2, 32, 32, 16, 24, 5, 2, 2, True, False),      # ConvTranspose2d (UP ops), even dilation: 3 empty phases
    (2, 32, 32, 16, 24, 3, 2, 1, True, True),
    (3, 1, 32, 40, 40, 7, 1, 1, False, False),      # stem: small-c_in weight gradient
    (2, 32, 2, 24, 24, 3, 1, 1, False, True),       # segmentation head (2 classes): straight-line 3x3 thin-N forward / thin-K data gradient
    (2, 32, 4, 30, 22, 3, 1, 1, False, True),       # ... 4 classes
    (3, 32, 2, 70, 50, 3, 1, 1, False, True),       # ... several pixel passes per thread, ragged last block
    (2, 2, 16, 12, 12, 3, 1, 1, True, False),       # ... the transposed-gather instantiations (ConvTranspose2d 3x3 stride 1 from 2 channels)
    (8, 32, 2, 256, 256, 3, 1, 1, False, True),     # ... at the BASELINE size (2 048 partial rows in the weight gradient)
    (2, 8, 8, 20, 20, 5, 1, 3, False, False),       # inner supernet edge
    (2, 3, 16, 20, 28, 3, 2, 1, False, False),      # thin-K gather: RGB stem, stride 2
    (2, 32, 4, 18, 30, 1, 1, 1, False, False),      # thin-N: 1x1 head, 4 classes
    (2, 16, 3, 24, 24, 3, 1, 1, False, True),       # thin-N with 3 classes (padded to 4), thin-K dgrad with c_in = 3 + ReLU mask
    (2, 2, 16, 12, 12, 3, 2, 1, True, False),       # thin-K transposed gather forward (ConvTranspose2d from 2 channels)
    (9, 1, 32, 64, 64, 7, 1, 1, False, False),      # stem at a size that takes 8 pixel passes per thread
    (2, 32, 32, 24, 64, 3, 1, 1, False, True),      # LDS weight gradient, 9 units: 8 row lanes x 9 accumulators
    (2, 32, 32, 16, 32, 1, 1, 1, False, False),     # 1x1: one unit, rows dealt to the 8 waves
    (2, 64, 32, 16, 40, 3, 1, 1, False, False),     # 18 units: 4 unit groups x 2 row lanes
    (2, 64, 32, 16, 32, 1, 1, 1, False, True),      # 2 units
    (1, 96, 32, 16, 32, 3, 1, 1, False, False),     # 27 units
    (1, 96, 16, 8, 32, 1, 1, 1, False, False),      # 3 units, 16 output channels (padded columns)
    (1, 128, 32, 8, 32, 1, 1, 1, False, True),      # 4 units
    (8, 32, 32, 64, 96, 3, 1, 1, False, True),      # LDS window kernel with the taps dealt to 2 wave groups
    (8, 16, 32, 72, 128, 3, 1, 1, False, False),    # 4x32 tiles, one wave group (between the two regimes)
    (2, 32, 32, 32, 32, 3, 1, 1, False, True),      # single-MFMA-row blocks (few tiles): 32 wide
    (8, 32, 32, 16, 16, 5, 1, 3, False, False),     # 16-wide maps: one MFMA row = 2 image rows
    (8, 32, 32, 8, 8, 5, 1, 2, False, True),        # 8-wide maps: one MFMA row = 4 image rows
    (3, 32, 16, 12, 20, 5, 1, 2, False, False),     # 16-wide tiles, ragged in both directions
    (2, 32, 32, 10, 9, 3, 1, 1, False, False),      # 8-wide tiles, ragged
    (2, 16, 32, 8, 8, 1, 1, 1, False, False),       # 1x1 on an 8x8 map (one-wave blocks)
    (8, 128, 32, 16, 16, 3, 1, 1, False, True),     # ShrinkBlock shape on a 16x16 map (narrow LDS weight gradient, 36 units)
    (8, 128, 32, 8, 8, 3, 1, 1, False, False),      # ... and on 8x8
    (2, 32, 32, 8, 8, 1, 1, 1, False, True),        # narrow 1x1 weight gradient
    (2, 8, 8, 24, 24, 1, 1, 1, False, True),        # 8 -> 8 (supernet inner edge): thin-K gather both ways, c8 weight gradient
    (2, 8, 8, 16, 16, 3, 2, 1, False, False),       # ... stride 2
    (2, 8, 8, 12, 12, 3, 2, 1, True, False),        # ... transposed
    (2, 4, 8, 16, 16, 3, 1, 1, False, False),       # 4 input channels
    (2, 8, 5, 16, 16, 3, 1, 1, False, True),        # 5 output channels (zero-padded columns in the c8 weight gradient)
    (4, 32, 8, 16, 16, 5, 1, 2, False, False),      # full-width supernet candidate on a 16x16 map: narrow LDS kernels, 8 outputs
    (4, 32, 8, 16, 16, 1, 1, 1, False, True),       # ... its 1x1 adapter
    (4, 32, 8, 8, 8, 3, 1, 1, False, False),        # ... on 8x8
    (2, 32, 8, 32, 32, 5, 2, 3, False, False),      # stride-2 candidate with 8 outputs (thin-N gather forward)
    (2, 32, 8, 16, 16, 5, 2, 2, True, False),       # transposed candidate with 8 outputs
    (2, 32, 32, 32, 32, 3, 2, 1, False, True),      # 3x3 stride 2 (se_conv_3 of a down cell): strided LDS weight gradient
    (4, 32, 32, 64, 64, 5, 2, 3, False, False),     # 5x5 d3 stride 2, several tiles per block
    (4, 32, 32, 32, 32, 5, 2, 3, True, False),      # ConvTranspose2d 5x5 d3 (dy on the fine grid)
    (2, 32, 32, 8, 8, 5, 2, 2, True, False),        # ... on an 8x8 input (8-wide strided tiles)
    (2, 3, 32, 40, 72, 7, 1, 1, False, False),      # RGB stem on the MFMA stem kernel (K = 147), ragged tiles
    (2, 1, 64, 16, 32, 7, 1, 1, False, True),       # stem with 64 outputs (two channel tiles), ReLU on load
    (2, 2, 32, 24, 40, 3, 1, 2, False, False),      # 2 input channels, dilated 3x3 (K = 18)
    (4, 8, 16, 128, 128, 5, 1, 3, False, False),    # big-map inner edge: 4-columns-per-thread thin-K gather, forward and (flipped) data gradient
    (4, 8, 8, 128, 132, 3, 1, 1, False, True),      # ... 3x3 with ReLU on load (the masked data gradient stays on the one-pixel form)
    (2, 8, 16, 24, 24, 5, 1, 2, False, False),      # stacked search candidates: two 8 -> 8 inner edges (c8 weight gradient, 16 columns)
    (2, 8, 12, 16, 16, 3, 1, 1, False, True),       # ... 12 columns (padded)
    (2, 32, 24, 32, 32, 5, 1, 3, False, False),     # three 32 -> 8 candidates stacked: 24 outputs; data gradient from 24 channels
    (2, 32, 24, 32, 32, 5, 2, 2, False, False),     # ... stride 2 (down cell)
    (2, 32, 24, 16, 16, 3, 2, 1, True, False),      # ... transposed (up cell)
    (2, 32, 24, 24, 24, 1, 1, 1, False, False),     # ... the stacked 1x1 adapters
    # the 8-channel inner edges on 16 x 16 x 4 MFMA tiles (conv_c8.hip): forward 8 -> 8 / 16, data gradient 8 / 16 -> 8, weight gradient
    (2, 8, 8, 20, 36, 5, 1, 3, False, False),       # ragged in both directions, several tiles
    (3, 8, 8, 8, 8, 5, 1, 2, False, False),         # one partly empty tile per image
    (4, 8, 16, 64, 64, 5, 1, 2, False, False),      # two stacked edges, 64 tiles (the weight gradient's blocks loop over none)
    (2, 8, 16, 13, 37, 5, 1, 3, False, False),      # stacked, ragged
    (4, 8, 16, 256, 256, 5, 1, 3, False, False),    # head cell size: 1 024 tiles, the weight gradient's 256 blocks take 4 each
    (2, 8, 8, 16, 16, 5, 1, 1, False, False),       # dilation 1
    # transposed gathers at stride 2 with the window in LDS (conv_t2.hip): ConvTranspose2d forward, stride-2 Conv2d data gradient
    (2, 32, 32, 9, 13, 5, 2, 3, True, False),       # ragged position tiles, dilation 3 (9 + 6 + 6 + 4 taps in the four phases)
    (2, 16, 32, 7, 20, 3, 2, 1, True, False),       # one 16-channel pass, 3x3
    (2, 64, 32, 12, 12, 5, 2, 2, True, False),      # four passes, dilation 2: three of the four output phases are zero
    (2, 32, 40, 10, 10, 3, 2, 1, True, False),      # two output-channel tiles
    (2, 32, 32, 24, 40, 5, 2, 1, False, False),     # stride-2 Conv2d 5x5 dilation 1: its data gradient
    # BASELINE sizes, single layers against the oracle (the whole nets at this size are property-checked only): the tile
    # seams of a 256 x 256 map, 8 images
    (8, 32, 32, 256, 256, 5, 1, 3, False, False),   # dil_3_conv_5 of the derived head cell
    (8, 32, 32, 256, 256, 5, 1, 2, False, True),    # dil_2_conv_5, ReLU on load
    (8, 128, 32, 256, 256, 3, 1, 1, False, False),  # post_process 128 -> 32
    (4, 32, 32, 128, 128, 5, 2, 3, True, False),    # ConvTranspose2d 5x5 d3 128 -> 256 (supernet head cell, stacked width)
]


@pytest.mark.parametrize('case', _CONV_CASES, ids=lambda c: 'n%d_%dto%d_%dx%d_k%ds%dd%d%s%s' % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], '_T' if c[8] else '', '_relu' if c[9] else ''))
def test_conv_kernels_vs_oracle(case):
    """Forward, data gradient, weight gradient and producer-side statistics of one convolution launch
    against the oracle's leaf arithmetic (torch CPU conv / conv_transpose)."""
    from oracle import senas_ref as R
    from senas_amd import functional as F
    n, ci, co, h, w, k, stride, dil, tr, relu = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn((ci, co, k, k) if tr else (co, ci, k, k), generator=g) * (1.0 / (ci * k * k) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = R.conv(torch.relu(xr) if relu else xr, wr, k, stride, dil, tr)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xg, wg = x.to(dev()).requires_grad_(True), wt.to(dev()).requires_grad_(True)
    pad = (k // 2) * dil
    y, st = F.conv2d(xg, wg, stride=stride, pad=pad, dil=dil, transposed=tr, out_pad=stride - 1 if tr else 0, in_relu=relu,
                     want_stats=True)
    close(y, ref.detach().numpy(), 'y', rel=2e-5)
    y.backward(gy.to(dev()))
    close(xg.grad, xr.grad.numpy(), 'dx', rel=2e-5)
    close(wg.grad, wr.grad.numpy(), 'dw', rel=5e-5)
    ref64 = ref.detach().double()
    exp = torch.stack([ref64.sum((2, 3)), (ref64 ** 2).sum((2, 3))], -1)
    close(st.float(), exp.float().numpy(), 'stats', rel=2e-5)


def test_weight_packer_and_graph_step_match_eager():
    """The cached weight images (one batched repack per step) and the HIP-graph replay must give the same
    step as plain eager calls that repack per launch: same loss trajectory, same weights after 3 steps."""
    import copy
    from senas_amd import functional as F
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.step import TrainStep
    torch.manual_seed(0)
    base = SenasModel(2, 1, c=16, depth=4, genotype=senas_node_4)
    _randomize(base, 5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 1, 64, 64, generator=g).to(dev())
    y = torch.randint(0, 2, (2, 64, 64), generator=g).to(dev())
    crit = SegmentationLosses('dice_ce')
    results = []
    for mode in ('eager', 'graph'):
        net = copy.deepcopy(base).to(dev()).train()
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        losses = []
        if mode == 'eager':
            assert not F.PACKED
            for _ in range(3):
                opt.zero_grad()
                loss = crit(net(x), y)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
                opt.step()
                losses.append(float(loss))
        else:
            step = TrainStep(net, crit, opt, x, y, use_graph=True)
            assert step.graphed and F.PACKED
            for _ in range(3):
                losses.append(float(step()))
            step.close()
        # (buffers included: the capture warm-up passes must leave the running statistics where they were)
        results.append((losses, {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}))
    (l0, w0), (l1, w1) = results
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-5 * abs(a), (l0, l1)
    for k in w0:
        close(w1[k], w0[k], 'weights after 3 steps ' + k, rel=2e-4)


@pytest.mark.parametrize('cfg', [dict(momentum=0.9, weight_decay=5e-4), dict(momentum=0.9, weight_decay=3e-4, nesterov=True),
                                 dict(momentum=0.0, weight_decay=0.0), dict(momentum=0.5, dampening=0.1, weight_decay=1e-2)],
                         ids=['momentum_wd', 'nesterov', 'plain', 'dampening'])
def test_fused_clip_sgd_matches_torch(cfg):
    """senas_sgd_clip_step == nn.utils.clip_grad_norm_ + torch.optim.SGD.step (train_model.py:284-289), 3 steps,
    one tensor larger than a block, one parameter that never gets a gradient, clipping active and inactive."""
    from senas_amd.optim import FusedClipSGD
    g = torch.Generator().manual_seed(11)
    shapes = [(32, 32, 5, 5), (7,), (2, 32, 3, 3), (128,), (1,), (32, 128, 3, 3)]
    base = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * sc for s in shapes] for sc in (3.0, 0.01, 1.0)]     # norm >> 5, << 5, ~
    ref = [torch.nn.Parameter(b.clone().to(dev())) for b in base] + [torch.nn.Parameter(torch.ones(3, device=dev()))]
    got = [torch.nn.Parameter(b.clone().to(dev())) for b in base] + [torch.nn.Parameter(torch.ones(3, device=dev()))]
    opt_r = torch.optim.SGD(ref, lr=0.05, **cfg)
    opt_g = torch.optim.SGD(got, lr=0.05, **cfg)
    for p in got[:-1]:
        p.grad = torch.zeros_like(p)                  # static gradient storage, as under graph replay
    fused = FusedClipSGD(opt_g, 5.0)
    for step in range(3):
        for p, q, gr in zip(ref, got, grads[step]):
            p.grad = gr.clone().to(dev())
            q.grad.copy_(gr)
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        opt_r.step()
        n_got = fused.step()
        close(n_got, n_ref.reshape(1).cpu().numpy(), 'total_norm', rel=1e-5)
        for k, (p, q) in enumerate(zip(ref, got)):
            close(q.detach(), p.detach().cpu().numpy(), 'param %d step %d' % (k, step), rel=2e-6)
            if p.grad is not None:
                close(q.grad, p.grad.cpu().numpy(), 'clipped grad %d step %d' % (k, step), rel=2e-6)
    assert set(opt_g.state_dict()['state'].keys()) == set(opt_r.state_dict()['state'].keys())


@pytest.mark.parametrize('tag', gio.index('multi_loss'))
def test_multi_loss_kernels(tag):
    """MultiSegmentationLosses (utils/loss/loss.py:30-43) over senas_dice_ce_fwd/_bwd against the reference's own value
    and per-output gradients (weight factors; fewer outputs than ``depth``)."""
    from senas_amd.loss import MultiSegmentationLosses
    z = gio.load('multi_loss')
    meta = json.loads(str(z[tag + '/meta']))
    logits = [torch.from_numpy(z[tag + '/logits%d' % i]).to(dev()).requires_grad_(True) for i in range(meta['outputs'])]
    tgt = torch.from_numpy(z[tag + '/target']).to(dev())
    loss = MultiSegmentationLosses('dice_ce', meta['depth'], meta['factors'])(logits, tgt)
    np.testing.assert_allclose(loss.item(), float(z[tag + '/loss']), rtol=3e-6)
    loss.backward()
    for i, l in enumerate(logits):
        np.testing.assert_allclose(l.grad.cpu().numpy(), z[tag + '/dlogits%d' % i], rtol=3e-5, atol=3e-9)
    with pytest.raises(ValueError):
        MultiSegmentationLosses('dice_ce', 3, [1.0, 2.0])


@pytest.mark.parametrize('tag', gio.index('loss_metric'))
def test_loss_and_metric_kernels(tag):
    """senas_dice_ce_fwd/_bwd and senas_seg_metric_update against the reference's own outputs (golden vectors
    from utils/loss/loss.py and utils/metrics.py) and against the oracle on a non-unit upstream gradient."""
    from oracle import senas_ref as R
    from senas_amd.loss import SegmentationLosses, soft_dice_loss
    from senas_amd.metrics import SegmentationMetric
    z = gio.load('loss_metric')
    lg = torch.from_numpy(z[tag + '/logits'])
    tgt = torch.from_numpy(z[tag + '/target'])
    logits = lg.to(dev()).requires_grad_(True)
    loss = SegmentationLosses('dice_ce')([logits], tgt.to(dev()))
    np.testing.assert_allclose(loss.item(), float(z[tag + '/loss']), rtol=2e-6)
    loss.backward()
    np.testing.assert_allclose(logits.grad.cpu().numpy(), z[tag + '/dlogits'], rtol=2e-5, atol=2e-9)
    # scaled upstream gradient + the dice-only entry point
    ref_in = lg.clone().requires_grad_(True)
    (R.soft_dice_loss(ref_in, tgt) * 3.0).backward()
    got_in = lg.to(dev()).requires_grad_(True)
    (soft_dice_loss(got_in, tgt.to(dev())) * 3.0).backward()
    np.testing.assert_allclose(got_in.grad.cpu().numpy(), ref_in.grad.numpy(), rtol=2e-5, atol=2e-9)
    m = SegmentationMetric(lg.shape[1])
    m.update(tgt.to(dev()), logits.detach())
    m.update(tgt.to(dev()), logits.detach() * 0.5 + 0.1)
    np.testing.assert_allclose(np.array(m.get()), z[tag + '/metric'], atol=2e-3)
    # hard counts are integers: bit-exact against the oracle
    cnt = None
    for t in (lg, lg * 0.5 + 0.1):
        c = R.hard_counts(t, tgt)
        cnt = c if cnt is None else tuple(a + b for a, b in zip(cnt, c))
    tp, fp, fn = m.counts()
    assert [list(map(int, v)) for v in (tp, fp, fn)] == [[int(q) for q in np.asarray(v).reshape(-1)] for v in cnt]


def _random_conv_cases(count, seed):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < count:
        k = int(rng.choice([1, 3, 5, 7]))
        stride = int(rng.choice([1, 1, 2]))
        dil = int(rng.choice([1, 2, 3])) if k > 1 else 1
        tr = bool(stride == 2 and rng.rand() < 0.4)
        ci = int(rng.choice([1, 3, 4, 8, 16, 24, 32, 64]))
        co = int(rng.choice([2, 3, 4, 8, 16, 32]))
        h, w = int(rng.randint(4, 41)), int(rng.randint(4, 41))
        if stride == 2 and not tr:
            h, w = h + (h & 1), w + (w & 1)                      # even inputs, as on the path
        n = int(rng.randint(1, 4))
        if k == 7 and ci > 4:
            continue                                             # 7x7 only exists on the 1/3-channel stem
        cases.append((n, ci, co, h, w, k, stride, dil, tr, bool(rng.rand() < 0.5)))
    return cases


@pytest.mark.parametrize('case', _random_conv_cases(48, 1234), ids=lambda c: 'n%d_%dto%d_%dx%d_k%ds%dd%d%s%s' % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], '_T' if c[8] else '', '_relu' if c[9] else ''))
def test_conv_dispatch_sweep(case):
    """Seeded random geometries across the dispatch space (thin / c8 / LDS window incl. narrow, strided and single-row
    forms / direct-global MFMA / fallbacks): whatever kernel a shape lands on must agree with the oracle."""
    test_conv_kernels_vs_oracle(case)


def _random_node_cases(count, seed):
    rng = np.random.RandomState(seed)
    cases = []
    for _ in range(count):
        c = int(rng.choice([4, 8, 12, 32, 64, 6]))
        cases.append(dict(n=int(rng.randint(1, 5)), c=c, h=int(rng.randint(2, 20)), w=int(rng.randint(2, 20)),
                          T=int(rng.choice([1, 2, 3, 6, 9, 12])), relu=bool(rng.rand() < 0.6), residual=bool(rng.rand() < 0.3),
                          mix=bool(rng.rand() < 0.6), training=bool(rng.rand() < 0.8), seed=int(rng.randint(1 << 30)),
                          se=bool(rng.rand() < 0.4), zero_term=bool(rng.rand() < 0.2)))
    return cases


_BIG_NODES = [dict(n=8, c=32, h=256, w=256, T=2, relu=True, residual=False, mix=False, training=True, seed=901, se=False, zero_term=False),
              dict(n=4, c=8, h=256, w=256, T=12, relu=True, residual=False, mix=True, training=True, seed=902, se=True, zero_term=True)]

# more addends than one launch describes (senas_amd/node.py: a partial sum + the rest on top of it as its residual): the 36-term
# node of a --meta_node_num 5 search cell, the first count past the limit, and 256 channels, where the combine kernel's LDS stage
# lowers the limit to 31 (node.max_terms)
_SPLIT_NODES = [dict(n=4, c=8, h=9, w=7, T=36, relu=True, residual=False, mix=True, training=True, seed=903, se=True, zero_term=True),
                dict(n=2, c=8, h=16, w=16, T=33, relu=True, residual=True, mix=True, training=True, seed=904, se=False, zero_term=False),
                dict(n=2, c=256, h=6, w=5, T=36, relu=True, residual=False, mix=True, training=True, seed=905, se=True, zero_term=True),
                dict(n=3, c=32, h=12, w=12, T=36, relu=False, residual=False, mix=False, training=False, seed=906, se=False, zero_term=False)]


@pytest.mark.parametrize('cfg', _random_node_cases(36, 77) + _BIG_NODES + _SPLIT_NODES, ids=lambda d: 'n%d_c%d_%dx%d_T%d%s%s%s%s' % (
    d['n'], d['c'], d['h'], d['w'], d['T'], '_relu' if d['relu'] else '', '_res' if d['residual'] else '',
    '_se' if d['se'] else '', '' if d['training'] else '_eval'))
def test_node_sweep_vs_torch(cfg):
    """The fused node (bn_combine) against a float64 torch formulation of the same arithmetic: every BatchNorm2d (train
    statistics + running-buffer update, or eval), SE gates, mixing weights, the 'none' op's bias-only term, residual,
    ReLU -- outputs and the gradients of every input."""
    import torch.nn as nn
    from senas_amd import functional as F
    from senas_amd.operations import SEBlock
    g = torch.Generator().manual_seed(cfg['seed'])
    n, c, h, w, T = cfg['n'], cfg['c'], cfg['h'], cfg['w'], cfg['T']
    zs = [torch.randn(n, c, h, w, generator=g) * (0.5 + t) + 0.3 * t for t in range(T)]
    if cfg['zero_term'] and T > 1:
        zs[1] = None                                       # the 'none' candidate: BN of zeros = its bias
    bns = []
    for t in range(T):
        bn = nn.BatchNorm2d(c)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        bn.train(cfg['training'])
        bns.append(bn)
    ses = [SEBlock(c) if (cfg['se'] and t % 2 == 0 and zs[t] is not None) else None for t in range(T)]
    for se in ses:
        if se is not None:
            with torch.no_grad():
                for lin in (se.excitation[0], se.excitation[2]):
                    lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.5)
    mix = torch.rand(T, generator=g) + 0.1 if cfg['mix'] else None
    res = torch.randn(n, c, h, w, generator=g) if cfg['residual'] else None
    gy = torch.randn(n, c, h, w, generator=g)

    # ---- float64 reference
    import copy
    ref_bns = [copy.deepcopy(b).double() for b in bns]
    ref_ses = [copy.deepcopy(s).double() if s is not None else None for s in ses]
    zr = [z.double().requires_grad_(True) if z is not None else None for z in zs]
    mr = mix.double().requires_grad_(True) if mix is not None else None
    rr = res.double().requires_grad_(True) if res is not None else None
    acc = rr if rr is not None else 0.0
    for t in range(T):
        zt = zr[t] if zr[t] is not None else torch.zeros(n, c, h, w, dtype=torch.float64)
        v = ref_bns[t](zt)
        if ref_ses[t] is not None:
            gate = ref_ses[t].excitation(v.mean((2, 3)))
            v = v * gate[:, :, None, None]
        acc = acc + (mr[t] * v if mr is not None else v)
    ref = torch.relu(acc) if cfg['relu'] else acc
    ref.backward(gy.double())

    # ---- HIP path
    dbns = [copy.deepcopy(b).to(dev()) for b in bns]
    dses = [copy.deepcopy(s).to(dev()) if s is not None else None for s in ses]
    zd = [z.to(dev()).requires_grad_(True) if z is not None else None for z in zs]
    md = mix.to(dev()).requires_grad_(True) if mix is not None else None
    rd = res.to(dev()).requires_grad_(True) if res is not None else None
    terms = [F.Term(zd[t], dbns[t], se=dses[t]) for t in range(T)]
    out = F.bn_combine(terms, mix=md, residual=rd, relu=cfg['relu'])
    out.backward(gy.to(dev()))
    close(out, ref.detach().float().numpy(), 'y', rel=2e-5)
    for t in range(T):
        if zd[t] is not None:
            close(zd[t].grad, zr[t].grad.float().numpy(), 'dz%d' % t, rel=1e-4)
        close(dbns[t].weight.grad, ref_bns[t].weight.grad.float().numpy(), 'dgamma%d' % t, rel=1e-4)
        close(dbns[t].bias.grad, ref_bns[t].bias.grad.float().numpy(), 'dbeta%d' % t, rel=1e-4)
        close(dbns[t].running_mean, ref_bns[t].running_mean.float().numpy(), 'running_mean%d' % t, rel=1e-5)
        close(dbns[t].running_var, ref_bns[t].running_var.float().numpy(), 'running_var%d' % t, rel=1e-5)
        if dses[t] is not None:
            for k in (0, 2):
                close(dses[t].excitation[k].weight.grad, ref_ses[t].excitation[k].weight.grad.float().numpy(), 'dse%d.%d' % (t, k), rel=1e-4)
    if md is not None:
        close(md.grad, mr.grad.float().numpy(), 'dmix', rel=1e-4)
    if rd is not None:
        close(rd.grad, rr.grad.float().numpy(), 'dres', rel=1e-5)


def _random_pointwise_cases(count, seed):
    rng = np.random.RandomState(seed)
    kinds = ['avg', 'max', 'bilinear', 'relu', 'dw3', 'dw5', 'dw3T', 'dw5T']
    cases = []
    for _ in range(count):
        kind = kinds[int(rng.randint(len(kinds)))]
        stride = 2 if kind.endswith('T') else int(rng.choice([1, 2]))
        h, w = int(rng.randint(3, 33)), int(rng.randint(3, 33))
        if stride == 2 and not kind.endswith('T'):
            h, w = h + (h & 1), w + (w & 1)
        cases.append((kind, int(rng.randint(1, 5)), int(rng.choice([3, 4, 8, 12, 32, 64])), h, w, stride, bool(rng.rand() < 0.5),
                      int(rng.randint(1 << 30))))
    return cases


@pytest.mark.parametrize('case', _random_pointwise_cases(40, 4242), ids=lambda c: '%s_n%d_c%d_%dx%d_s%d%s' % (c[0], c[1], c[2], c[3], c[4], c[5], '_relu' if c[6] else ''))
def test_pool_resample_depthwise_sweep(case):
    """Seeded random shapes for the HBM-bound kernels (3x3 average / max pooling, bilinear x2, ReLU, depthwise 3x3 / 5x5
    incl. stride 2 and transposed): outputs, producer-side statistics, input and weight gradients against torch (float64)."""
    import torch.nn.functional as tf
    from senas_amd import functional as F
    kind, n, c, h, w, stride, relu, seed = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    xr = x.double().requires_grad_(True)
    xin = torch.relu(xr) if relu else xr
    xd = x.to(dev()).requires_grad_(True)
    wt = wr = wd = None
    if kind == 'avg':
        ref = tf.avg_pool2d(xin, 3, stride, 1, count_include_pad=False)
        got, st = F.avg_pool3(xd, stride, in_relu=relu, want_stats=True)
    elif kind == 'max':
        ref = tf.max_pool2d(xin, 3, stride, 1)
        got, st = F.max_pool3(xd, stride, in_relu=relu, want_stats=True)
    elif kind == 'bilinear':
        ref = tf.interpolate(xin, scale_factor=2, mode='bilinear', align_corners=False)
        got, st = F.bilinear2x(F.relu(xd) if relu else xd, want_stats=True)
    elif kind == 'relu':
        ref, got, st = torch.relu(xr), F.relu(xd), None
    else:
        k = 3 if '3' in kind else 5
        tr = kind.endswith('T')
        wt = torch.randn(c, 1, k, k, generator=g) * 0.3
        wr, wd = wt.double().requires_grad_(True), wt.to(dev()).requires_grad_(True)
        if tr:
            ref = tf.conv_transpose2d(xin, wr, stride=2, padding=k // 2, output_padding=1, groups=c)
        else:
            ref = tf.conv2d(xin, wr, stride=stride, padding=k // 2, groups=c)
        got, st = F.conv2d(xd, wd, stride=stride, pad=k // 2, dil=1, transposed=tr, out_pad=1 if tr else 0, groups=c,
                           in_relu=relu, want_stats=True)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy.double())
    got.backward(gy.to(dev()))
    close(got, ref.detach().float().numpy(), 'y', rel=2e-5)
    close(xd.grad, xr.grad.float().numpy(), 'dx', rel=2e-5)
    if wd is not None:
        close(wd.grad, wr.grad.float().numpy(), 'dw', rel=5e-5)
    if st is not None:
        r64 = ref.detach()
        exp = torch.stack([r64.sum((2, 3)), (r64 ** 2).sum((2, 3))], -1)
        close(st.float(), exp.float().numpy(), 'stats', rel=2e-5)


@pytest.mark.parametrize('case', [(2, 2, 16, 16), (1, 3, 9, 13), (3, 4, 32, 20), (2, 5, 7, 7), (1, 8, 33, 31), (4, 6, 24, 24)],
                         ids=lambda c: 'n%d_c%d_%dx%d' % c)
def test_loss_metric_class_sweep(case):
    """Dice+CE and metric kernels for 2..8 classes and odd shapes against the oracle (loss, gradient, hard counts)."""
    from oracle import senas_ref as R
    from senas_amd.loss import SegmentationLosses
    from senas_amd.metrics import SegmentationMetric
    n, c, h, w = case
    g = torch.Generator().manual_seed(n * 1000 + c * 100 + h)
    lg = torch.randn(n, c, h, w, generator=g) * 2.0
    tgt = torch.randint(0, c, (n, h, w), generator=g)
    ref_in = lg.clone().requires_grad_(True)
    ref = R.dice_ce_loss(ref_in, tgt)
    ref.backward()
    got_in = lg.to(dev()).requires_grad_(True)
    got = SegmentationLosses('dice_ce')([got_in], tgt.to(dev()))
    got.backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=5e-6)
    np.testing.assert_allclose(got_in.grad.cpu().numpy(), ref_in.grad.numpy(), rtol=5e-5, atol=2e-9)
    m = SegmentationMetric(c)
    m.update(tgt.to(dev()), got_in.detach())
    tp, fp, fn = m.counts()
    rtp, rfp, rfn = R.hard_counts(lg, tgt)
    assert [list(map(int, v)) for v in (tp, fp, fn)] == [[int(q) for q in np.asarray(v).reshape(-1)] for v in (rtp, rfp, rfn)]
    assert abs(m.get()[0] - round(100.0 * float(R.mean_pix_accuracy(lg, tgt)), 3)) < 2e-3


def _random_genotype(rng, nodes):
    from senas_amd.operations import DownOps, NormOps, UpOps

    def cell(kind):
        gene = []
        for i in range(nodes):
            for idx in sorted(rng.choice(2 + i, 2, replace=False)):
                if idx >= 2:
                    ops = NormOps
                elif kind == 'down':
                    ops = DownOps
                else:
                    ops = UpOps if idx > 0 else NormOps
                gene.append((ops[int(rng.randint(len(ops)))], int(idx)))
        return gene
    return cell('down'), cell('up')


@pytest.mark.parametrize('seed', [11, 12, 13, 14, 15, 16])
def test_derived_random_genotypes_vs_oracle(seed):
    """Derived networks built from RANDOM genotypes (every candidate op in every legal position, incl. 'none' and
    'identity'; 3 or 4 nodes; with and without pruned up cells), c=8, depth 4, 2x1x32x32: logits, loss and every
    parameter gradient against the oracle."""
    from oracle import senas_ref as R
    from senas_amd.genotype import Genotype
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    rng = np.random.RandomState(seed)
    nodes = int(rng.choice([3, 4]))
    down, up = _random_genotype(rng, nodes)
    gamma = [int(v) for v in rng.randint(0, 2, 3)]                # depth 4: 3 gates
    if gamma[1] == 1 and gamma[2] == 0:
        gamma[2] = 1                                             # keep the second skip row monotone (as NAS.genotype() emits)
    geno = Genotype(down=down, down_concat=range(2, 2 + nodes), up=up, up_concat=range(2, 2 + nodes), gamma=gamma)
    net = SenasModel(2, 1, c=8, depth=4, genotype=geno)
    _randomize(net, seed)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    gio.share_stem(sd)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1, 32, 32, generator=gen)
    y = torch.randint(0, 2, (2, 32, 32), generator=gen)
    ref = R.derived_forward(sd, x, R.Genotype(*geno), depth=4)[-1]
    ref_loss = R.dice_ce_loss(ref, y)
    ref_loss.backward()
    net = net.to(dev()).train()
    out = net(x.to(dev()))
    loss = SegmentationLosses('dice_ce')(out, y.to(dev()))
    loss.backward()
    close(out[-1], ref.detach().numpy(), 'logits %s' % (geno,), rel=1e-3)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * abs(float(ref_loss.detach()))
    # Gradients through ReLU + train-mode BN on 4x4 maps are ill-conditioned (DESIGN.md "gradient parity"): a
    # pre-activation within rounding distance of zero flips a ReLU and moves every upstream gradient by percents.  Judge
    # every parameter gradient against the float64 oracle; the yardstick is how far the oracle's own gradient moves
    # under float32-sized noise: the float32 oracle, and (only when that is not enough) float64 oracles with weights
    # perturbed by 1e-6 relative.
    def oracle64(perturb_seed=None):
        g = torch.Generator().manual_seed(1000 + (perturb_seed or 0))
        sd64 = {}
        for k, v in sd.items():
            w = v.detach().double() if v.is_floating_point() else v.detach().clone()
            if v.is_floating_point() and v.requires_grad:
                if perturb_seed is not None:
                    w = w * (1.0 + 1e-6 * torch.randn(w.shape, generator=g, dtype=torch.float64))
                w.requires_grad_(True)
            sd64[k] = w
        gio.share_stem(sd64)
        R.dice_ce_loss(R.derived_forward(sd64, x.double(), R.Genotype(*geno), depth=4)[-1], y).backward()
        return {k: v.grad.numpy() for k, v in sd64.items() if v.is_floating_point() and v.grad is not None}

    def l2(a, b):
        return float(np.sqrt(((a - b) ** 2).sum()))
    e64 = oracle64()
    got = grads_of(net)
    names = [k for k in got if k in e64]
    assert 'stem0.0.weight' in names and len(names) > 20
    norms = {k: float(np.sqrt((e64[k] ** 2).sum())) for k in names}
    floor = 1e-6 * max(norms.values())                            # conv biases ahead of a BN: zero gradient, only noise
    names = [k for k in names if norms[k] > floor]
    noise = {k: l2(sd[k].grad.numpy().astype(np.float64), e64[k]) / norms[k] for k in names}
    errs = {k: l2(got[k], e64[k]) / norms[k] for k in names}
    bad = [k for k in names if errs[k] > max(1e-3, 10.0 * noise[k])]
    for ps in range(6):
        if not bad:
            break
        alt = oracle64(ps)
        for k in names:
            noise[k] = max(noise[k], l2(alt[k], e64[k]) / norms[k])
        bad = [k for k in names if errs[k] > max(1e-3, 10.0 * noise[k])]
    assert not bad, 'gradients off: ' + ', '.join('%s %.2e (oracle noise %.2e)' % (k, errs[k], noise[k]) for k in bad[:6])


def _channel_mask_dropout2d(x, p=0.5, training=True, inplace=False):
    """A deterministic stand-in for torch.nn.functional.dropout2d, the same on the CPU (oracle) and on the device: whole
    (image, channel) planes are dropped by a fixed rule of (n, c, height), the rest scaled by 1 / (1 - p)."""
    if not training or p == 0:
        return x
    n, c = x.shape[:2]
    idx = torch.arange(n, device=x.device).view(n, 1) * 7 + torch.arange(c, device=x.device).view(1, c) * 3 + x.shape[2]
    keep = (idx % 4 != 0).to(x.dtype) / (1.0 - p)
    return x * keep.view(n, c, 1, 1)


@pytest.mark.parametrize('seed', [21, 22])
def test_derived_dropout_vs_oracle(seed, monkeypatch):
    """SenasModel(dropout_prob > 0) (models/senas_model.py:80,110-111,133-134 -> utils/operations.py:121-122): a Dropout2d
    in front of every convolution of the candidate ops.  With one deterministic mask on both sides, training-mode logits,
    loss and gradients must match the oracle (placement of the dropouts, the shifted child indices / state_dict keys);
    in eval mode the network with dropout equals the one without."""
    from oracle import senas_ref as R
    from senas_amd.genotype import Genotype
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    monkeypatch.setattr(torch.nn.functional, 'dropout2d', _channel_mask_dropout2d)
    rng = np.random.RandomState(seed)
    down, up = _random_genotype(rng, 3)
    if seed == 21:      # every weighted candidate at least once
        down[0], down[1], up[1], up[2] = ('dep_sep_conv_5', down[0][1]), ('se_conv_3', down[1][1]), ('dil_2_conv_5', up[1][1]), ('dep_sep_conv_3', up[2][1])
    geno = Genotype(down=down, down_concat=range(2, 5), up=up, up_concat=range(2, 5), gamma=[1, 1, 1])
    net = SenasModel(2, 1, c=8, depth=4, dropout_prob=0.25, genotype=geno)
    _randomize(net, seed)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert any(k.endswith('_ops.0.1.weight') for k in sd), 'the convolution behind a Dropout2d is child 1'
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    gio.share_stem(sd)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 1, 32, 32, generator=gen)
    y = torch.randint(0, 2, (2, 32, 32), generator=gen)
    ref = R.derived_forward(sd, x, R.Genotype(*geno), depth=4, dropout_prob=0.25)[-1]
    ref_loss = R.dice_ce_loss(ref, y)
    ref_loss.backward()
    plain = R.derived_forward({k: v.detach() for k, v in sd.items()}, x, R.Genotype(*geno), depth=4, dropout_prob=0.25, training=False)[-1]
    net = net.to(dev()).train()
    out = net(x.to(dev()))
    loss = SegmentationLosses('dice_ce')(out, y.to(dev()))
    loss.backward()
    close(out[-1], ref.detach().numpy(), 'logits with dropout', rel=1e-3)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * abs(float(ref_loss.detach()))
    got = grads_of(net)
    checked = 0
    for k in ('stem0.0.weight', 'head_block.0.segmentation_head.1.weight'):
        if k in got and sd[k].grad is not None:
            close(torch.from_numpy(got[k]), sd[k].grad.numpy(), 'grad ' + k, rel=2e-2)
            checked += 1
    assert checked >= 1
    # the mask really bites (training differs from eval), and in eval mode the dropouts are the identity
    assert float((plain - ref.detach()).abs().max()) > 1e-3
    net.eval()
    with torch.no_grad():
        ev = net(x.to(dev()))[-1]
    close(ev, plain.numpy(), 'eval logits (dropout is the identity)', rel=1e-3)


# ---------------------------------------------------------------------------------------------- inference pass (8f-4)
@pytest.mark.parametrize('folded', [False, True])
@pytest.mark.parametrize('graphed', [False, True])
def test_evaluator_vs_oracle(graphed, folded):
    """The validation / testing pass (experiments/testing_model.py:150-190): eval-mode logits, loss, arg-max masks and
    pixAcc / mIoU / Dice over three batches, eager and HIP-graph replayed, against the oracle run in eval mode."""
    from oracle import senas_ref as R
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.infer import Evaluator, FoldedEvaluator
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    net = SenasModel(2, 1, c=16, depth=4, genotype=senas_node_4)
    _randomize(net, 5)
    # running statistics a trained net would carry (not the 0 / 1 of a fresh module)
    gen = torch.Generator().manual_seed(5)
    for k, v in net.state_dict().items():
        if k.endswith('running_mean'):
            v.copy_(0.2 * torch.randn(v.shape, generator=gen))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=gen))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    gio.share_stem(sd)
    xs = torch.randn(3, 2, 1, 64, 64, generator=gen)
    ys = torch.randint(0, 2, (3, 2, 64, 64), generator=gen)
    net = net.to(dev())
    ev = (FoldedEvaluator if folded else Evaluator)(net, 2, xs[0].to(dev()), ys[0].to(dev()), SegmentationLosses('dice_ce'),
                                                    use_graph=graphed)
    assert (ev.graph is not None) == graphed
    ref_loss, tp, fp, fn, acc = 0.0, 0, 0, 0, 0.0
    for b in range(3):
        logits, mask = ev(xs[b].to(dev()), ys[b].to(dev()))
        with torch.no_grad():
            ref = R.derived_forward(sd, xs[b], R.Genotype(*senas_node_4), depth=4, training=False)[-1]
        close(logits, ref.numpy(), 'eval logits, batch %d' % b, rel=1e-3)
        top2 = ref.topk(2, dim=1).values
        sure = (top2[:, 0] - top2[:, 1]) > 1e-3 * ref.abs().max()
        assert bool((mask.cpu() == ref.argmax(1))[sure].all())
        assert bool((mask == logits.argmax(1)).all())
        ref_loss += float(R.dice_ce_loss(ref, ys[b]))
    mean_loss, pix, miou, dice = ev.result()
    assert abs(mean_loss - ref_loss / 3) <= 1e-4 * abs(ref_loss / 3)
    if folded:
        assert ev.folded.fused_launches > 20                      # batch-norm really rides in convolution epilogues
    # the metric against the reference formulas on the evaluator's own logits is covered by the metric tests; here:
    # the running-statistics buffers must be untouched by an eval pass, and the figures finite and in range
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            assert torch.equal(v.cpu(), sd[k]), k
    assert 0.0 <= pix <= 100.0 and 0.0 <= miou <= 100.0 and 0.0 <= dice <= 100.0
    ev.reset()
    ev(xs[0].to(dev()), ys[0].to(dev()))
    first = ev.result()
    ev.reset()
    ev(xs[0].to(dev()), ys[0].to(dev()))
    assert ev.result() == first                                   # reset really clears the device accumulators
    if folded:
        # validation between epochs: the weights moved; refresh() must bring the folded constants (and a captured graph,
        # which reads them in place) up to date
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.mul_(1.1)
                    m.running_mean.add_(0.05)
                elif isinstance(m, nn.Conv2d):
                    m.weight.mul_(0.97)
        ev.refresh()
        logits, _ = ev(xs[1].to(dev()), ys[1].to(dev()))
        with torch.no_grad():
            want = net(xs[1].to(dev()))[-1]
        close(logits, want.cpu().numpy(), 'folded logits after refresh', rel=1e-3)
    ev.packer.uninstall()


@pytest.mark.parametrize('seed', [11, 12, 13, 14, 15, 16])
def test_folded_inference_random_genotypes(seed):
    """The batch-norm-folded inference forward on derived nets of RANDOM genotypes (every candidate op in every legal
    position, 'none' and 'identity' included, 3 / 4 nodes, pruned rows), c=16, depth 4, 2x1x64x64, non-trivial running
    statistics: logits against the oracle in eval mode, and against the module forward in eval mode."""
    from oracle import senas_ref as R
    from senas_amd.genotype import Genotype
    from senas_amd.infer import FoldedForward
    from senas_amd.senas_model import SenasModel
    rng = np.random.RandomState(seed)
    nodes = int(rng.choice([3, 4]))
    down, up = _random_genotype(rng, nodes)
    gamma = [int(v) for v in rng.randint(0, 2, 3)]
    if gamma[1] == 1 and gamma[2] == 0:
        gamma[2] = 1
    geno = Genotype(down=down, down_concat=range(2, 2 + nodes), up=up, up_concat=range(2, 2 + nodes), gamma=gamma)
    net = SenasModel(2, 1, c=16, depth=4, genotype=geno)
    _randomize(net, seed)
    gen = torch.Generator().manual_seed(seed)
    for k, v in net.state_dict().items():
        if k.endswith('running_mean'):
            v.copy_(0.2 * torch.randn(v.shape, generator=gen))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=gen))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    gio.share_stem(sd)
    x = torch.randn(2, 1, 64, 64, generator=gen)
    with torch.no_grad():
        ref = R.derived_forward(sd, x, R.Genotype(*geno), depth=4, training=False)[-1]
    net = net.to(dev()).eval()
    ff = FoldedForward(net, 2)
    with torch.no_grad():
        got = ff(x.to(dev()))[-1]
        plain = net(x.to(dev()))[-1]
    close(got, ref.numpy(), 'folded logits %s' % (geno,), rel=1e-3)
    close(got, plain.cpu().numpy(), 'folded vs module forward', rel=1e-3)
    assert ff.fused_launches > 0


# ------------------------------------------------------------------------------ BASELINE sizes: size-independent properties
def test_full_size_batch_properties():
    """BASELINE configs[1] at full size (README genotype, c=32, depth 5, 8x1x256x256) is too big for the CPU oracle to
    finish in seconds, so it is checked through properties the network has by construction:
    train mode -- batch statistics do not depend on the order of the images, so permuting the batch permutes the logits
    and leaves the loss and every parameter gradient where they were;
    eval mode -- an image's logits do not depend on its batch mates, and the folded inference forward agrees with the
    module forward."""
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.infer import FoldedForward
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4)
    _randomize(net, 3)
    net = net.to(dev()).train()
    crit = SegmentationLosses('dice_ce')
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(8, 1, 256, 256, generator=gen).to(dev())
    y = torch.randint(0, 2, (8, 256, 256), generator=gen).to(dev())
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=dev())
    runs = []
    for xx, yy in ((x, y), (x[perm].contiguous(), y[perm].contiguous())):
        net.zero_grad(set_to_none=True)
        out = net(xx)[-1]
        loss = crit([out], yy)
        loss.backward()
        runs.append((out.detach(), float(loss.detach()), grads_of(net)))
    (o0, l0, g0), (o1, l1, g1) = runs
    close(o1, o0[perm].cpu().numpy(), 'logits of the permuted batch', rel=1e-4)
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    assert set(g0) == set(g1) and len(g0) > 300
    for k in g0:
        a, b = g0[k].astype(np.float64), g1[k].astype(np.float64)
        norm = np.sqrt((a ** 2).sum())
        if norm > 1e-8:
            # (a ReLU input within fp32 noise of zero may flip between the two summation orders: L2, not max)
            assert np.sqrt(((a - b) ** 2).sum()) <= 1e-2 * norm, 'gradient %s moved under a batch permutation' % k
    net.eval()
    with torch.no_grad():
        full = net(x)[-1]
        one = net(x[2:3].contiguous())[-1]
        folded = FoldedForward(net, 8)(x)[-1]
    close(one, full[2:3].cpu().numpy(), 'eval logits of image 2 alone vs inside the batch', rel=1e-5)
    close(folded, full.cpu().numpy(), 'folded inference forward vs module forward', rel=1e-4)
    assert bool((folded.argmax(1) == full.argmax(1)).float().mean() > 0.9999)


def test_full_size_supernet_batch_properties():
    """BASELINE configs[2] at full size (NAS supernet, c=32, depth 5, 3 nodes, 4x1x256x256): permuting the batch permutes
    the logits and leaves the loss, the architecture gradients and the derived genotype where they were; the frozen-weight
    architecture pass of SearchStep produces the same architecture gradients as the full backward."""
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    torch.manual_seed(11)
    net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev()).to(dev()).train()
    with torch.no_grad():
        for p in net.arch_parameters():
            p.copy_(0.3 * torch.randn(p.shape, device=dev()))
    crit = SegmentationLosses('dice_ce')
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(4, 1, 256, 256, generator=gen).to(dev())
    y = torch.randint(0, 2, (4, 256, 256), generator=gen).to(dev())
    perm = torch.tensor([2, 0, 3, 1], device=dev())
    geno = net.genotype()
    weights = [p for p in net.parameters() if not any(p is a for a in net.arch_parameters())]
    runs = []
    for xx, yy, frozen in ((x, y, False), (x[perm].contiguous(), y[perm].contiguous(), False), (x, y, True)):
        net.zero_grad(set_to_none=True)
        for p in weights:
            p.requires_grad_(not frozen)
        out = net(xx)[-1]
        loss = crit([out], yy)
        loss.backward()
        runs.append((out.detach(), float(loss.detach()), [a.grad.detach().cpu().numpy().astype(np.float64) for a in net.arch_parameters()]))
        if frozen:
            assert all(p.grad is None for p in weights)           # no weight-gradient kernel ran
    for p in weights:
        p.requires_grad_(True)
    (o0, l0, a0), (o1, l1, a1), (o2, l2, a2) = runs
    close(o1, o0[perm].cpu().numpy(), 'supernet logits of the permuted batch', rel=1e-4)
    assert abs(l0 - l1) <= 1e-5 * abs(l0) and abs(l0 - l2) <= 1e-6 * abs(l0)
    for k, (g0, g1, g2) in enumerate(zip(a0, a1, a2)):
        scale = np.abs(g0).max()
        assert np.abs(g0 - g1).max() <= 1e-2 * scale, 'architecture gradient %d moved under a batch permutation' % k
        assert np.abs(g0 - g2).max() <= 1e-5 * scale, 'architecture gradient %d differs with frozen weights' % k
    assert net.genotype() == geno


def test_eager_training_beside_an_installed_packer_sees_the_new_weights():
    """An Evaluator leaves its weight packer installed; a hand-written training loop (torch optimizer, no step driver)
    on the same model must not be served the evaluator's images once the weights have moved."""
    import copy
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.infer import Evaluator
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    torch.manual_seed(22)
    x = torch.randn(2, 1, 64, 64, device=dev())
    y = torch.randint(0, 2, (2, 64, 64), device=dev())
    crit = SegmentationLosses('dice_ce')
    net = SenasModel(2, 1, c=32, depth=4, genotype=senas_node_4).to(dev())
    ev = Evaluator(net, 2, x, y, crit, use_graph=False)
    ev(x, y)
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=5e-2, momentum=0.9)
    for _ in range(2):
        opt.zero_grad()
        crit(net(x), y).backward()
        opt.step()
    with torch.no_grad():
        got = net(x)[-1]
        want = copy.deepcopy(net)(x)[-1]
    close(got, want.cpu().numpy(), 'eager forward after eager steps beside an installed packer', rel=1e-5)
    ev.packer.uninstall()


@pytest.mark.parametrize('kind', ['derived', 'supernet'])
def test_eager_forward_after_a_graphed_step_sees_the_new_weights(kind):
    """A validation forward right after a training step (the drivers do exactly that): the step driver's cached weight
    images -- MFMA fragment images, stacked search-cell weights -- are one optimizer step old at that point and must not
    be used.  The eager forward has to agree with an independent copy of the model that never saw a packer."""
    import copy
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep, TrainStep
    torch.manual_seed(21)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(2, 1, 64, 64, generator=gen).to(dev())
    y = torch.randint(0, 2, (2, 64, 64), generator=gen).to(dev())
    crit = SegmentationLosses('dice_ce')
    if kind == 'derived':
        net = SenasModel(2, 1, c=32, depth=4, genotype=senas_node_4).to(dev()).train()
        opt = torch.optim.SGD(net.parameters(), lr=5e-2, momentum=0.9)
        step = TrainStep(net, crit, opt, x, y, use_graph=True)
        run = step
    else:
        net = NAS(1, 32, 2, 3, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev()).to(dev()).train()
        opt = torch.optim.SGD(net.parameters(), lr=5e-2, momentum=0.9)
        opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-2)
        step = SearchStep(net, crit, opt, opt_a, x, y, use_graph=True)

        def run():
            return step(x, y, x, y)
    for _ in range(2):
        run()
    with torch.no_grad():
        got = net(x)[-1]
        twin = copy.deepcopy(net)                  # same weights at fresh addresses: no cached image can match
        want = twin(x)[-1]
    close(got, want.cpu().numpy(), 'eager forward after a graphed %s step' % kind, rel=1e-5)
    run()                                          # and the driver keeps working after the interleaved eager pass
    step.close()


@pytest.mark.parametrize('case', [(3, 2, 32, 12, 20, True, False), (6, 4, 32, 16, 16, True, True), (2, 2, 8, 9, 7, True, False),
                                  (8, 1, 64, 8, 8, True, True), (1, 3, 16, 5, 5, True, False), (4, 2, 32, 8, 12, False, False)])
def test_bnrelu_multi_vs_torch(case):
    """senas_bnrelu_multi_fwd / _bwd (k independent BatchNorm2d + ReLU in one launch) against float64 torch: outputs, running
    buffers, dz / dgamma / dbeta; with the upstream gradient arriving as a channel slice of a wider tensor (strided)."""
    from senas_amd import functional as F
    k, n, c, h, w, training, strided = case
    gen = torch.Generator().manual_seed(sum(case[:5]))
    bns = [nn.BatchNorm2d(c) for _ in range(k)]
    zs, refs = [], []
    for bn in bns:
        with torch.no_grad():
            bn.weight.copy_(1.0 + 0.3 * torch.randn(c, generator=gen))
            bn.bias.copy_(0.3 * torch.randn(c, generator=gen))
            bn.running_mean.copy_(0.2 * torch.randn(c, generator=gen))
            bn.running_var.copy_(0.5 + torch.rand(c, generator=gen))
        bn.train(training)
        z = (0.5 + torch.randn(n, c, h, w, generator=gen)) * 1.5
        zs.append(z)
        ref_bn = nn.BatchNorm2d(c).double()
        ref_bn.load_state_dict({kk: (v.double() if v.is_floating_point() else v.clone()) for kk, v in bn.state_dict().items()})
        ref_bn.train(training)
        refs.append(ref_bn)
    wide = [torch.randn(n, 3 * c, h, w, generator=gen) for _ in range(k)]
    # float64 reference
    want = []
    for z, ref_bn, wd in zip(zs, refs, wide):
        zz = z.double().requires_grad_(True)
        y = torch.relu(ref_bn(zz))
        (y * wd[:, c:2 * c].double()).sum().backward()
        want.append((y.detach(), zz.grad, ref_bn.weight.grad, ref_bn.bias.grad, ref_bn.running_mean, ref_bn.running_var))
    dbns = [bn.to(dev()) for bn in bns]
    dz_in = [z.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True) for z in zs]
    assert F.bnrelu_multi_ok(dz_in, dbns) == training          # eval-mode modules only without autograd
    if not training:
        with torch.no_grad():
            ys = F.bnrelu_multi(dz_in, dbns, [None] * k)
        for y, wnt in zip(ys, want):
            close(y, wnt[0].numpy(), 'eval output', rel=2e-5)
        return
    ys = F.bnrelu_multi(dz_in, dbns, [None] * k)
    loss = 0
    for y, wd in zip(ys, wide):
        wdev = wd.to(dev()).contiguous(memory_format=torch.channels_last)
        if strided:
            cat = torch.cat([torch.zeros_like(y), y, torch.zeros_like(y)], dim=1)        # dy of y = a channel slice of d cat
            loss = loss + (cat * wdev).sum()
        else:
            loss = loss + (y * wdev[:, c:2 * c]).sum()
    loss.backward()
    for t in range(k):
        y, dz, dg, db, rm, rv = want[t]
        close(ys[t], y.numpy(), 'output %d' % t, rel=2e-5)
        close(dz_in[t].grad, dz.numpy(), 'dz %d' % t, rel=2e-4)
        close(dbns[t].weight.grad, dg.numpy(), 'dgamma %d' % t, rel=2e-4)
        close(dbns[t].bias.grad, db.numpy(), 'dbeta %d' % t, rel=2e-4)
        close(dbns[t].running_mean, rm.numpy(), 'running_mean %d' % t, rel=2e-5)
        close(dbns[t].running_var, rv.numpy(), 'running_var %d' % t, rel=2e-5)
        assert int(dbns[t].num_batches_tracked) == 1


@pytest.mark.parametrize('case', [(3, 2, 32, 16, 20, 5, 1, False), (2, 2, 32, 16, 16, 3, 2, False), (3, 2, 32, 8, 12, 5, 2, True),
                                  (4, 1, 8, 12, 12, 3, 1, False), (2, 4, 16, 9, 11, 5, 1, False), (3, 2, 32, 12, 12, 3, 2, True),
                                  # kernel size 0: dep_sep_conv_3 and dep_sep_conv_5 of the same edges share the launches (3, 5, 3, 5, ...)
                                  (6, 4, 32, 16, 16, 0, 1, False), (6, 2, 32, 16, 24, 0, 2, False), (6, 2, 32, 8, 12, 0, 2, True),
                                  (2, 2, 8, 12, 12, 0, 1, False), (4, 3, 8, 9, 11, 0, 1, False), (8, 1, 16, 8, 8, 0, 2, True),
                                  # stride 2 on small maps: the data gradient requests the tap slots of three problems together
                                  (3, 2, 16, 10, 14, 5, 2, False), (5, 1, 8, 9, 11, 0, 2, False), (7, 2, 32, 8, 8, 0, 2, False),
                                  # past 128 blocks of outputs the data gradient is one thread per output again (no problem split)
                                  (6, 4, 8, 128, 144, 0, 1, False), (6, 4, 8, 128, 144, 0, 2, False)])
def test_dwconv_multi_vs_single(case):
    """senas_dwconv_pair_* (k depthwise convolutions of one input, 3x3 and 5x5 mixed: one forward launch, one data-gradient
    launch summing over the problems, one weight-gradient launch + sums) against the same convolutions run one by one
    through torch (float64)."""
    from senas_amd import functional as F
    k, n, c, h, w, ks0, stride, tr = case
    gen = torch.Generator().manual_seed(sum(case[:7]))

    def mk(t):
        ks = ks0 if ks0 else (3, 5)[t % 2]
        if tr:
            return nn.ConvTranspose2d(c, c, ks, stride=stride, padding=ks // 2, output_padding=stride - 1, groups=c, bias=False)
        return nn.Conv2d(c, c, ks, stride=stride, padding=ks // 2, groups=c, bias=False)
    convs = [mk(t) for t in range(k)]
    for cv in convs:
        with torch.no_grad():
            cv.weight.copy_(torch.randn(cv.weight.shape, generator=gen) * 0.3)
    x = torch.randn(n, c, h, w, generator=gen)
    x64 = x.double().requires_grad_(True)
    refs = [copy_module64(cv) for cv in convs]
    outs64 = [r(x64) for r in refs]
    gs = [torch.randn(o.shape, generator=gen) for o in outs64]
    sum((o * g.double()).sum() for o, g in zip(outs64, gs)).backward()
    dconvs = [cv.to(dev()) for cv in convs]
    xd = x.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    parts = F.dwconv_multi(xd, dconvs, True)
    assert parts is not None
    sum((z * g.to(dev())).sum() for (z, _), g in zip(parts, gs)).backward()
    for t in range(k):
        z, st = parts[t]
        o = outs64[t].detach()
        close(z, o.numpy(), 'output %d' % t, rel=5e-5)
        want_st = torch.stack([o.sum((2, 3)), (o * o).sum((2, 3))], dim=-1)
        close(st.float(), want_st.numpy(), 'statistics %d' % t, rel=1e-4)
        close(dconvs[t].weight.grad, refs[t].weight.grad.numpy(), 'dw %d' % t, rel=2e-4)
    close(xd.grad, x64.grad.numpy(), 'dx (sum over the problems)', rel=2e-4)


@pytest.mark.parametrize('case', [(3, 2, 32, 16, 16, 1, False), (3, 4, 8, 12, 20, 1, False), (2, 2, 16, 8, 8, 2, False), (3, 2, 32, 8, 12, 2, True),
                                  (2, 3, 32, 8, 8, 1, False)])
def test_dwconv_multi2_vs_two_single_input_launches(case):
    """functional.dwconv_multi2 (the DepSepConv candidates of BOTH input states of a search cell in one forward and one
    weight-gradient launch, senas_dwconv_pair_fwd_xs / _bwd_weight_xs): outputs, statistics, both input gradients and all
    weight gradients equal to functional.dwconv_multi run once per input (edges, n, c, h, w, stride, transposed; each
    edge brings a 3x3 and a 5x5 problem)."""
    from senas_amd import functional as F
    edges, n, c, h, w, stride, tr = case
    torch.manual_seed(sum(case[:6]))

    def convs():
        out = []
        for _ in range(edges):
            for ks in (3, 5):
                if tr:
                    m = torch.nn.ConvTranspose2d(c, c, ks, stride=stride, padding=ks // 2, output_padding=stride - 1, groups=c, bias=False)
                else:
                    m = torch.nn.Conv2d(c, c, ks, stride=stride, padding=ks // 2, groups=c, bias=False)
                out.append(m.to(dev()))
        return out
    ca, cb = convs(), convs()
    xa = torch.randn(n, c, h, w, device=dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xb = torch.randn(n, c, h, w, device=dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ref_a, ref_b = F.dwconv_multi(xa, ca, True), F.dwconv_multi(xb, cb, True)
    assert ref_a is not None and ref_b is not None
    wts = [torch.randn_like(z) for z, _ in ref_a + ref_b]
    sum((z * t).sum() for (z, _), t in zip(ref_a + ref_b, wts)).backward()
    want = [xa.grad.clone(), xb.grad.clone()] + [m.weight.grad.clone() for m in ca + cb]
    want_out = [(z.detach().clone(), st.clone()) for z, st in ref_a + ref_b]
    xa.grad = xb.grad = None
    for m in ca + cb:
        m.weight.grad = None
    both = F.dwconv_multi2(xa, ca, xb, cb, True)
    assert both is not None, 'off the paired path'
    got = both[0] + both[1]
    for (z, st), (wz, wst) in zip(got, want_out):
        close(z, wz.cpu().numpy(), 'output', rel=1e-6)
        assert torch.allclose(st, wst, rtol=1e-10, atol=1e-9), 'statistics differ'         # (fp64 atomics: the order may differ)
    sum((z * t).sum() for (z, _), t in zip(got, wts)).backward()
    have = [xa.grad, xb.grad] + [m.weight.grad for m in ca + cb]
    for i, (a, b) in enumerate(zip(have, want)):
        close(a, b.cpu().numpy(), 'gradient %d' % i, rel=2e-6)


def copy_module64(m):
    import copy
    return copy.deepcopy(m).double()


@pytest.mark.parametrize('case', [(6, 4, 32, 8, 16, 16), (4, 2, 8, 8, 12, 20), (2, 2, 32, 8, 8, 8), (8, 1, 16, 4, 9, 7), (3, 4, 64, 8, 32, 32)])
def test_pw_multi_vs_torch(case):
    """senas_pw_multi_* (k independent 1x1 convolutions sharing their launches) against float64 torch: outputs, producer-side
    statistics, every dx and dw."""
    from senas_amd import functional as F
    k, n, cin, cout, h, w = case
    gen = torch.Generator().manual_seed(sum(case))
    convs = [nn.Conv2d(cin, cout, 1, bias=False) for _ in range(k)]
    for cv in convs:
        with torch.no_grad():
            cv.weight.copy_(torch.randn(cv.weight.shape, generator=gen) * 0.3)
    xs = [torch.randn(n, cin, h, w, generator=gen) for _ in range(k)]
    gs = [torch.randn(n, cout, h, w, generator=gen) for _ in range(k)]
    x64 = [x.double().requires_grad_(True) for x in xs]
    refs = [copy_module64(cv) for cv in convs]
    outs = [r(x) for r, x in zip(refs, x64)]
    sum((o * g.double()).sum() for o, g in zip(outs, gs)).backward()
    dconvs = [cv.to(dev()) for cv in convs]
    xd = [x.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
    parts = F.pw_multi(xd, dconvs, True)
    assert parts is not None
    sum((z * g.to(dev())).sum() for (z, _), g in zip(parts, gs)).backward()
    for t in range(k):
        z, st = parts[t]
        o = outs[t].detach()
        close(z, o.numpy(), 'output %d' % t, rel=5e-5)
        close(st.float(), torch.stack([o.sum((2, 3)), (o * o).sum((2, 3))], dim=-1).numpy(), 'statistics %d' % t, rel=1e-4)
        close(xd[t].grad, x64[t].grad.numpy(), 'dx %d' % t, rel=2e-4)
        close(dconvs[t].weight.grad, refs[t].weight.grad.numpy(), 'dw %d' % t, rel=2e-4)


@pytest.mark.parametrize('nodes,stacked', [(4, True), (2, True), (3, False), (5, True)])
def test_supernet_other_node_counts_vs_oracle(nodes, stacked):
    """``--meta_node_num`` (experiments/search_arc.py:38-44): 5 nodes -- the last node has six inputs, 36 addends: more than one
    node launch describes (SENAS_MAX_TERMS 32), so it runs as a partial sum + the rest on top (node.bn_combine); five edges
    leave states 0 and 1, beyond the widest stack.  The search cell with 4 nodes (states 0 and 1 feed FOUR edges: the widest stacks and batched launches: 32 -> 32
    stacked candidates, 8 DepSepConv candidates per state, 30 terms on the last node), with 2 nodes, and with the
    state-major execution switched off (every candidate launched on its own): forward + backward + architecture gradients
    against the CPU oracle, c=32, depth 3, 2x1x32x32."""
    from oracle import senas_ref as R
    from senas_amd.cell import Cell
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    kw = dict(input_c=1, c=32, num_classes=2, depth=3, meta_node_num=nodes)
    torch.manual_seed(3)
    net = NAS(use_sharing=False, double_down_channel=False, multi_gpus=False, device=torch.device('cpu'), **kw)
    if nodes != 5:
        _randomize(net, 19 + nodes)
    # (5 nodes: the reference's own initialisation, which NAS() has applied -- with the wide random weights of _randomize the
    # six-input node's architecture gradients are chaotic: 2.5e-2 off at an oracle spread that one perturbation sample does
    # not catch; at the reference's scale the same network agrees to 4e-4, tools/diag_nodes5.py)
    with torch.no_grad():
        for p in net.arch_parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel() + nodes)) * 0.5)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    gio.share_stem(sd, 'net.')
    gen = torch.Generator().manual_seed(20 + nodes)
    x = torch.randn(2, 1, 32, 32, generator=gen)
    y = torch.randint(0, 2, (2, 32, 32), generator=gen)
    ref = R.nas_forward(sd, x, depth=3, nodes=nodes)[-1]
    ref_loss = R.dice_ce_loss(ref, y)
    ref_loss.backward()
    net = net.to(dev()).train()
    was = Cell.stacked
    Cell.stacked = stacked
    try:
        out = net(x.to(dev()))
        loss = SegmentationLosses('dice_ce')(out, y.to(dev()))
        loss.backward()
    finally:
        Cell.stacked = was
    close(out[-1], ref.detach().numpy(), 'logits', rel=1e-3)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * abs(float(ref_loss.detach()))
    got = grads_of(net)
    keys = ('alphas_dn', 'alphas_up', 'alphas_dn_nm', 'alphas_up_nm', 'betas_dn', 'betas_up', 'net.stem0.0.weight')
    # the oracle's own conditioning: the same pass with inputs and weights perturbed by 1e-6 (what another summation order
    # amounts to) -- the deeper the cell, the further its architecture gradients move (5 nodes: a few 1e-2)
    g = torch.Generator().manual_seed(77)
    sd2 = {k: ((v.detach() * (1 + 1e-6 * torch.randn(v.shape, generator=g))).requires_grad_(True)
               if (v.is_floating_point() and 'running' not in k) else v.detach().clone()) for k, v in sd.items()}
    gio.share_stem(sd2, 'net.')
    R.dice_ce_loss(R.nas_forward(sd2, x * (1 + 1e-6 * torch.randn(x.shape, generator=g)), depth=3, nodes=nodes)[-1], y).backward()
    worst = {}
    for k in keys:
        e = sd[k].grad.numpy().astype(np.float64)
        err = float(np.sqrt(((got[k] - e) ** 2).sum()) / np.sqrt((e ** 2).sum()))
        spread = float(np.sqrt(((sd2[k].grad.numpy().astype(np.float64) - e) ** 2).sum()) / np.sqrt((e ** 2).sum()))
        worst[k] = (err, spread)
        assert err <= max(1e-2, 4 * spread), 'grad %s: L2 rel err %.2e, the oracle itself moves by %.2e under 1e-6 noise' % (k, err, spread)
    from conftest import record_margin
    record_margin('test_supernet_other_node_counts_vs_oracle[%d-%s]' % (nodes, stacked),
                  **{k: {'gpu_vs_oracle': a, 'oracle_spread_under_1e-6': b} for k, (a, b) in worst.items()})
    ref_geno = R.derive_genotype(sd, depth=3, nodes=nodes)
    got_geno = net.genotype()
    assert (list(got_geno.down), list(got_geno.up)) == (list(ref_geno.down), list(ref_geno.up))


def test_blend2_vs_torch():
    """senas_blend2_fwd / _bwd (gamma-gated blend of two skip candidates) against float64 torch."""
    from senas_amd import functional as F
    gen = torch.Generator().manual_seed(31)
    x1, x2, w = (torch.randn(3, 32, 20, 12, generator=gen) for _ in range(3))
    gam = torch.randn(6, 2, generator=gen)
    a64, b64, g64 = x1.double().requires_grad_(True), x2.double().requires_grad_(True), gam.double().requires_grad_(True)
    row = torch.softmax(g64, -1)[4]
    ((a64 * row[0] + b64 * row[1]) * w.double()).sum().backward()
    cl = torch.channels_last
    a, b = (t.to(dev()).contiguous(memory_format=cl).requires_grad_(True) for t in (x1, x2))
    g = gam.to(dev()).requires_grad_(True)
    y = F.blend2(a, b, torch.softmax(g, -1)[4])
    (y * w.to(dev())).sum().backward()
    with torch.no_grad():
        r = torch.softmax(g64, -1)[4]
        close(y, (x1.double() * r[0] + x2.double() * r[1]).numpy(), 'blend', rel=1e-5)
    close(a.grad, a64.grad.numpy(), 'dx1', rel=1e-5)
    close(b.grad, b64.grad.numpy(), 'dx2', rel=1e-5)
    close(g.grad, g64.grad.numpy(), 'dgamma', rel=1e-4)


@pytest.mark.parametrize('case', [(2, 2, 32, 12, 20), (3, 4, 32, 16, 16), (4, 1, 8, 9, 7), (6, 2, 16, 5, 5), (8, 1, 64, 8, 4), (4, 4, 32, 128, 128)])
def test_skip_stack_vs_torch(case):
    """senas_skipcat_fwd / _bwd -- in0 of a supernet up cell: the column's down-path output and the gamma-gated blends of
    neighbouring outputs, concatenated (search/senas_search.py:96-103) -- against the reference's own composition in float64
    torch (output, every input gradient, d gamma); bit-equal to the blend + torch.cat composition it replaces in the forward
    direction; and with one input that wants no gradient."""
    from senas_amd import functional as F
    from senas_amd.grid import gamma_index
    m, n, c, h, w = case
    gen = torch.Generator().manual_seed(sum(case))
    xs = [torch.randn(n, c, h, w, generator=gen) for _ in range(m)]
    wgt = torch.randn(n, m * c, h, w, generator=gen)
    gam = torch.randn(sum(range(m + 1)), 2, generator=gen)
    idx = [0] + [gamma_index(k, 1) for k in range(1, m)]
    xs64 = [x.double().requires_grad_(True) for x in xs]
    g64 = gam.double().requires_grad_(True)
    t64 = torch.softmax(g64, -1)
    y64 = torch.cat([xs64[0]] + [t64[idx[k], 0] * xs64[k - 1] + t64[idx[k], 1] * xs64[k] for k in range(1, m)], dim=1)
    (y64 * wgt.double()).sum().backward()
    cl = torch.channels_last
    for frozen in (None, m - 1):
        gx = [x.to(dev()).contiguous(memory_format=cl).requires_grad_(k != frozen) for k, x in enumerate(xs)]
        g = gam.to(dev()).requires_grad_(True)
        rows = F.GammaRows(torch.softmax(g, -1))
        y = F.skip_stack(gx, rows, idx)
        assert y.shape == (n, m * c, h, w) and y.is_contiguous(memory_format=cl)
        (y * wgt.to(dev())).sum().backward()
        close(y, y64.detach().numpy(), 'stack', rel=1e-5)
        for k in range(m):
            if k == frozen:
                assert gx[k].grad is None
            else:
                close(gx[k].grad, xs64[k].grad.numpy(), 'dx%d' % k, rel=1e-5)
        close(g.grad, g64.grad.numpy(), 'dgamma', rel=1e-4)
        with torch.no_grad():
            old = torch.cat([gx[0]] + [F.blend2_row(gx[k - 1], gx[k], rows, idx[k]) for k in range(1, m)], dim=1)
        assert torch.equal(old, y)


@pytest.mark.parametrize('case', [(3, 2, 32, 8, 12, 20, True, False), (6, 4, 32, 8, 16, 16, True, True), (2, 2, 8, 8, 9, 7, True, False),
                                  (8, 1, 64, 4, 8, 8, True, True), (1, 3, 16, 8, 5, 5, True, False), (4, 2, 32, 8, 8, 12, False, False),
                                  (6, 4, 32, 8, 64, 64, True, False),
                                  # one channel quad (every lane of a row folds into one), 12 problems, 8 images (an accumulator each)
                                  (2, 2, 4, 4, 6, 10, True, False), (12, 2, 8, 8, 16, 16, True, False), (3, 8, 16, 8, 12, 12, True, True)])
def test_dstail_vs_torch(case):
    """senas_dstail_fwd / _bwd -- BatchNorm2d + ReLU + 1x1 convolution of k DepSepConv candidates as one pass, the activated
    tensor recomputed instead of stored -- against float64 torch: outputs, producer statistics, running buffers, dz1,
    dgamma1, dbeta1, dW (folded by the last block); with the upstream gradient arriving as a channel slice (strided)."""
    from senas_amd import functional as F
    k, n, cin, cout, h, w, training, strided = case
    gen = torch.Generator().manual_seed(sum(case[:6]))
    bns = [nn.BatchNorm2d(cin) for _ in range(k)]
    convs = [nn.Conv2d(cin, cout, 1, bias=False) for _ in range(k)]
    zs, want = [], []
    wide = [torch.randn(n, 3 * cout, h, w, generator=gen) for _ in range(k)]
    for bn, conv, wd in zip(bns, convs, wide):
        with torch.no_grad():
            bn.weight.copy_(1.0 + 0.3 * torch.randn(cin, generator=gen))
            bn.bias.copy_(0.3 * torch.randn(cin, generator=gen))
            bn.running_mean.copy_(0.2 * torch.randn(cin, generator=gen))
            bn.running_var.copy_(0.5 + torch.rand(cin, generator=gen))
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (1.0 / cin ** 0.5))
        bn.train(training)
        z = (0.3 + torch.randn(n, cin, h, w, generator=gen)) * 1.5
        zs.append(z)
        rbn, rconv = nn.BatchNorm2d(cin).double(), nn.Conv2d(cin, cout, 1, bias=False).double()
        rbn.load_state_dict({kk: (v.double() if v.is_floating_point() else v.clone()) for kk, v in bn.state_dict().items()})
        rconv.weight.data.copy_(conv.weight.double())
        rbn.train(training)
        zz = z.double().requires_grad_(True)
        y = rconv(torch.relu(rbn(zz)))
        (y * wd[:, cout:2 * cout].double()).sum().backward()
        want.append((y.detach(), zz.grad, rbn.weight.grad, rbn.bias.grad, rconv.weight.grad, rbn.running_mean, rbn.running_var))
    dbns, dconvs = [bn.to(dev()) for bn in bns], [c.to(dev()) for c in convs]
    z_in = [z.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True) for z in zs]
    if not training:
        assert F.dstail(z_in, dbns, [None] * k, dconvs, False) is None      # eval mode only without autograd
        with torch.no_grad():
            outs = F.dstail(z_in, dbns, [None] * k, dconvs, False)
        for (y, st), wnt in zip(outs, want):
            assert st is None
            close(y, wnt[0].numpy(), 'eval output', rel=2e-5)
        return
    outs = F.dstail(z_in, dbns, [None] * k, dconvs, True)
    loss = 0
    for (y, st), wd in zip(outs, wide):
        wdev = wd.to(dev()).contiguous(memory_format=torch.channels_last)
        if strided:
            cat = torch.cat([torch.zeros_like(y), y, torch.zeros_like(y)], dim=1)        # dy of y = a channel slice of d cat
            loss = loss + (cat * wdev).sum()
        else:
            loss = loss + (y * wdev[:, cout:2 * cout]).sum()
    loss.backward()
    for t in range(k):
        y, dz, dg, db, dw, rm, rv = want[t]
        close(outs[t][0], y.numpy(), 'output %d' % t, rel=2e-5)
        exp = torch.stack([y.sum((2, 3)), (y ** 2).sum((2, 3))], -1)
        close(outs[t][1].float(), exp.float().numpy(), 'stats %d' % t, rel=2e-5)
        close(z_in[t].grad, dz.numpy(), 'dz1 %d' % t, rel=2e-4)
        close(dbns[t].weight.grad, dg.numpy(), 'dgamma %d' % t, rel=2e-4)
        close(dbns[t].bias.grad, db.numpy(), 'dbeta %d' % t, rel=2e-4)
        close(dconvs[t].weight.grad, dw.numpy(), 'dW %d' % t, rel=2e-4)
        close(dbns[t].running_mean, rm.numpy(), 'running_mean %d' % t, rel=2e-5)
        close(dbns[t].running_var, rv.numpy(), 'running_var %d' % t, rel=2e-5)
        assert int(dbns[t].num_batches_tracked) == 1


@pytest.mark.parametrize('cfg', [dict(n=4, c=8, h=16, w=16, T=12, se=True, zero=True, training=True, relu=True),
                                 dict(n=4, c=8, h=64, w=64, T=24, se=True, zero=True, training=True, relu=True),
                                 dict(n=3, c=8, h=9, w=7, T=18, se=True, zero=False, training=False, relu=True),
                                 dict(n=2, c=32, h=8, w=8, T=12, se=True, zero=True, training=True, relu=False),
                                 dict(n=5, c=12, h=5, w=6, T=9, se=False, zero=False, training=True, relu=True)],
                         ids=lambda d: 'n%d_c%d_%dx%d_T%d%s' % (d['n'], d['c'], d['h'], d['w'], d['T'], '' if d['training'] else '_eval'))
def test_wide_node_is_the_two_launch_node_bit_for_bit(cfg):
    """csrc/node.hip node_wide_fwd_kernel (many-term nodes on small maps: preparation as a prologue of the one launch) against
    the prepare + combine pair it replaces there (SENAS_NODE_WIDE=0): output, ReLU-masked gradients of every input, running
    statistics and step counters -- the forward pass bit for bit, the gradients (fp64 atomics in the backward reduce) to 1e-6."""
    import copy
    import torch.nn as nn
    from senas_amd import functional as F
    from senas_amd.operations import SEBlock
    g = torch.Generator().manual_seed(77)
    n, c, h, w, T = cfg['n'], cfg['c'], cfg['h'], cfg['w'], cfg['T']
    zs = [torch.randn(n, c, h, w, generator=g) * (0.5 + 0.1 * t) + 0.05 * t for t in range(T)]
    if cfg['zero']:
        zs[1] = None
    bns = []
    for t in range(T):
        bn = nn.BatchNorm2d(c)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        bn.train(cfg['training'])
        bns.append(bn)
    ses = [SEBlock(c) if (cfg['se'] and t % 3 == 0 and zs[t] is not None) else None for t in range(T)]
    mix = torch.rand(T, generator=g) + 0.1
    gy = torch.randn(n, c, h, w, generator=g)
    res = []
    keep = os.environ.get('SENAS_NODE_WIDE')
    # the producer-side statistics once, for both forms (they are accumulated with fp64 atomics: two runs differ in their last bits)
    stats = [F.chan_stats(z.to(dev()).contiguous(memory_format=torch.channels_last)) if z is not None else None for z in zs]
    torch.cuda.synchronize()
    try:
        for mode in ('1', '0'):
            os.environ['SENAS_NODE_WIDE'] = mode
            dbns = [copy.deepcopy(b).to(dev()) for b in bns]
            dses = [copy.deepcopy(s_).to(dev()) if s_ is not None else None for s_ in ses]
            zd = [z.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True) if z is not None else None for z in zs]
            md = mix.to(dev()).requires_grad_(True)
            out = F.bn_combine([F.Term(zd[t], dbns[t], se=dses[t], stats=stats[t]) for t in range(T)], mix=md, relu=cfg['relu'])
            out.backward(gy.to(dev()))
            torch.cuda.synchronize()
            res.append((out.detach().clone(), [b.running_mean.clone() for b in dbns], [b.running_var.clone() for b in dbns],
                        [int(b.num_batches_tracked) for b in dbns], [z.grad.clone() for z in zd if z is not None], md.grad.clone(),
                        [b.weight.grad.clone() for b in dbns]))
    finally:
        if keep is None:
            os.environ.pop('SENAS_NODE_WIDE', None)
        else:
            os.environ['SENAS_NODE_WIDE'] = keep
    a, b = res
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    for x, y_ in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.equal(x, y_)
    assert a[3] == b[3]
    for x, y_ in zip(a[4] + [a[5]] + a[6], b[4] + [b[5]] + b[6]):
        assert float((x - y_).abs().max()) <= 1e-6 * (float(y_.abs().max()) + 1e-12)


@pytest.mark.parametrize('kind,size', [('up', 32), ('down', 32), ('up', 8), ('down', 64)])
def test_planar_stacked_outputs_change_nothing_but_the_layout(kind, size):
    """The stacked convolutions of a search cell with their 8-channel parts written PLANAR (senas_conv2d_fwd_planar: every edge's
    slice a dense tensor, read in full lines by its node kernel) against the interleaved form (functional.PLANAR = False): the
    same kernels compute the same values -- the cell's output bit for bit, every gradient to the order of atomics (1e-6)."""
    from senas_amd import functional as F
    from senas_amd.cell import Cell
    torch.manual_seed(5)
    cell = Cell(3, 1, 32 if kind == 'down' else 64, 32, 32, kind).to(dev()).train()
    g = torch.Generator().manual_seed(6)
    if kind == 'down':
        in0 = torch.randn(2, 32, 2 * size, 2 * size, generator=g)
        in1 = torch.randn(2, 32, size, size, generator=g)
    else:
        in0 = torch.randn(2, 64, size, size, generator=g)
        in1 = torch.randn(2, 32, size // 2, size // 2, generator=g)
    nops = 6
    k = sum(2 + i for i in range(3))
    w_norm = torch.softmax(torch.randn(k, nops, generator=g), -1).to(dev())
    w_chg = torch.softmax(torch.randn(k, nops, generator=g), -1).to(dev())
    betas = torch.softmax(torch.randn(k, generator=g), -1).to(dev())
    res = []
    keep = F.PLANAR
    try:
        for planar in (True, False):
            F.PLANAR = planar
            F.PLANAR_LAUNCHES[0] = 0
            for p in cell.parameters():
                p.grad = None
            a = in0.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            b = in1.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wn, wc, bt = (t.clone().requires_grad_(True) for t in (w_norm, w_chg, betas))
            out = cell(a, b, wn, wc, bt)
            out.backward(torch.ones_like(out) * 0.01 + out.detach() * 0.1)
            torch.cuda.synchronize()
            res.append((F.PLANAR_LAUNCHES[0], out.detach().clone(), a.grad.clone(), b.grad.clone(), wn.grad.clone(), wc.grad.clone(), bt.grad.clone(),
                        {k_: p.grad.clone() for k_, p in cell.named_parameters() if p.grad is not None}))
    finally:
        F.PLANAR = keep
    on, off = res
    assert on[0] >= 4 and off[0] == 0, (on[0], off[0])                    # the dilated pairs and the SE stacks of both input states
    assert torch.equal(on[1], off[1]), float((on[1] - off[1]).abs().max())
    for x_, y_ in zip(on[2:7], off[2:7]):
        assert float((x_ - y_).abs().max()) <= 1e-6 * (float(y_.abs().max()) + 1e-12)
    assert set(on[7]) == set(off[7])
    top = max(float(v.abs().max()) for v in off[7].values())
    for k_, v in off[7].items():
        assert float((on[7][k_] - v).abs().max()) <= 1e-6 * max(float(v.abs().max()), 1e-3 * top), k_


@pytest.mark.parametrize('cfg', [dict(n=4, c=32, h=16, w=16, T=2, se=False, zero=False, training=True, relu=True, res=False, mix=False),
                                 dict(n=8, c=32, h=128, w=128, T=2, se=False, zero=False, training=True, relu=True, res=False, mix=False),
                                 dict(n=3, c=8, h=9, w=7, T=4, se=True, zero=True, training=True, relu=True, res=True, mix=True),
                                 dict(n=2, c=32, h=33, w=20, T=1, se=False, zero=False, training=True, relu=False, res=False, mix=False),
                                 dict(n=5, c=6, h=5, w=6, T=3, se=False, zero=False, training=True, relu=True, res=True, mix=True),
                                 dict(n=2, c=64, h=12, w=12, T=2, se=True, zero=False, training=False, relu=True, res=False, mix=True)],
                         ids=lambda d: 'n%d_c%d_%dx%d_T%d%s' % (d['n'], d['c'], d['h'], d['w'], d['T'], '' if d['training'] else '_eval'))
def test_fused_apply_is_the_three_launch_backward_bit_for_bit(cfg):
    """csrc/node.hip node_apply_fused_kernel (few-term nodes: the backward preparation as a prologue of the apply launch, the
    thread's first element requested before it) against reduce + prepare + apply (SENAS_NODE_FUSED_APPLY=0) on the SAME reduced
    sums: every input gradient, the residual's, every batch-norm / SE / mixing gradient -- bit for bit (the two forms run the same
    device functions on the same operands; the reduce launch in front of both is shared, so its atomics do not enter)."""
    import copy
    import torch.nn as nn
    from senas_amd import functional as F
    from senas_amd.operations import SEBlock
    g = torch.Generator().manual_seed(91)
    n, c, h, w, T = cfg['n'], cfg['c'], cfg['h'], cfg['w'], cfg['T']
    zs = [torch.randn(n, c, h, w, generator=g) * (0.5 + 0.2 * t) + 0.05 * t for t in range(T)]
    if cfg['zero'] and T > 1:
        zs[1] = None
    bns = []
    for t in range(T):
        bn = nn.BatchNorm2d(c)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        bn.train(cfg['training'])
        bns.append(bn)
    ses = [SEBlock(c) if (cfg['se'] and t % 2 == 0 and zs[t] is not None) else None for t in range(T)]
    mix = torch.rand(T, generator=g) + 0.1 if cfg['mix'] else None
    resid = torch.randn(n, c, h, w, generator=g) if cfg['res'] else None
    gy = torch.randn(n, c, h, w, generator=g)
    cl = lambda t: t.to(dev()).contiguous(memory_format=torch.channels_last)
    stats = [F.chan_stats(cl(z)) if z is not None else None for z in zs]
    torch.cuda.synchronize()
    res = []
    keep = os.environ.get('SENAS_NODE_FUSED_APPLY')
    try:
        for mode in ('1', '0'):
            os.environ['SENAS_NODE_FUSED_APPLY'] = mode
            dbns = [copy.deepcopy(b).to(dev()) for b in bns]
            dses = [copy.deepcopy(s_).to(dev()) if s_ is not None else None for s_ in ses]
            zd = [cl(z).requires_grad_(True) if z is not None else None for z in zs]
            md = mix.to(dev()).requires_grad_(True) if mix is not None else None
            rd = cl(resid).requires_grad_(True) if resid is not None else None
            out = F.bn_combine([F.Term(zd[t], dbns[t], se=dses[t], stats=stats[t]) for t in range(T)], mix=md, residual=rd, relu=cfg['relu'])
            out.backward(cl(gy))
            torch.cuda.synchronize()
            grads = [z.grad.clone() for z in zd if z is not None] + [b.weight.grad.clone() for b in dbns] + [b.bias.grad.clone() for b in dbns]
            grads += [s_.excitation[k].weight.grad.clone() for s_ in dses if s_ is not None for k in (0, 2)]
            if md is not None:
                grads.append(md.grad.clone())
            if rd is not None:
                grads.append(rd.grad.clone())
            res.append((out.detach().clone(), grads))
    finally:
        if keep is None:
            os.environ.pop('SENAS_NODE_FUSED_APPLY', None)
        else:
            os.environ['SENAS_NODE_FUSED_APPLY'] = keep
    (o1, g1), (o0, g0) = res
    assert torch.equal(o1, o0) and len(g1) == len(g0) and len(g1) >= T
    for i, (a, b) in enumerate(zip(g1, g0)):
        # the reduce launch's fp64 atomics differ in their last bits from run to run: what comes out in fp32 is compared to 2e-7 of
        # the tensor scale (one ulp), and must be EQUAL wherever the reduced sums were
        assert float((a - b).abs().max()) <= 2e-7 * (float(b.abs().max()) + 1e-30), (i, float((a - b).abs().max()), float(b.abs().max()))
