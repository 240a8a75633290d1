"""GPU checks of the macro grid on several HIP streams (senas_amd/grid.py: Lanes), of the lane scheduler that replays the
captured passes (csrc/sched.hip, senas_amd/lanesched.py) and of the fused architecture tables against the torch path.
The reference walks the grid cell by cell on one stream (search/senas_search.py:96-107, models/senas_model.py:160-175);
every schedule that respects the data flow must give its results."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _batch(n=2, size=64, seed=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 1, size, size, generator=g).to(dev()), torch.randint(0, 2, (n, size, size), generator=g).to(dev())


def _make(kind):
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    torch.manual_seed(0)
    if kind == 'nas.c8.d5':
        return NAS(1, 8, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev()).train()
    if kind == 'nas.c32.d4.sup.share':
        return NAS(1, 32, 2, 4, meta_node_num=3, use_sharing=True, double_down_channel=False, supervision=True).to(dev()).train()
    if kind == 'nas.c8.d3.dd':
        return NAS(1, 8, 2, 3, meta_node_num=3, use_sharing=False, double_down_channel=True).to(dev()).train()
    if kind == 'derived.c32.d5':
        return SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4).to(dev()).train()
    if kind == 'derived.c8.d4.all':
        return SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4._replace(gamma=[1] * 6), supervision=True).to(dev()).train()
    raise KeyError(kind)


def _pass(net, crit, x, y):
    for p in net.parameters():
        p.grad = None
    out = net(x)
    crit(out, y).backward()
    torch.cuda.synchronize()
    return [o.detach().clone() for o in out], {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


@pytest.fixture
def lanes_switch():
    from senas_amd import grid
    keep = grid.Lanes.enabled
    yield grid.Lanes
    grid.Lanes.enabled = keep


@pytest.mark.parametrize('kind', ['nas.c8.d5', 'nas.c32.d4.sup.share', 'nas.c8.d3.dd', 'derived.c32.d5', 'derived.c8.d4.all'])
def test_lanes_give_the_serial_schedules_results(kind, lanes_switch):
    """Columns of up cells on their own streams against every cell on the caller's stream: identical logits (no atomics
    in the forward pass), every gradient equal up to the summation order of atomics."""
    from senas_amd.loss import MultiSegmentationLosses, SegmentationLosses
    net = _make(kind)
    crit = MultiSegmentationLosses('dice_ce', len(net(_batch()[0]))) if ('sup' in kind or 'all' in kind) else SegmentationLosses('dice_ce')
    x, y = _batch()
    state = copy.deepcopy(net.state_dict())
    res = []
    for on in (False, True, True):
        lanes_switch.enabled = on
        net.load_state_dict(state)
        res.append(_pass(net, crit, x, y))
    (o0, g0), (o1, g1), (o2, g2) = res
    assert set(g0) == set(g1) == set(g2) and len(g0) > 100
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    for ga, gb in ((g0, g1), (g1, g2)):
        worst = max(float((ga[k] - gb[k]).abs().max() / (ga[k].abs().max() + 1e-30)) for k in ga)
        assert worst < 2e-5, worst


def test_lane_scheduler_replays_a_captured_multi_stream_graph():
    """csrc/sched.hip on a small hand-made capture: two side streams forked off the capture stream, a hand-over through it,
    a memset node, a join -- replayed by the scheduler (never instantiated by the runtime) on fresh inputs."""
    from senas_amd.lanesched import LaneSchedule
    a = torch.zeros(1 << 16, device=dev())
    out = torch.zeros_like(a)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main, capture_error_mode='thread_local'):
            b = a * 2.0
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                c = torch.sin(b) + 1.0
                for _ in range(5):
                    c = c * 1.5 - 0.25
            with torch.cuda.stream(s2):
                d = torch.zeros_like(b)              # a memset node
                d = d + b * b
            main.wait_stream(s1)                     # hand c over to s2 through the origin stream
            s2.wait_stream(main)
            with torch.cuda.stream(s2):
                e = d - c
            main.wait_stream(s2)
            out.copy_(e + b)
    sched = LaneSchedule(g, max_lanes=3)
    info = sched.info()
    assert info['lanes'] >= 2 and info['kernel_nodes'] >= 10 and info['segments'] >= 3, info
    for seed in (1, 2, 3):
        a.copy_(torch.randn(a.shape, generator=torch.Generator().manual_seed(seed)).to(dev()))
        with torch.cuda.stream(main):
            sched.launch()
        torch.cuda.synchronize()
        b_ = a * 2.0
        c_ = torch.sin(b_) + 1.0
        for _ in range(5):
            c_ = c_ * 1.5 - 0.25
        want = (b_ * b_) - c_ + b_
        assert torch.allclose(out, want, rtol=1e-6, atol=1e-6), float((out - want).abs().max())
    sched.close()


@pytest.mark.parametrize('kind', ['search', 'train'])
def test_step_drivers_on_lanes_track_the_serial_eager_step(kind, lanes_switch):
    """The captured step (lanes + lane scheduler + weight-gradient lane) against the same driver run eagerly on one stream.
    After ONE optimizer step over ONE pass nothing has been amplified yet: every tensor's UPDATE (clip, SGD with momentum and
    weight decay on the captured pass's gradients) must agree to 1e-4 of the update's scale (the order of atomics in the
    gradients: ~5e-6; a tensor whose update is below 1 % of the largest one is held on that scale; one unit in the last
    place of the stored fp32 weight, through which the update is read, is taken off first), batch-norm running
    statistics to 1e-5.  For the search driver that first step is the weight step alone (before ``alpha_begin``,
    experiments/search_arc.py:262-266); its architecture pass on lanes is held to 5e-5 by the gradient test below.  A full
    search step is two passes with an optimizer between them: the second pass's forward already sees architecture weights
    that differ in their last bits, and a piecewise-linear network turns that into 1e-3 on a tensor with a small gradient
    (measured: 1.5e-8 absolute on an update of 1.4e-5) -- so the steps after the first are held by their losses (2e-5): the
    drivers keep working on the same trajectory."""
    from conftest import record_margin
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep, TrainStep
    crit = SegmentationLosses('dice_ce')
    x, y = _batch(2, 64, 5)
    xv, yv = _batch(2, 64, 6)
    runs = []
    for lanes, graphed in ((False, False), (True, True)):
        lanes_switch.enabled = lanes
        torch.manual_seed(1)
        if kind == 'search':
            # (c = 32, the benchmark's width: the c = 8 network's 2-channel inner edges run on the fallback kernels that add with fp32
            # atomics, whose order alone moved a 100-element weight gradient by 1.4e-4 of the bound's scale between two runs)
            net = NAS(1, 32, 2, 4, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev()).train()
            ow = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
            # (plain SGD on the architecture tensors: Adam's normalised step turns the last bits of a near-zero gradient into a
            # full step -- the trajectory tests against the oracle use Adam)
            oa = torch.optim.SGD(net.arch_parameters(), lr=1e-2)
            drv = SearchStep(net, crit, ow, oa, x.clone(), y.clone(), use_graph=graphed)
            step = lambda first=[True]: drv(x, y, *(() if first.pop() else (xv, yv))) if first else drv(x, y, xv, yv)
            sched = drv.fb.sched
        else:
            net = SenasModel(2, 1, c=32, depth=4, genotype=senas_node_4).to(dev()).train()
            opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
            drv = TrainStep(net, crit, opt, x, y, use_graph=graphed)
            step = drv
            sched = drv.fb.sched
        if graphed:
            assert sched is not None and sched.info()['lanes'] >= 2, 'the captured pass does not run on the lane scheduler'
        start = {k: v.detach().clone() for k, v in net.state_dict().items()}
        losses = [float(step())]
        torch.cuda.synchronize()
        after1 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        losses += [float(step()) for _ in range(2)]
        torch.cuda.synchronize()
        runs.append((losses, start, after1))
        drv.close()
    (l0, b0, s0), (l1, b1, s1) = runs
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-5 * abs(a), (l0, l1)
    params = set(k for k, _ in net.named_parameters())
    upd0 = {k: s0[k] - b0[k] for k in s0 if k in params}
    upd1 = {k: s1[k] - b1[k] for k in s1 if k in params}
    top = max(float(v.abs().max()) for v in upd0.values())
    assert top > 0 and len(upd0) > 100
    worst = (0.0, '')
    for k, u in upd0.items():
        assert bool(torch.equal(b0[k], b1[k])), k                        # (same start)
        scale = max(float(u.abs().max()), 1e-2 * top)
        # The update is read off as (weight after) - (weight before), both fp32: ONE unit in the last place of a weight of
        # magnitude 0.1 is 7.5e-9, 1.6e-4 of an update of 4.7e-5 -- which is what the gradients' agreement to 2e-7
        # (tools/diag_lanes_vs_serial.py 32) turns into whenever it tips a rounding.  That one unit of the stored weight is
        # taken off before the comparison; what the schedules may differ in beyond it stays held to 1e-4.
        ulp = torch.finfo(torch.float32).eps * s0[k].abs()
        err = float(((u - upd1[k]).abs() - ulp).clamp_min(0).max()) / scale
        worst = (err, k) if err > worst[0] else worst
        assert err <= 1e-4, (k, err, scale)
    for k in s0:
        if k not in params and s0[k].is_floating_point():               # running statistics of every BatchNorm2d
            scale = float(s0[k].abs().max()) + 1e-12
            assert float((s0[k] - s1[k]).abs().max()) <= 1e-5 * scale + 1e-9, (k, float((s0[k] - s1[k]).abs().max()), scale)
    record_margin('test_step_drivers_on_lanes_track_the_serial_eager_step[%s]' % kind, tensors=len(upd0), bound=1e-4,
                  worst_update_error=worst[0], worst_tensor=worst[1], losses_serial=l0, losses_lanes=l1)


def test_a_lane_scheduled_pass_gives_the_serial_gradients(lanes_switch):
    """ONE pass (no optimizer in between): every parameter gradient of the captured weight pass on lanes -- lane scheduler, weight
    gradients on their own lane -- against the same pass launched eagerly on one stream: 5e-5 of the tensor scale (measured
    5.5e-6: the order of atomics), all 2 252 tensors of a depth-4 supernet.  The ARCHITECTURE pass likewise (its own capture,
    its own schedule, weights frozen, no weight-gradient lane): its seven gradients, same bound."""
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep
    x, y = _batch(2, 64, 5)
    res, res_arch = {}, {}
    for name, lanes, graphed in (('serial-eager', False, False), ('lanes-graph', True, True)):
        lanes_switch.enabled = lanes
        torch.manual_seed(1)
        net = NAS(1, 8, 2, 4, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev()).train()
        with torch.no_grad():                           # (architecture tensors away from their 1e-3 initialisation: gradients of every size)
            for p in net.arch_parameters():
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev()) * 0.3)
        ow = torch.optim.SGD(net.parameters(), lr=0.0)
        oa = torch.optim.SGD(net.arch_parameters(), lr=0.0)
        drv = SearchStep(net, SegmentationLosses('dice_ce'), ow, oa, x.clone(), y.clone(), grad_clip=0.0, use_graph=graphed)
        assert (drv.fb.sched is not None) == graphed and (drv.fb_arch.sched is not None) == graphed
        for _ in range(2):
            drv.fb_arch()
        torch.cuda.synchronize()
        res_arch[name] = [p.grad.detach().clone() for p in net.arch_parameters()]
        for _ in range(2):
            drv.fb()
        torch.cuda.synchronize()
        res[name] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        drv.close()
    base, got = res['serial-eager'], res['lanes-graph']
    assert set(base) == set(got) and len(base) > 2000
    top = max(float(v.abs().max()) for v in base.values())
    for k, v in base.items():
        scale = max(float(v.abs().max()), 1e-3 * top)
        assert float((got[k] - v).abs().max()) <= 5e-5 * scale, (k, float((got[k] - v).abs().max()), scale)
    names = ['alphas_dn', 'alphas_up', 'alphas_dn_nm', 'alphas_up_nm', 'betas_dn', 'betas_up', 'gamma']
    top = max(float(v.abs().max()) for v in res_arch['serial-eager'])
    assert top > 0
    for k, a, b in zip(names, res_arch['serial-eager'], res_arch['lanes-graph']):
        scale = max(float(a.abs().max()), 1e-3 * top)
        assert float(a.abs().max()) > 0, k
        assert float((a - b).abs().max()) <= 5e-5 * scale, ('architecture pass', k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize('c,size', [(8, 64), (32, 64)])
def test_replays_of_a_lane_scheduled_pass_agree(c, size):
    """Every replay of the captured passes of a depth-5 search step gives the first replay's gradients (to the order of atomics).
    A dependency lost between capture and replay does not show in the FIRST replay -- it still finds the warm-up pass's values
    in memory -- but as garbage from the second on: found in round 4 when the scheduler cut the relay markers' chain and with it
    the origin stream's own waits (csrc/sched.hip, note at relay_marker_kernel; c = 8: the 2-channel inner edges of the first
    down cell, whose weight gradients are summed on the origin stream at the end of the pass)."""
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    from senas_amd.step import SearchStep
    torch.manual_seed(1)
    net = NAS(1, c, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev()).train()
    x, y = _batch(2, size, 5)
    ow = torch.optim.SGD(net.parameters(), lr=0.0)
    oa = torch.optim.SGD(net.arch_parameters(), lr=0.0)
    drv = SearchStep(net, SegmentationLosses('dice_ce'), ow, oa, x.clone(), y.clone(), grad_clip=0.0)
    assert drv.fb.sched is not None and drv.fb_arch.sched is not None
    for fb in (drv.fb_arch, drv.fb):
        runs = []
        for _ in range(4):
            fb()
            torch.cuda.synchronize()
            runs.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        top = max(float(v.abs().max()) for v in runs[0].values())
        assert top > 0 and all(bool(torch.isfinite(v).all()) for v in runs[0].values())
        for r in runs[1:]:
            for k, v in runs[0].items():
                scale = max(float(v.abs().max()), 1e-3 * top)
                assert float((r[k] - v).abs().max()) <= 1e-4 * scale, (k, float((r[k] - v).abs().max()), scale)
    drv.close()


def test_a_network_with_dropout_keeps_the_serial_schedule():
    """torch's graph replay advances the Philox offsets of captured dropout draws; the lane scheduler replays launches as they
    are -- so a driver over a network with dropout captures one stream and uses torch's replay."""
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.step import TrainStep
    torch.manual_seed(2)
    net = SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4, dropout_prob=0.2).to(dev()).train()
    x, y = _batch(2, 64, 7)
    drv = TrainStep(net, SegmentationLosses('dice_ce'), torch.optim.SGD(net.parameters(), lr=1e-3), x, y)
    assert drv.fb.sched is None and drv.fb.graph is not None
    a, b = float(drv()), float(drv())
    assert a == a and b == b and a != b            # finite, and a fresh mask per replay
    drv.close()


@pytest.mark.parametrize('sharing,depth,nodes', [(False, 4, 3), (True, 4, 3), (False, 2, 3), (True, 3, 4), (False, 3, 4), (False, 3, 5)])
def test_arch_tables_against_the_torch_path(sharing, depth, nodes):
    """senas_arch_mix_fwd / _bwd (all softmaxes, the overlapping beta windows, both mixing matrices, the seven parameter
    gradients -- search/senas_search.py:252-260, search/cell.py:33-36,100-106) against NAS._mixing_weights + _EdgeMix on the
    same network: tables to 1e-6, gradients to 1e-5 of their scale.  use_sharing: alphas_up_nm IS alphas_dn_nm (the NORM rows of
    both kinds land in one gradient); depth 2: gamma is empty."""
    from senas_amd import functional as F
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    torch.manual_seed(11)
    net = NAS(1, 32 if nodes == 5 else 8, 2, depth, meta_node_num=nodes, use_sharing=sharing, double_down_channel=False).to(dev()).train()
    with torch.no_grad():
        for p in net.arch_parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev()) * 0.5)
    crit = SegmentationLosses('dice_ce')
    x, y = _batch(2, 32 if depth > 2 else 16, 9)
    # tables
    tabs = F.ArchTables(nodes, net.alphas_dn, net.alphas_up, net.alphas_dn_nm, net.alphas_up_nm, net.betas_dn, net.betas_up, net.gamma).args()
    want = net._mixing_weights()
    for a, b in zip(tabs, want):
        assert a.shape == b.shape
        if a.numel():
            assert float((a.detach() - b.detach()).abs().max()) < 1e-6
    state = copy.deepcopy(net.state_dict())
    grads = []
    for plain in (True, False):
        net.load_state_dict(state)
        net._plain_arch = plain
        for p in net.parameters():
            p.grad = None
        crit(net(x), y).backward()
        torch.cuda.synchronize()
        grads.append([None if p.grad is None else p.grad.detach().clone() for p in net.arch_parameters()])
    net._plain_arch = False
    names = ['alphas_dn', 'alphas_up', 'alphas_dn_nm', 'alphas_up_nm', 'betas_dn', 'betas_up', 'gamma']
    for name, a, b in zip(names, *grads):
        if a is None or a.numel() == 0:
            assert b is None or b.numel() == 0 or float(b.abs().max()) == 0.0, name
            continue
        assert b is not None, name
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9, (name, float((a - b).abs().max()), scale)


def test_a_foreign_capture_of_the_forward_pass_stays_on_one_stream():
    """Somebody else's ``torch.cuda.graph`` around the network's forward pass (no step driver, no Evaluator): the columns must
    NOT fork inside that capture -- the runtime's own executor would be handed a multi-branch graph (SIGSEGV in
    hip::Graph::UpdateStreams, profiles/r4_graph_executor.txt).  The captured pass is single-branch, replays, and gives the
    eager pass's logits bit for bit."""
    from senas_amd import functional as F
    from senas_amd.grid import Lanes
    net = _make('derived.c32.d5')
    x, _ = _batch()
    assert Lanes.enabled
    with torch.no_grad():
        want = net(x)[-1].clone()                               # eager: on lanes
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        before = set(F.LANES)
        g = torch.cuda.CUDAGraph()
        xs = x.clone()
        with torch.cuda.graph(g):
            F.LANES.clear()
            out = net(xs)[-1]
            forked = len(F.LANES)
        F.LANES.update(before)
        assert forked == 0, 'the forward pass forked onto %d lanes inside a foreign capture' % forked
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


@pytest.mark.parametrize('graphed', [False, True])
def test_a_module_applied_twice_under_the_weight_gradient_lane(graphed):
    """One convolution applied TWICE in a pass under a step driver with lanes on: its first gradient goes into
    the parameter's view of the flat buffer through the weight-gradient lane's queue, its second one through autograd, which
    ADDS into the view -- the queued kernel, which OVERWRITES the view, must have run by then (functional.wgrad_dest flushes the
    queue on the current stream).  Every gradient against the same pass without a driver (plain autograd, serial schedule)."""
    from senas_amd import functional as F
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.grid import Lanes
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_model import SenasModel
    from senas_amd.step import TrainStep

    class Twice(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4)
            self.again = torch.nn.Conv2d(2, 2, 3, padding=1, bias=False)

        def forward(self, x):
            y = self.net(x)[-1]
            for _ in range(2):
                y = F.conv2d(y, self.again.weight, pad=1)[0]
            return [y]

    torch.manual_seed(3)
    model = Twice().to(dev()).train()
    x, y = _batch(2, 64, 8)
    crit = SegmentationLosses('dice_ce')
    keep = Lanes.enabled
    try:
        Lanes.enabled = False
        for p in model.parameters():
            p.grad = None
        crit(model(x), y).backward()
        torch.cuda.synchronize()
        want = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        Lanes.enabled = True
        drv = TrainStep(model, crit, torch.optim.SGD(model.parameters(), lr=0.0), x, y, grad_clip=0.0, use_graph=graphed)
        assert drv.fb.wlane is not None
        for _ in range(2):
            drv.fb()
        torch.cuda.synchronize()
        got = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        drv.close()
    finally:
        Lanes.enabled = keep
    top = max(float(v.abs().max()) for v in want.values())
    assert float(want['again.weight'].abs().max()) > 0
    for k, v in want.items():
        scale = max(float(v.abs().max()), 1e-3 * top)
        assert float((got[k] - v).abs().max()) <= 5e-5 * scale, (k, float((got[k] - v).abs().max()), scale)
