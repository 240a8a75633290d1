"""CPU-side checks (run in the `-m "not gpu"` suite): the C-ABI library loads and exports every
symbol include/senas_hip.h declares, the module tree emits the reference's state_dict keys, the
genotype logic is bit-exact, and the product path refuses to compute without a GPU."""
import json
import os
import re

import numpy as np
import pytest
import torch

import golden_io as gio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = None


def _kinds():
    from senas_amd.operations import OpType
    return {'up': OpType.UP, 'down': OpType.DOWN, 'norm': OpType.NORM}


# ------------------------------------------------------------------ C ABI
def test_abi_header_matches_binding_and_library():
    from senas_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'senas_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(senas_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()                      # raises if libsenas_hip.so is missing or lacks a symbol
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.senas_abi_version() >= 13
    # argument counts of the binding agree with the header
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r'\b%s\s*\(([^;]*?)\)\s*;' % name, hdr, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(',') if p.strip() and p.strip() != 'void']
        assert len(params) == len(args), (name, len(params), len(args))


def test_invalid_arguments_are_reported_not_executed():
    """Entry points validate before launching: a null pointer / inconsistent geometry returns
    SENAS_EINVAL with a message (no GPU needed -- nothing is launched)."""
    import ctypes as C
    from senas_amd import _lib
    lib = _lib.lib()
    g = _lib.ConvGeom(1, 8, 8, 4, 9, 9, 4, 3, 3, 1, 1, 1, 0, 1)      # ho/wo inconsistent with the rest
    assert lib.senas_conv2d_fwd(C.byref(g), None, None, None, 0, None, None, None, None) == -1
    assert b'geometry' in lib.senas_last_error()
    assert lib.senas_relu_fwd(16, None, None, None) == -1
    assert lib.senas_chan_stats(1, 16, 300, None, None, None) == -1
    with pytest.raises(_lib.SenasHipError):
        _lib.check(-1, 'probe')


def test_no_cpu_fallback():
    from senas_amd._lib import SenasHipError
    from senas_amd.operations import OPS, OpType
    op = OPS['dil_3_conv_5'](8, 8, OpType.NORM, 0)
    with pytest.raises(SenasHipError):
        op(torch.zeros(1, 8, 8, 8))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'senas_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in re.sub(r'""".*?"""', '', src, flags=re.S), f


# ------------------------------------------------------------------ state_dict interchange
def _keys_shapes(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


@pytest.mark.parametrize('tag', gio.index('prims'))
def test_primitive_state_dict_keys(tag):
    from senas_amd.operations import OPS
    z = gio.load('prims')
    kind, name, ci, co = tag.split('.')
    mod = OPS[name](int(ci), int(co), _kinds()[kind], 0)
    exp = {k: tuple(v.shape) for k, v in gio.sub(z, tag + '/sd0/').items()}
    assert _keys_shapes(mod) == exp


@pytest.mark.parametrize('tag', gio.index('nets'))
def test_net_state_dict_keys(tag):
    from senas_amd.genotype import Genotype
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    z = gio.load('nets')
    kw = json.loads(str(z[tag + '/kw']))
    if tag.startswith('nas'):
        net = NAS(use_sharing=False, double_down_channel=False, multi_gpus=False, device=torch.device('cpu'), **kw)
    else:
        net = SenasModel(genotype=gio.geno_from_json(z[tag + '/genotype'], Genotype), **kw)
    exp = {k: tuple(s) for k, s, _ in json.loads(str(z[tag + '/sd0/index']))}
    got = {k: s for k, s in _keys_shapes(net).items() if not k.endswith('num_batches_tracked')}
    assert got == exp
    # same key ORDER as the reference (checkpoints are ordered dicts)
    assert [k for k in net.state_dict() if not k.endswith('num_batches_tracked')] == \
        [k for k, _, _ in json.loads(str(z[tag + '/sd0/index']))]
    # named_parameters() reports exactly the names the reference's optimizer sees
    assert {k for k, _ in net.named_parameters()} == set(gio.digest(z, tag + '/grad/'))


def test_full_width_parameter_counts():
    """supernet 1 967 552 weights (+246 arch scalars), derived net 2 164 128."""
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import NAS
    nas = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device=torch.device('cpu'))
    arch = sum(p.numel() for p in nas.arch_parameters())
    assert arch == 246
    assert sum(p.numel() for p in nas.parameters()) - arch == 1967552
    assert len(nas.state_dict()) == 6718
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4)
    assert sum(p.numel() for p in net.parameters()) == 2164128
    assert len(net.state_dict()) == 807
    assert sum(p.numel() for p in SenasModel(4, 3, c=32, depth=5, genotype=senas_node_4).parameters()) == 2167840


# ------------------------------------------------------------------ genotype logic (bit-exact)
@pytest.mark.parametrize('tag', [t for t in gio.index('genoparse') if t.startswith('parse')])
def test_geno_parser(tag):
    from senas_amd.genotype import GenoParser
    z = gio.load('genoparse')
    nodes = int(tag.split('.')[1])
    for cell in ('down', 'up'):
        exp = [(a, int(b)) for a, b in json.loads(str(z[tag + '/' + cell]))]
        got = GenoParser(nodes).parse(z[tag + '/w1'], z[tag + '/w2'], cell)
        assert [(a, int(b)) for a, b in got] == exp, (tag, cell)


@pytest.mark.parametrize('tag', [t for t in gio.index('genoparse') if t.startswith('nasgeno')])
def test_nas_genotype(tag):
    from senas_amd.genotype import Genotype
    from senas_amd.senas_search import NAS
    z = gio.load('genoparse')
    _, depth, nodes, _ = tag.split('.')
    net = NAS(1, 4, 2, int(depth), meta_node_num=int(nodes), use_sharing=False, double_down_channel=False,
              device=torch.device('cpu'))
    with torch.no_grad():
        for k in ('alphas_dn', 'alphas_up', 'alphas_dn_nm', 'alphas_up_nm', 'betas_dn', 'betas_up', 'gamma'):
            getattr(net, k).copy_(torch.from_numpy(z[tag + '/' + k]))
    assert net.genotype() == gio.geno_from_json(z[tag + '/genotype'], Genotype)


def test_genotype_text_round_trip():
    """train_model.py:118 eval()s the genotype text in the namespace of the genotype module."""
    from senas_amd import genotype as G
    from senas_amd.geno_searched import senas_node_4
    txt = ("Genotype(down=[('se_conv_3', 1), ('avg_pool', 0), ('dil_3_conv_5', 2), ('dep_sep_conv_5', 1), "
           "('dil_3_conv_5', 2), ('avg_pool', 0), ('avg_pool', 1), ('dil_3_conv_5', 3)], down_concat=range(2, 6), "
           "up=[('up_sample', 1), ('dil_3_conv_5', 0), ('dil_3_conv_5', 0), ('dil_2_conv_5', 2), ('dil_3_conv_5', 1), "
           "('dil_2_conv_5', 2), ('dep_sep_conv_3', 0), ('dil_2_conv_5', 4)], up_concat=range(2, 6), gamma=[0, 0, 0, 1, 1, 1])")
    assert eval('G.%s' % txt) == senas_node_4     # README.md:44
    assert repr(senas_node_4) == txt


# ------------------------------------------------------------------ loss / metric: device kernels only
def test_loss_and_metric_have_no_cpu_path():
    """Dice+CE and the metric are HIP kernels; CPU tensors must be refused, not computed."""
    from senas_amd._lib import SenasHipError
    from senas_amd.loss import SegmentationLosses
    from senas_amd.metrics import SegmentationMetric
    logits, tgt = torch.zeros(1, 2, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long)
    with pytest.raises(SenasHipError):
        SegmentationLosses('dice_ce')([logits], tgt)
    with pytest.raises(SenasHipError):
        SegmentationMetric(2).update(tgt, logits)
    with pytest.raises(NotImplementedError):
        SegmentationLosses('focal')


# ------------------------------------------------------------------ checkpoint interchange (scope row f-3)
def test_checkpoint_round_trip_in_reference_format(tmp_path):
    """A search-phase checkpoint in the reference's dictionary layout (search_arc.py:227-238), built around the
    REFERENCE's own state_dict from the golden file, loads into this package's modules, survives a save/load
    through the reference's file name, and comes back with identical keys, order and values."""
    from senas_amd import checkpoint as ck
    from senas_amd.senas_search import NAS
    z = gio.load('nets')
    tag = 'nas.c8.d4'
    kw = json.loads(str(z[tag + '/kw']))
    ref_sd = gio.add_missing_counters(gio.torch_sd(gio.unpack(z, tag + '/sd0/'), requires_grad=False))
    net = NAS(use_sharing=False, double_down_channel=False, multi_gpus=False, device=torch.device('cpu'), **kw)
    opt_w = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=3e-4)
    opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt_w, 10)
    ref_ckpt = {'epoch': 3, 'dur_time': 12.5, 'cur_patience': 0, 'geno_type': 'g', 'model_state': ref_sd,
                'arch_optimizer': opt_a.state_dict(), 'model_optimizer': opt_w.state_dict(),
                # the reference writes these keys (alphas_dict(), betas_dict()) ...
                'alphas_dict': {k: ref_sd[k] + 0.5 for k in ('alphas_dn', 'alphas_dn_nm', 'alphas_up', 'alphas_up_nm')},
                'betas_dict': {k: ref_sd[k] - 0.25 for k in ('betas_dn', 'betas_up')}, 'scheduler': sched.state_dict()}
    epoch, dur, geno = ck.load_search_state(ref_ckpt, net, opt_a, opt_w, sched)
    assert (epoch, dur, geno) == (3, 12.5, 'g')
    assert torch.equal(net.alphas_dn, ref_sd['alphas_dn'] + 0.5) and torch.equal(net.betas_up, ref_sd['betas_up'] - 0.25)
    assert net.arch_parameters()[0] is net.alphas_dn            # still the tensors the optimizer holds
    out = ck.search_state(net, opt_a, opt_w, sched, epoch=3, dur_time=1.0, geno_type=str(net.genotype()))
    assert list(out.keys()) == ['epoch', 'dur_time', 'cur_patience', 'geno_type', 'model_state', 'arch_optimizer',
                                'model_optimizer', 'alphas_dict', 'betas_dict', 'scheduler']
    path = ck.save_checkpoint(out, True, str(tmp_path))
    assert os.path.basename(path) == 'checkpint.pth.tar' and os.path.exists(os.path.join(str(tmp_path), 'model_best.pth.tar'))
    back = torch.load(path, map_location='cpu', weights_only=False)
    assert [k for k in back['model_state'] if not k.endswith('num_batches_tracked')] == \
        [k for k in ref_sd if not k.endswith('num_batches_tracked')]
    for k, v in ref_sd.items():
        if k.startswith(('alphas', 'betas')):
            continue
        assert torch.equal(back['model_state'][k], v), k
    # legacy key names (what the reference's load_params reads) are accepted as well
    ck.load_search_state({'model_state': back['model_state'], 'alphas_dict': {'alphas_down': ref_sd['alphas_dn']},
                          'betas_dict': {'betas_down': ref_sd['betas_dn']}}, net)
    assert torch.equal(net.alphas_dn, ref_sd['alphas_dn'])


def test_train_checkpoint_layout(tmp_path):
    from senas_amd import checkpoint as ck
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    net = SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    st = ck.train_state(net, opt, epoch=0, best_dice_coeff=0.5)
    assert list(st.keys()) == ['epoch', 'dur_time', 'model_state', 'model_optimizer', 'best_pixAcc', 'best_mIoU',
                               'best_dice_coeff', 'best_loss']
    other = SenasModel(2, 1, c=8, depth=4, genotype=senas_node_4)
    assert ck.load_train_state(st, other, torch.optim.SGD(other.parameters(), lr=0.01, momentum=0.9)) == 1
    assert ck.load_train_state(net.state_dict(), other) == 0            # bare state_dict (testing_model.py)
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), other.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


# ------------------------------------------------------------------ state-major search cell (host-side pieces)
def test_stacked_weight_assembles_and_routes_gradients():
    """functional.StackedWeight: the stacked buffer is the concatenation of the per-edge parameters (dim 0 for Conv2d,
    dim 1 for ConvTranspose2d) and every parameter receives exactly its slice of the gradient of the buffer."""
    import torch
    from senas_amd import functional as F
    for dim, shape in ((0, (8, 32, 3, 3)), (1, (32, 8, 5, 5))):
        params = [torch.nn.Parameter(torch.randn(shape)) for _ in range(3)]
        sw = F.StackedWeight(params, dim)
        w = sw.tensor()
        assert torch.equal(w.detach(), torch.cat([p.detach() for p in params], dim=dim))
        g = torch.randn(w.shape)
        (w * g).sum().backward()
        for i, p in enumerate(params):
            assert torch.equal(p.grad, g.narrow(dim, i * shape[dim], shape[dim]))
        with torch.no_grad():
            params[1].add_(1.0)                       # unmanaged: the next use sees the new value
        assert torch.equal(sw.tensor().detach(), torch.cat([p.detach() for p in params], dim=dim))


def test_grad_landing_hands_the_buffer_on_only_when_every_part_was_written_in_place():
    import torch
    from senas_amd import functional as F
    land = F.GradLanding(3, (2, 8, 4, 4))
    like = torch.zeros(1)
    parts = [land.part(e, like) for e in range(3)]
    assert parts[1].data_ptr() == parts[0].data_ptr() + 4 * 8 and parts[0].shape == (2, 8, 4, 4)
    assert parts[0].stride(3) == 24                    # pixel stride of a part = channels of the whole buffer
    whole = land.take(parts)
    assert whole is not None and whole.shape == (2, 24, 4, 4) and land.buf is None
    parts = [land.part(e, like) for e in range(3)]
    assert land.take([parts[0], parts[1].clone(), parts[2]]) is None      # a foreign tensor: fall back to concatenation
    assert land.take([None, None, None]) is None


def test_search_cell_plan_covers_every_candidate_once():
    """Cell._plan: per state, the jobs (stacked convolutions, batched DepSepConv groups, per-edge leftovers) together with
    the 'none' terms account for every candidate of every edge that reads the state, with one alias per reader."""
    from senas_amd.cell import Cell
    for nodes in (2, 3, 4):
        for kind in ('down', 'up'):
            cell = Cell(nodes, 1, 32, 32, 32, kind)
            seen = set()
            for j in range(2 + nodes):
                edges = cell._out_edges(j)
                assert len(edges) == sum(1 for i in range(nodes) if j < 2 + i)
                jobs, fixed = cell._plan(j)
                assert all(a >= 1 for _, a in jobs)
                for e, p, t in fixed:
                    assert t.z is None and (e, p) not in seen
                    seen.add((e, p))
                if not edges:
                    assert not jobs and not fixed
            # every stacked group: same-geometry weights, k = number of edges leaving the state
            for sw in cell.stacked_weights():
                if sw.pad_to is not None:            # the zero-padded post_process weight (24 -> 32 input channels)
                    assert len(sw.params) == 1 and sw.buffer().shape[1] == 32
                else:
                    assert 2 <= len(sw.params) <= 4 and len({tuple(p.shape) for p in sw.params}) == 1


# ------------------------------------------------------------------ weights_init / YAML entry points (host side)
def test_weights_init_statistics():
    """utils/utils.py:240-250: kaiming-normal (fan_out, relu) convolutions, unit / zero batch-norm, xavier-normal linears."""
    import math
    import torch
    import torch.nn as nn
    from senas_amd.utils import weights_init
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(64, 96, 5, bias=False), nn.BatchNorm2d(96), nn.ConvTranspose2d(96, 48, 3, bias=False),
                        nn.Linear(256, 128, bias=True))
    with torch.no_grad():
        for p in net.parameters():
            p.fill_(7.0)
    net.apply(weights_init)
    conv, bn, convt, lin = net
    # kaiming_normal_(mode='fan_out'): std = sqrt(2 / (weight.shape[0] * receptive field)) -- torch's fan_out for both layouts
    for m in (conv, convt):
        fan_out = m.weight.shape[0] * m.weight[0][0].numel()
        want = math.sqrt(2.0 / fan_out)
        assert abs(float(m.weight.std()) / want - 1.0) < 0.03 and abs(float(m.weight.mean())) < 0.05 * want
    assert bool((bn.weight == 1).all()) and bool((bn.bias == 0).all())
    want = math.sqrt(2.0 / (256 + 128))
    assert abs(float(lin.weight.std()) / want - 1.0) < 0.03 and bool((lin.bias == 0).all())
    # and it matches what NAS applies to its own net at construction (search/senas_search.py:136)
    from senas_amd.senas_search import NAS
    nas = NAS(1, 8, 2, 2, meta_node_num=2, use_sharing=False, double_down_channel=False, device='cpu')
    bns = [m for m in nas.modules() if isinstance(m, nn.BatchNorm2d)]
    assert bns and all(bool((m.weight == 1).all()) and bool((m.bias == 0).all()) for m in bns)


def test_shipped_config_matches_the_reference_values():
    """senas_amd/configs/senas_promise12.yml restates configs/senas/senas_promise12.yml:10-67; the loader reads the
    reference's own file too (its ``!!python/tuple`` betas included).  The comparison against the reference file runs
    where /root/reference exists (the build container); the shipped values are pinned literally everywhere."""
    from senas_amd.run import DEFAULT_CONFIG, load_config, _parse_genotype
    from senas_amd.geno_searched import senas_node_4
    cfg = load_config(DEFAULT_CONFIG)
    s, t = cfg['searching'], cfg['training']
    assert (s['init_channels'], s['depth'], s['epoch'], s['batch_size'], s['alpha_begin'], s['meta_node_num'], s['grad_clip']) == (32, 5, 100, 8, 15, 3, 5)
    assert (s['sharing_normal'], s['double_down_channel'], s['deep_supervision']) == (False, False, False)
    assert s['model_optimizer'] == {'name': 'sgd', 'lr': 5e-3, 'weight_decay': 3e-4, 'momentum': 0.9}
    assert s['arch_optimizer']['lr'] == 1e-4 and tuple(s['arch_optimizer']['betas']) == (0.5, 0.999)
    assert (t['geno_type'], t['init_channels'], t['depth'], t['batch_size'], t['grad_clip']) == ('senas', 32, 5, 12, 5)
    assert t['model_optimizer'] == {'name': 'sgd', 'lr': 6e-3, 'weight_decay': 5e-4, 'momentum': 0.9}
    ref = '/root/reference/configs/senas/senas_promise12.yml'
    if os.path.exists(ref):
        r = load_config(ref)
        for blk in ('searching', 'training'):
            for k, v in cfg[blk].items():
                if k in r[blk]:
                    rv = r[blk][k]
                    if isinstance(v, dict):
                        for kk, vv in v.items():
                            if kk in (rv or {}):
                                assert list(vv) == list(rv[kk]) if isinstance(vv, (list, tuple)) else vv == rv[kk], (blk, k, kk)
                    else:
                        assert v == rv, (blk, k)
    assert _parse_genotype(str(senas_node_4)) == senas_node_4
    with pytest.raises(ValueError):
        _parse_genotype('__import__("os").system("true")')


def test_dropout_keeps_the_reference_child_indices():
    """utils/operations.py:118-130: with dp > 0 a Dropout2d is child 0 of the op's Sequential and the convolution child 1
    (DepSepConv: 0 1 2 3 | 4 5 6) -- the state_dict keys of SenasModel(dropout_prob > 0) shift accordingly."""
    from senas_amd.operations import ConvBn, ConvBnSe, DepSepConv
    import torch.nn as nn
    a, b, c = ConvBn(8, 8, 3, dropout=0.1), ConvBnSe(8, 8, 3, dropout=0.1), DepSepConv(8, 8, 3, dropout=0.1)
    assert isinstance(a[0], nn.Dropout2d) and a.conv is a[1] and a.norm is a[2]
    assert isinstance(b[0], nn.Dropout2d) and b.conv is b[1] and b.norm is b[2] and b.se is b[3]
    assert [type(m).__name__ for m in c] == ['Dropout2d', 'Conv2d', 'BatchNorm2d', 'ReLU', 'Dropout2d', 'Conv2d', 'BatchNorm2d']
    assert c.dw is c[1] and c.norm1 is c[2] and c.pw is c[5] and c.norm2 is c[6]
    plain = DepSepConv(8, 8, 3)
    assert plain.dw is plain[0] and plain.norm1 is plain[1] and plain.pw is plain[3] and plain.norm2 is plain[4] and plain.drop is None
    assert sorted(k for k in a.state_dict() if k.endswith('weight')) == ['1.weight', '2.weight']


# ------------------------------------------------------------------ macro-grid schedule (grid.MacroGrid._walk_grid)
class _CountingPlan(object):
    """FanPlan's interface, counting: how often every key is put and got, and in which order the cells are applied."""

    def __init__(self):
        self.puts, self.gets = {}, {}
        self.dry = True

    def put(self, key, value=None):
        self.puts[key] = self.puts.get(key, 0) + 1
        return key

    def get(self, key):
        self.gets[key] = self.gets.get(key, 0) + 1
        return None


@pytest.mark.parametrize('depth', [3, 5])
def test_supernet_dry_walk_reads_every_column_tensor_once_per_up_cell(depth):
    """SenasSearch's own skip lists (functional.skip_stack: one launch stacks the column's down-path output and the gamma-gated
    blends of neighbouring outputs, search/senas_search.py:96-103): up cell (i, j) reads every tensor (k, j), k < i, ONCE -- the
    reference's loop reads the inner ones twice, as the second operand of one blend and the first of the next -- plus its in1."""
    from senas_amd.senas_search import SenasSearch
    net = SenasSearch(1, 8, 2, depth, meta_node_num=3, double_down_channel=False, supervision=False)
    plan = _CountingPlan()
    net._walk(plan, None, None)
    want = {'s0': 3}                                         # stem1, down cell 1, the head
    for j in range(depth):
        for k in range(depth - j):
            r = (depth - 1 - j) - k                          # skip input of the up cells above it in its column
            r += 1 if (j >= 1 and k + 1 <= depth - 1 - (j - 1)) else 0      # in1 of up cell (k + 1, j - 1)
            if k == 0:
                r += (1 if j + 1 < depth else 0) + (1 if j + 2 < depth else 0)   # the down cells that read it
            if (k, j) == (depth - 1, 0):
                r += 1                                       # the head
            want[('o', k, j)] = r
    assert plan.gets == want


@pytest.mark.parametrize('kind,depth,supervision', [('search', 2, False), ('search', 3, True), ('search', 5, False), ('search', 6, True),
                                                   ('derived', 5, False), ('derived', 4, True), ('derived', 3, False)])
def test_macro_grid_schedule_reads_what_the_references_loop_reads(kind, depth, supervision):
    """The level-major schedule (down cell j + 1, then at once up cell (1, j); then levels 2, 3, ...) must hand every cell
    exactly the tensors the reference's loop does (search/senas_search.py:96-107, models/senas_model.py:160-175: j descending, i
    ascending, overwriting a running list).  Both schedules are run dry over keys: the same cells, each with the same in1 and the
    same skip list; every tensor read as often in both (a reader count that differs between the dry and the live walk of the
    FanPlan ends in StopIteration on the GPU)."""
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.grid import gamma_index
    from senas_amd.senas_model import SenasModel
    from senas_amd.senas_search import SenasSearch
    if kind == 'search':
        net = SenasSearch(1, 8, 2, depth, meta_node_num=3, double_down_channel=False, supervision=supervision)
    else:
        gamma = [1] * 6 if supervision else list(senas_node_4.gamma)
        net = SenasModel(2, 1, c=8, depth=depth, genotype=senas_node_4._replace(gamma=gamma), supervision=supervision)
    # --- this package's schedule, dry
    plan = _CountingPlan()
    seen = []

    def skips(plan_, G, i, j, live):
        if kind == 'search':
            out = [plan_.get(G[0][j])]
            for k in range(1, i):
                plan_.get(G[k - 1][j]); plan_.get(G[k][j])
                out.append(('blend', k, j))
            return out
        return [plan_.get(G[k][j]) or G[k][j] for k in range(i) if G[k][j] is not None]

    orig_put = plan.put

    def put(key, value=None):
        seen.append(key)
        return orig_put(key, value)
    plan.put = put
    net._walk_grid(plan, None, lambda *a: None, skips)
    # --- the reference's loop, dry, over the same keys
    want_gets, cells = {}, []

    def get(key):
        want_gets[key] = want_gets.get(key, 0) + 1
        return key
    s0 = 's0'
    get(s0)
    outs = [('o', 0, 0)]
    for j in range(1, depth):
        get(s0 if j == 1 else outs[j - 2]); get(outs[j - 1])
        outs.append(('o', 0, j))
        cells.append(('o', 0, j))
    for j in reversed(range(depth - 1)):
        for i in range(1, depth - j):
            if net.blocks[i][j] is None:
                outs[i + j] = None
                continue
            if kind == 'search':
                get(outs[j])
                for k in range(1, i):
                    get(outs[j + k - 1]); get(outs[j + k])
            else:
                for t in range(j, i + j):
                    if outs[t] is not None:
                        get(outs[t])
            get(outs[i + j])
            outs[i + j] = ('o', i, j)
            cells.append(('o', i, j))
    for o in (outs if supervision else outs[-1:]):
        get(s0); get(o)
    assert plan.gets == want_gets
    assert sorted(k for k in seen if k != 's0' and k != ('o', 0, 0)) == sorted(cells)
    assert all(v == 1 for v in plan.puts.values())
    # the order respects the data flow: a cell is applied after the cells it reads
    pos = {k: n for n, k in enumerate(seen)}
    for (_, i, j) in cells:
        if i >= 1:
            assert pos[('o', i - 1, j + 1)] < pos[('o', i, j)]
            assert all(pos[('o', k, j)] < pos[('o', i, j)] for k in range(i) if ('o', k, j) in pos)
    assert gamma_index(2, 1) == 4


def _plan(n, edges, max_lanes, solo=None):
    """senas_sched_plan (csrc/sched.hip: the lane scheduler's plan, host arithmetic only) on a DAG numbered topologically."""
    import ctypes as C
    from senas_amd import _lib
    m = len(edges)
    I = C.c_int32
    frm, to = (I * max(m, 1))(*[a for a, _ in edges]), (I * max(m, 1))(*[b for _, b in edges])
    so = (C.c_uint8 * n)(*(solo or [0] * n))
    node_lane, node_seg, seg_lane, dep_begin, deps = (I * n)(), (I * n)(), (I * n)(), (I * (n + 1))(), (I * max(m, 1))()
    nseg = I(0)
    _lib.check(_lib.lib().senas_sched_plan(n, m, frm, to, so, max_lanes, node_lane, node_seg, C.byref(nseg), seg_lane, dep_begin, deps),
               'senas_sched_plan')
    k = nseg.value
    return list(node_lane), list(node_seg), [seg_lane[s] for s in range(k)], [[deps[i] for i in range(dep_begin[s], dep_begin[s + 1])] for s in range(k)]


def _check_plan(n, edges, max_lanes, solo=None):
    """The invariants every policy of the lane scheduler must keep (a policy change is then testable without a GPU):
    every node in exactly one segment, a segment on one lane, nodes of a segment consecutive in issue order on their lane; segment
    indices = issue order, every wait points to an EARLIER segment of another lane; and every edge of the DAG is enforced -- by
    stream order (same lane, the parent's segment not later) or by a wait: some segment of the child's lane, issued no later than
    the child's, waits for a segment of the parent's lane issued no earlier than the parent's."""
    lane, seg, seg_lane, deps = _plan(n, edges, max_lanes, solo)
    k = len(seg_lane)
    assert all(0 <= l < max_lanes for l in lane) and lane[0] == 0
    assert all(0 <= s < k for s in seg) and sorted(set(seg)) == list(range(k))
    first = {}
    for v in range(n):
        assert seg_lane[seg[v]] == lane[v]
        first.setdefault(seg[v], v)
    assert [first[s] for s in range(k)] == sorted(first.values())            # segments are numbered by their first node: the issue order
    for s in range(k):                                                        # a segment's nodes are consecutive among its lane's nodes
        mine = [v for v in range(n) if seg[v] == s]
        on_lane = [v for v in range(n) if lane[v] == seg_lane[s]]
        i = on_lane.index(mine[0])
        assert on_lane[i:i + len(mine)] == mine
        for d in deps[s]:
            assert 0 <= d < s and seg_lane[d] != seg_lane[s]
    for s, flag in enumerate(solo or []):
        if flag:
            assert sum(1 for v in range(n) if seg[v] == seg[s]) == 1
    for a, b in edges:
        if lane[a] == lane[b]:
            assert seg[a] <= seg[b]
            continue
        ok = any(seg_lane[d] == lane[a] and d >= seg[a]
                 for s in range(seg[b] + 1) if seg_lane[s] == lane[b] for d in deps[s])
        assert ok, ('edge %d -> %d is not enforced' % (a, b), lane[a], lane[b], seg[a], seg[b])
    return lane, seg, seg_lane, deps


def test_lane_scheduler_plan_on_synthetic_dags():
    """csrc/sched.hip's chain cover + segment cut through senas_sched_plan (no device): dependencies preserved, issue order
    topological, for a chain, a fork / join diamond, the macro grid's shape (a spine with columns hanging off it and handing
    over to each other through it) and seeded random DAGs, at 1 .. 6 lanes."""
    import numpy as np
    # a chain: one lane, one segment
    lane, seg, seg_lane, deps = _check_plan(6, [(i, i + 1) for i in range(5)], 4)
    assert set(lane) == {0} and len(seg_lane) == 1 and deps == [[]]
    # a diamond with three-node arms
    edges = [(0, 1), (1, 2), (2, 3), (0, 4), (4, 5), (5, 6), (3, 7), (6, 7)]
    lane, seg, seg_lane, deps = _check_plan(8, edges, 4)
    assert len(set(lane)) == 2 and lane[1] == lane[2] == lane[3] and lane[4] == lane[5] == lane[6] != lane[1]
    assert len(set(_check_plan(8, edges, 1)[0])) == 1                        # one lane: everything in issue order on it
    # a solo node splits its chain
    lane, seg, seg_lane, deps = _check_plan(5, [(i, i + 1) for i in range(4)], 2, solo=[0, 0, 1, 0, 0])
    assert len(seg_lane) == 3
    # the macro grid's shape: spine nodes s_0..s_4 in a chain; column j forks off s_j, runs 3 nodes, joins the spine's tail
    n, edges = 0, []
    spine = list(range(5))
    n = 5
    edges += [(i, i + 1) for i in range(4)]
    tails = []
    for j in range(4):
        col = list(range(n, n + 3))
        n += 3
        edges += [(spine[j], col[0]), (col[0], col[1]), (col[1], col[2])]
        tails.append(col[2])
    edges += [(t, n) for t in tails] + [(spine[-1], n)]
    n += 1
    order = sorted(range(n))                                                  # (already topological: every edge goes up)
    assert all(a < b for a, b in edges)
    for L in (1, 2, 3, 4, 6):
        _check_plan(n, edges, L)
    # random DAGs
    rng = np.random.RandomState(5)
    for trial in range(40):
        n = int(rng.randint(2, 60))
        edges = set()
        for b in range(1, n):
            for a in rng.choice(b, size=min(b, int(rng.randint(0, 4))), replace=False):
                edges.add((int(a), b))
        solo = [int(rng.rand() < 0.05) for _ in range(n)]
        _check_plan(n, sorted(edges), int(rng.randint(1, 7)), solo)


def _contract(n, edges, kind):
    import ctypes as C
    from senas_amd import _lib
    I = C.c_int32
    m = len(edges)
    frm, to = (I * max(m, 1))(*[a for a, _ in edges]), (I * max(m, 1))(*[b for _, b in edges])
    cap = 4 * m * max(1, m) + 16
    of, ot, om = (I * cap)(), (I * cap)(), I(0)
    _lib.check(_lib.lib().senas_sched_contract(n, m, frm, to, (I * n)(*kind), C.byref(om), of, ot, cap), 'senas_sched_contract')
    return sorted((of[i], ot[i]) for i in range(om.value))


def test_typed_markers_give_a_reader_its_producer_and_the_origin_stream_everything():
    """csrc/sched.hip contract_markers (through senas_sched_contract, no device) on the shape grid.Lanes.hand captures: two
    hand-overs through the origin stream, A -> B and C -> D, with the origin stream's own kernels before, between and after.
    Reader B must depend on A only (not on the origin stream's history), reader D on C only (not on A, which the chain of
    relay markers carries along); the origin stream's next kernel keeps EVERY wait (A, C and its own previous kernel).
    An untyped relay (no PRODUCER parent, no CONSUMER child: round 4's captures) contracts as it always did."""
    # nodes (topological): 0 m0 (origin kernel) | 1 a (lane A kernel) | 2 Pa (PRODUCER on A) | 3 R1 (RELAY: parents m0, Pa)
    # 4 b0 (lane B kernel) | 5 Cb (CONSUMER on B: parents b0, R1) | 6 b1 (B's reader kernel: parent Cb)
    # 7 c (lane C kernel) | 8 Pc (PRODUCER) | 9 R2 (RELAY: parents R1 [stream order], Pc) | 10 d0 | 11 Cd (CONSUMER: d0, R2) | 12 d1
    # 13 m1 (origin stream's next kernel: parent R2)
    R, P, Cn, K = 0, 1, 2, -1
    kind = [K, K, P, R, K, Cn, K, K, P, R, K, Cn, K, K]
    edges = [(0, 3), (1, 2), (2, 3), (4, 5), (3, 5), (5, 6), (7, 8), (3, 9), (8, 9), (10, 11), (9, 11), (11, 12), (9, 13)]
    got = _contract(14, edges, kind)
    par = {}
    for a, b in got:
        par.setdefault(b, set()).add(a)
    assert par[6] == {4, 1}                        # b1 <- b0 (its lane), a (the producer): not m0
    assert par[12] == {10, 7}                      # d1 <- d0, c: neither m0 nor a
    assert par[13] == {0, 1, 7}                    # the origin stream's next kernel: everything
    assert all(kind[a] == K and kind[b] == K for a, b in got)
    # untyped: relay with plain parents and children
    kind2 = [K, K, R, K, K]
    got2 = _contract(5, [(0, 2), (1, 2), (2, 3), (2, 4)], kind2)
    assert got2 == [(0, 3), (0, 4), (1, 3), (1, 4)]
    # an empty node in a chain
    assert _contract(3, [(0, 1), (1, 2)], [K, 3, K]) == [(0, 2)]


def _plan2(n, edges, streams, node_us=None, solo=None):
    import ctypes as C
    from senas_amd import _lib
    I = C.c_int32
    m = len(edges)
    frm, to = (I * max(m, 1))(*[a for a, _ in edges]), (I * max(m, 1))(*[b for _, b in edges])
    so = (C.c_uint8 * n)(*(solo or [0] * n))
    us = (C.c_double * n)(*node_us) if node_us is not None else None
    node_seg, seg_stream, seg_issue, dep_begin, deps = (I * n)(), (I * n)(), (I * n)(), (I * (n + 1))(), (I * max(m, 1))()
    nseg = I(0)
    _lib.check(_lib.lib().senas_sched_plan2(n, m, frm, to, so, us, streams, node_seg, C.byref(nseg), seg_stream, seg_issue, dep_begin, deps),
               'senas_sched_plan2')
    k = nseg.value
    return (list(node_seg), [seg_stream[s] for s in range(k)], [seg_issue[s] for s in range(k)],
            [[deps[i] for i in range(dep_begin[s], dep_begin[s + 1])] for s in range(k)])


def _check_plan2(n, edges, streams, node_us=None, solo=None):
    """Invariants of the "critical" plan: segments are linear runs (consecutive nodes joined by an edge), only a segment's first node
    has parents outside it and only its last node children outside it, so every edge is enforced by "wait for the parent's whole
    segment"; a segment's dependencies are exactly the segments of its first node's parents; the issue order is a permutation and
    topological (every dependency is issued earlier -- which also makes a same-stream dependency the stream's own order)."""
    seg, stream, issue, deps = _plan2(n, edges, streams, node_us, solo)
    k = len(stream)
    assert sorted(issue) == list(range(k)) and all(0 <= q < streams for q in stream)
    when = {s: i for i, s in enumerate(issue)}
    nodes = {}
    for v in range(n):
        nodes.setdefault(seg[v], []).append(v)
    assert sorted(nodes) == list(range(k))
    es = set(edges)
    for s_, vs in nodes.items():
        for a, b in zip(vs, vs[1:]):
            assert (a, b) in es
        want = set()
        for a, b in edges:
            if b in vs and a not in vs:
                assert b == vs[0], ('a parent outside the segment in front of a node that is not its first', a, b)
                assert a == nodes[seg[a]][-1], ('the parent is not the last node of its segment', a, b)
                want.add(seg[a])
            if a in vs and b not in vs:
                assert a == vs[-1]
        assert set(deps[s_]) == want
        for d in deps[s_]:
            assert when[d] < when[s_]
    for s_, flag in enumerate(solo or []):
        if flag:
            assert len(nodes[seg[s_]]) == 1
    return seg, stream, issue, deps


def test_critical_path_plan_on_synthetic_dags():
    """csrc/sched.hip cut_pieces + list_schedule through senas_sched_plan2 (no device): the invariants above on a chain, a
    diamond, the macro grid's shape and seeded random DAGs; and the policy itself on a case with a known answer -- a long
    chain beside short independent pieces on two streams: the long chain starts first and keeps a stream to itself."""
    import numpy as np
    seg, stream, issue, deps = _check_plan2(6, [(i, i + 1) for i in range(5)], 4)
    assert len(stream) == 1
    edges = [(0, 1), (1, 2), (2, 3), (0, 4), (4, 5), (5, 6), (3, 7), (6, 7)]
    seg, stream, issue, deps = _check_plan2(8, edges, 4)
    assert len(stream) == 4 and stream[seg[1]] != stream[seg[4]]             # the two arms side by side
    # known answer: node 0 forks into a 6-node chain (1..6, 10 us each) and four single nodes (7..10, 10 us each); all join in 11
    edges = [(0, 1)] + [(i, i + 1) for i in range(1, 6)] + [(0, 7), (0, 8), (0, 9), (0, 10)] + [(6, 11), (7, 11), (8, 11), (9, 11), (10, 11)]
    seg, stream, issue, deps = _check_plan2(12, edges, 2, node_us=[10.0] * 12)
    chain = seg[1]
    assert all(seg[v] == chain for v in range(1, 7))
    assert issue.index(chain) == 1                                           # right behind the fork: the longest remaining path first
    others = [seg[v] for v in (7, 8, 9, 10)]
    assert all(stream[o] != stream[chain] for o in others)                   # the short pieces share the other stream
    rng = np.random.RandomState(7)
    for trial in range(40):
        n = int(rng.randint(2, 70))
        edges = set()
        for b in range(1, n):
            for a in rng.choice(b, size=min(b, int(rng.randint(0, 4))), replace=False):
                edges.add((int(a), b))
        solo = [int(rng.rand() < 0.05) for _ in range(n)]
        _check_plan2(n, sorted(edges), int(rng.randint(1, 6)), node_us=[float(rng.rand() * 50 + 1) for _ in range(n)], solo=solo)
