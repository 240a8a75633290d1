"""The BASELINE-depth supernet and derived net (c = 32, depth 5) gradient check that CAN fail.

Whole-net gradients of this family are chaotic in the reference itself: for NAS(c=32, depth=5) at the reference's
initialisation scale a 1e-6 relative perturbation of the inputs and weights moves the MEDIAN gradient tensor by 1e-2 of its
scale and 95 % of the tensors by more than 2.5e-4 -- every one of 7 seeds, at 2x1x64x64 and at 4x1x128x128
(tests/golden/make_golden.py round4-search), so no whole-net fixture of this depth can hold a gradient to
1e-3.  A single cell is well conditioned.  So the network is checked CELL BY CELL in its own operating regime: the oracle runs
the whole pass once and hands every cell of the HIP network the inputs the oracle's cell saw and the gradient the oracle's cell
output received; the cell's output, both input gradients and every parameter gradient must then match the oracle's to 5e-5 of
the tensor scale (north_star: 1e-3).  A systematic error of a few 1e-3 in any cell of the depth-5 network -- its 2 x 2 maps at
the bottom, its 128-channel skip inputs at the top -- fails here; the wiring BETWEEN the cells is what the logits, loss and
trajectory tests hold (search/senas_search.py:96-107, search/cell.py:92-110, models/senas_model.py:50-64,160-175)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL = 5e-5


def dev():
    return torch.device('cuda:0')


def _oracle_leaves(net, prefix):
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    a, b = prefix + 'blocks.0.0.', prefix + 'stem1.'
    for k in list(sd):
        if k.startswith(a):
            sd[k] = sd[b + k[len(a):]]                      # stem1 is registered twice in the reference (one module)
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k and not k.startswith(a):
            v.requires_grad_(True)
    return sd


def _record_cells(R, name):
    """Patch the oracle's cell function so that every call records (prefix, in0, in1, out) with gradients retained; the inputs
    are passed through a fresh node each, so that their .grad is the gradient THROUGH THIS CELL only."""
    orig = getattr(R, name)
    records = []

    def wrapped(p, in0, in1, *rest, **kw):
        a, b = in0 * 1.0, in1 * 1.0
        a.retain_grad()
        b.retain_grad()
        y = orig(p, a, b, *rest, **kw)
        y.retain_grad()
        records.append((p.prefix, a, b, y))
        return y

    setattr(R, name, wrapped)
    return orig, records


def _cell_module(net, prefix, root):
    mod = net
    for part in (root + prefix).strip('.').split('.'):
        mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
    return mod


def _check_cells(kind, records, sd, net, root, call, margins):
    worst = {}
    for prefix, a, b, y in records:
        cell = _cell_module(net, prefix, root)
        for p in cell.parameters():
            p.grad = None
        in0 = a.detach().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        in1 = b.detach().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        out = call(cell, in0, in1)
        ys = float(y.detach().abs().max())
        e_out = float((out.detach().cpu() - y.detach()).abs().max()) / ys
        out.backward(y.grad.to(dev()))
        torch.cuda.synchronize()
        errs = {'out': e_out}
        gtop = max(float(a.grad.abs().max()), float(b.grad.abs().max()))
        errs['d_in0'] = float((in0.grad.cpu() - a.grad).abs().max()) / gtop
        errs['d_in1'] = float((in1.grad.cpu() - b.grad).abs().max()) / gtop
        exp = {k[len(prefix):]: v.grad for k, v in sd.items() if k.startswith(prefix) and v.requires_grad and v.grad is not None}
        got = {k: p.grad for k, p in cell.named_parameters() if p.grad is not None}
        assert set(got) == set(exp) and len(exp) > 20, (prefix, sorted(set(got) ^ set(exp))[:4])
        ptop = max(float(v.abs().max()) for v in exp.values())
        for k, e in exp.items():
            scale = max(float(e.abs().max()), 1e-2 * ptop)           # (analytically-zero gradients: on the scale of the cell's largest)
            errs['dw'] = max(errs.get('dw', 0.0), float((got[k].cpu() - e).abs().max()) / scale)
        worst[prefix] = errs
        for what, v in errs.items():
            assert v <= REL, '%s cell %s %s: %.2e of the tensor scale (map %s)' % (kind, prefix, what, v, tuple(y.shape))
    margins.update({'cells': len(records), 'bound': REL,
                    'worst': {w: max(e[w] for e in worst.values()) for w in ('out', 'd_in0', 'd_in1', 'dw')},
                    'worst_cell': max(worst, key=lambda k: max(worst[k].values()))})


def test_depth5_supernet_cell_by_cell():
    from conftest import record_margin
    from oracle import senas_ref as R            # checker only
    from senas_amd.loss import SegmentationLosses
    from senas_amd.senas_search import NAS
    torch.manual_seed(21)
    net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev()).train()
    g = torch.Generator().manual_seed(22)
    x, tgt = torch.randn(2, 1, 64, 64, generator=g), torch.randint(0, 2, (2, 64, 64), generator=g)
    sd = _oracle_leaves(net, 'net.')
    orig, records = _record_cells(R, 'search_cell')
    try:
        outs = R.nas_forward(sd, x, depth=5, nodes=3)
        R.dice_ce_loss(outs[-1], tgt).backward()
    finally:
        R.search_cell = orig
    assert len(records) == 4 + 10 + 1                                  # down cells, up cells, the head's cell
    # the whole pass first: logits and loss to north_star's bar (the wiring between the cells)
    got = net(x.to(dev()))
    scale = float(outs[-1].detach().abs().max())
    assert float((got[-1].detach().cpu() - outs[-1].detach()).abs().max()) <= 1e-3 * scale
    loss = SegmentationLosses('dice_ce')(got, tgt.to(dev()))
    assert abs(float(loss) - float(R.dice_ce_loss(outs[-1], tgt))) <= 1e-4 * abs(float(loss))
    w = [t.detach() for t in net._mixing_weights()]                    # (alphas_dn_nm, alphas_up_nm, alphas_dn, alphas_up, betas_dn, betas_up, gamma)
    args = {'down': (w[0], w[2], w[4]), 'up': (w[1], w[3], w[5])}
    kind_of = lambda cell: 'down' if cell in list(net.net.blocks[0]) else 'up'
    margins = {}
    _check_cells('supernet', records, sd, net, '', lambda cell, a, b: cell(a, b, *args[kind_of(cell)]), margins)
    record_margin('test_depth5_supernet_cell_by_cell', **margins)


def test_depth5_derived_cell_by_cell():
    from conftest import record_margin
    from oracle import senas_ref as R            # checker only
    from senas_amd.geno_searched import senas_node_4
    from senas_amd.senas_model import SenasModel
    from senas_amd.utils import weights_init
    torch.manual_seed(23)
    net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4).to(dev()).train()
    net.apply(weights_init)
    g = torch.Generator().manual_seed(24)
    x, tgt = torch.randn(2, 1, 64, 64, generator=g), torch.randint(0, 2, (2, 64, 64), generator=g)
    sd = _oracle_leaves(net, '')
    orig, records = _record_cells(R, 'build_cell')
    try:
        outs = R.derived_forward(sd, x, R.Genotype(*senas_node_4), depth=5)
        R.dice_ce_loss(outs[-1], tgt).backward()
    finally:
        R.build_cell = orig
    assert len(records) == 4 + 7 + 1                                   # down cells, the up cells the genotype keeps, the head's cell
    got = net(x.to(dev()))
    scale = float(outs[-1].detach().abs().max())
    assert float((got[-1].detach().cpu() - outs[-1].detach()).abs().max()) <= 1e-3 * scale
    margins = {}
    _check_cells('derived', records, sd, net, '', lambda cell, a, b: cell(a, b), margins)
    record_margin('test_depth5_derived_cell_by_cell', **margins)


def test_depth5_supernet_between_the_cells():
    """The pieces BETWEEN the cells of the depth-5 c = 32 supernet, teacher-forced from the same kind of recorded oracle pass as
    the cells above: the stems (x -> s0, c0: search/senas_search.py:30-33), in0 of every up cell of level >= 2 -- the column's
    down-path output and the gamma-gated blends below it, concatenated (:98-102; functional.skip_stack) -- with its gradient for
    every tensor of the column and for the softmax(gamma) table, and the head's ReLU + 3x3 convolution (:5-13).  Output, input
    gradients and parameter gradients to 5e-5 of the tensor scale.  Together with the cell-by-cell test every kernel launch of
    the depth-5 pass has a gradient check at this width."""
    from conftest import record_margin
    from oracle import senas_ref as R            # checker only
    from senas_amd import functional as F
    from senas_amd.grid import gamma_index
    from senas_amd.senas_search import NAS
    torch.manual_seed(31)
    depth = 5
    net = NAS(1, 32, 2, depth, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev()).train()
    with torch.no_grad():                                               # (a gamma table away from 0.5 / 0.5)
        net.gamma.copy_(torch.randn(net.gamma.shape, generator=torch.Generator().manual_seed(32)).to(dev()) * 0.7)
    g = torch.Generator().manual_seed(33)
    x, tgt = torch.randn(2, 1, 64, 64, generator=g), torch.randint(0, 2, (2, 64, 64), generator=g)
    sd = _oracle_leaves(net, 'net.')
    orig_cell, records = _record_cells(R, 'search_cell')
    orig_stem, orig_head = R.stem, R.relu_conv
    stems, heads = [], []

    def stem(p, xx, training):
        s0, c0 = orig_stem(p, xx, training)
        a, b = s0 * 1.0, c0 * 1.0                                       # what the REST of the network reads (stem1 read s0 itself)
        a.retain_grad()
        b.retain_grad()
        stems.append((a, b))
        return a, b

    def relu_conv(p, y):
        a = y * 1.0
        a.retain_grad()
        out = orig_head(p, a)
        out.retain_grad()
        heads.append((a, out))
        return out

    R.stem, R.relu_conv = stem, relu_conv
    try:
        outs = R.nas_forward(sd, x, depth=depth, nodes=3)
        R.dice_ce_loss(outs[-1], tgt).backward()
    finally:
        R.search_cell, R.stem, R.relu_conv = orig_cell, orig_stem, orig_head
    worst = {}

    def err(got, want, scale=None):
        scale = float(want.abs().max()) if scale is None else scale
        return float((got.detach().cpu() - want).abs().max()) / max(scale, 1e-30)

    def param_errs(module, prefix):
        exp = {k[len(prefix):]: v.grad for k, v in sd.items() if k.startswith(prefix) and v.requires_grad and v.grad is not None}
        got = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
        assert set(got) == set(exp) and exp, (prefix, sorted(set(got) ^ set(exp))[:4])
        ptop = max(float(v.abs().max()) for v in exp.values())
        return max(err(got[k], e, max(float(e.abs().max()), 1e-2 * ptop)) for k, e in exp.items())

    # ---- stems: x -> (s0, c0), driven back with the gradients the rest of the oracle's network sent into them
    (s0_ref, c0_ref), = stems
    for p in net.parameters():
        p.grad = None
    grid = net.net
    xd = x.to(dev())
    s0 = grid.stem0(xd)
    c0 = grid.stem1(s0)
    torch.autograd.backward([s0, c0], [s0_ref.grad.to(dev()), c0_ref.grad.to(dev())])
    torch.cuda.synchronize()
    worst['stems'] = {'s0': err(s0, s0_ref.detach()), 'c0': err(c0, c0_ref.detach()),
                      'dw': max(param_errs(grid.stem0, 'net.stem0.'), param_errs(grid.stem1, 'net.stem1.'))}
    # ---- in0 of the up cells of level >= 2
    rec = {p: (a, b, y) for p, a, b, y in records}
    table_ref = torch.softmax(sd['gamma'].detach(), dim=-1)

    def column(i, j):
        """O(0, j) .. O(i-1, j): what sits at resolution level j when up cell (i, j) runs (grid.MacroGrid)."""
        first = c0_ref.detach() if j == 0 else rec['net.blocks.0.%d.' % j][2].detach()
        return [first] + [rec['net.blocks.%d.%d.' % (k, j)][2].detach() for k in range(1, i)]

    stacks = {}
    for i in range(2, depth):
        for j in range(depth - i):
            a_ref = rec['net.blocks.%d.%d.' % (i, j)][0]                # in0 as the oracle's cell saw it, with its gradient
            col = [t.clone().requires_grad_(True) for t in column(i, j)]
            tab = table_ref.clone().requires_grad_(True)
            idx = [0] + [gamma_index(k, j) for k in range(1, i)]
            want = torch.cat([col[0]] + [tab[idx[k]][0] * col[k - 1] + tab[idx[k]][1] * col[k] for k in range(1, i)], 1)
            assert float((want.detach() - a_ref.detach()).abs().max()) <= 1e-6 * float(a_ref.detach().abs().max())   # the restated formula IS the oracle's in0
            want.backward(a_ref.grad)
            cold = [t.detach().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in col]
            tabd = table_ref.to(dev()).requires_grad_(True)
            rows = F.GammaRows(tabd)
            got = F.skip_stack(cold, rows, idx)
            got.backward(a_ref.grad.to(dev()))
            torch.cuda.synchronize()
            gtop = max(float(t.grad.abs().max()) for t in col)
            e = {'out': err(got, a_ref.detach()),
                 'd_col': max(err(cd.grad, cr.grad, gtop) for cd, cr in zip(cold, col)),
                 'd_gamma_rows': err(tabd.grad, tab.grad)}
            stacks['%d.%d' % (i, j)] = e
    worst['skip_stacks'] = {w: max(e[w] for e in stacks.values()) for w in ('out', 'd_col', 'd_gamma_rows')}
    assert len(stacks) == 6
    # ---- the head's ReLU + 3x3 convolution
    (h_in, h_out), = heads
    head = grid.head_block[-1].segmentation_head
    for p in head.parameters():
        p.grad = None
    hin = h_in.detach().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    hout = head(hin)
    hout.backward(h_out.grad.to(dev()))
    torch.cuda.synchronize()
    worst['head_conv'] = {'out': err(hout, h_out.detach()), 'd_in': err(hin.grad, h_in.grad),
                          'dw': param_errs(head, 'net.head_block.0.segmentation_head.')}
    record_margin('test_depth5_supernet_between_the_cells', bound=REL, **worst)
    for piece, errs in worst.items():
        for what, v in errs.items():
            assert v <= REL, '%s %s: %.2e of the tensor scale' % (piece, what, v)
