"""CPU oracle for the SENAS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  Nothing under ``senas_amd/`` imports it; the product path fails loudly
when the HIP library is missing instead of falling back to anything in here.

What it is: a *functional* restatement (plain ``torch.nn.functional`` on CPU, fp32) of the
reference's supernet / derived-net forward pass, written over a flat ``state_dict`` whose keys
are exactly the ones the reference's modules emit.  Backward comes from torch autograd over
these functional ops, as in the reference.  The arithmetic itself (conv, batch-norm, pooling,
bilinear resize) lives in the third-party dependency ``torch`` (reference pin ``torch==1.8.1``,
requirements.txt:5; this container runs torch 2.10 CPU); the call sites restated here are the
reference's own (file:line cited per function, relative to /root/reference).

Pinning: the reference has no tests for this path.  The oracle is pinned
by golden vectors produced by importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
checks every one of them.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# utils/operations.py:23-48 -- column order of the alpha tables
DOWN_OPS = ('avg_pool', 'se_conv_3', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3', 'dep_sep_conv_5')
UP_OPS = ('up_sample', 'se_conv_3', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3', 'dep_sep_conv_5')
NORM_OPS = ('identity', 'none', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3', 'dep_sep_conv_5')
OPS_OF = {'up': UP_OPS, 'down': DOWN_OPS, 'norm': NORM_OPS}

# utils/genotype.py:5
Genotype = namedtuple('Genotype', ['down', 'down_concat', 'up', 'up_concat', 'gamma'])

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class View:
    """Prefix view on a flat state dict (key -> tensor)."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str = ''):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, key: str) -> torch.Tensor:
        return self.sd[self.prefix + key]

    def has(self, key: str) -> bool:
        return (self.prefix + key) in self.sd

    def sub(self, name) -> 'View':
        return View(self.sd, '%s%s.' % (self.prefix, name))


# ----------------------------------------------------------------------------- leaf arithmetic
def batch_norm(p: View, x, training: bool):
    """nn.BatchNorm2d(c, affine=True) -- utils/operations.py:133-134 (eps 1e-5, momentum 0.1)."""
    if training and p.has('num_batches_tracked'):
        p['num_batches_tracked'].add_(1)
    return F.batch_norm(x, p['running_mean'], p['running_var'], p['weight'], p['bias'],
                        training, BN_MOMENTUM, BN_EPS)


def conv(x, w, k: int, stride: int = 1, dil: int = 1, transposed: bool = False, groups: int = 1):
    """build_weight -- utils/operations.py:118-130: bias-free, padding = (k//2)*dilation,
    ConvTranspose2d with output_padding = 1 for the UP type (stride 2)."""
    pad = (k // 2) * dil
    if transposed:
        return F.conv_transpose2d(x, w, None, stride, pad, stride - 1, groups, dil)
    return F.conv2d(x, w, None, stride, pad, dil, groups)


def _geometry(kind: str):
    """build_ops -- utils/operations.py:58-60."""
    return (1 if kind == 'norm' else 2), kind == 'up'


def dropout2d(x, dp, training):
    """nn.Dropout2d(dp, inplace=False) in front of a convolution when dp > 0 -- utils/operations.py:121-122 (whole
    channels of an image are zeroed, the rest scaled by 1 / (1 - dp); the identity in eval mode)."""
    return F.dropout2d(x, dp, training) if dp > 0 else x


def conv_bn(p: View, x, k, kind, dil, training, dp=0.0):
    """ConvBn -- utils/operations.py:89-95 (Sequential: [dropout,] conv, norm: the child indices shift by d = 1 with
    dropout, build_weight :118-130)."""
    stride, tr = _geometry(kind)
    d = 1 if dp > 0 else 0
    return batch_norm(p.sub(d + 1), conv(dropout2d(x, dp, training), p['%d.weight' % d], k, stride, dil, tr), training)


def se_block(p: View, x):
    """SEBlock -- utils/operations.py:186-203: squeeze, Linear-ReLU-Linear-Sigmoid, scale."""
    n, c = x.shape[:2]
    y = x.mean(dim=(2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, p['excitation.0.weight'])), p['excitation.2.weight']))
    return x * y.view(n, c, 1, 1)


def conv_bn_se(p: View, x, k, kind, training, dp=0.0):
    """ConvBnSe -- utils/operations.py:98-104 ([dropout,] conv, norm, se)."""
    return se_block(p.sub((1 if dp > 0 else 0) + 2), conv_bn(p, x, k, kind, 1, training, dp))


def dep_sep_conv(p: View, x, k, kind, training, dp=0.0):
    """DepSepConv -- utils/operations.py:107-115 ([dropout,] dw conv, norm, relu, [dropout,] 1x1 conv, norm: children
    0 1 2 3 4 without dropout, 1 2 3 5 6 with)."""
    stride, tr = _geometry(kind)
    c = x.shape[1]
    d = 1 if dp > 0 else 0
    y = conv(dropout2d(x, dp, training), p['%d.weight' % d], k, stride, 1, tr, groups=c)
    y = F.relu(batch_norm(p.sub(d + 1), y, training))
    y = conv(dropout2d(y, dp, training), p['%d.weight' % (2 * d + 3)], 1)
    return batch_norm(p.sub(2 * d + 4), y, training)


def adapter(p: View, y, training):
    """AdapterBlock tail -- utils/operations.py:178-183: optional 1x1 conv, then norm."""
    if p.has('conv.weight'):
        y = F.conv2d(y, p['conv.weight'])
    return batch_norm(p.sub('norm'), y, training)


def avg_pool3(x, stride):
    return F.avg_pool2d(x, 3, stride, 1, count_include_pad=False)


def bilinear_x2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def candidate(name: str, kind: str, p: View, x, training: bool, dp: float = 0.0):
    """OPS[name](c_in, c_ot, op_type, dp).forward(x) -- utils/operations.py:8-21,57-78 (dp reaches the weighted
    candidates only: build_ops :61-77; the search cell passes 0, search/cell.py:29)."""
    stride, _ = _geometry(kind)
    if name == 'none':            # ZeroOp(stride=1): x.mul(0.) -- operations.py:9,155-164
        return adapter(p, x.mul(0.), training)
    if name == 'identity':
        return adapter(p, x, training)
    if name == 'avg_pool':
        return adapter(p, avg_pool3(x, stride), training)
    if name == 'max_pool':
        return adapter(p, F.max_pool2d(x, 3, stride, 1), training)
    if name == 'up_sample':
        return adapter(p, bilinear_x2(x), training)
    if name == 'conv_3':
        return conv_bn(p, x, 3, kind, 1, training, dp)
    if name == 'se_conv_3':
        return conv_bn_se(p, x, 3, kind, training, dp)
    if name == 'dil_3_conv_5':
        return conv_bn(p, x, 5, kind, 3, training, dp)
    if name == 'dil_2_conv_5':
        return conv_bn(p, x, 5, kind, 2, training, dp)
    if name == 'dep_sep_conv_3':
        return dep_sep_conv(p, x, 3, kind, training, dp)
    if name == 'dep_sep_conv_5':
        return dep_sep_conv(p, x, 5, kind, training, dp)
    raise NotImplementedError(name)


# ----------------------------------------------------------------------------- cells
def mixed_op(p: View, x, kind: str, alpha_row, training):
    """MixedOp.forward -- search/cell.py:32-36 (k == 1, so the PC-DARTS branch :38-42 is dead)."""
    out = None
    for k, name in enumerate(OPS_OF[kind]):
        term = alpha_row[k] * candidate(name, kind, p.sub('_ops.%d' % k), x, training)
        out = term if out is None else out + term
    return out


def edge_kind(cell_type: str, j: int) -> str:
    """search/cell.py:81-89 and models/senas_model.py:38-46: which OpType edge j (input j) gets."""
    if j >= 2:
        return 'norm'
    if cell_type == 'down':
        return 'down'
    return 'up' if j > 0 else 'norm'


def preprocess0(p: View, x, cell_type, training):
    """down: build_rectify -- operations.py:141-152; up: ShrinkBlock -- operations.py:206-218."""
    x = F.relu(x)
    if cell_type == 'down':
        if p.has('1.weight'):          # c_in0 != c_in1: 1x1 stride-2 conv
            x = F.conv2d(x, p['1.weight'], None, 2)
        else:
            x = avg_pool3(x, 2)
        return batch_norm(p.sub(2), x, training)
    x = F.conv2d(x, p['conv.weight'], None, 1, 1)
    return batch_norm(p.sub('norm'), x, training)


def post_process(p: View, x, training):
    """RectifyBlock -- operations.py:221-232: 3x3 conv + norm."""
    return batch_norm(p.sub('norm'), F.conv2d(x, p['conv.weight'], None, 1, 1), training)


def search_cell(p: View, in0, in1, w_norm, w_chg, betas, cell_type, nodes, training):
    """Cell.forward -- search/cell.py:92-110."""
    states = [preprocess0(p.sub('preprocess0'), in0, cell_type, training), F.relu(in1)]
    off = 0
    for _ in range(nodes):
        acc = None
        for j, h in enumerate(states):
            kind = edge_kind(cell_type, j)
            row = (w_norm if kind == 'norm' else w_chg)[off + j]
            e = betas[off + j] * mixed_op(p.sub('_ops.%d' % (off + j)), h, kind, row, training)
            acc = e if acc is None else acc + e
        off += len(states)
        states.append(F.relu(acc))
    return post_process(p.sub('post_process'), torch.cat(states[-nodes:], 1), training)


def build_cell(p: View, in0, in1, genotype: Genotype, cell_type, training, dropout_prob=0.0):
    """BuildCell.forward -- models/senas_model.py:50-64 (dropout_prob goes to every candidate op, :38-46)."""
    gene = genotype.up if cell_type == 'up' else genotype.down
    concat = genotype.up_concat if cell_type == 'up' else genotype.down_concat
    states = [preprocess0(p.sub('preprocess0'), in0, cell_type, training), F.relu(in1)]
    for i in range(len(gene) // 2):
        hs = []
        for e in (2 * i, 2 * i + 1):
            name, idx = gene[e]
            hs.append(candidate(name, edge_kind(cell_type, idx), p.sub('_ops.%d' % e), states[idx], training, dropout_prob))
        states.append(F.relu(hs[0] + hs[1]))
    return post_process(p.sub('post_process'), torch.cat([states[i] for i in concat], 1), training)


# ----------------------------------------------------------------------------- macro grid
def stem0(p: View, x, training):
    """stem0 = ConvBn(in_channels, c, kernel_size=7) -- search/senas_search.py:30, senas_model.py:93."""
    return conv_bn(p, x, 7, 'norm', 1, training)


def stem1(p: View, s0, training):
    """stem1 = Sequential(ReLU, MaxPool(3,2,1), BasicBlock) -- senas_search.py:31-33;
    BasicBlock -- operations.py:235-268 (conv-bn-relu-conv-bn + residual, no ReLU after the add)."""
    b = p.sub(2)
    y = F.max_pool2d(F.relu(s0), 3, 2, 1)
    t = F.relu(batch_norm(b.sub('bn1'), F.conv2d(y, b['conv1.weight'], None, 1, 1), training))
    t = batch_norm(b.sub('bn2'), F.conv2d(t, b['conv2.weight'], None, 1, 1), training)
    return t + y


def relu_conv(p: View, x):
    """ReLUConv(c, nclass, kernel_size=3): ReLU then bias-free 3x3 conv, no norm -- operations.py:81-86."""
    return F.conv2d(F.relu(x), p['1.weight'], None, 1, 1)


def stem(p: View, x, training):
    s0 = stem0(p.sub('stem0'), x, training)
    return s0, stem1(p.sub('stem1'), s0, training)


def softmax_arch(sd, nodes):
    """NAS.forward prologue -- search/senas_search.py:246-260."""
    a = {k: F.softmax(sd[k], dim=-1) for k in ('alphas_dn_nm', 'alphas_up_nm', 'alphas_dn', 'alphas_up')}
    for k in ('betas_dn', 'betas_up'):
        # senas_search.py:254-257: ``offset = len(betas_dn)`` is the length of a *list of
        # tensors*, i.e. the node index i -- the slices are [0:2], [1:4], [2:6] (overlapping),
        # not the per-node partitions [0:2], [2:5], [5:9].  Restated as the reference computes it.
        a[k] = torch.cat([F.softmax(sd[k][i:i + 2 + i], dim=-1) for i in range(nodes)])
    a['gamma'] = F.softmax(sd['gamma'], dim=-1)
    return a


def nas_forward(sd, x, depth=5, nodes=3, supervision=False, training=True):
    """NAS.forward + SenasSearch.forward -- search/senas_search.py:246-260,76-112."""
    a = softmax_arch(sd, nodes)
    net = View(sd, 'net.')
    s0, c0 = stem(net, x, training)
    outs = [c0]
    for j in range(1, depth):
        prev = s0 if j == 1 else outs[-2]
        outs.append(search_cell(net.sub('blocks.0.%d' % j), prev, outs[-1], a['alphas_dn_nm'], a['alphas_dn'],
                                a['betas_dn'], 'down', nodes, training))
    g = a['gamma']
    for j in reversed(range(depth - 1)):
        for i in range(1, depth - j):
            parts = [outs[j]]
            for k in range(1, i):
                gi = sum(range(k + j)) + j
                parts.append(outs[j + k - 1] * g[gi][0] + outs[j + k] * g[gi][1])
            outs[i + j] = search_cell(net.sub('blocks.%d.%d' % (i, j)), torch.cat(parts, 1), outs[i + j],
                                      a['alphas_up_nm'], a['alphas_up'], a['betas_up'], 'up', nodes, training)
    head = net.sub('head_block.0')

    def run_head(o):
        y = search_cell(head.sub('up_cell'), s0, o, a['alphas_up_nm'], a['alphas_up'], a['betas_up'], 'up', nodes,
                        training)
        return relu_conv(head.sub('segmentation_head'), y)

    return [run_head(o) for o in outs] if supervision else [run_head(outs[-1])]


def derived_forward(sd, x, genotype: Genotype, depth=5, supervision=False, training=True, dropout_prob=0.0):
    """SenasModel.forward -- models/senas_model.py:146-179 (dropout_prob: the down and up cells, :110-111,133-134; the
    head's cell is built without, :143)."""
    net = View(sd, '')
    s0, c0 = stem(net, x, training)
    outs: List[Optional[torch.Tensor]] = [c0]
    for j in range(1, depth):
        prev = s0 if j == 1 else outs[-2]
        outs.append(build_cell(net.sub('blocks.0.%d' % j), prev, outs[-1], genotype, 'down', training, dropout_prob))
    for j in reversed(range(depth - 1)):
        for i in range(1, depth - j):
            gi = sum(range(i + j)) + j
            if i + j < depth - 1 and genotype.gamma[gi] == 0:
                outs[i + j] = None
                continue
            in0 = torch.cat([outs[t] for t in range(j, i + j) if outs[t] is not None], 1)
            outs[i + j] = build_cell(net.sub('blocks.%d.%d' % (i, j)), in0, outs[i + j], genotype, 'up', training, dropout_prob)
    head = net.sub('head_block.0')

    def run_head(o):
        y = build_cell(head.sub('up_cell'), s0, o, genotype, 'up', training)
        return relu_conv(head.sub('segmentation_head'), y)

    return [run_head(o) for o in outs] if supervision else [run_head(outs[-1])]


# ----------------------------------------------------------------------------- genotype derivation
def parse_cell(w_norm: np.ndarray, w_chg: np.ndarray, cell_type: str, nodes: int):
    """GenoParser.parse -- utils/genotype.py:13-90.  Per node: the best non-'none' op of every
    incoming edge, then the two strongest edges (ascending strength; ties resolved like
    ``sorted`` on (weight, op, idx) tuples)."""
    chg_names = UP_OPS if cell_type == 'up' else DOWN_OPS
    n_chg = 2 if cell_type == 'down' else 1
    gene, start = [], 0
    for i in range(nodes):
        n_in = 2 + i
        if cell_type == 'down':
            chg_rows = [(start + e, e) for e in range(n_chg)]                       # -> input idx e
            nrm_rows = [(start + e, e) for e in range(n_chg, n_in)]
        else:
            chg_rows = [(start + 1, 1)]
            nrm_rows = [(start, 0)] + [(start + e, e) for e in range(2, n_in)]

        def best(table, names, r):
            ks = [k for k in range(table.shape[1]) if names[k] != 'none']
            kb = ks[0]
            for k in ks[1:]:
                if table[r][k] > table[r][kb]:
                    kb = k
            return table[r][kb], names[kb]

        def top2(rows, table, names):
            cand = [best(table, names, r) + (idx,) for r, idx in rows]
            order = sorted(range(len(cand)), key=lambda t: -cand[t][0])[:2]   # stable, like the reference
            return [cand[t] for t in order]

        items = top2(nrm_rows, w_norm, NORM_OPS) + top2(chg_rows, w_chg, chg_names)
        # genotype.py:77-82: rescale when the two op lists differ in length (both are 6 today)
        if len(NORM_OPS) != len(chg_names) and nrm_rows and chg_rows:
            raise NotImplementedError('op lists of different length')
        gene += [(name, idx) for (_, name, idx) in sorted(items)[-2:]]
        start += n_in
    return gene


def derive_genotype(sd, depth=5, nodes=3) -> Genotype:
    """NAS.genotype -- search/senas_search.py:203-244."""
    with torch.no_grad():
        a = softmax_arch({k: v.detach().cpu() for k, v in sd.items() if not k.startswith('net.')}, nodes)
        tabs = {}
        for cell, (nm, chg, b) in {'down': ('alphas_dn_nm', 'alphas_dn', 'betas_dn'),
                                   'up': ('alphas_up_nm', 'alphas_up', 'betas_up')}.items():
            t_nm, t_chg = a[nm].clone(), a[chg].clone()
            for j in range(t_nm.shape[0]):
                t_nm[j, :] = t_nm[j, :] * a[b][j].item()
                t_chg[j, :] = t_chg[j, :] * a[b][j].item()
            tabs[cell] = (t_nm.numpy(), t_chg.numpy())
        g = a['gamma']
        drop = set(torch.topk(g[:, 1], len(g) // 2, largest=False).indices.tolist())
        hard = [0 if i in drop else v for i, v in enumerate(g.argmax(1).tolist())]
    path = []
    for i in range(1, depth - 1):
        seg = hard[sum(range(i)): sum(range(i)) + i]
        if 1 in seg:
            first = seg.index(1)
            seg = seg[:first] + [1] * (len(seg) - first)
        path += seg
    concat = range(2, nodes + 2)
    return Genotype(down=parse_cell(*tabs['down'], 'down', nodes), down_concat=concat,
                    up=parse_cell(*tabs['up'], 'up', nodes), up_concat=concat, gamma=path)


# ----------------------------------------------------------------------------- loss and metric
def soft_dice_loss(logits, target, smooth=1e-5):
    """SoftDiceLoss (batch_dice True, do_bg False, softmax applied inside) -- utils/loss/loss.py:45-70,173-228."""
    prob = F.softmax(logits, 1)
    onehot = torch.zeros_like(prob).scatter_(1, target.long().unsqueeze(1), 1)
    axes = [0] + list(range(2, logits.dim()))
    tp = (prob * onehot).sum(axes)
    fp = (prob * (1 - onehot)).sum(axes)
    fn = ((1 - prob) * onehot).sum(axes)
    dc = (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8)
    return 1 - dc[1:].mean()


def dice_ce_loss(logits, target, smooth=1e-5):
    """DiceCrossEntropyLoss (weights 1/1) = nn.CrossEntropyLoss + SoftDiceLoss -- utils/loss/loss.py:124-159."""
    return F.cross_entropy(logits, target.long()) + soft_dice_loss(logits, target, smooth)


def multi_dice_ce_loss(outputs, target, depth, weight_factors=None, smooth=1e-5):
    """MultiSegmentationLosses('dice_ce', depth, weight_factors) -- utils/loss/loss.py:30-43: the weighted sum of the
    per-output losses (zip stops at the shorter of factors / outputs) divided by the NUMBER OF OUTPUTS."""
    factors = [1] * depth if weight_factors is None else list(weight_factors)
    assert len(factors) == depth
    total = 0
    for wf, logits in zip(factors, outputs):
        total = total + wf * dice_ce_loss(logits, target, smooth)
    return total / len(outputs)


def hard_counts(logits, label):
    """confusion_matrix -- utils/metrics.py:145-162: per-class (1..C-1) hard TP/FP/FN over the batch."""
    seg = F.softmax(logits, 1).argmax(1)
    out = []
    for c in range(1, logits.shape[1]):
        p, t = seg == c, label == c
        out.append(((p & t).sum().item(), (p & ~t).sum().item(), (~p & t).sum().item()))
    tp, fp, fn = (np.array(v, dtype=np.float32) for v in zip(*out))
    return tp, fp, fn


def dice_from_counts(tp, fp, fn):
    """SegmentationMetric.dice / percentage -- utils/metrics.py:8,60-64,99-104."""
    s = np.spacing(1)
    return round(100.0 * float(np.mean((2 * tp + s) / (2 * tp + fp + fn + s))), 3)


def miou_from_counts(tp, fp, fn):
    s = np.spacing(1)
    return round(100.0 * float(np.mean((tp + s) / (tp + fp + fn + s))), 3)


def mean_pix_accuracy(logits, target):
    """mean_pix_accuracy -- utils/metrics.py:127-142 (bitwise AND of the arg-max with target>0)."""
    s = np.spacing(1)
    predict = logits.argmax(1)
    labeled = (target > 0).float().sum((1, 2))
    correct = (predict & (target > 0)).float().sum((1, 2))
    return ((correct + s) / (labeled + s)).mean()
