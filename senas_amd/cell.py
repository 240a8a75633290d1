"""Search-phase cell: MixedOp (alpha-weighted sum of the candidate ops of one edge) and Cell
(beta-weighted DAG of MixedOps).  Same constructors, forward signatures and parameter names as the
reference's ``search/cell.py`` (``MixedOp`` :5-43, ``Cell`` :46-110).

Execution differs: a node is ONE ``bn_combine`` launch over the raw outputs of all candidates of
all incoming edges -- batch-norm, SE gate, alpha and beta weights, the edge sum and the node ReLU
are per-channel coefficients of that single pass instead of ~13 elementwise passes per edge.
"""
import torch
import torch.nn as nn

from . import functional as F
from .operations import OPS, OpType, RectifyBlock, ShrinkBlock, build_activation, build_rectify


class MixedOp(nn.Module):
    def __init__(self, c_in, c_out, op_type):
        super().__init__()
        self._op_type = op_type
        self.k = 1                       # PC-DARTS style partial channels are off in SENAS (k == 1)
        self.c_out = c_out
        self.c_part = c_out // self.k
        self._ops = nn.ModuleList(OPS[name](c_in, self.c_part, op_type, dp=0) for name in op_type.value['ops'])

    def pick(self, alpha_normal, alpha_up_dn):
        return alpha_normal if self._op_type == OpType.NORM else alpha_up_dn

    def terms(self, x):
        """x: the edge's input, or one alias of it per candidate (functional.fan_out)."""
        xs = x if isinstance(x, (list, tuple)) else [x] * len(self._ops)
        return [op.raw(xi) for op, xi in zip(self._ops, xs)]

    def forward(self, x, alpha_normal, alpha_up_dn):
        return F.bn_combine(self.terms(x), mix=self.pick(alpha_normal, alpha_up_dn))


class Cell(nn.Module):
    def __init__(self, meta_node_num, double_down, c_in0, c_in1, c_out, cell_type):
        super().__init__()
        self.k = 4                       # "shrink": every edge works on c_out / k channels while searching
        self._meta_node_num = meta_node_num
        self._input_num = 2
        if cell_type == 'down':
            self.preprocess0 = build_rectify(c_in0, c_in1, cell_type)
            c_part = (c_out // double_down) // self.k
        else:
            self.preprocess0 = ShrinkBlock(c_in0, c_in1)
            c_part = c_out // self.k
        self.preprocess1 = build_activation(False)
        self.node_activation = build_activation()
        self.post_process = RectifyBlock(c_part * meta_node_num, c_out, cell_type=cell_type)

        def edge_type(j):
            if j >= self._input_num:
                return OpType.NORM
            if cell_type == 'down':
                return OpType.DOWN
            return OpType.UP if j > 0 else OpType.NORM

        self._ops = nn.ModuleList()
        for i in range(meta_node_num):
            for j in range(self._input_num + i):
                self._ops.append(MixedOp(c_in1 if j < self._input_num else c_part, c_part, edge_type(j)))

    def _node_mixes(self, weights_norm, weights_chg, betas):
        """Per node, the concatenated ``beta_j * alpha_row_j`` of its incoming edges (search/cell.py:100-106).
        They depend on the architecture tensors and the cell KIND only, so all cells of a kind share them within
        a forward pass: the list is parked on the ``betas`` tensor (fresh softmax outputs every pass, so nothing
        goes stale) -- 2 x nodes small tensors per pass instead of one set per cell (15x fewer tiny kernels,
        forward and backward)."""
        kinds = tuple(edge._op_type == OpType.NORM for edge in self._ops)
        key = (id(weights_norm), id(weights_chg), kinds)
        cache = betas.__dict__.setdefault('_senas_mix', {}) if hasattr(betas, '__dict__') else {}
        if key not in cache:
            mixes, offset = [], 0
            for i in range(self._meta_node_num):
                cnt = self._input_num + i
                rows = [betas[offset + j] * (weights_norm if kinds[offset + j] else weights_chg)[offset + j] for j in range(cnt)]
                mixes.append(torch.cat(rows))
                offset += cnt
            cache[key] = mixes
        return cache[key]

    def forward(self, in0, in1, weights_norm, weights_chg, betas):
        mixes = self._node_mixes(weights_norm, weights_chg, betas)
        # a state feeds every candidate of every outgoing edge (18 consumers for the two cell inputs): hand each
        # consumer its own alias, so the gradients meet in ONE n-ary sum instead of n-1 autograd accumulations
        nodes, nin = self._meta_node_num, self._input_num
        uses = [0] * (nin + nodes)
        offset = 0
        for i in range(nodes):
            for j in range(nin + i):
                uses[j] += len(self._ops[offset + j]._ops)
            offset += nin + i
        states = []

        def add_state(h):
            k = len(states)
            states.append(iter(F.fan_out(h, uses[k] + (1 if k >= nin else 0))))

        add_state(self.preprocess0(in0))
        add_state(self.preprocess1(in1))
        offset = 0
        for i in range(nodes):
            terms = []
            for j in range(nin + i):
                edge = self._ops[offset + j]
                terms += edge.terms([next(states[j]) for _ in edge._ops])
            offset += nin + i
            add_state(F.bn_combine(terms, mix=mixes[i], relu=True))
        return self.post_process(torch.cat([next(states[nin + i]) for i in range(nodes)], dim=1))
