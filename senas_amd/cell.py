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
        return [op.raw(x) for op in self._ops]

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
        states = [self.preprocess0(in0), self.preprocess1(in1)]
        mixes = self._node_mixes(weights_norm, weights_chg, betas)
        offset = 0
        for i in range(self._meta_node_num):
            terms = []
            for j, h in enumerate(states):
                terms += self._ops[offset + j].terms(h)
            offset += len(states)
            states.append(F.bn_combine(terms, mix=mixes[i], relu=True))
        return self.post_process(torch.cat(states[-self._meta_node_num:], dim=1))
