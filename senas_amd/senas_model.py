"""Derived ("expanded") network of the SENAS train phase: ``BuildCell`` wires the ops a
``Genotype`` names, ``SenasModel`` stacks them on the macro grid, pruning the up cells whose gamma
entry is 0.  Same public surface as the reference's ``models/senas_model.py`` (BuildCell :4-64,
Head :67-75, SenasModel :78-179).
"""
import torch
import torch.nn as nn

from . import functional as F
from .grid import FanPlan, MacroGrid, NoPlan, gamma_index
from .operations import OPS, OpType, RectifyBlock, ReLUConv, ShrinkBlock, build_activation, build_rectify


class BuildCell(nn.Module):
    def __init__(self, genotype, double_down, c_in0, c_in1, c_out, cell_type, dropout_prob=0):
        super().__init__()
        if cell_type == 'down':
            self.preprocess0 = build_rectify(c_in0, c_in1, cell_type)
            c_part = c_out // double_down
            gene, concat = genotype.down, genotype.down_concat
        else:
            self.preprocess0 = ShrinkBlock(c_in0, c_in1)
            c_part = c_out
            gene, concat = genotype.up, genotype.up_concat
        self.preprocess1 = build_activation(False)
        self.node_activation = build_activation()
        self.post_process = RectifyBlock(c_part * len(concat), c_out, cell_type=cell_type)
        self.dropout_prob = dropout_prob
        self._concat = concat
        self._multiplier = len(concat)
        self._input_num = 2
        self._num_meta_node = len(gene) // 2
        self._indices = tuple(idx for _, idx in gene)

        def edge_type(idx):
            if idx >= self._input_num:
                return OpType.NORM
            if cell_type == 'down':
                return OpType.DOWN
            return OpType.UP if idx > 0 else OpType.NORM

        self._ops = nn.ModuleList(
            OPS[name](c_in1 if idx < self._input_num else c_part, c_part, edge_type(idx), dropout_prob)
            for name, idx in gene)

    paired = True       # dense stride-1 candidates that read the SAME state run two to a launch (class-wide switch)

    def _pairs(self):
        """{state: [(edge a, edge b)]}: ConvBn candidates (plain Conv2d, same kernel size and channels, no dropout) that read
        one state -- two nodes taking the same op from the same state, or dil_3_conv_5 beside dil_2_conv_5.  Each pair is ONE
        forward launch and one for its two data gradients (functional.conv2d_pair): on the small maps of the deep cells a
        launch of either alone leaves most CUs idle."""
        from .operations import ConvBn
        plan = self.__dict__.get('_pair_plan')
        if plan is None:
            by_key = {}
            for e, idx in enumerate(self._indices):
                m = self._ops[e]
                if type(m) is ConvBn and type(m.conv) is nn.Conv2d and m.drop is None and m.conv.groups == 1:
                    c = m.conv
                    by_key.setdefault((idx, tuple(c.weight.shape), c.stride), []).append(e)
            plan = {}
            for (idx, _, _), es in by_key.items():
                for i in range(0, len(es) - 1, 2):
                    plan.setdefault(idx, []).append((es[i], es[i + 1]))
            self.__dict__['_pair_plan'] = plan
        return plan

    def forward(self, in0, in1):
        # every consumer of a state (ops reading it, the output concat) gets its own alias, so that the state's
        # gradient is ONE n-ary sum (functional.fan_out) instead of n-1 autograd accumulations
        total = self._input_num + self._num_meta_node
        uses = [sum(1 for idx in self._indices if idx == k) + (1 if k in self._concat else 0) for k in range(total)]
        states = []
        ready = {}                       # edge -> Term of a candidate that already ran beside its partner
        pairs = self._pairs() if self.paired else {}

        def add_state(h):
            k = len(states)
            it = iter(F.fan_out(h, uses[k]))
            states.append(it)
            for ea, eb in pairs.get(k, ()):
                ma, mb = self._ops[ea], self._ops[eb]
                ca, cb = ma.conv, mb.conv
                (za, sta), (zb, stb) = F.conv2d_pair(next(it), next(it), ca.weight, cb.weight, ca.stride[0], ca.padding[0],
                                                    ca.dilation[0], cb.padding[0], cb.dilation[0], want_stats=ma.norm.training)
                ready[ea], ready[eb] = F.Term(za, ma.norm, stats=sta), F.Term(zb, mb.norm, stats=stb)

        add_state(self.preprocess0(in0))
        add_state(self.preprocess1(in1))
        # Nodes that sit in the output concatenation write their result straight into its channel slice (node.bn_combine
        # cat=...): no torch.cat launch; a node that nothing else reads is not stored anywhere else.
        concat = list(self._concat)
        direct = all(k >= self._input_num for k in concat) and len(set(concat)) == len(concat)
        catbuf = None
        for i in range(self._num_meta_node):
            pair = [ready.pop(e) if e in ready else self._ops[e].raw(next(states[self._indices[e]])) for e in (2 * i, 2 * i + 1)]
            k = self._input_num + i
            cat = None
            if direct and k in concat:
                ref = next((t.z for t in pair if t.z is not None), None)
                if ref is not None and ref.shape[1] % 4 == 0:
                    n, c, h, w = ref.shape
                    if catbuf is None:
                        catbuf = F.new_nhwc(n, c * len(concat), h, w, ref)
                    cat = (catbuf, concat.index(k) * c, uses[k] > 1)
                else:
                    direct = False
            add_state(F.bn_combine(pair, relu=True, cat=cat))               # ReLU(op_a(.) + op_b(.)) in one pass
        outs = [next(states[i]) for i in concat]
        if direct and catbuf is not None:
            return self.post_process(F.cat_slices(catbuf, catbuf.shape[1] // len(concat), outs))
        return self.post_process(torch.cat(outs, dim=1))


class Head(nn.Module):
    def __init__(self, genotype, double_down, c_in0, c_in1, nclass):
        super().__init__()
        self.up_cell = BuildCell(genotype, double_down, c_in0, c_in1, c_in1, cell_type='up')
        self.segmentation_head = ReLUConv(c_in1, nclass, kernel_size=3)

    def forward(self, s0, ot):
        return self.segmentation_head(self.up_cell(s0, ot))


class SenasModel(MacroGrid):
    def __init__(self, nclass, in_channels, c=32, depth=5, dropout_prob=0, supervision=False, genotype=None,
                 double_down_channel=False):
        double = 2 if double_down_channel else 1
        gamma = genotype.gamma

        def make_cell(kind, c0, c1, co, i, j):
            if kind == 'up' and i + j < depth - 1 and gamma[gamma_index(i, j)] == 0:
                return None                                        # pruned skip cell
            return BuildCell(genotype, double, c0, c1, co, cell_type=kind, dropout_prob=dropout_prob)

        super().__init__(in_channels, c, nclass, depth, double_down_channel, make_cell=make_cell,
                         make_head=lambda c0, c1, ncls: Head(genotype, double, c0, c1, ncls))
        self._supervision = supervision
        self._meta_node_num = len(genotype.down_concat)
        self.gamma = gamma

    def forward(self, x):
        if self.cut is not None:              # two-part backward (several ranks): the cut tensors are re-leafed, no aliases
            return self._walk(NoPlan(), x)
        plan = self.__dict__.get('_fan_plan')
        if plan is None:                      # dry run of the schedule: how many readers every tensor has (grid.FanPlan)
            plan = self.__dict__['_fan_plan'] = FanPlan()
            self._walk(plan, None)
        return self._walk(plan.start(), x)

    def _walk(self, plan, x):
        """The reference's forward pass (models/senas_model.py:150-179) on the shared schedule (grid.MacroGrid._walk_grid):
        in0 of up cell (i, j) is the concatenation of the column's outputs below it that the genotype keeps (:165-170)."""
        def run(module, kind, a, b):
            return module(a, b)

        def skips(plan, G, i, j, live, fetch=None):
            col = [(k, plan.get(G[k][j])) for k in range(i) if G[k][j] is not None]
            return [fetch(k, t) if (live and fetch is not None) else t for k, t in col]

        return self._walk_grid(plan, x, run, skips)
