"""Supernet of the SENAS search phase: ``SenasSearch`` (macro grid of search cells), ``NAS``
(architecture parameters alpha / beta / gamma, their softmaxes, genotype derivation) and
``Architecture`` (first-order arch step).  Same public surface as the reference's
``search/senas_search.py`` (Head :5-13, SenasSearch :16-112, NAS :115-279, Architecture :282-303).
"""
import torch
import torch.nn as nn
import torch.nn.functional as tf

from . import functional as F
from .cell import Cell
from .genotype import GenoParser, Genotype
from .grid import FanPlan, Lanes, MacroGrid, NoPlan, gamma_index
from .operations import DownOps, NormOps, ReLUConv, UpOps
from .utils import weights_init


class Head(nn.Module):
    def __init__(self, meta_node_num, double_down, c_in0, c_in1, nclass):
        super().__init__()
        self.up_cell = Cell(meta_node_num, double_down, c_in0, c_in1, c_in1, cell_type='up')
        self.segmentation_head = ReLUConv(c_in1, nclass, kernel_size=3)

    def forward(self, s0, ot, weights_up_norm, weights_up, betas_up):
        return self.segmentation_head(self.up_cell(s0, ot, weights_up_norm, weights_up, betas_up))


class SenasSearch(MacroGrid):
    def __init__(self, in_channels, c, nclass, depth, meta_node_num=3, double_down_channel=False, supervision=False):
        double = 2 if double_down_channel else 1
        super().__init__(in_channels, c, nclass, depth, double_down_channel,
                         make_cell=lambda kind, c0, c1, co, i, j: Cell(meta_node_num, double, c0, c1, co, cell_type=kind),
                         make_head=lambda c0, c1, ncls: Head(meta_node_num, double, c0, c1, ncls))
        self._supervision = supervision
        self._meta_node_num = meta_node_num

    def forward(self, x, alpha_dn_nm, alpha_up_nm, alpha_dn, alpha_up, beta_dn, beta_up, gamma):
        arch = (alpha_dn_nm, alpha_up_nm, alpha_dn, alpha_up, beta_dn, beta_up, gamma)
        if self.cut is not None:              # two-part backward (several ranks): the cut tensors are re-leafed, no aliases
            return self._walk(NoPlan(), x, arch)
        plan = self.__dict__.get('_fan_plan')
        if plan is None:                      # dry run of the schedule below: how many readers every tensor has
            plan = self.__dict__['_fan_plan'] = FanPlan()
            self._walk(plan, None, None)
        return self._walk(plan.start(), x, arch)

    def _walk(self, plan, x, arch):
        """The reference's forward pass (search/senas_search.py:96-107) on the shared schedule (grid.MacroGrid._walk_grid):
        in0 of up cell (i, j) is the concatenation of the column's down-path output and the gamma-gated blends of
        neighbouring outputs below it in the column (:98-102)."""
        rows = None
        if x is not None:
            alpha_dn_nm, alpha_up_nm, alpha_dn, alpha_up, beta_dn, beta_up, gamma = arch
            rows = getattr(gamma, '_senas_rows', None) or F.GammaRows(gamma)      # the blends read their gamma pair in place
            args = {'down': (alpha_dn_nm, alpha_dn, beta_dn), 'up': (alpha_up_nm, alpha_up, beta_up),
                    'head': (alpha_up_nm, alpha_up, beta_up)}
            # the mixing matrices of both cell kinds are made HERE, on the caller's stream (not by the first cell that asks,
            # on its lane): their backward nodes wait for every lane (functional.join_lanes), which only this stream may do
            self.blocks[0][1]._mix_slots(*args['down'])
            self.head_block[-1].up_cell._mix_slots(*args['up'])

        def run(module, kind, a, b):
            return module(a, b, *args[kind])

        def skips(plan, G, i, j, live, fetch=None):
            # the column's down-path output, then the gamma-gated blends of neighbouring skip candidates (:98-102) -- stacked
            # by one launch that reads every tensor of the column once (functional.skip_stack).  ``fetch(k, t)``: tensor k of
            # the column as the CURRENT stream may read it (grid.MacroGrid._walk_grid: the hand-over between lanes)
            col = [plan.get(G[k][j]) for k in range(i)]
            if live and fetch is not None:
                col = [fetch(k, t) for k, t in enumerate(col)]
            if not live or i == 1:
                return col
            if i > F.SKIP_MAX or col[0].shape[1] % 4 != 0:
                return [col[0]] + [F.blend2_row(col[k - 1], col[k], rows, gamma_index(k, j)) for k in range(1, i)]
            return [F.skip_stack(col, rows, [0] + [gamma_index(k, j) for k in range(1, i)])]

        return self._walk_grid(plan, x, run, skips)


def _node_softmax(beta, nodes):
    # The reference takes ``offset = len(list_of_slices)`` (search/senas_search.py:254-257), i.e. the
    # NODE index, so the softmax windows are [i : 2i+2] -- [0:2], [1:4], [2:6] -- and overlap.
    # Kept bit-for-bit: the searched genotypes and every checkpoint depend on it.
    return torch.cat([tf.softmax(beta[i:2 * i + 2], dim=-1) for i in range(nodes)], dim=0)


class NAS(nn.Module):
    def __init__(self, input_c, c, num_classes, depth, meta_node_num=4, use_sharing=True, double_down_channel=True,
                 use_softmax_head=False, supervision=False, multi_gpus=False, device='cuda'):
        super().__init__()
        self._use_sharing = use_sharing
        self._meta_node_num = meta_node_num
        self._depth = depth
        self.net = SenasSearch(input_c, c, num_classes, depth, meta_node_num, double_down_channel, supervision)
        self.net.apply(weights_init)
        # Multi-GPU is data parallelism with one process per GPU (senas_amd.parallel); the in-module
        # scatter/replicate path of the reference (:262-279) is not reproduced.
        self.device_ids = [0]
        self._init_alphas()

    def _init_alphas(self):
        k = sum(2 + i for i in range(self._meta_node_num))
        self.alphas_dn = nn.Parameter(1e-3 * torch.randn(k, len(DownOps)))
        self.alphas_up = nn.Parameter(1e-3 * torch.randn(k, len(UpOps)))
        self.alphas_dn_nm = nn.Parameter(1e-3 * torch.randn(k, len(NormOps)))
        self.alphas_up_nm = self.alphas_dn_nm if self._use_sharing else nn.Parameter(1e-3 * torch.randn(k, len(NormOps)))
        self.betas_dn = nn.Parameter(1e-3 * torch.randn(k))
        self.betas_up = nn.Parameter(1e-3 * torch.randn(k))
        self.gamma = nn.Parameter(1e-3 * torch.randn(sum(range(self._depth - 1)), 2))
        self._arch_parameters = [self.alphas_dn, self.alphas_up, self.alphas_dn_nm, self.alphas_up_nm, self.betas_dn,
                                 self.betas_up, self.gamma]

    def arch_parameters(self):
        return self._arch_parameters

    def alphas_dict(self):
        return {'alphas_dn': self.alphas_dn, 'alphas_dn_nm': self.alphas_dn_nm, 'alphas_up': self.alphas_up,
                'alphas_up_nm': self.alphas_up_nm}

    def betas_dict(self):
        return {'betas_dn': self.betas_dn, 'betas_up': self.betas_up}

    def load_params(self, alphas_dict, betas_dict):
        """Accepts the keys ``alphas_dict()`` / ``betas_dict()`` write.  (The reference reads a
        different key set than it writes -- senas_search.py:170-198 -- so its search resume raises
        KeyError; the legacy names are accepted too.)"""
        def pick(d, *names):
            for nme in names:
                if nme in d:
                    return d[nme]
            raise KeyError(names[0])
        self.alphas_dn = pick(alphas_dict, 'alphas_dn', 'alphas_down')
        self.alphas_up = pick(alphas_dict, 'alphas_up')
        self.alphas_dn_nm = pick(alphas_dict, 'alphas_dn_nm', 'alphas_normal_down')
        self.alphas_up_nm = pick(alphas_dict, 'alphas_up_nm', 'alphas_normal_up')
        self.betas_dn = pick(betas_dict, 'betas_dn', 'betas_down')
        self.betas_up = pick(betas_dict, 'betas_up')
        self._arch_parameters = [self.alphas_dn, self.alphas_up, self.alphas_dn_nm, self.alphas_up_nm, self.betas_dn,
                                 self.betas_up, self.gamma]

    def _mixing_weights(self):
        return (tf.softmax(self.alphas_dn_nm, dim=-1), tf.softmax(self.alphas_up_nm, dim=-1),
                tf.softmax(self.alphas_dn, dim=-1), tf.softmax(self.alphas_up, dim=-1),
                _node_softmax(self.betas_dn, self._meta_node_num), _node_softmax(self.betas_up, self._meta_node_num),
                tf.softmax(self.gamma, dim=-1))

    def forward(self, x):
        if x.is_cuda and self.alphas_dn.is_cuda and self.net.cut is None and not getattr(self, '_plain_arch', False):
            # one launch for every softmax and both mixing matrices (functional.ArchTables); the positional arguments keep
            # the reference's meaning (search/senas_search.py:259-260).  (Not under a two-part backward: its node would be
            # entered from both parts.)
            return self.net(x, *F.ArchTables(self._meta_node_num, self.alphas_dn, self.alphas_up, self.alphas_dn_nm, self.alphas_up_nm,
                                             self.betas_dn, self.betas_up, self.gamma).args())
        return self.net(x, *self._mixing_weights())

    def genotype(self):
        """Arg-max architecture (host-side, numpy): per-edge op tables scaled by the edge's beta,
        two strongest edges per node; gamma: drop the weaker half, arg-max, make each skip row monotone."""
        with torch.no_grad():
            a_dn_nm, a_up_nm, a_dn, a_up, b_dn, b_up, gamma = (t.detach().float().cpu() for t in self._mixing_weights())
        for tab, beta in ((a_dn_nm, b_dn), (a_dn, b_dn), (a_up_nm, b_up), (a_up, b_up)):
            for j in range(tab.shape[0]):
                tab[j, :] = tab[j, :] * beta[j].item()
        parser = GenoParser(self._meta_node_num)
        gene_down = parser.parse(a_dn_nm.numpy(), a_dn.numpy(), cell_type='down')
        gene_up = parser.parse(a_up_nm.numpy(), a_up.numpy(), cell_type='up')
        weakest = set(torch.topk(gamma[:, 1], len(gamma) // 2, largest=False).indices.tolist())
        keep = [0 if i in weakest else g for i, g in enumerate(gamma.argmax(1).tolist())]
        path = []
        for i in range(1, self._depth - 1):
            row = keep[sum(range(i)): sum(range(i)) + i]
            if 1 in row:
                row = row[:row.index(1)] + [1] * (len(row) - row.index(1))
            path += row
        concat = range(2, self._meta_node_num + 2)
        return Genotype(down=gene_down, down_concat=concat, up=gene_up, up_concat=concat, gamma=path)


class Architecture(object):
    """First-order DARTS architecture step: one forward/backward on a validation batch, Adam on alpha/beta/gamma."""

    def __init__(self, model, arch_optimizer, criterion):
        self.model, self.optimizer, self.criterion = model, arch_optimizer, criterion

    def step(self, input_valid, target_valid):
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(input_valid), target_valid)
        loss.backward()
        self.optimizer.step()
        return loss
