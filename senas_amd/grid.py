"""The UNet++-style macro grid shared by the supernet and the derived network.

Level 0 is the down path (stem1 + depth-1 down cells); level i >= 1 holds the up cells
``(i, j)``, j = 0 .. depth-i-1.  Up cell (i, j) takes as in0 the concatenation of the outputs that
currently sit at positions j .. i+j-1 of the running output list (all at resolution level j) and
as in1 the output at position i+j, and overwrites position i+j.
Reference: search/senas_search.py:16-112 and models/senas_model.py:78-179 build and walk the same
grid; here the bookkeeping lives in one place.
"""
import torch.nn as nn

from .operations import ConvBn, Stem1


class FanPlan(object):
    """Readers of the tensors a macro-grid forward pass produces.  A tensor with several consumers (a cell output that
    is a skip input of later cells, the in1 of the next cell and a blend operand) would have its gradient accumulated by
    autograd in n - 1 binary adds; with the number of readers known up front every reader gets its own alias and the
    gradient is ONE n-ary sum (functional.fan_out).  The schedule is static, so the counts come from a dry run of the
    same forward code: ``put`` / ``get`` count in dry mode and hand out aliases in live mode."""

    def __init__(self):
        self.counts, self.live, self.dry = {}, {}, True

    def put(self, key, value=None):
        if self.dry:
            self.counts[key] = 0
        else:
            from . import functional as F
            n = self.counts[key]
            self.live[key] = iter(F.fan_out(value, n)) if n > 1 else iter([value] * max(n, 1))
        return key

    def get(self, key):
        if self.dry:
            self.counts[key] += 1
            return None
        return next(self.live[key])

    def start(self):
        """Switch to live mode for one forward pass (the counts stay)."""
        self.dry = False
        self.live = {}
        return self

    def __deepcopy__(self, memo):
        fresh = FanPlan()                 # (the aliases of the last pass are not part of a copied model)
        fresh.counts, fresh.dry = dict(self.counts), self.dry
        return fresh


def gamma_index(i, j):
    """Index into the flat gamma table of the skip feeding row i+j from column j."""
    return sum(range(i + j)) + j


class MacroGrid(nn.Module):
    """Owns ``stem0``, ``stem1``, ``blocks`` and ``head_block`` with the reference's names.

    make_cell(cell_type, c_in0, c_in1, c_out, i, j) -> nn.Module or None (None = pruned up cell)
    make_head(c_in0, c_in1, nclass) -> nn.Module
    """

    def __init__(self, in_channels, c, nclass, depth, double_down_channel, make_cell, make_head):
        super().__init__()
        assert depth >= 2, 'depth must >= 2'
        self._depth = depth
        self._double_down_channel = double_down_channel
        double = 2 if double_down_channel else 1
        self.blocks = nn.ModuleList()
        self.stem0 = ConvBn(in_channels, c, kernel_size=7)
        self.stem1 = Stem1(c, c)
        widths = [[c]]                       # widths[level][j] = channels produced at grid slot (level, j)
        row = nn.ModuleList([self.stem1])
        c_in0, c_in1, c_cur = c, c, c
        for _ in range(1, depth):
            c_cur = int(double * c_cur)
            row.append(make_cell('down', c_in0, c_in1, c_cur, 0, len(row)))
            widths[0].append(c_cur)
            c_in0, c_in1 = c_in1, c_cur
        self.blocks.append(row)
        for i in range(1, depth):
            row, wrow = nn.ModuleList(), []
            for j in range(depth - i):
                c_skip = sum(widths[k][j] for k in range(i))
                cell = make_cell('up', c_skip, widths[i - 1][j + 1], widths[0][j], i, j)
                row.append(cell)
                wrow.append(widths[0][j] if cell is not None else 0)
            self.blocks.append(row)
            widths.append(wrow)
        self.head_block = nn.ModuleList([make_head(c, widths[-1][0], nclass)])
        # step drivers may cut the autograd graph at the outputs of the down path (senas_amd.step: backward is then run
        # -- and captured -- in two parts, so that the up-path gradients are all-reduced while the rest still runs)
        self.cut = None

    def down_parameters(self):
        """Parameters of the down path (stems, down cells): the ones whose gradients backward produces last."""
        out, seen = [], set()
        for m in [self.stem0, self.stem1] + list(self.blocks[0]):
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    def _down_done(self, s0, outs):
        """Called by the subclasses' forward with the stem output and the down-path outputs."""
        if self.cut is None:
            return s0, outs
        cut = self.cut([s0] + list(outs))
        return cut[0], list(cut[1:])
