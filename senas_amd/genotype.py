"""Genotype record and the alpha-table -> genotype parser (host-side integer/arg-max logic).

Mirrors the reference surface ``utils/genotype.py`` (``Genotype`` at :5, ``GenoParser.parse`` at
:13-90): same constructor, same ``parse(weights1, weights2, cell_type)`` arguments, same
``[(op_name, input_idx), ...]`` result, bit-exact including tie behaviour.  This is CPU work in
the reference as well (numpy on a 9x6 table) and stays on the host here.
"""
from collections import namedtuple

import numpy as np

from .operations import DownOps, NormOps, UpOps

Genotype = namedtuple('Genotype', ['down', 'down_concat', 'up', 'up_concat', 'gamma'])


def _strongest(table, rows, names, keep):
    """For each row: value and name of its best non-'none' column (first maximum wins, as a strict
    '>' scan does), then the ``keep`` rows with the largest value, in descending stable order."""
    cols = np.array([k for k, n in enumerate(names) if n != 'none'])
    picked = []
    for r in rows:
        k = cols[int(np.argmax(table[r][cols]))]
        picked.append((table[r][k], names[k]))
    order = np.argsort(-np.array([v for v, _ in picked], dtype=table.dtype), kind='stable')[:keep]
    return [(picked[t][0], picked[t][1], t) for t in order]


class GenoParser:
    def __init__(self, meta_node_num=4):
        self._meta_node_num = meta_node_num

    def parse(self, weights1, weights2, cell_type):
        """weights1: normal-op table, weights2: up/down-op table (rows = edges, already scaled by
        beta); returns two (op, input) pairs per intermediate node."""
        weights1, weights2 = np.asarray(weights1), np.asarray(weights2)
        change_names = UpOps if cell_type == 'up' else DownOps
        if len(change_names) != len(NormOps):
            raise NotImplementedError('op lists of different length need the reference rescale '
                                      '(utils/genotype.py:77-82)')
        gene, first = [], 0
        for node in range(self._meta_node_num):
            fan_in = 2 + node
            if cell_type == 'down':        # inputs 0,1 are reduced; the rest are normal edges
                change = [(first, 0), (first + 1, 1)]
                normal = [(first + e, e) for e in range(2, fan_in)]
            else:                          # only input 1 is up-sampled
                change = [(first + 1, 1)]
                normal = [(first, 0)] + [(first + e, e) for e in range(2, fan_in)]
            found = []
            for rows, table, names in ((normal, weights1, NormOps), (change, weights2, change_names)):
                if rows:
                    found += [(w, op, rows[t][1]) for w, op, t in
                              _strongest(table, [r for r, _ in rows], names, 2)]
            gene += [(op, src) for _, op, src in sorted(found)[-2:]]
            first += fan_in
        return gene
