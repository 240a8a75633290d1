"""Built-in genotypes, by name (``senas_node_2``, ``senas_node_3``, ``senas_node_4``, ``senas`` = ``senas_node_4``).

The values are data published by the reference (``models/geno_searched.py:3-10``, README.md:44); they live in
``data/genotypes.json`` and are turned into :class:`~senas_amd.genotype.Genotype` tuples at import time, so
``geno_searched.senas_node_4`` / ``getattr(geno_searched, name)`` work exactly as the reference's module attributes
(``experiments/train_model.py:117``).
"""
import json
import os

from .genotype import Genotype

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'genotypes.json')


def _load():
    with open(_PATH) as f:
        table = json.load(f)
    made = {}
    for name, g in table['genotypes'].items():
        made[name] = Genotype(down=[(op, int(idx)) for op, idx in g['down']], down_concat=range(*g['down_concat']),
                              up=[(op, int(idx)) for op, idx in g['up']], up_concat=range(*g['up_concat']),
                              gamma=list(g['gamma']))
    for alias, target in table['aliases'].items():
        made[alias] = made[target]
    return made


globals().update(_load())
__all__ = ['senas', 'senas_node_2', 'senas_node_3', 'senas_node_4']
