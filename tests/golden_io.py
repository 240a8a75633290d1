"""Loader for the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the reference itself)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return _cache[name]


def index(name):
    return json.loads(str(load(name)['index']))


def sub(npz, prefix):
    """All entries under ``prefix`` with the prefix stripped."""
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


def unpack(npz, prefix):
    """Inverse of make_golden._pack: {name: ndarray}."""
    flat = npz[prefix + 'flat']
    out = {}
    for name, shape, off in json.loads(str(npz[prefix + 'index'])):
        n = int(np.prod(shape)) if shape else 1
        out[name] = flat[off:off + n].reshape(shape).copy()
    return out


def digest(npz, prefix):
    return dict(zip(json.loads(str(npz[prefix + 'names'])), npz[prefix + 'digest']))


def torch_sd(arrays, device='cpu', requires_grad=True):
    """ndarray dict -> tensor dict; float non-buffer entries become leaves that require grad."""
    sd = {}
    for k, v in arrays.items():
        t = torch.from_numpy(np.array(v)).to(device)
        buf = k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked')
        if requires_grad and t.is_floating_point() and not buf:
            t.requires_grad_(True)
        sd[k] = t
    return sd


def add_missing_counters(sd):
    """flat-packed state dicts leave out the int64 ``num_batches_tracked`` buffers (all zero)."""
    for k in list(sd):
        if k.endswith('running_mean'):
            nk = k[:-len('running_mean')] + 'num_batches_tracked'
            if nk not in sd:
                sd[nk] = torch.zeros((), dtype=torch.long, device=sd[k].device)
    return sd


def share_stem(sd, prefix=''):
    """Make ``<prefix>blocks.0.0.*`` the very same tensors as ``<prefix>stem1.*`` (one module, two names)."""
    a, b = prefix + 'blocks.0.0.', prefix + 'stem1.'
    for k in list(sd):
        if k.startswith(a):
            sd[k] = sd[b + k[len(a):]]
    return sd


def geno_from_json(txt, Genotype):
    d = json.loads(str(txt))
    return Genotype(down=[(a, int(b)) for a, b in d['down']], down_concat=range(d['down_concat'][0], d['down_concat'][-1] + 1),
                    up=[(a, int(b)) for a, b in d['up']], up_concat=range(d['up_concat'][0], d['up_concat'][-1] + 1),
                    gamma=list(d['gamma']))


def check_grads(expected, got, rtol, atol, what=''):
    """``expected``: entries of a '<tag>/grad/' section (full tensors, or '#head' / '#sum' digests);
    ``got``: name -> ndarray."""
    seen = set()
    for k, e in expected.items():
        if k.endswith('#head'):
            name = k[:-5]
            g = got[name].reshape(-1)[:e.size]
            np.testing.assert_allclose(g, e, rtol=rtol, atol=atol + rtol * float(np.abs(e).max()),
                                       err_msg='%s %s head' % (what, name))
        elif k.endswith('#sum'):
            name = k[:-4]
            g = got[name].astype(np.float64)
            np.testing.assert_allclose(np.sqrt((g ** 2).sum()), e[1], rtol=rtol, err_msg='%s %s l2' % (what, name))
            np.testing.assert_allclose(g.sum(), e[0], rtol=rtol, atol=atol * max(1.0, e[1]) * 8, err_msg='%s %s sum' % (what, name))
        else:
            name = k
            np.testing.assert_allclose(got[name], e, rtol=rtol, atol=atol + rtol * float(np.abs(e).max()),
                                       err_msg='%s %s' % (what, name))
        seen.add(name)
    return seen


def alias_shared_stem(got, prefix=''):
    """The reference registers the stem1 module twice (``stem1`` and ``blocks.0.0``); its
    ``named_parameters()`` reports the shared tensors under ``blocks.0.0``.  Give gradients found
    under ``<prefix>stem1.`` that second name too."""
    for k in list(got):
        if k.startswith(prefix + 'stem1.'):
            got.setdefault(prefix + 'blocks.0.0.' + k[len(prefix) + 6:], got[k])
    return got


def check_digest(expected, got, rtol, atol_scale=1e-5, what=''):
    """expected: name -> (sum, l2) float64; got: name -> ndarray.  Tensors whose norm is noise
    compared with the largest one (gradients that are analytically zero, e.g. a bias in front of
    another batch-norm) are compared absolutely against that scale."""
    top = max(float(v[1]) for v in expected.values())
    for name, (s, l2) in expected.items():
        g = np.asarray(got[name]).astype(np.float64)
        np.testing.assert_allclose(np.sqrt((g ** 2).sum()), l2, rtol=rtol, atol=atol_scale * top,
                                   err_msg='%s %s l2' % (what, name))
        np.testing.assert_allclose(g.sum(), s, rtol=rtol, atol=atol_scale * (top + l2 * np.sqrt(g.size)) + 1e-12,
                                   err_msg='%s %s sum' % (what, name))
