"""clip_grad_norm_ + torch.optim.SGD.step() as two launches over all parameter tensors
(``senas_sgd_clip_step``), for the graph-replayed steps where gradient addresses are static.

The torch optimizer object stays the owner of the hyper-parameters (``param_groups[0]`` is read on every
step, so LR schedulers keep working) and of the state (``state[p]['momentum_buffer']``), so
``optimizer.state_dict()`` / ``load_state_dict()`` interchange with a run that steps through torch.
Reference call sites: experiments/train_model.py:284-289, experiments/search_arc.py:280-285.
"""
import ctypes as C

import torch

from . import _lib
from . import functional as F


class _Item(C.Structure):
    """senas_sgd_item (include/senas_hip.h)."""
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('buf', C.c_void_p), ('numel', C.c_int64)]


def supported(optimizer):
    if type(optimizer) is not torch.optim.SGD or len(optimizer.param_groups) != 1:
        return False
    g = optimizer.param_groups[0]
    return not g.get('maximize', False) and all(p.dtype == torch.float32 and p.is_contiguous() for p in g['params'])


class FusedClipSGD(object):
    def __init__(self, optimizer, grad_clip):
        if not supported(optimizer):
            raise ValueError('FusedClipSGD: needs a single-group torch.optim.SGD over contiguous fp32 parameters')
        self.opt, self.grad_clip = optimizer, float(grad_clip or 0.0)
        self.table = None

    def _build(self):
        group = self.opt.param_groups[0]
        items, self.first = [], False
        for p in group['params']:
            if p.grad is None:
                continue
            if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise _lib.SenasHipError('FusedClipSGD: non-contiguous or non-fp32 gradient')
            buf = None
            if group['momentum'] != 0:
                st = self.opt.state[p]
                if st.get('momentum_buffer') is None:
                    st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    self.first = True
                buf = st['momentum_buffer']
            items.append(_Item(p.data_ptr(), p.grad.data_ptr(), buf.data_ptr() if buf is not None else None, p.numel()))
        if not items:
            raise _lib.SenasHipError('FusedClipSGD: no parameter has a gradient')
        self.n = len(items)
        self.max_numel = max(it.numel for it in items)
        dev = group['params'][0].device
        raw = bytes((_Item * self.n)(*items))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.partial = torch.zeros(64, device=dev, dtype=torch.float64)
        self.total_norm = torch.zeros(1, device=dev, dtype=torch.float32)

    def step(self):
        """Gradient addresses must be the ones seen at the first call (HIP-graph replays guarantee that)."""
        if self.table is None:
            self._build()
        g = self.opt.param_groups[0]
        _lib.check(_lib.lib().senas_sgd_clip_step(self.table.data_ptr(), self.n, self.max_numel, self.partial.data_ptr(),
                                                  self.grad_clip, g['lr'], g['momentum'], g['dampening'], g['weight_decay'],
                                                  int(g['nesterov']), int(self.first), self.total_norm.data_ptr(), F._stream()),
                   'senas_sgd_clip_step')
        self.first = False
        return self.total_norm
