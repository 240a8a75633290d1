"""clip_grad_norm_ + torch.optim.SGD.step() as three launches over all parameter tensors
(``senas_sgd_clip_step``), for the step drivers, where gradient addresses are static (views of the flat gradient
buffer, ``gradsink.GradSink``).

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
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('buf', C.c_void_p), ('numel', C.c_int64), ('first', C.c_int64)]


def supported(optimizer):
    if type(optimizer) is not torch.optim.SGD or len(optimizer.param_groups) != 1:
        return False
    g = optimizer.param_groups[0]
    return not g.get('maximize', False) and all(p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda for p in g['params'])


class FusedClipSGD(object):
    def __init__(self, optimizer, grad_clip):
        if not supported(optimizer):
            raise ValueError('FusedClipSGD: needs a single-group torch.optim.SGD over contiguous fp32 CUDA parameters')
        self.opt, self.grad_clip = optimizer, float(grad_clip or 0.0)
        self.table = None
        self.key = None
        # optimizer.load_state_dict() replaces the momentum buffers: the device table must not outlive them
        if hasattr(optimizer, 'register_load_state_dict_post_hook'):
            optimizer.register_load_state_dict_post_hook(lambda *_: self.invalidate())

    def invalidate(self):
        """Forget the device table of (parameter, gradient, momentum buffer) addresses; the next step rebuilds it."""
        self.table = None

    def _addresses(self):
        group = self.opt.param_groups[0]
        state = self.opt.state
        return tuple((p.grad.data_ptr() if p.grad is not None else 0,
                      state[p]['momentum_buffer'].data_ptr() if state.get(p, {}).get('momentum_buffer') is not None else 0)
                     for p in group['params'])

    def _build(self):
        group = self.opt.param_groups[0]
        items, any_first = [], False
        for p in group['params']:
            if p.grad is None:
                continue
            if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise _lib.SenasHipError('FusedClipSGD: non-contiguous or non-fp32 gradient')
            buf, first = None, 0
            if group['momentum'] != 0:
                st = self.opt.state[p]
                if st.get('momentum_buffer') is None:          # per tensor: a resumed / partial state keeps its other buffers
                    st['momentum_buffer'] = torch.empty_like(p, memory_format=torch.contiguous_format)
                    first, any_first = 1, True
                buf = st['momentum_buffer']
            items.append(_Item(p.data_ptr(), p.grad.data_ptr(), buf.data_ptr() if buf is not None else None, p.numel(), first))
        if not items:
            raise _lib.SenasHipError('FusedClipSGD: no parameter has a gradient')
        self.n = len(items)
        self.max_numel = max(it.numel for it in items)
        dev = group['params'][0].device
        raw = bytes((_Item * self.n)(*items))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        slots = 1 + self.n * ((self.max_numel + 1023) // 1024)
        self.partial = torch.zeros(slots, device=dev, dtype=torch.float64)
        self.total_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.key = self._addresses()
        self.steps_since_check = 0
        return any_first

    def step(self, check=False):
        """One clip + SGD step.  The table of addresses is rebuilt after ``optimizer.load_state_dict()`` and whenever a
        gradient or momentum buffer turns out to have moved (checked on the first steps and every 64th after, or with
        ``check=True``; under a step driver the gradients are views of the flat buffer and never move)."""
        if self.table is not None:
            self.steps_since_check += 1
            if check or self.steps_since_check <= 2 or self.steps_since_check % 64 == 0:
                if self._addresses() != self.key:
                    self.table = None
        had_first = self._build() if self.table is None else False
        g = self.opt.param_groups[0]
        _lib.check(_lib.lib().senas_sgd_clip_step(self.table.data_ptr(), self.n, self.max_numel, self.partial.data_ptr(),
                                                  self.grad_clip, g['lr'], g['momentum'], g['dampening'], g['weight_decay'],
                                                  int(g['nesterov']), 0, self.total_norm.data_ptr(), F._stream()),
                   'senas_sgd_clip_step')
        self.opt._opt_called = True        # (what lr schedulers look at to warn about a step order they cannot see here)
        if had_first:                      # those buffers now hold the first gradient: drop the flags
            self.table = None
        return self.total_norm
