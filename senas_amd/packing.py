"""Packed-weight cache: the MFMA kernels read convolution weights from a fragment image
(`conv_mfma.hip`); weights change once per optimizer step, so the images of ALL dense convolutions of a
model are refreshed by ONE launch at the start of a forward pass (`senas_pack_batched`) instead of one
repack launch per convolution call (~200 per derived step, ~1 500 per supernet pass).

Without a packer every `senas_conv2d_*` call repacks its own weights -- same results, more launches.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib
from . import functional as F


class _Item(C.Structure):
    """senas_pack_item (include/senas_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('d0', C.c_int32), ('d1', C.c_int32), ('taps', C.c_int32),
                ('swap', C.c_int32), ('elems', C.c_int64)]


class _CopyItem(C.Structure):
    """senas_copy_item (include/senas_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('rows', C.c_int64), ('row_len', C.c_int64), ('src_stride', C.c_int64),
                ('dst_stride', C.c_int64), ('accumulate', C.c_int64)]


def copy_table(items, device):
    """Device table for senas_copy_rows_batched: (table tensor, item count, largest item in floats)."""
    raw = bytes((_CopyItem * len(items))(*items))
    return (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device), len(items), max(it.rows * it.row_len for it in items))


class _Image(object):
    """One cached fragment image: weak reference to the tensor it was made from, the image, and that tensor's version
    counter when the image was last refreshed on the host's watch."""
    __slots__ = ('ref', 'img', 'version', 'packer')

    def __init__(self, w, img, packer):
        self.ref, self.img, self.version, self.packer = weakref.ref(w), img, -1, packer


class WeightPacker(object):
    def __init__(self, model):
        L = _lib.lib()
        self.gen = -1                # functional.WEIGHT_GEN at the last refresh: the images count only in that generation
        self.entries = []            # (weight parameter, direction, image tensor)
        items, lp_items = [], []
        seen = set()
        # stacked weight buffers of the search cells (cell.py): filled by refresh(), then packed like any other weight
        self.stacks = [sw for m in model.modules() if hasattr(m, 'stacked_weights') for sw in m.stacked_weights()]
        copies = []
        for sw in self.stacks:
            buf = sw.buffer()
            for p, dst in zip(sw.params, sw.slices()):
                if not p.is_contiguous():
                    raise _lib.SenasHipError('stacked weights must be contiguous parameters')
                if sw.dim == 0:                                   # Conv2d [co][ci][kh][kw]: one dense block
                    copies.append(_CopyItem(p.data_ptr(), dst.data_ptr(), 1, p.numel(), p.numel(), p.numel(), 0))
                else:                                             # ConvTranspose2d [ci][co][kh][kw]: one row per input channel
                    copies.append(_CopyItem(p.data_ptr(), dst.data_ptr(), p.shape[0], p.numel() // p.shape[0], p.numel() // p.shape[0], buf.stride(0), 0))
        self.n_copies = len(copies)
        self.max_copy = max((it.rows * it.row_len for it in copies), default=0)
        if copies:
            raw = bytes((_CopyItem * len(copies))(*copies))
            self.copy_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.stacks[0].buffer().device)
        weights = [(m.weight, isinstance(m, nn.ConvTranspose2d)) for m in model.modules()
                   if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and m.groups == 1]
        weights += [(sw.buffer(), sw.dim == 1) for sw in self.stacks]
        for w, tr in weights:
            if id(w) in seen:
                continue
            seen.add(id(w))
            ci, co = (w.shape[0], w.shape[1]) if tr else (w.shape[1], w.shape[0])
            g = F.ConvGeom(1, 8, 8, ci, 8, 8, co, w.shape[2], w.shape[3], 1, 0, 1, int(tr), 1)   # only channel/tap fields matter
            for direction in (0, 1):
                d0, d1, swap, elems = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
                _lib.check(L.senas_conv2d_pack_layout(C.byref(g), direction, C.byref(d0), C.byref(d1), C.byref(swap),
                                                      C.byref(elems)), 'senas_conv2d_pack_layout')
                if elems.value == 0:
                    continue
                img = torch.empty(elems.value, device=w.device, dtype=torch.float32)
                self.entries.append((w, direction, img))
                items.append(_Item(w.data_ptr(), img.data_ptr(), d0.value, d1.value, w.shape[2] * w.shape[3], swap.value,
                                   elems.value))
                # the bf16 image of the same weight for the math mode this packer is built in (csrc/conv_bf.hip)
                terms = F.MATH_TERMS
                if terms:
                    _lib.check(L.senas_conv2d_pack_layout_lp(C.byref(g), direction, terms, C.byref(d0), C.byref(d1), C.byref(swap),
                                                             C.byref(elems)), 'senas_conv2d_pack_layout_lp')
                    if elems.value:
                        img = torch.empty(elems.value, device=w.device, dtype=torch.float32)
                        self.entries.append((w, (terms, direction), img))
                        lp_items.append(_Item(w.data_ptr(), img.data_ptr(), d0.value, d1.value, w.shape[2] * w.shape[3], swap.value,
                                              elems.value))
        self.n_lp = len(lp_items)
        self.max_lp = max((it.elems for it in lp_items), default=0)
        if self.n_lp:
            raw = bytes((_Item * self.n_lp)(*lp_items))
            self.table_lp = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.entries[0][0].device)
        self.n = len(items)
        self.max_elems = max((it.elems for it in items), default=0)
        if self.n:
            raw = bytes((_Item * self.n)(*items))
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.entries[0][0].device)
        self.images = {(w.data_ptr(), d): _Image(w, img, self) for w, d, img in self.entries}

    def refresh(self):
        """Refill the stacked buffers (one launch) and repack every weight (one launch).  Call after the
        optimizer changed the weights, before the next forward; it is part of the captured graph when the step is graphed."""
        if self.n_copies:
            _lib.check(_lib.lib().senas_copy_rows_batched(self.copy_table.data_ptr(), self.n_copies, self.max_copy, F._stream()),
                       'senas_copy_rows_batched')
        if self.n:
            _lib.check(_lib.lib().senas_pack_batched(self.table.data_ptr(), self.n, self.max_elems, F._stream()),
                       'senas_pack_batched')
        if self.n_lp:
            _lib.check(_lib.lib().senas_pack_batched_lp(self.table_lp.data_ptr(), self.n_lp, self.max_lp, F._stream()),
                       'senas_pack_batched_lp')
        self.mark_refreshed()

    def mark_refreshed(self):
        """Host-side bookkeeping of a refresh (also called after a graph replay that started with the refresh launches)."""
        for ent in self.images.values():
            w = ent.ref()
            ent.version = w._version if w is not None else -1
        for sw in self.stacks:
            sw.mark_filled()
        self.gen = F.WEIGHT_GEN

    def stale(self):
        """Have the weights moved since the last refresh (an optimizer step of either kind)?"""
        if self.gen != F.WEIGHT_GEN:
            return True
        for ent in self.images.values():
            w = ent.ref()
            if w is None or w._version != ent.version:
                return True
        return any(sw.filled != tuple(p._version for p in sw.params) for sw in self.stacks)

    def install(self):
        F.PACKED = self.images
        for sw in self.stacks:
            sw.managed, sw.packer = True, self  # refresh() keeps the buffer current: no cat launch per use

    def uninstall(self):
        if F.PACKED is self.images:
            F.PACKED = {}
        for sw in self.stacks:
            if sw.packer is self:
                sw.managed, sw.packer = False, None
