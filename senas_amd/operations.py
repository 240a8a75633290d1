"""Candidate operations and cell building blocks of the SENAS search space, MI355X edition.

Same plugin surface as the reference's ``utils/operations.py``: the ``OPS`` registry
(``OPS[name](c_in, c_ot, op_type, dp) -> nn.Module``, :8-21), the ``DownOps / UpOps / NormOps``
lists whose order is the alpha column order (:23-48), ``OpType`` (:51-54), ``build_ops`` (:57-78)
and the block classes (``ConvBn``, ``ConvBnSe``, ``DepSepConv``, ``AdapterBlock``, ``SEBlock``,
``ZeroOp``, ``ShrinkBlock``, ``RectifyBlock``, ``ReLUConv``, ``BasicBlock``, ``build_rectify``).
Every module owns the same parameters/buffers under the same names, so ``state_dict`` keys --
and therefore checkpoints -- are interchangeable with the reference.

What differs is the execution: modules here never run torch.nn kernels.  A candidate op exposes
``raw(x) -> Term`` -- its convolution / pooling output *before* batch-norm, plus the BatchNorm2d
(and SE block) that still has to be applied -- and the consumer (a cell node, ``MixedOp``, or the
module's own ``forward``) folds normalisation, SE gating, alpha/beta mixing, the node sum and the
ReLU into one pass (``functional.bn_combine``).  The ReLU in front of a convolution is applied on
load inside the convolution kernel.
"""
from enum import Enum

import torch.nn as nn

from . import functional as F
from .functional import Term

DownOps = ['avg_pool', 'se_conv_3', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3', 'dep_sep_conv_5']
UpOps = ['up_sample', 'se_conv_3', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3', 'dep_sep_conv_5']
NormOps = ['identity', 'none', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3', 'dep_sep_conv_5']


class OpType(Enum):
    UP = {'id': 1, 'ops': UpOps}
    DOWN = {'id': 2, 'ops': DownOps}
    NORM = {'id': 3, 'ops': NormOps}


def get_same_padding(kernel_size):
    """utils/utils.py:21-29."""
    if isinstance(kernel_size, tuple):
        return tuple(get_same_padding(k) for k in kernel_size)
    if not isinstance(kernel_size, int) or kernel_size % 2 == 0:
        raise AssertionError('kernel size should be an odd int, got %r' % (kernel_size,))
    return kernel_size // 2


# ----------------------------------------------------------------------------------------------- leaves
def run_conv(conv, x, in_relu=False, want_stats=True):
    """Launch the HIP convolution described by an nn.Conv2d / nn.ConvTranspose2d parameter holder.
    ``want_stats``: also produce the per-image channel sums the following BatchNorm2d needs in training mode (pass
    ``bn.training``: in eval mode the running statistics are used and the sums would be wasted work)."""
    tr = isinstance(conv, nn.ConvTranspose2d)
    if conv.bias is not None:
        raise F.SenasHipError('convolutions on the SENAS path are bias-free')
    return F.conv2d(x, conv.weight, stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], transposed=tr,
                    out_pad=conv.output_padding[0] if tr else 0, groups=conv.groups, in_relu=in_relu,
                    want_stats=want_stats)


class ReLU(nn.Module):
    """nn.ReLU stand-in that runs the HIP kernel (``inplace`` is accepted and ignored: the op is
    out of place, values are identical)."""

    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return F.relu(x)


def build_activation(inplace=True):
    return ReLU(inplace=inplace)


def build_norm(c_ot, affine):
    if not affine:
        raise NotImplementedError('the SENAS path only builds affine batch-norm')
    return nn.BatchNorm2d(c_ot, affine=True)


def build_weight(c_in, c_ot, kernel_size, stride, dilation, use_transpose, output_padding, dropout_rate, groups=1):
    """utils/operations.py:118-130: [Dropout2d(p) when p > 0,] bias-free Conv2d / ConvTranspose2d.  The Dropout2d keeps its
    place in the Sequential (the reference's numeric child names, hence its state_dict keys, shift by one with it)."""
    pad = get_same_padding(kernel_size) * dilation
    ops = [nn.Dropout2d(dropout_rate, inplace=False)] if dropout_rate > 0 else []
    if use_transpose:
        return ops + [nn.ConvTranspose2d(c_in, c_ot, kernel_size, stride=stride, padding=pad, output_padding=output_padding,
                                         groups=groups, bias=False, dilation=dilation)]
    return ops + [nn.Conv2d(c_in, c_ot, kernel_size, stride=stride, padding=pad, dilation=dilation, groups=groups, bias=False)]


def _dropped(drop, x, in_relu=False):
    """x as the convolution behind a Dropout2d sees it, and whether the ReLU is still to be applied on load.  The mask is
    torch's (one Bernoulli draw per (image, channel), scaled by 1 / (1 - p), graph-safe Philox state); an identity in
    eval mode.  Only the derived network can ask for it (models/senas_model.py:40-46; the search cell builds dp = 0)."""
    if drop is None or not drop.training or drop.p == 0:
        return x, in_relu
    if in_relu:
        x = F.relu(x)
    return drop(x), False


class ZeroOp(nn.Module):
    """x.mul(0.) -- feeds an all-zero tensor to the adapter, so the op contributes its batch-norm
    bias only.  Never materialised here (see AdapterBlock.raw)."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        raise F.SenasHipError('ZeroOp is folded into AdapterBlock; it has no standalone kernel')


class SEBlock(nn.Module):
    """Squeeze-and-excitation parameters (c -> mid -> c, no bias).  The squeeze comes from the
    per-image channel sums the convolution already produced; the gate is applied in bn_combine."""

    def __init__(self, c, r=16):
        super().__init__()
        self.mid = c // r if c > r else 1
        self.squeeze = nn.AdaptiveAvgPool2d(1)
        self.excitation = nn.Sequential(nn.Linear(c, self.mid, bias=False), nn.ReLU(inplace=True),
                                        nn.Linear(self.mid, c, bias=False), nn.Sigmoid())

    def forward(self, x):
        raise F.SenasHipError('SEBlock is fused into its ConvBnSe parent')


class _Op(nn.Sequential):
    """A candidate op stored as a Sequential (to keep the reference's numeric child names)."""

    def raw(self, x):
        raise NotImplementedError

    def forward(self, x):
        return F.bn_combine([self.raw(x)])


class ConvBn(_Op):
    def __init__(self, c_in, c_ot, kernel_size=3, stride=1, dilation=1, transpose=False, output_padding=0, affine=True,
                 dropout=0):
        super().__init__(*build_weight(c_in, c_ot, kernel_size, stride, dilation, transpose, output_padding, dropout),
                         build_norm(c_ot, affine))
        self._d = 1 if dropout > 0 else 0                # children: [Dropout2d,] conv, norm

    drop = property(lambda self: self[0] if self._d else None)
    conv = property(lambda self: self[self._d])
    norm = property(lambda self: self[self._d + 1])

    def raw(self, x, in_relu=False):
        x, in_relu = _dropped(self.drop, x, in_relu)
        z, st = run_conv(self.conv, x, in_relu, want_stats=self.norm.training)
        return Term(z, self.norm, stats=st)


class ConvBnSe(_Op):
    def __init__(self, c_in, c_ot, kernel_size=3, stride=1, dilation=1, transpose=False, output_padding=0, affine=True,
                 dropout=0):
        super().__init__(*build_weight(c_in, c_ot, kernel_size, stride, dilation, transpose, output_padding, dropout),
                         build_norm(c_ot, affine), SEBlock(c_ot))
        self._d = 1 if dropout > 0 else 0                # children: [Dropout2d,] conv, norm, se

    drop = property(lambda self: self[0] if self._d else None)
    conv = property(lambda self: self[self._d])
    norm = property(lambda self: self[self._d + 1])
    se = property(lambda self: self[self._d + 2])

    def raw(self, x):
        x, _ = _dropped(self.drop, x)
        z, st = run_conv(self.conv, x)
        return Term(z, self.norm, se=self.se, stats=st)


class DepSepConv(_Op):
    def __init__(self, c_in, c_ot, kernel_size=3, stride=1, dilation=1, transpose=False, output_padding=0, affine=True,
                 dropout=0):
        depth = build_weight(c_in, c_in, kernel_size, stride, dilation, transpose, output_padding, dropout, groups=c_in)
        point = build_weight(c_in, c_ot, 1, 1, 1, False, 0, dropout)
        super().__init__(*depth, build_norm(c_in, affine), build_activation(), *point, build_norm(c_ot, affine))
        self._d = 1 if dropout > 0 else 0                # children: [Dropout2d,] dw conv, norm, relu, [Dropout2d,] 1x1 conv, norm

    drop = property(lambda self: self[0] if self._d else None)
    dw = property(lambda self: self[self._d])
    norm1 = property(lambda self: self[self._d + 1])
    drop2 = property(lambda self: self[self._d + 3] if self._d else None)
    pw = property(lambda self: self[2 * self._d + 3])
    norm2 = property(lambda self: self[2 * self._d + 4])

    def raw(self, x):
        x, _ = _dropped(self.drop, x)
        z1, st1 = run_conv(self.dw, x, want_stats=self.norm1.training)
        mid = F.bn_combine([Term(z1, self.norm1, stats=st1)], relu=True)
        mid, _ = _dropped(self.drop2, mid)
        z2, st = run_conv(self.pw, mid, want_stats=self.norm2.training)
        return Term(z2, self.norm2, stats=st)


class AdapterBlock(nn.Module):
    """Parameter-free resampling op (or identity / zero) + optional 1x1 channel adapter + norm."""

    def __init__(self, c_in, c_ot, module):
        super().__init__()
        self.c_in, self.c_ot, self.module = c_in, c_ot, module
        if c_in != c_ot:
            self.conv = nn.Conv2d(c_in, c_ot, kernel_size=1, bias=False)
        self.norm = build_norm(c_ot, True)

    def _resample(self, x, want_stats=False):
        """y, or (y, stats-or-None) with want_stats (the resampling kernels produce the next norm's statistics)."""
        m = self.module
        if isinstance(m, nn.Identity):
            # (a search-cell node leaves its own channel sums on its output: node.bn_combine(out_stats=True))
            return (x, getattr(x, '_senas_stats', None)) if want_stats else x
        if isinstance(m, nn.AvgPool2d):
            if (m.kernel_size, m.padding, m.count_include_pad) != (3, 1, False):
                raise NotImplementedError('only AvgPool2d(3, s, 1, count_include_pad=False) is on the path')
            return F.avg_pool3(x, m.stride, want_stats=want_stats)
        if isinstance(m, nn.MaxPool2d):
            if (m.kernel_size, m.padding) != (3, 1):
                raise NotImplementedError('only MaxPool2d(3, s, 1) is on the path')
            return F.max_pool3(x, m.stride, want_stats=want_stats)
        if isinstance(m, nn.Upsample):
            if m.scale_factor != 2 or m.mode != 'bilinear' or m.align_corners:
                raise NotImplementedError('only Upsample(x2, bilinear, align_corners=False) is on the path')
            return F.bilinear2x(x, want_stats=want_stats)
        raise NotImplementedError('AdapterBlock around %s' % type(m).__name__)

    def raw(self, x):
        has_conv = self.c_in != self.c_ot
        if isinstance(self.module, ZeroOp):
            if self.module.stride != 1:
                raise NotImplementedError('ZeroOp(stride != 1) is not used by OPS')
            return Term(None, self.norm, passengers=[self.conv.weight] if has_conv else [])
        if has_conv:
            z, st = run_conv(self.conv, self._resample(x), want_stats=self.norm.training)
            return Term(z, self.norm, stats=st)
        if not self.norm.training:
            return Term(self._resample(x), self.norm)
        y, st = self._resample(x, want_stats=True)
        return Term(y, self.norm, stats=st)

    def forward(self, x):
        t = self.raw(x)
        if t.z is None:     # 'none': the output is the batch-norm bias broadcast over the input's grid
            return F.bn_combine([t], residual=F.zero_feature(x, self.c_ot))
        return F.bn_combine([t])


def build_ops(op_name, op_type, c_in=None, c_ot=None, dp=0):
    stride = 1 if op_type == OpType.NORM else 2
    up = op_type == OpType.UP
    geo = dict(stride=stride, transpose=up, output_padding=1 if up else 0, dropout=dp)
    if op_name == 'avg_pool':
        return AdapterBlock(c_in, c_ot, nn.AvgPool2d(3, stride=stride, padding=1, count_include_pad=False))
    if op_name == 'max_pool':
        return AdapterBlock(c_in, c_ot, nn.MaxPool2d(3, stride=stride, padding=1))
    if op_name == 'conv_3':
        return ConvBn(c_in, c_ot, kernel_size=3, **geo)
    if op_name == 'se_conv_3':
        return ConvBnSe(c_in, c_ot, kernel_size=3, **geo)
    if op_name in ('dil_3_conv_5', 'dil_2_conv_5'):
        return ConvBn(c_in, c_ot, kernel_size=5, dilation=int(op_name[4]), **geo)
    if op_name in ('dep_sep_conv_3', 'dep_sep_conv_5'):
        return DepSepConv(c_in, c_ot, kernel_size=int(op_name[-1]), **geo)
    raise NotImplementedError(op_name)


def _registered(name):
    return lambda c_in, c_ot, op_type, dp: build_ops(name, op_type, c_in, c_ot, dp=dp)


OPS = {
    'none': lambda c_in, c_ot, op_type, dp: AdapterBlock(c_in, c_ot, ZeroOp(stride=1)),
    'identity': lambda c_in, c_ot, op_type, dp: AdapterBlock(c_in, c_ot, nn.Identity()),
    'up_sample': lambda c_in, c_ot, op_type, dp: AdapterBlock(
        c_in, c_ot, nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)),
}
for _name in ('avg_pool', 'max_pool', 'conv_3', 'se_conv_3', 'dil_3_conv_5', 'dil_2_conv_5', 'dep_sep_conv_3',
              'dep_sep_conv_5'):
    OPS[_name] = _registered(_name)


# ----------------------------------------------------------------------------------------------- cell blocks
class ReLUConv(nn.Sequential):
    """ReLU then a bias-free conv, no norm (the segmentation head)."""

    def __init__(self, c_in, c_ot, kernel_size=3, stride=1, dilation=1, transpose=False, output_padding=0, dropout=0):
        super().__init__(build_activation(False),
                         *build_weight(c_in, c_ot, kernel_size, stride, dilation, transpose, output_padding, dropout))

    def forward(self, x):
        return run_conv(self[1], x, in_relu=True, want_stats=False)[0]


class _Rectify(nn.Sequential):
    """build_rectify's Sequential(act, resample-or-1x1-conv, norm) with the ReLU folded into the
    resampling / convolution kernel's load."""

    def forward(self, x):
        op = self[1]
        if isinstance(op, (nn.Conv2d, nn.ConvTranspose2d)):
            z, st = run_conv(op, x, in_relu=True, want_stats=self[2].training)
            return F.bn_combine([Term(z, self[2], stats=st)])
        if isinstance(op, nn.AvgPool2d):
            z, st = F._AvgPool3.apply(x, op.stride, True, self[2].training)
            return F.bn_combine([Term(z, self[2], stats=st)])
        z, st = F._Bilinear2x.apply(F.relu(x), self[2].training)
        return F.bn_combine([Term(z, self[2], stats=st)])


def build_rectify(c_in, c_ot, cell_type):
    act = build_activation(False)
    if cell_type == 'up':
        mid = (nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) if c_in == c_ot else
               nn.ConvTranspose2d(c_in, c_ot, kernel_size=1, stride=2, output_padding=1, bias=False))
    else:
        mid = (nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False) if c_in == c_ot else
               nn.Conv2d(c_in, c_ot, kernel_size=1, stride=2, bias=False))
    return _Rectify(act, mid, build_norm(c_ot, True))


class ShrinkBlock(nn.Module):
    """ReLU -> 3x3 conv (c_in -> c_ot) -> norm: squeezes the concatenated skip inputs of an up cell."""

    def __init__(self, c_in, c_ot):
        super().__init__()
        self.act = build_activation(False)
        self.conv = nn.Conv2d(c_in, c_ot, kernel_size=3, padding=1, bias=False)
        self.norm = build_norm(c_ot, True)

    def forward(self, x):
        z, st = run_conv(self.conv, x, in_relu=True, want_stats=self.norm.training)
        return F.bn_combine([Term(z, self.norm, stats=st)])


class RectifyBlock(nn.Module):
    """3x3 conv + norm on the concatenated node outputs of a cell."""

    def __init__(self, c_in, c_ot, cell_type='down'):
        super().__init__()
        self.cell_type = cell_type
        self.conv = nn.Conv2d(c_in, c_ot, kernel_size=3, padding=1, bias=False)
        self.norm = build_norm(c_ot, True)

    def forward(self, x):
        z, st = run_conv(self.conv, x, want_stats=self.norm.training)
        return F.bn_combine([Term(z, self.norm, stats=st)])


class BasicBlock(nn.Module):
    """ResNet basic block as the stem uses it: conv-bn-relu-conv-bn + residual, no final ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, previous_dilation=1, norm_layer=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                               bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = build_activation(True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=previous_dilation,
                               dilation=previous_dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        z1, s1 = run_conv(self.conv1, x, want_stats=self.bn1.training)
        a = F.bn_combine([Term(z1, self.bn1, stats=s1)], relu=True)
        z2, s2 = run_conv(self.conv2, a, want_stats=self.bn2.training)
        res = x if self.downsample is None else self.downsample(x)
        return F.bn_combine([Term(z2, self.bn2, stats=s2)], residual=res)


class Stem1(nn.Sequential):
    """Sequential(ReLU, MaxPool2d(3, 2, 1), BasicBlock): the ReLU rides on the pooling kernel's load."""

    def __init__(self, c_in, c_ot):
        super().__init__(build_activation(False), nn.MaxPool2d(3, stride=2, padding=1),
                         BasicBlock(c_in, c_ot, stride=1, dilation=1, previous_dilation=1, norm_layer=nn.BatchNorm2d))

    def forward(self, x):
        return self[2](F.max_pool3(x, 2, in_relu=True))
