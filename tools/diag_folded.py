#!/usr/bin/env python3
"""Folded inference vs the module's own eval forward, one candidate op at a time (pairs them with an identity term)."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd import functional as F  # noqa: E402
from senas_amd.infer import FoldedForward  # noqa: E402
from senas_amd.operations import OPS, OpType  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for kind in (OpType.NORM, OpType.DOWN, OpType.UP):
        for name in kind.value['ops']:
            if name == 'none':
                continue
            op = OPS[name](c, c, kind, 0).to(dev).eval()
            other = OPS['dil_3_conv_5'](c, c, kind, 0).to(dev).eval()
            with torch.no_grad():
                for m in list(op.modules()) + list(other.modules()):
                    if isinstance(m, nn.BatchNorm2d):
                        m.running_mean.normal_(0, 0.3)
                        m.running_var.uniform_(0.5, 1.5)
                        m.weight.normal_(1, 0.3)
                        m.bias.normal_(0, 0.3)
            holder = nn.ModuleList([op, other])
            ff = FoldedForward(holder, 2)
            x = torch.randn(2, c, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                ref1 = op(x)
                got1 = ff._finish([ff.term(op, x)], relu=False)
                ref2 = F.bn_combine([op.raw(x), other.raw(x)], relu=True)
                got2 = ff._finish([ff.term(op, x), ff.term(other, x)], relu=True)
            e1 = float((got1 - ref1).abs().max() / ref1.abs().max())
            e2 = float((got2 - ref2).abs().max() / ref2.abs().max())
            print('%-5s %-16s single %.2e   node %.2e   fused launches %d fallback %d' % (kind.name, name, e1, e2, ff.fused_launches,
                                                                                           ff.fallback_launches), flush=True)


if __name__ == '__main__':
    main()
