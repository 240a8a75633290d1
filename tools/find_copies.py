#!/usr/bin/env python3
"""Which Python call sites launch device-to-device copies / small torch kernels in one eager train step."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from senas_amd.geno_searched import senas_node_4  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402
from senas_amd.senas_model import SenasModel  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    if len(sys.argv) > 1 and sys.argv[1] == '--search':
        from senas_amd.senas_search import NAS
        net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False, device=dev).to(dev)
    else:
        net = SenasModel(2, 1, c=32, depth=5, genotype=senas_node_4).to(dev)
    crit = SegmentationLosses('dice_ce')
    x = torch.randn(2, 1, 64, 64, device=dev)
    y = torch.randint(0, 2, (2, 64, 64), device=dev)
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        crit(net(x), y).backward()
    sites = collections.Counter()
    orig_copy = torch.Tensor.copy_
    orig_clone = torch.Tensor.clone
    orig_contig = torch.Tensor.contiguous

    def where():
        st = traceback.extract_stack()[:-2]
        return ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-4:]))

    def copy_(self, *a, **k):
        sites['copy_ ' + where()] += 1
        return orig_copy(self, *a, **k)

    def clone(self, *a, **k):
        sites['clone ' + where()] += 1
        return orig_clone(self, *a, **k)

    def contiguous(self, *a, **k):
        out = orig_contig(self, *a, **k)
        if out.data_ptr() != self.data_ptr():
            sites['contiguous(copy) ' + where()] += 1
        return out

    torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = copy_, clone, contiguous
    net.zero_grad(set_to_none=True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        crit(net(x), y).backward()
        torch.cuda.synchronize()
    torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = orig_copy, orig_clone, orig_contig
    for k, v in sites.most_common(30):
        print('%4d  %s' % (v, k))
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=70, max_name_column_width=70))


if __name__ == '__main__':
    main()
