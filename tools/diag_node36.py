import sys
sys.path.insert(0, '/root/repo')
import torch
import torch.nn as nn
from senas_amd import functional as F
dev = torch.device('cuda:0')
torch.manual_seed(0)
for T in (24, 36, 40):
    n, c, h, w = 2, 8, 16, 16
    bns = [nn.BatchNorm2d(c).to(dev).train() for _ in range(T)]
    for b in bns:
        with torch.no_grad():
            b.weight.copy_(1 + 0.2 * torch.randn(c)); b.bias.copy_(0.1 * torch.randn(c))
    zs = [torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(T)]
    mix = torch.rand(T, device=dev, requires_grad=True)
    terms = [F.Term(z, b) for z, b in zip(zs, bns)]
    y = F.bn_combine(terms, mix=mix, relu=True)
    g = torch.randn_like(y)
    y.backward(g)
    got = [y.detach().clone(), mix.grad.clone()] + [z.grad.clone() for z in zs] + [b.weight.grad.clone() for b in bns]
    for p in [mix] + zs + [b.weight for b in bns] + [b.bias for b in bns]:
        p.grad = None
    # torch reference in fp64
    ref_bns = [nn.BatchNorm2d(c).to(dev).double().train() for _ in range(T)]
    for rb, b in zip(ref_bns, bns):
        rb.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in b.state_dict().items()})
    z64 = [z.detach().double().requires_grad_(True) for z in zs]
    m64 = mix.detach().double().requires_grad_(True)
    acc = sum(m64[t] * ref_bns[t](z64[t]) for t in range(T))
    yr = torch.relu(acc)
    yr.backward(g.double())
    ref = [yr.detach(), m64.grad] + [z.grad for z in z64] + [rb.weight.grad for rb in ref_bns]
    worst = max(float((a.double() - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(got, ref))
    print('T=%d worst rel err %.2e (y %.1e, dmix %.1e)' % (T, worst, float((got[0].double() - ref[0]).abs().max() / ref[0].abs().max()),
                                                            float((got[1].double() - ref[1]).abs().max() / ref[1].abs().max())))
