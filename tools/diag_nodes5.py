import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import golden_io as gio
from oracle import senas_ref as R
from senas_amd.loss import SegmentationLosses
from senas_amd.senas_search import NAS
from senas_amd.cell import Cell
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
torch.manual_seed(3)
net = NAS(input_c=1, c=32, num_classes=2, depth=3, meta_node_num=nodes, use_sharing=False, double_down_channel=False)
with torch.no_grad():
    for p in net.arch_parameters():
        p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel() + nodes)) * 0.5)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
for k, v in sd.items():
    if v.is_floating_point() and 'running' not in k:
        v.requires_grad_(True)
gio.share_stem(sd, 'net.')
gen = torch.Generator().manual_seed(20 + nodes)
x = torch.randn(2, 1, 32, 32, generator=gen); y = torch.randint(0, 2, (2, 32, 32), generator=gen)
R.dice_ce_loss(R.nas_forward(sd, x, depth=3, nodes=nodes)[-1], y).backward()
net = net.to(dev).train()
for flag in ('default', 'nostack', 'nofuse'):
    Cell.stacked = flag != 'nostack'
    Cell.fused_tail = flag != 'nofuse'
    for p in net.parameters(): p.grad = None
    out = net(x.to(dev)); SegmentationLosses('dice_ce')(out, y.to(dev)).backward()
    for k in ('alphas_dn', 'alphas_dn_nm', 'alphas_up', 'alphas_up_nm', 'betas_dn', 'betas_up'):
        e = sd[k].grad.numpy().astype(np.float64); g = dict(net.named_parameters())[k].grad.cpu().numpy()
        rows = np.abs(g - e).reshape(e.shape[0], -1).max(1) / np.abs(e).max()
        print(flag, k, 'L2 %.1e' % (np.sqrt(((g - e) ** 2).sum()) / np.sqrt((e ** 2).sum())), 'rows>1e-3:', [i for i, r in enumerate(rows) if r > 1e-3])
Cell.stacked = Cell.fused_tail = True
