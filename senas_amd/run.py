"""YAML-driven search / train loops around the step drivers: what
``experiments/search_arc.py:37-48,111-148,252-299`` and ``experiments/train_model.py:42-54,117-151,264-305`` do per
epoch -- build the model and the optimizers from the config's ``searching`` / ``training`` block, honour ``alpha_begin``,
step the cosine schedule once per epoch, log the genotype, save a reference-format checkpoint -- over a synthetic slice
source (seed 1 + rank; datasets are out of scope).  Entry points: ``python -m senas_amd.search``, ``python -m senas_amd.train``.

One process per GPU: under ``torch.distributed.run`` the ranks shard the batch and all-reduce gradients over RCCL
(``senas_amd.step``); alone, it is a single-GPU run.
"""
import argparse
import ast
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs', 'senas_promise12.yml')


class _Loader(yaml.SafeLoader):
    """SafeLoader that also reads the ``!!python/tuple`` the reference's configs use for Adam's betas."""


_Loader.add_constructor('tag:yaml.org,2002:python/tuple', lambda loader, node: tuple(loader.construct_sequence(node)))


def load_config(path):
    with open(path) as f:
        return yaml.load(f, Loader=_Loader)


class SyntheticSlices(object):
    """``images`` random slices and labels resident on the device, served in shuffled batches: x float
    [N, C, size, size], y int64 [N, size, size] (what Promise12.__getitem__ yields, utils/datasets/promise12.py:392-418)."""

    def __init__(self, images, in_channels, nclass, size, batch, seed, device):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(images, in_channels, size, size, generator=g).to(device)
        self.y = torch.randint(0, nclass, (images, size, size), generator=g).to(device)
        self.batch, self.gen = batch, torch.Generator().manual_seed(seed + 7919)

    def __len__(self):
        return self.x.shape[0] // self.batch

    def __iter__(self):
        order = torch.randperm(self.x.shape[0], generator=self.gen).to(self.x.device)
        for i in range(len(self)):
            idx = order[i * self.batch:(i + 1) * self.batch]
            yield self.x[idx], self.y[idx]


def _optimizer(params, spec):
    kw = {k: v for k, v in spec.items() if k != 'name'}
    name = spec['name'].lower()
    if name == 'sgd':
        return torch.optim.SGD(params, **kw)
    if name == 'adam':
        if 'betas' in kw:
            kw['betas'] = tuple(kw['betas'])
        return torch.optim.Adam(params, **kw)
    raise NotImplementedError('optimizer %r (the shipped senas configs use sgd / adam)' % spec['name'])


def _dist_setup():
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)
    return rank, world, dev


def _args(phase, argv):
    ap = argparse.ArgumentParser(prog='python -m senas_amd.%s' % phase)
    ap.add_argument('--config', default=DEFAULT_CONFIG, help='YAML with the reference\'s searching / training blocks')
    ap.add_argument('--epochs', type=int, default=None, help='override the block\'s epoch count')
    ap.add_argument('--steps-per-epoch', type=int, default=None, help='cap the batches per epoch')
    ap.add_argument('--batch-size', type=int, default=None)
    ap.add_argument('--size', type=int, default=None, help='slice edge of the synthetic source')
    ap.add_argument('--images', type=int, default=None, help='slices in the synthetic source')
    ap.add_argument('--save', default=None, help='directory for the reference-format checkpoint')
    ap.add_argument('--no-graph', action='store_true')
    if phase == 'train':
        ap.add_argument('--genotype', default=None, help='"Genotype(down=[...], ...)" text, as train_model.py:118 takes it')
    return ap.parse_args(argv)


def _source(cfg, args, nclass, in_ch, batch, rank, dev):
    syn = (cfg.get('data') or {}).get('synthetic') or {}
    size = args.size or syn.get('size', 256)
    images = args.images or syn.get('images', 64)
    return SyntheticSlices(max(images, batch), in_ch, nclass, size, batch, 1 + rank, dev)


def search(argv=None):
    """search_arc.py: two optimizers, ``Architecture.step`` on a validation batch from ``alpha_begin`` on, weight step on a
    training batch, cosine schedule per epoch, genotype after every epoch."""
    from . import checkpoint
    from .loss import MultiSegmentationLosses, SegmentationLosses
    from .models import DATASET_SHAPES
    from .parallel import broadcast_parameters
    from .senas_search import NAS
    from .step import SearchStep
    args = _args('search', argv)
    cfg = load_config(args.config)
    blk = cfg['searching']
    rank, world, dev = _dist_setup()
    torch.manual_seed(cfg.get('seed', 0))
    nclass, in_ch = DATASET_SHAPES[str(cfg['data']['dataset']).lower()]
    supervision = bool(blk.get('deep_supervision'))
    model = NAS(in_ch, blk['init_channels'], nclass, blk['depth'], meta_node_num=blk['meta_node_num'],
                use_sharing=blk['sharing_normal'], double_down_channel=blk['double_down_channel'], supervision=supervision,
                device=dev).to(dev).train()
    if world > 1:
        broadcast_parameters(model)
    # search_arc.py:107
    crit = MultiSegmentationLosses(blk['loss']['name'], blk['depth']) if supervision else SegmentationLosses(blk['loss']['name'])
    opt_w = _optimizer(model.parameters(), blk['model_optimizer'])
    opt_a = _optimizer(model.arch_parameters(), blk['arch_optimizer'])
    epochs = args.epochs if args.epochs is not None else blk['epoch']
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt_w, blk['epoch'])
    batch = args.batch_size or blk['batch_size']
    per_rank = max(1, batch // world)
    # train_portion splits the training set into weight-training and architecture-validation halves (search_arc.py:81-99)
    train = _source(cfg, args, nclass, in_ch, per_rank, rank, dev)
    valid = _source(cfg, args, nclass, in_ch, per_rank, rank + 1000, dev)
    x0, y0 = next(iter(train))
    step = SearchStep(model, crit, opt_w, opt_a, x0.clone(), y0.clone(), world_size=world, grad_clip=blk['grad_clip'],
                      use_graph=not args.no_graph)
    log = []
    t0 = time.time()
    for epoch in range(epochs):
        vit = iter(valid)
        losses = []
        for i, (x, y) in enumerate(train):
            if args.steps_per_epoch is not None and i >= args.steps_per_epoch:
                break
            xv = yv = None
            if epoch >= blk['alpha_begin']:                # the architecture moves only once the weights are warm
                try:
                    xv, yv = next(vit)
                except StopIteration:
                    vit = iter(valid)
                    xv, yv = next(vit)
            losses.append(step(x, y, xv, yv).clone())        # (a graphed step returns ONE static buffer: copy it)
        sched.step()
        mean = float(torch.stack(losses).mean()) if losses else float('nan')
        geno = model.genotype()
        log.append({'epoch': epoch, 'loss': mean, 'lr': opt_w.param_groups[0]['lr'], 'genotype': str(geno)})
        if rank == 0:
            print(json.dumps(log[-1]), flush=True)
        if args.save and rank == 0:
            os.makedirs(args.save, exist_ok=True)
            checkpoint.save_checkpoint(checkpoint.search_state(model, opt_a, opt_w, sched, epoch, time.time() - t0, 0, str(geno)),
                                       False, args.save)
    step.close()
    return log


def train(argv=None):
    """train_model.py: the derived network of ``--genotype`` (or the config's ``geno_type`` name), SGD + clip + cosine
    schedule per epoch."""
    from . import checkpoint, geno_searched
    from .genotype import Genotype  # noqa: F401  (the namespace ``--genotype`` text is evaluated in)
    from .loss import MultiSegmentationLosses, SegmentationLosses
    from .models import DATASET_SHAPES, get_segmentation_model
    from .parallel import broadcast_parameters
    from .step import TrainStep
    from .utils import weights_init
    args = _args('train', argv)
    cfg = load_config(args.config)
    blk = cfg['training']
    rank, world, dev = _dist_setup()
    torch.manual_seed(cfg.get('seed', 0))
    nclass, in_ch = DATASET_SHAPES[str(cfg['data']['dataset']).lower()]
    if args.genotype:
        geno = _parse_genotype(args.genotype)
    else:
        geno = getattr(geno_searched, blk['geno_type'])
    supervision = bool(blk.get('deep_supervision'))
    model = get_segmentation_model(cfg['model']['arch'], dataset=cfg['data']['dataset'], c=blk['init_channels'], depth=blk['depth'],
                                   supervision=supervision, genotype=geno, double_down_channel=blk['double_down_channel'])
    model.apply(weights_init)
    model = model.to(dev).train()
    if world > 1:
        broadcast_parameters(model)
    # train_model.py:111
    crit = MultiSegmentationLosses(blk['loss']['name'], blk['depth']) if supervision else SegmentationLosses(blk['loss']['name'])
    opt = _optimizer(model.parameters(), blk['model_optimizer'])
    sched = None
    if (blk.get('lr_schedule') or {}).get('name') == 'cos':
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, blk['lr_schedule']['T_max'])
    epochs = args.epochs if args.epochs is not None else blk['epoch']
    batch = args.batch_size or blk['batch_size']
    per_rank = max(1, batch // world)
    data = _source(cfg, args, nclass, in_ch, per_rank, rank, dev)
    x0, y0 = next(iter(data))
    x_buf, y_buf = x0.clone(), y0.clone()
    step = TrainStep(model, crit, opt, x_buf, y_buf, world_size=world, grad_clip=blk['grad_clip'], use_graph=not args.no_graph)
    log = []
    t0 = time.time()
    for epoch in range(epochs):
        losses = []
        for i, (x, y) in enumerate(data):
            if args.steps_per_epoch is not None and i >= args.steps_per_epoch:
                break
            x_buf.copy_(x, non_blocking=True)
            y_buf.copy_(y, non_blocking=True)
            losses.append(step().clone())
        if sched is not None:
            sched.step()
        mean = float(torch.stack(losses).mean()) if losses else float('nan')
        log.append({'epoch': epoch, 'loss': mean, 'lr': opt.param_groups[0]['lr']})
        if rank == 0:
            print(json.dumps(log[-1]), flush=True)
        if args.save and rank == 0:
            os.makedirs(args.save, exist_ok=True)
            checkpoint.save_checkpoint(checkpoint.train_state(model, opt, epoch, time.time() - t0, best_loss=mean), False, args.save)
    step.close()
    return log


def _parse_genotype(text):
    """``Genotype(down=[('op', idx), ...], down_concat=range(2, 6), up=[...], up_concat=range(2, 6), gamma=[...])`` -- the
    text train_model.py:118 ``eval``s; parsed here without eval."""
    from .genotype import Genotype
    tree = ast.parse(text.strip(), mode='eval').body
    if not (isinstance(tree, ast.Call) and getattr(tree.func, 'id', None) == 'Genotype' and not tree.args):
        raise ValueError('not a Genotype(...) expression')
    fields = {}
    for kw in tree.keywords:
        v = kw.value
        if isinstance(v, ast.Call) and getattr(v.func, 'id', None) == 'range':
            fields[kw.arg] = range(*[ast.literal_eval(a) for a in v.args])
        else:
            fields[kw.arg] = ast.literal_eval(v)
    return Genotype(**fields)


if __name__ == '__main__':
    sys.exit('use python -m senas_amd.search or python -m senas_amd.train')
