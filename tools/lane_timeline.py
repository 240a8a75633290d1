#!/usr/bin/env python3
"""What really overlaps in a step that runs on several HIP streams (senas_amd.grid.Lanes) under HIP-graph replay: device
time stamps in stream order at the boundaries of every cell, forward and backward (senas_stamp; functional.stamp), read
after ONE replay of the step.  A tracing profiler serialises the hardware queues; these stamps do not.

    python tools/lane_timeline.py search|train [--serial] [--steady] [--segments]

--steady: the stamps of the LAST of six back-to-back replays (the host has run ahead of the device, as in a training loop)
instead of those of one replay launched on an idle device (whose first cells also wait for the host to issue their segments).
--segments: also the lane scheduler's plan (SENAS_SCHED_DUMP): every segment with its lane, its stream of the pool, the segments
it waits for and the cell boundaries (stamps) it holds, with the time each stamp showed.

Prints per pass: every cell with the start / end of its forward and of its backward part (us from the first stamp of the
pass), the sum of the cells' own durations and the span they cover.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from senas_amd import functional as F  # noqa: E402
from senas_amd import grid  # noqa: E402
from senas_amd.loss import SegmentationLosses  # noqa: E402


def report(rows, title):
    cells, order = {}, []
    for name, f, b in rows:
        cell, point = name.rsplit('.', 1)
        if cell not in cells:
            cells[cell] = {}
            order.append(cell)
        cells[cell][point] = (f, b)
    t0 = min(v for c in cells.values() for p in c.values() for v in p if v)
    us = lambda t: (t - t0) / 100.0 if t else float('nan')
    print('\n== %s' % title)
    print('%-8s %10s %10s %8s | %10s %10s %8s' % ('cell', 'fwd start', 'fwd end', 'us', 'bwd start', 'bwd end', 'us'))
    fsum = bsum = 0.0
    last = 0.0
    for cell in order:
        c = cells[cell]
        fs = min(us(c[p][0]) for p in ('in0', 'in1') if p in c)
        fe = us(c['out'][0])
        bs = us(c['out'][1])
        be = max([us(c[p][1]) for p in ('in0', 'in1') if p in c and c[p][1]] or [float('nan')])
        fsum += fe - fs
        bsum += (be - bs) if be == be else 0.0
        last = max(last, fe, be if be == be else 0.0)
        print('%-8s %10.1f %10.1f %8.1f | %10.1f %10.1f %8.1f' % (cell, fs, fe, fe - fs, bs, be, be - bs))
    print('cells: forward %.1f us + backward %.1f us = %.1f us of cell time inside a span of %.1f us' % (fsum, bsum, fsum + bsum, last))


def split_passes(rows):
    """The stamps of the two captured passes of a search step (architecture pass first)."""
    first = rows[0][0]
    cut = [k for k, r in enumerate(rows) if r[0] == first]
    return [rows[a:b] for a, b in zip(cut, cut[1:] + [len(rows)])]


def segments(path, rec):
    """The scheduler's dump against the stamps: per captured pass, every segment with its lane, stream, the segments it waits for
    and the stamps it holds (name:direction@us from the pass's first stamp)."""
    base = rec.buf.data_ptr()
    v = rec.buf.cpu().tolist()
    passes, cur = [], None
    for line in open(path):
        w = line.split()
        if w[0] == 'sched':
            cur = []
            passes.append((line.strip(), cur))
        elif w[0] == 'seg':
            d, st = w.index('deps'), w.index('stamps')
            marks = [(int(ptr) - base) // 8 for ptr in w[st + 1:]]
            marks = [k for k in marks if 0 <= k // 2 < len(rec.names) and v[k]]
            cur.append(dict(k=int(w[1]), lane=int(w[3]), stream=int(w[5]), nodes=int(w[7]), deps=[int(x) for x in w[d + 1:st]], marks=marks))
    for head, segs in passes:
        if not any(sg['marks'] for sg in segs):
            continue
        print('\n== ' + head)
        z = min(v[k] for sg in segs for k in sg['marks'])
        for sg in segs:
            ms = ' '.join('%s:%s@%.0f' % (rec.names[k // 2], 'fb'[k % 2], (v[k] - z) / 100.0) for k in sg['marks'])
            print('seg %3d lane %d stream %d nodes %4d waits %-14s %s' % (sg['k'], sg['lane'], sg['stream'], sg['nodes'],
                                                                          ','.join(map(str, sg['deps'])) or '-', ms))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'search'
    dump = None
    if '--segments' in sys.argv:
        dump = '/tmp/senas_sched_dump_%d.txt' % os.getpid()
        os.environ['SENAS_SCHED_DUMP'] = dump
    grid.Lanes.enabled = '--serial' not in sys.argv
    dev = torch.device('cuda:0')
    F.STAMPS = rec = F.StampRecorder(dev)
    crit = SegmentationLosses('dice_ce')
    if what == 'search':
        from senas_amd.senas_search import NAS
        from senas_amd.step import SearchStep
        torch.manual_seed(0)
        net = NAS(1, 32, 2, 5, meta_node_num=3, use_sharing=False, double_down_channel=False).to(dev).train()
        opt_w = torch.optim.SGD(net.parameters(), lr=5e-3, weight_decay=3e-4, momentum=0.9)
        opt_a = torch.optim.Adam(net.arch_parameters(), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-3)
        xt, yt = bench.synthetic(4, 1, 2, 256, 1, dev)
        xv, yv = bench.synthetic(4, 1, 2, 256, 101, dev)
        drv = SearchStep(net, crit, opt_w, opt_a, xt.clone(), yt.clone())
        step = lambda: drv(xt, yt, xv, yv)
    else:
        from senas_amd.step import TrainStep
        net = bench.build_derived(dev)
        opt = torch.optim.SGD(net.parameters(), lr=6e-3, weight_decay=5e-4, momentum=0.9)
        x, y = bench.synthetic(8, 1, 2, 256, 1, dev)
        drv = TrainStep(net, crit, opt, x, y)
        step = drv
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    print('%s step with stamps, lanes=%s: %.3f ms' % (what, grid.Lanes.enabled, e0.elapsed_time(e1) / 10))
    rec.buf.zero_()
    torch.cuda.synchronize()
    for _ in range(6 if '--steady' in sys.argv else 1):
        step()
    torch.cuda.synchronize()
    rows = rec.read()
    if '--order' in sys.argv:
        # every stamp of the replay in time order (name, direction): tools/cell_kernels.py matches them with the stamp kernels of
        # a rocprofv3 --kernel-trace of this very command to cut the trace into cells
        import json
        seq = sorted([(f, n, 'f') for n, f, b in rows if f] + [(b, n, 'b') for n, f, b in rows if b])
        with open(sys.argv[sys.argv.index('--order') + 1], 'w') as fh:
            json.dump([[n, d] for _, n, d in seq], fh)
    if dump is not None and os.path.exists(dump):
        segments(dump, rec)
    passes = split_passes(rows)
    titles = ['architecture pass (weights frozen)', 'weight pass'] if what == 'search' else ['train step']
    for rows_, title in zip(passes, titles):
        report(rows_, title)


if __name__ == '__main__':
    main()
