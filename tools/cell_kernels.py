#!/usr/bin/env python3
"""Kernel time per cell: cut a rocprofv3 --kernel-trace of `tools/lane_timeline.py <search|train> --serial --order order.json` into
cells with the stamp kernels (senas_stamp) that bracket every cell's forward and backward part.

    python tools/cell_kernels.py <trace dir> order.json [--ordered] [cell ...] > table.txt

The LAST replay of the run is the one whose stamps order.json lists (in time order); the last len(order) stamp kernels of the
trace are matched with it one to one.  Prints, per requested cell (default: all, summary only) and direction, the kernels between
the cell's first and last stamp: family, launches, total and average time.  Serial schedule only: with lanes the cells interleave.
"""
import collections
import csv
import glob
import json
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('senas::', '').strip()
    if name.startswith('at::native'):
        return 'torch:' + name.split('<')[0].split('::')[-1]
    return name.split('<')[0]


def main():
    files = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)
    order = json.load(open(sys.argv[2]))
    ordered = '--ordered' in sys.argv
    want = [a for a in sys.argv[3:] if a != '--ordered']
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                         int(r.get('Grid_Size_X') or 0) * int(r.get('Grid_Size_Y') or 1) * int(r.get('Grid_Size_Z') or 1)))
    rows.sort()
    stamps = [i for i, r in enumerate(rows) if 'stamp_kernel' in r[2]]
    if len(stamps) < len(order):
        sys.exit('the trace holds %d stamp kernels, order.json lists %d' % (len(stamps), len(order)))
    stamps = stamps[-len(order):]
    at = {}
    nth = collections.Counter()                      # the two passes of a search step repeat the stamp names: number them
    for idx, (name, d) in zip(stamps, order):
        nth[(name, d)] += 1
        cell, point = name.rsplit('.', 1)
        at.setdefault((cell, d, nth[(name, d)]), []).append(idx)
    cells = sorted((min(idxs), cell, d, k, min(idxs), max(idxs)) for (cell, d, k), idxs in at.items())
    print('%-10s %-4s %5s %9s %9s' % ('cell', 'dir', 'pass', 'launches', 'kernel us'))
    tables = []
    seen = {}
    for _, cell, d, k, lo, hi in cells:
        seen[(cell, d)] = k
        ks = [r for r in rows[lo + 1:hi] if 'stamp_kernel' not in r[2]]
        tot = sum(r[1] - r[0] for r in ks) / 1e3
        print('%-10s %-4s %5d %9d %9.1f' % (cell, 'fwd' if d == 'f' else 'bwd', k, len(ks), tot))
        if cell in want and ordered:
            # every launch in issue order: start offset from the cell's first kernel, duration, idle gap in front of it
            print('\n== %s %s (pass %d), in order: %d launches, %.1f us of kernel time, %.1f us first start to last end' % (
                cell, 'forward' if d == 'f' else 'backward', k, len(ks), tot, (ks[-1][1] - ks[0][0]) / 1e3 if ks else 0.0))
            prev = None
            for s, e, n, g in ks:
                print('   %8.1f  %7.1f us  gap %6.1f  %s/%d' % ((s - ks[0][0]) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, short(n)[:60], g))
                prev = e
        if cell in want:
            fam = collections.defaultdict(lambda: [0, 0.0])
            for s, e, n, g in ks:
                a = fam['%s/%d' % (short(n), g)]
                a[0] += 1
                a[1] += (e - s) / 1e3
            tables.append((cell, d, seen[(cell, d)], tot, sorted(fam.items(), key=lambda kv: -kv[1][1])))
    for cell, d, k, tot, fam in tables:
        print('\n== %s %s (pass %d): %.1f us of kernel time' % (cell, 'forward' if d == 'f' else 'backward', k, tot))
        for name, (cnt, us) in fam:
            print('   %-44s x%-3d %9.1f us  avg %7.1f' % (name[:44], cnt, us, us / cnt))


if __name__ == '__main__':
    main()
