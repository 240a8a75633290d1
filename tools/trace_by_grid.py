#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel symbol, grid size): per-shape launch counts and
average durations -- separates e.g. the node kernels of a 256x256 map from those of a 16x16 map.

    python tools/trace_by_grid.py <dir with *_kernel_trace.csv> [substring filter] > summary.txt
"""
import collections
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('senas::', '').strip()
            if name.startswith('at::native'):                     # keep what tells torch's elementwise kernels apart
                full = r['Kernel_Name']
                for key in ('Functor_add', 'FillFunctor', 'MulFunctor', 'copy', 'CatArray', 'sigmoid', 'softmax', 'reduce_kernel',
                            'div', 'sub', 'neg', 'index', 'sum', 'exp', 'threshold', 'clamp'):
                    if key.lower() in full.lower():
                        name = 'torch:' + key
                        break
            if flt and flt not in name:
                continue
            grid = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
            a = agg[(name, grid)]
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print('total us %.1f' % tot)
    for (name, grid), (cnt, us) in rows:
        print('%-44s grid %-22s calls %5d  total_us %10.1f  avg_us %8.2f' % (name[:44], 'x'.join(str(g) for g in grid), cnt, us, us / cnt))


if __name__ == '__main__':
    main()
