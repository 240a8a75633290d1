#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel symbol, grid size): per-shape launch counts and
average durations -- separates e.g. the node kernels of a 256x256 map from those of a 16x16 map.

    python tools/trace_by_grid.py <dir with *_kernel_trace.csv> [substring filter] [--steady N] > summary.txt

--steady N: count only the launches between the first pair of marker kernels (bench.py under SENAS_TRACE_MARKER=1 brackets
its timed region with torch.cuda._sleep) and print per-step figures for the N steps in between -- one-time set-up
launches (optimizer-state clones, capture warm-up, buffer snapshots) stay out.
"""
import collections
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)
    args = sys.argv[2:]
    steady = None
    if '--steady' in args:
        i = args.index('--steady')
        steady = int(args[i + 1])
        del args[i:i + 2]
    flt = args[0] if args else ''
    agg = collections.defaultdict(lambda: [0, 0.0])
    window = None
    if steady:
        marks = sorted(int(r['Start_Timestamp']) for f in files for r in csv.DictReader(open(f)) if 'spin' in r['Kernel_Name'] or 'sleep' in r['Kernel_Name'].lower())
        if len(marks) >= 2:
            window = (marks[0], marks[1])
    for f in files:
        for r in csv.DictReader(open(f)):
            if window and not (window[0] < int(r['Start_Timestamp']) < window[1]):
                continue
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('senas::', '').strip()
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
    if steady and window:
        n = sum(v[0] for _, v in rows)
        print('steady state, %d steps: %.1f launches per step, %.3f ms of kernel time per step' % (steady, n / steady, tot / steady / 1e3))
        fam = collections.defaultdict(lambda: [0, 0.0])
        for (name, grid), (cnt, us) in rows:
            a = fam[name.split('<')[0]]
            a[0] += cnt
            a[1] += us
        for name, (cnt, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            print('  family %-40s %7.1f launches/step %8.3f ms/step  avg %6.2f us' % (name[:40], cnt / steady, us / steady / 1e3, us / cnt))
    print('total us %.1f' % tot)
    for (name, grid), (cnt, us) in rows:
        print('%-44s grid %-22s calls %5d  total_us %10.1f  avg_us %8.2f' % (name[:44], 'x'.join(str(g) for g in grid), cnt, us, us / cnt))


if __name__ == '__main__':
    main()
