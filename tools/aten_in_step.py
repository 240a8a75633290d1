#!/usr/bin/env python3
"""List the kernels that are NOT libsenas_hip's inside the steady-state window of a rocprofv3 --kernel-trace of
tools/search_profile.py / bench.py (SENAS_TRACE_MARKER=1): full symbol, grid, launches per step, time per step.
    python tools/aten_in_step.py <trace dir> <steps>"""
import collections
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)
    steps = int(sys.argv[2])
    rows = [r for f in files for r in csv.DictReader(open(f))]
    marks = sorted(int(r['Start_Timestamp']) for r in rows if 'spin' in r['Kernel_Name'] or 'sleep' in r['Kernel_Name'].lower())
    lo, hi = marks[0], marks[1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    total = [0, 0.0]
    for r in rows:
        t = int(r['Start_Timestamp'])
        if not lo < t < hi:
            continue
        us = (int(r['End_Timestamp']) - t) / 1e3
        total[0] += 1
        total[1] += us
        if 'senas::' in r['Kernel_Name']:
            continue
        a = agg[(r['Kernel_Name'][:150], r.get('Grid_Size_X') or r.get('Grid_Size'))]
        a[0] += 1
        a[1] += us
    print('window: %.1f launches/step, %.3f ms/step; foreign kernels: %.1f launches/step, %.3f ms/step' % (
        total[0] / steps, total[1] / steps / 1e3, sum(v[0] for v in agg.values()) / steps, sum(v[1] for v in agg.values()) / steps / 1e3))
    for (name, grid), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print('%6.1f/step %7.1f us/step  grid %-8s %s' % (cnt / steps, us / steps, grid, name))


if __name__ == '__main__':
    main()
