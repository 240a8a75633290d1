#!/usr/bin/env python3
"""Which kernels run right before / after a given kernel in a rocprofv3 --kernel-trace CSV (by start time) --
to find out who issues anonymous runtime kernels such as __amd_rocclr_copyBuffer.

    python tools/trace_neighbors.py <dir> copyBuffer [context launches]
"""
import collections
import csv
import glob
import sys


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('senas::', '')[:60]))
    rows.sort()
    pat = sys.argv[2]
    before, after = collections.Counter(), collections.Counter()
    for i, (_, name) in enumerate(rows):
        if pat in name:
            before[rows[i - 1][1] if i else '-'] += 1
            after[rows[i + 1][1] if i + 1 < len(rows) else '-'] += 1
    if len(sys.argv) > 3:                       # context of the LAST run of matches: the <n> launches around it
        n = int(sys.argv[3])
        last = max(i for i, (_, name) in enumerate(rows) if pat in name)
        first = last
        while first > 0 and pat in rows[first - 1][1]:
            first -= 1
        for i in range(max(0, first - n), min(len(rows), last + n + 1)):
            print('  %12d  %s' % (rows[i][0] - rows[first][0], rows[i][1]))
    print('before:')
    for k, v in before.most_common(8):
        print('  %5d  %s' % (v, k))
    print('after:')
    for k, v in after.most_common(8):
        print('  %5d  %s' % (v, k))


if __name__ == '__main__':
    main()
