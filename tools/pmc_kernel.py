#!/usr/bin/env python3
"""Per-launch averages of whatever counters a set of rocprofv3 --pmc passes collected, for kernels whose symbol contains
a substring (tuning aid: one convolution launched repeatedly by tools/conv_bench.py, one pass per counter group).

    python tools/pmc_kernel.py <dir with the passes' output> conv_pipe_kernel
"""
import collections
import csv
import glob
import sys


def main():
    acc = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(acc):
        v = acc[k]
        print('%-34s launches %4d   avg per launch %16.1f' % (k, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main()
