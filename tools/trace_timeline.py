#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV for a step that runs on several hardware queues (senas_amd.grid.Lanes):
does kernel time overlap, which queue is busy when, and where is the time in which only one queue runs.

    python tools/trace_timeline.py <dir with *_kernel_trace.csv> --steady N [--bins 40]

Only the launches between the two marker kernels of bench.py / tools/search_profile.py (SENAS_TRACE_MARKER=1).  Prints:
wall time per step, sum of kernel time per step (> wall: overlap), time with 0 / 1 / 2 / ... kernels in flight, busy time per
queue, and a coarse timeline of ONE step (the second of the window): per time bin, the busy share of every queue and the
kernel family that took most of the bin on it.
"""
import collections
import csv
import glob
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('senas::', '').strip()
    return name.split('<')[0]


def main():
    files = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)
    args = sys.argv[2:]
    steady = int(args[args.index('--steady') + 1]) if '--steady' in args else 1
    bins = int(args[args.index('--bins') + 1]) if '--bins' in args else 40
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'],
                         int(r.get('Grid_Size_X') or 0) * int(r.get('Grid_Size_Y') or 1) * int(r.get('Grid_Size_Z') or 1)))
    marks = sorted(r[0] for r in rows if 'spin' in r[3] or 'sleep' in r[3].lower())
    if len(marks) < 2:
        sys.exit('no marker kernels in the trace (SENAS_TRACE_MARKER=1)')
    lo, hi = marks[0], marks[1]
    rows = sorted(r for r in rows if lo < r[0] < hi and not ('spin' in r[3] or 'sleep' in r[3].lower()))
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = (t1 - t0) / 1e6
    ksum = sum(r[1] - r[0] for r in rows) / 1e6
    print('window: %d launches, %d steps: wall %.3f ms/step, kernel time %.3f ms/step (ratio %.2f)' %
          (len(rows), steady, wall / steady, ksum / steady, ksum / wall))
    # time with k kernels in flight
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[depth] += t - last
        last, depth = t, depth + d
    print('kernels in flight:  ' + '  '.join('%d: %.3f ms/step' % (k, v / 1e6 / steady) for k, v in sorted(hist.items())))
    byq = collections.defaultdict(float)
    cnt = collections.Counter()
    for s, e, q, *_ in rows:
        byq[q] += (e - s) / 1e6
        cnt[q] += 1
    for q in sorted(byq, key=lambda q: -byq[q]):
        print('  queue %-4s %6d launches/step  busy %.3f ms/step' % (q, cnt[q] // steady, byq[q] / steady))
    # one step: the window cut in `steady` equal parts by launch count is wrong when steps differ; cut by time instead
    step_len = (t1 - t0) / steady
    k = 1 if steady > 1 else 0
    a, b = t0 + k * step_len, t0 + (k + 1) * step_len
    width = (b - a) / bins
    queues = sorted(byq, key=lambda q: -byq[q])
    print('\ntimeline of step %d (%.3f ms, %d bins of %.0f us): per queue busy %% and dominant kernel family' % (k, (b - a) / 1e6, bins, width / 1e3))
    for i in range(bins):
        x0, x1 = a + i * width, a + (i + 1) * width
        cells = []
        for q in queues:
            busy, fam = 0.0, collections.Counter()
            for s, e, qq, name, grid in rows:
                if qq != q or e <= x0 or s >= x1:
                    continue
                d = min(e, x1) - max(s, x0)
                busy += d
                fam['%s/%d' % (short(name)[:18], grid)] += d
            cells.append('%3d%% %-28s' % (100 * busy / width, fam.most_common(1)[0][0] if fam else '-'))
        print('%7.3f ms | %s' % ((x0 - a) / 1e6, ' | '.join(cells)))


if __name__ == '__main__':
    main()
