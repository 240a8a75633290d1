#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into profiles/r2_pmc_traffic.json:
average FETCH_SIZE / WRITE_SIZE (KB) per launch, keyed by kernel symbol (senas:: prefix and argument list
stripped, i.e. the name bench.py / senas_conv2d_kernel_name use).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW profiles/r2_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import sys


def load(path, counter):
    files = glob.glob(path + '/**/*_counter_collection.csv', recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace("void ", "").replace("senas::", "").strip()
            acc[name].append(float(r['Counter_Value']))
    return acc


def main():
    """pmc_traffic.py <FETCH dir> <WRITE dir> <out.json> [--steps N] [--all]: --steps records over how many steps of the
    profiled program the launches were counted (per-step totals = sum / N); --all keeps every kernel symbol (torch's too)."""
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else None
    keep_all = '--all' in sys.argv
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over bench.py; KB per launch, '
                     'averaged over all launches of the kernel symbol; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950',
           'kernels': {}}
    for name in sorted(set(fetch) | set(write)):
        if not keep_all and not any(s in name for s in ('conv', 'wgrad', 'node_', 'chan_stats', 'pool', 'bilinear', 'relu')):
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        out['kernels'][name] = {'launches': max(len(f), len(w)),
                                'fetch_kb_avg': sum(f) / len(f) if f else 0.0,
                                'write_kb_avg': sum(w) / len(w) if w else 0.0}
    if steps:
        out['steps'] = steps
        out['hbm_bytes_per_step'] = int(sum((2 * v['fetch_kb_avg'] + v['write_kb_avg']) * 1024 * v['launches'] for v in out['kernels'].values()) / steps)
        fam = {}
        for k, v in out['kernels'].items():
            key = k.split('<')[0]
            a = fam.setdefault(key, [0.0, 0])
            a[0] += (2 * v['fetch_kb_avg'] + v['write_kb_avg']) * 1024 * v['launches'] / steps
            a[1] += v['launches'] / steps
        out['families'] = {k: {'hbm_bytes_per_step': int(b), 'launches_per_step': round(n, 1)} for k, (b, n) in
                           sorted(fam.items(), key=lambda kv: -kv[1][0])}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in out['kernels'].items():
        print('%-40s launches %5d  fetch %10.1f KB  write %10.1f KB' % (k, v['launches'], v['fetch_kb_avg'], v['write_kb_avg']))


if __name__ == '__main__':
    main()
