#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE run of bench.py into a per-kernel
table of matrix-pipe utilisation:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv \
        -d gpurun_out/pmcM -- python3 bench.py --steps 3 --warmup 1 --search-steps 0 --no-cpu-baseline
    python tools/pmc_mfma.py gpurun_out/pmcM profiles/r2_pmc_mfma.json

mfma_util = 100 * SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs): the share of (CU, SIMD) cycles of
the dispatch in which the matrix pipe was executing.  rocprofv3 reports both counters summed over the 8 XCDs: MFMA_BUSY is
then the chip total (checked: 170.0 M per launch of the dominant convolution = its 2.65 M wave-level v_mfma_f32_32x32x2_f32
x 64 cycles), GRBM_GUI_ACTIVE is 8 x the dispatch's cycles (2.09 M per 107 us launch = 8 x 261 k), hence the / 8.
"""
import collections
import csv
import glob
import json
import sys

CUS = 256


def main():
    files = glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('senas::', '').strip()
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE over bench.py; per kernel symbol, '
                     'summed over all its launches; mfma_util = 100 * MFMA_BUSY / (GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)', 'kernels': {}}
    rows = []
    for name, c in acc.items():
        busy, gui = sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), sum(c.get('GRBM_GUI_ACTIVE', []))
        if gui <= 0:
            continue
        rec = {'launches': len(c.get('GRBM_GUI_ACTIVE', [])), 'mfma_busy_cycles': busy, 'gui_active_cycles': gui,
               'sq_busy_cu_cycles': sum(c.get('SQ_BUSY_CU_CYCLES', [])), 'mfma_util_pct': round(100.0 * busy / (gui / 8.0 * CUS * 4), 2)}
        out['kernels'][name] = rec
        rows.append((gui, name, rec))
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for gui, name, rec in sorted(rows, reverse=True)[:25]:
        print('%-60s launches %5d  gui_active %12.0f  mfma_util %6.2f %%' % (name[:60], rec['launches'], gui, rec['mfma_util_pct']))


if __name__ == '__main__':
    main()
