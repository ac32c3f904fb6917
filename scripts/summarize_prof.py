#!/usr/bin/env python3
"""Condenses a rocprofv3 output tree (scripts/profile_bench.sh) into a per-kernel text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for k in ('halo_prep_kernel', 'halo_scatter_kernel', 'tile_regrid_kernel', 'regrid_kernel', 'sum2_kernel'):
        if k in name:
            if k == 'halo_scatter_kernel':
                mode = {'ILi0E': 'OFFSETS', 'ILi1E': 'PAINT', 'ILi2E': 'COUNT'}
                for m, v in mode.items():
                    if m in name:
                        return 'halo_scatter<%s,%s>' % (v, 'f64' if (m + 'd') in name else 'f32')
                if '<0' in name or '<(int)0' in name:
                    return 'halo_scatter<OFFSETS>'
            return k
    return name[:60]


print("== kernel trace stats ==")
for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-40s calls %6s  avg %12.1f ns  total %14s ns  %6s %%" % (
                short(row['Name']), row['Calls'], float(row['AverageNs']), row['TotalDurationNs'], row['Percentage']))

print("\n== PMC (mean per dispatch) ==")
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-28s %18.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
