#!/usr/bin/env python3
"""Condenses a rocprofv3 output tree (scripts/profile_bench.sh) into a per-kernel text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    """kernel name as the summaries print it"""
    tags = (('grid_gather_regrid_kernel', 'grid_gather_regrid'), ('grid_block_lists_kernel', 'grid_block_lists'), ('gather_sums_kernel', 'gather_sums'),
            ('grid_regrid_kernel', 'grid_regrid'), ('grid_scatter_kernel', 'grid_scatter'), ('grid_prep_kernel', 'grid_prep'),
            ('snap_displace_kernel', 'snap_displace'), ('snap_halo_prep_kernel', 'snap_halo_prep'),
            ('deposit_keys_kernel', 'deposit_keys'), ('deposit_split_kernel<1', 'deposit_split<1>'), ('deposit_split_kernel<2', 'deposit_split<2>'),
            ('deposit_count_kernel', 'deposit_count'), ('deposit_tiles_kernel', 'deposit_tiles'), ('fft_c2c_strided_kernel<true', 'fft_c2c_strided+pk_bins'),
            ('fft_c2c_strided_kernel', 'fft_c2c_strided'),
            ('fft_r2c_lines_kernel', 'fft_r2c_lines'), ('pk_bin_kernel', 'pk_bin'),
            ('tile_scatter2f_kernel', 'tile_scatter2'), ('tile_scatter2_kernel', 'tile_scatter2'), ('tile_scatter_kernel', 'tile_scatter(generic)'), ('tile_regrid3_kernel', 'tile_regrid3'),
            ('halo_prep_kernel', 'halo_prep'), ('halo_scatter_kernel', 'halo_scatter'), ('regrid_far_kernel', 'regrid_far'),
            ('regrid_kernel', 'regrid(algo0)'), ('sum2_kernel', 'sum2'), ('sum_tiles_kernel', 'sum_tiles'), ('tile_scan_kernel', 'tile_scan'),
            ('tile_place_kernel', 'tile_place'), ('tile_reach_kernel', 'tile_reach'), ('tile_apron_kernel', 'tile_apron'))
    for k, t in tags:
        if k in name:
            mode = ''
            for m, v in (('<0', 'OFFSETS'), ('<(int)0', 'OFFSETS'), ('<1', 'PAINT'), ('<(int)1', 'PAINT'), ('<2', 'COUNT'), ('<(int)2', 'COUNT')):
                if k + m in name.replace('bfgx::', '').replace('void ', ''):
                    mode = '<' + v + (',f64' if 'double' in name else ',f32') + '>'
                    break
            if not mode and 'regrid3' in k:
                # <ACC, real, PASS>: 0 = lean gather (reach of one ring), 2 = gather with the ring walk, 1 = repair of an overflowing far list
                ps = {'0': 'lean', '1': 'repair', '2': 'walk'}
                mode = '<f64' if 'regrid3_kernel<double' in name else '<f32'
                for d in '012':
                    if (', %s>(' % d) in name or (', (int)%s>(' % d) in name:
                        mode += ',' + ps[d]
                mode += '>'
            elif not mode and 'scatter' in k:
                mode = '<f64>' if 'double' in name else '<f32>'
            return t + mode
    return name[:60]


print("== kernel trace stats ==")
for f in glob.glob(os.path.join(out, 'trace*', '**', '*kernel_stats.csv'), recursive=True):
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

# machine-readable: per-launch HBM traffic and VALU instruction counts of the two big kernels (bench.py reads this file)
import json
cfg = None
for f in glob.glob(os.path.join(out, 'bench_trace.json')):
    try:
        cfg = json.loads(open(f).read().strip().splitlines()[-1])['config']
    except Exception:
        pass


def mean(k, c):
    v = agg.get(k, {}).get(c)
    return sum(v) / len(v) if v else None


pick = {'offsets': 'tile_scatter2<OFFSETS,f32>', 'regrid': 'tile_regrid3<f32,lean>', 'paint': 'tile_scatter2<PAINT,f64>', 'prep': 'halo_prep'}
res = {"source": "rocprofv3 --pmc passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline` (scripts/profile_bench.sh), mean per "
                 "dispatch; traffic = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 bytes: on gfx950 FETCH_SIZE reports half the bytes of wide "
                 "coalesced reads (MI355X_MICROARCH.md, HBM section) -- the correction is applied although part of these kernels' reads are "
                 "narrow gathers, so the figure is an upper bound; valu_wave_insts = SQ_INSTS_VALU",
       "config": None if cfg is None else {"halos_per_gpu": cfg.get('halos_per_gpu'), "nside": cfg.get('nside'), "algo": 1, "mode": "baryonify"},
       "kernels": {}, "valu_wave_insts": {}, "fetch_kb": {}, "write_kb": {}}
for key, kn in pick.items():
    fs, ws, vi = mean(kn, 'FETCH_SIZE'), mean(kn, 'WRITE_SIZE'), mean(kn, 'SQ_INSTS_VALU')
    if fs is not None and ws is not None:
        res["kernels"][key] = fs * 1024 * 2 + ws * 1024
        res["fetch_kb"][key], res["write_kb"][key] = fs, ws
    if vi is not None:
        res["valu_wave_insts"][key] = vi
json.dump(res, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
