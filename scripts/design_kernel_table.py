"""Regenerates the current-state kernel table of DESIGN.md section 4 from the committed evidence of one round:

    python scripts/design_kernel_table.py r05 > /tmp/table.md          # reads profiles/r05_*

For every configuration of the headline step (table x precision) one row per kernel: average duration from the rocprofv3 --kernel-trace --stats
csv of `bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras <args>`, HBM bytes and wave instructions per launch from the separate --pmc
passes (profiles/traffic_*.json), the algorithmic bytes of SURVEY 8(d) and the fraction of the 8 TB/s roof they make.  Nothing in here is typed in
by hand: a number in DESIGN section 4 that this script does not print is history."""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(HERE, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'

N_PAIRS, N_HALOS, NPIX = 53_984_077, 1_000_000, 12 * 1024 * 1024

# (label, kernel-stats csv suffix, traffic json, pix_offsets bytes per component)
CONFIGS = [("S19 table (benchmark, SURVEY 8d-ii), parity-grade: the DEFAULT and the headline `value`", 's19_parity', 'traffic_s19.json', 8),
           ("S19 table, fp64 throughout (`value_acc_f64`)", 's19_f64', 'traffic_s19_f64.json', 8),
           ("S19 table, fp32 pair math (`value_f32`; 2e-5 mean(map): outside the contract on this table)", 's19_f32', 'traffic_s19_f32.json', 4),
           ("closed-form table (plumbing, SURVEY 8d-i), default = fp32 pair math (`value_closed_form`)", 'cf', 'traffic_c2.json', 4),
           ("closed-form table, fp64 throughout", 'cf_f64', 'traffic_c2_f64.json', 8)]

GROUPS = [('K0 `halo_prep_kernel`', ['halo_prep_kernel'], 'prep', lambda b: N_HALOS * 48),
          ('binning: `tile_scan` + `tile_place` + memset', ['tile_scan', 'tile_place', 'fillBufferAligned'], None, lambda b: None),
          ('K1 `tile_scatter2f_kernel`', ['tile_scatter2f_kernel'], 'offsets', lambda b: N_PAIRS * 3 * b + N_HALOS * 32),
          ('K2 `tile_apron` + `tile_regrid3<.., 0 / 2 / 1>` (lean, walking, far + sums)', ['tile_apron_kernel', 'tile_regrid3_kernel'], 'regrid',
           lambda b: NPIX * (3 * b + 8 + 4 * 8 + 8))]


def stats(name):
    f = os.path.join(P, '%s_bench_%s_kernel_stats.csv' % (tag, name))
    if not os.path.exists(f):
        return None
    rows = list(csv.DictReader(open(f)))
    steps = 0
    for r in rows:
        if 'tile_regrid3_kernel' in r['Name'] and ', 1' in r['Name'].split('(')[0][-12:]:
            steps = int(r['Calls'])
    steps = steps or 105
    return rows, steps


for label, name, tfile, accb in CONFIGS:
    st = stats(name)
    if st is None:
        continue
    rows, steps = st
    tj = None
    tf = os.path.join(P, tfile)
    if os.path.exists(tf):
        tj = json.load(open(tf))
    print("**%s** (`profiles/%s_bench_%s_kernel_stats.csv`, %d steps%s)\n" % (label, tag, name, steps, ", `profiles/%s`" % tfile if tj else ""))
    print("| kernel | us per step | algorithmic MB (SURVEY 8d) | of the 8 TB/s roof | HBM MB per step (PMC) | wave instructions per step |")
    print("|---|---|---|---|---|---|")
    total = 0.0
    for gl, frags, key, alg in GROUPS:
        us = sum(float(r['TotalDurationNs']) for r in rows if any(fr in r['Name'] for fr in frags) and 'tile_scatter2_kernel<2' not in r['Name']) / 1e3 / steps
        if us == 0:
            continue
        total += us
        a = alg(accb)
        hb = tj['kernels'].get(key) if (tj and key) else None
        vi = tj.get('valu_wave_insts', {}).get(key) if (tj and key) else None
        print("| %s | %.1f | %s | %s | %s | %s |" % (gl, us, "%.0f" % (a / 1e6) if a else "--", "%.3f" % (a / (us * 1e-6) / 8e12) if a else "--",
                                                  "%.0f" % (hb / 1e6) if hb else "--", "%.3g" % vi if vi else "--"))
    print("| sum of the kernels | %.1f | | | | |\n" % total)
