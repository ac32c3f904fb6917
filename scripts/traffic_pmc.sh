#!/bin/bash
# GPU box: HBM traffic and VALU instruction counts per bench kernel group, from rocprofv3 --pmc passes of bench.py (separate passes for
# FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: MI355X_MICROARCH.md, HBM / rocprofv3 section) -> gpurun_out/traffic_<tag>.json, to be copied to
# profiles/traffic_<tag>.json, which bench.py reads for `roofline.traffic`.
#   scripts/traffic_pmc.sh c2                                   scripts/traffic_pmc.sh c3 --mode paint --nside 2048
#   scripts/traffic_pmc.sh c4 --halos 1250000 --nside 2048      scripts/traffic_pmc.sh grid3d --mode grid3d     scripts/traffic_pmc.sh snapshot --mode snapshot
set -o pipefail
TAG=$1; shift
OUT=/tmp/traffic_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out
export TMPDIR=/tmp
W=2; K=6
ARGS="--steps $K --warmup $W --no-cpu-baseline --no-extras --no-kernel-events $@"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 bench.py $ARGS > $OUT/$C.json 2> $OUT/$C.err || echo "pass $C failed: $(tail -2 $OUT/$C.err)"
done
python3 - "$OUT" "$TAG" $((W + K)) "$*" <<'PY' | tee gpurun_out/traffic_$TAG.json
import csv, glob, json, sys, collections
out, tag, nsteps, args = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
# bench.py kernel_ms key <- kernel name fragments (the groups bench.py times between one pair of events)
GROUPS = {'offsets': ['tile_scatter2_kernel<0', 'tile_scatter2_kernel<(int)0', 'tile_scatter2f_kernel<0', 'tile_scatter2f_kernel<(int)0', 'grid_gather_regrid_kernel'],
          'paint': ['tile_scatter2_kernel<1', 'tile_scatter2_kernel<(int)1', 'tile_scatter2f_kernel<1', 'tile_scatter2f_kernel<(int)1'],
          'regrid': ['tile_regrid3_kernel', 'grid_copy_sum_kernel'], 'prep': ['halo_prep_kernel', 'grid_prep_kernel'],
          'deposit': ['deposit_keys', 'deposit_split', 'deposit_count', 'deposit_tiles', 'deposit_scan', 'deposit_atomic'],
          'pk': ['fft_r2c_lines', 'fft_c2c_strided'], 'displace': ['snap_displace_kernel']}
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].replace('bfgx::', '').replace('void ', '')
        for g, frags in GROUPS.items():
            if any(fr in name for fr in frags):
                tot[row['Counter_Name']][g] += float(row['Counter_Value'])
                break
fetch = {g: v / nsteps for g, v in tot['FETCH_SIZE'].items()}
write = {g: v / nsteps for g, v in tot['WRITE_SIZE'].items()}
if 'displace' in fetch and 'deposit' in fetch:          # the fused snapshot flow is timed (and keyed by bench.py) as one group
    for d in (fetch, write, tot['SQ_INSTS_VALU']):
        d['displace+deposit'] = d.get('displace', 0.0) + d.get('deposit', 0.0)
bench = json.loads([l for l in open(out + '/FETCH_SIZE.json') if l.startswith('{')][0])
print(json.dumps({
    "source": "rocprofv3 --pmc passes (one counter per pass) of `python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events %s` "
              "(scripts/traffic_pmc.sh), summed over the kernels of a group and divided by the 8 steps of the run; traffic = FETCH_SIZE * 1024 * 2 + "
              "WRITE_SIZE * 1024 bytes: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -- the "
              "correction is applied although part of these kernels' reads are narrow gathers, so the figure is an upper bound; valu_wave_insts = SQ_INSTS_VALU" % args,
    "tag": tag, "config": bench.get("config", {}), "metric": bench.get("metric"),
    "kernels": {g: fetch.get(g, 0.0) * 2048 + write.get(g, 0.0) * 1024 for g in sorted(set(fetch) | set(write))},
    "valu_wave_insts": {g: v / nsteps for g, v in tot['SQ_INSTS_VALU'].items()},
    "fetch_kb": fetch, "write_kb": write}, indent=1))
PY
