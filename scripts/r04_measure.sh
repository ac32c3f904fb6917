#!/bin/bash
# GPU box: the round's bench lines + rocprofv3 kernel stats + per-mode PMC traffic files -> gpurun_out/r04m/ (copied to profiles/ by hand)
# usage: scripts/r04_measure.sh <tag>
TAG=${1:-r04}
O=gpurun_out/${TAG}m; mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_c2_line.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python3 bench.py --mode paint --nside 2048 --no-cpu-baseline > $O/bench_paint_c3_line.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python3 bench.py --halos 1250000 --nside 2048 --no-cpu-baseline --no-extras > $O/bench_c4_per_gpu_line.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python3 bench.py --mode grid3d --steps 30 --warmup 3 > $O/grid3d_bench_line.json 2> $O/grid3d.err; echo "grid3d rc=$?"
python3 bench.py --mode snapshot --steps 30 --warmup 3 --no-cpu-baseline > $O/snapshot_bench_line.json 2> $O/snapshot.err; echo "snapshot rc=$?"
BFGX_FORCE_EXCHANGE=1 python3 bench.py --steps 300 --no-extras --no-cpu-baseline > $O/rccl_single_rank_spatial_line.json 2> $O/forcex.err; echo "forcex rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/trace_c2.json 2> $O/trace_c2.err; echo "trace rc=$?"
find $O/trace_c2 -name "*kernel_stats.csv" -exec cp {} $O/bench_c2_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_grid3d -- python3 bench.py --mode grid3d --steps 10 --warmup 2 --no-cpu-baseline > $O/trace_grid3d.json 2> $O/trace_grid3d.err
find $O/trace_grid3d -name "*kernel_stats.csv" -exec cp {} $O/grid3d_kernel_stats.csv \;
# kernel trace + PMC passes of the same step on the Schneider19 table (every K2 tile in the walking kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s19 -- python3 bench.py --table s19 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_s19_line.json 2> $O/trace_s19.err; echo "trace s19 rc=$?"
find $O/trace_s19 -name "*kernel_stats.csv" -exec cp {} $O/bench_s19_kernel_stats.csv \;
rm -rf $O/trace_c2 $O/trace_grid3d $O/trace_s19
scripts/traffic_pmc.sh s19 --table s19 > /dev/null 2>&1
KERNEL=tile_regrid3 scripts/pmc_kernel.sh ${TAG}_s19 --table s19 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_s19.txt $O/bench_s19_pmc.txt
scripts/traffic_pmc.sh c2 > /dev/null 2>&1; scripts/traffic_pmc.sh c3 --mode paint --nside 2048 > /dev/null 2>&1
scripts/traffic_pmc.sh c4 --halos 1250000 --nside 2048 > /dev/null 2>&1
scripts/traffic_pmc.sh grid3d --mode grid3d > /dev/null 2>&1; scripts/traffic_pmc.sh snapshot --mode snapshot > /dev/null 2>&1
cp gpurun_out/traffic_*.json $O/
scripts/pmc_kernel.sh ${TAG}_c2 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_c2.txt $O/bench_c2_pmc.txt
for f in $O/*_line.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); r=d.get('roofline',{})
print('$f'.split('/')[-1], round(d['ms_per_step'],4), d.get('kernel_ms'), 'frac', round(r.get('frac',0),4), 'traffic', r.get('traffic'))"; done
