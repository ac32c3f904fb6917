#!/bin/bash
# GPU box: the round's bench lines + rocprofv3 kernel stats + per-mode PMC traffic files -> gpurun_out/<tag>m/ (copied to profiles/ by hand)
# usage: [QUICK=1] scripts/r05_measure.sh <tag>      (QUICK=1: the shell lines, traces and traffic files only)
TAG=${1:-r05}
O=gpurun_out/${TAG}m; mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_c2_line.json 2> $O/bench_c2.err; echo "c2 (S19 table, default precision) rc=$?"
# kernel traces of the SAME command in the three precisions of the headline table and on the closed-form table (program directly after `--`)
trace() {       # trace <name> <bench args...>
  N=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras "$@" > $O/trace_$N.json 2> $O/trace_$N.err; echo "trace $N rc=$?"
  find $O/trace_$N -name "*kernel_stats.csv" -exec cp {} $O/bench_${N}_kernel_stats.csv \;
  rm -rf $O/trace_$N
}
trace s19_parity --table s19
trace s19_f64 --table s19 --precision f64
trace s19_f32 --table s19 --precision f32
trace cf --table closed-form
trace cf_f64 --table closed-form --precision f64
# HBM traffic + VALU counts per kernel group (one counter per --pmc pass), matched by bench.py through its config keys (table, precision)
scripts/traffic_pmc.sh s19 --table s19 > /dev/null 2>&1
scripts/traffic_pmc.sh s19_f64 --table s19 --precision f64 > /dev/null 2>&1
scripts/traffic_pmc.sh s19_f32 --table s19 --precision f32 > /dev/null 2>&1
scripts/traffic_pmc.sh c2 --table closed-form > /dev/null 2>&1
scripts/traffic_pmc.sh c2_f64 --table closed-form --precision f64 > /dev/null 2>&1
cp gpurun_out/traffic_s19.json gpurun_out/traffic_s19_f64.json gpurun_out/traffic_s19_f32.json gpurun_out/traffic_c2.json gpurun_out/traffic_c2_f64.json $O/
scripts/pmc_kernel.sh ${TAG}_s19 --table s19 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_s19.txt $O/bench_s19_parity_pmc_k1.txt
KERNEL=tile_regrid3 scripts/pmc_kernel.sh ${TAG}_s19k2 --table s19 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_s19k2.txt $O/bench_s19_parity_pmc_k2.txt
scripts/pmc_kernel.sh ${TAG}_s19f64 --table s19 --precision f64 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_s19f64.txt $O/bench_s19_f64_pmc_k1.txt
if [ "$QUICK" != 1 ]; then
python3 bench.py --mode paint --nside 2048 --no-cpu-baseline > $O/bench_paint_c3_line.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python3 bench.py --halos 1250000 --nside 2048 --no-cpu-baseline --no-extras > $O/bench_c4_per_gpu_line.json 2> $O/bench_c4.err; echo "c4 share rc=$?"
python3 bench.py --mode grid3d --steps 30 --warmup 3 > $O/grid3d_bench_line.json 2> $O/grid3d.err; echo "grid3d rc=$?"
python3 bench.py --mode snapshot --steps 30 --warmup 3 --no-cpu-baseline > $O/snapshot_bench_line.json 2> $O/snapshot.err; echo "snapshot rc=$?"
BFGX_FORCE_EXCHANGE=1 python3 bench.py --steps 300 --no-extras --no-cpu-baseline > $O/rccl_single_rank_spatial_line.json 2> $O/forcex.err; echo "forcex rc=$?"
BFGX_FORCE_EXCHANGE=1 python3 bench.py --steps 300 --no-extras --no-cpu-baseline --table closed-form > $O/rccl_single_rank_spatial_cf_line.json 2> $O/forcex_cf.err; echo "forcex cf rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_grid3d -- python3 bench.py --mode grid3d --steps 10 --warmup 2 --no-cpu-baseline > $O/trace_grid3d.json 2> $O/trace_grid3d.err
find $O/trace_grid3d -name "*kernel_stats.csv" -exec cp {} $O/grid3d_kernel_stats.csv \;
rm -rf $O/trace_grid3d
scripts/traffic_pmc.sh c3 --mode paint --nside 2048 > /dev/null 2>&1
scripts/traffic_pmc.sh c4 --halos 1250000 --nside 2048 > /dev/null 2>&1
scripts/traffic_pmc.sh grid3d --mode grid3d > /dev/null 2>&1; scripts/traffic_pmc.sh snapshot --mode snapshot > /dev/null 2>&1
cp gpurun_out/traffic_*.json $O/
fi
for f in $O/*_line.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); r=d.get('roofline',{})
print('$f'.split('/')[-1], round(d['ms_per_step'],4), d.get('kernel_ms'), 'frac', round(r.get('frac',0),4), 'traffic', r.get('traffic'))"; done
