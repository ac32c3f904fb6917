export BFGX_DIST_BACKEND=gloo GLOO_SOCKET_IFNAME=lo BFGX_BENCH_CHECK=1
for N in 4; do      # (6 ranks + the launcher agent = 7 processes on the card: over the box limit of 6)
  echo "== --gpus $N (full config 2, default flags)"
  timeout -k 10 500 python3 bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/full_reh_$N.err > gpurun_out/full_reh_$N.json || { tail -5 gpurun_out/full_reh_$N.err; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open('gpurun_out/full_reh_$N.json') if l.startswith('{')][0])
print(d['n_gpus'], d['scaling'], round(d['ms_per_step'],3), d.get('mass_conserved'), d.get('check'), 'weak', {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('value_weak') or {}).items() if k in ('ms_per_step','mass_conserved','halos_total')})
PY
done
