#!/bin/bash
# GPU box (one GPU): every multi-rank bench mode as the driver would type it, ranks sharing device 0 over gloo (timings mean nothing)
export BFGX_DIST_BACKEND=gloo GLOO_SOCKET_IFNAME=lo
run() { echo "== $*"; timeout -k 10 600 python3 bench.py "$@" 2> /tmp/reh.err | python3 -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')]
if not l: print('NO LINE'); sys.exit(1)
d=json.loads(l[0]); print(d['n_gpus'], d['scaling'], round(d['ms_per_step'],3), d.get('mass_conserved'), d['config'].get('parallelism','')[:60], 'weak:' , (d.get('value_weak') or {}).get('mass_conserved'))" || { tail -5 /tmp/reh.err; }; }
run --gpus 2 --halos 200000 --nside 256 --steps 3 --warmup 1
run --gpus 4 --halos 200000 --nside 256 --steps 3 --warmup 1
run --gpus 3 --halos 100000 --nside 128 --steps 3 --warmup 1 --mode paint
run --gpus 2 --halos 100000 --nside 256 --steps 3 --warmup 1 --exchange slices
run --gpus 2 --halos 100000 --nside 256 --steps 3 --warmup 1 --exchange reduce
run --gpus 2 --halos 100000 --nside 256 --steps 3 --warmup 1 --scaling weak
run --gpus 2 --halos 300000 --nside 256 --steps 2 --warmup 1 --config 4
run --gpus 4 --mode grid3d --ngrid 128 --grid-halos 2000 --steps 2 --warmup 1
run --gpus 2 --halos 100000 --nside 256 --steps 3 --warmup 1 --table s19
