#!/bin/bash
# GPU box: A/B of config 5 (bench.py --mode grid3d) variants selected by environment.  usage: scripts/grid_ab.sh "ENV=.." ...   (bench args in $BARGS)
for V in "$@"; do
  env $V timeout -k 10 200 python3 bench.py --mode ${MODE:-grid3d} --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline $BARGS 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$V]', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})
"
done
