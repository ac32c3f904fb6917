#!/bin/bash
# GPU box: A/B of K1 variants selected by environment.  usage: scripts/k1_ab.sh "ENV1=.. ENV2=.." "ENV.." ...   (bench args in $BARGS)
for V in "$@"; do
  env $V timeout -k 10 120 python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-extras $BARGS 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$V]', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()}, d['mass_conserved'])
"
done
