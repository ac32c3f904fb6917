#!/bin/bash
# GPU box: run bench.py once per library given (paths relative to baryonification_amd/csrc), print ms_per_step + kernel_ms
cd "$(dirname "$0")/.."
D=$PWD/baryonification_amd/csrc
LIBS="$1"; shift
for L in $LIBS; do
  echo -n "$L  "
  BFGX_LIB=$D/$L python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()}, d.get('mass_conserved'))"
done
