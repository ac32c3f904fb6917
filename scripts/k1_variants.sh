#!/bin/bash
# K1 experiment builds side by side (GPU box): scripts/k1_variants.sh <lib suffixes...>   e.g.  scripts/k1_variants.sh base w10
# every variant is baryonification_amd/csrc/variants/libbfgx_<suffix>.so ("base" = the shipped libbfgx.so); prints ms_per_step and kernel_ms
cd "$(dirname "$0")/.."
D=$PWD/baryonification_amd/csrc
for V in "$@"; do
  L=$D/variants/libbfgx_$V.so; [ $V = base ] && L=$D/libbfgx.so
  echo -n "$V  "
  BFGX_LIB=$L python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()}, d['mass_conserved'])"
done
