#!/bin/bash
# Timing-only ablation builds of the fast tile kernel (never shipped): build/libbfgx_abl<N>.so with -DBFGX_ABL2=N
#   1: no chunk loop (zero-fill + ring table + flush only)   2: + entry phase   3: + ring-row phase   4: + row compaction   0: full
# Usage (build container): scripts/ablate2.sh build ; (GPU box) scripts/ablate2.sh run [bench args]
cd "$(dirname "$0")/.."
D=baryonification_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $D/build
  for N in ${ABL_LIST:-1 2 3 4}; do
    hipcc -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -DBFGX_ABL2=$N ${ABL_EXTRA} \
      -o $D/build/libbfgx_abl$N.so $D/bfgx_api.hip &
  done
  wait
  ls -la $D/build/libbfgx_abl*.so
else
  shift
  for N in 0 ${ABL_LIST:-1 2 3 4}; do
    L=$PWD/$D/build/libbfgx_abl$N.so; [ $N = 0 ] && L=$PWD/$D/libbfgx.so
    echo -n "ABL2=$N  "
    BFGX_BENCH_NOSTATUS=1 BFGX_LIB=$L python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'])"
  done
fi
