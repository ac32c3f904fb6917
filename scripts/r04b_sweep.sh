#!/bin/bash
# GPU box: the shell-path configurations under the two K1 forms (BFGX_K1_FLUID=0/1)
export STEPS=100
for F in 1 0; do
  echo "== BFGX_K1_FLUID=$F"
  BARGS="" scripts/k1_ab.sh "BFGX_K1_FLUID=$F"
  BARGS="--table s19" scripts/k1_ab.sh "BFGX_K1_FLUID=$F"
  BARGS="--mode paint --nside 2048" STEPS=60 scripts/k1_ab.sh "BFGX_K1_FLUID=$F"
  BARGS="--halos 1250000 --nside 2048" STEPS=60 scripts/k1_ab.sh "BFGX_K1_FLUID=$F"
  BARGS="--nside 4096" STEPS=20 scripts/k1_ab.sh "BFGX_K1_FLUID=$F"
  BARGS="--nside 512" scripts/k1_ab.sh "BFGX_K1_FLUID=$F"
done
