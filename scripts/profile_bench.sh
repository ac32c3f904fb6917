#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then PMC passes.
# Usage: scripts/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $BENCH_ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?"
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$N -- python3 bench.py $BENCH_ARGS > $OUT/bench_pmc_$N.json 2> $OUT/bench_pmc_$N.err
  echo "pmc $N rc=$?"
done
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
