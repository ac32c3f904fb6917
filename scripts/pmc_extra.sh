#!/bin/bash
# extra PMC passes for latency diagnosis: scripts/pmc_extra.sh <tag>
TAG=${1:-x}; OUT=gpurun_out/pmcx_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline"
i=0
for PASS in "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS" \
            "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU" \
            "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > /dev/null 2> $OUT/p$i.err
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        if 'tile_scatter_kernel<0' in k or 'tile_regrid' in k:
            agg[k[:50]][row['Counter_Name']].append(float(row['Counter_Value']))
for k in agg:
    print(k)
    for c in sorted(agg[k]): print('   %-34s %16.1f'%(c, sum(agg[k][c])/len(agg[k][c])))
PY
