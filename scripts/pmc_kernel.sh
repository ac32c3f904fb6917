#!/bin/bash
# GPU box: rocprofv3 --pmc passes over a short bench.py run, per-launch averages of the counters for the kernels whose name contains
# $KERNEL (default tile_scatter2).   scripts/pmc_kernel.sh <tag> [bench args...]  -> gpurun_out/pmc_<tag>.txt
set -o pipefail
TAG=${1:-x}; shift
KERNEL=${KERNEL:-tile_scatter2}
OUT=/tmp/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events $@"
i=0
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS" \
            ${EXTRA_PASS:+"$EXTRA_PASS"}; do
  i=$((i+1))
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
done
python3 - "$OUT" "$KERNEL" <<'PY' | tee gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row['Kernel_Name']:
            name = row['Kernel_Name'].split('(')[0][-70:]
            a = acc[(name, row['Counter_Name'])]
            a[0] += float(row['Counter_Value']); a[1] += 1
byk = collections.defaultdict(dict)
for (k, c), (v, n) in acc.items():
    byk[k][c] = (v / n, n)
for k, d in byk.items():
    print(k)
    for c in sorted(d):
        print("    %-28s %16.1f   (n=%d)" % (c, d[c][0], d[c][1]))
PY
