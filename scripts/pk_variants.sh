#!/bin/bash
# GPU box: scripts/pk_variants.sh <suffix...>  -- P(k) call time and per-kernel times (rocprofv3) for library variants build/libbfgx_<suffix>.so
cd "$(dirname "$0")/.."
D=$PWD/baryonification_amd/csrc
export TMPDIR=/tmp
for V in "$@"; do
  L=$D/build/libbfgx_$V.so; [ $V = base ] && L=$D/libbfgx.so
  echo "== $V"
  BFGX_LIB=$L python3 scripts/pk_time.py ${PKN:-512} 20 2>&1 | tail -1
  rm -rf /tmp/pkv_$V
  BFGX_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pkv_$V -- python3 scripts/pk_time.py ${PKN:-512} 10 > /dev/null 2>&1
  python3 - /tmp/pkv_$V <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fft' in r['Name'] or 'pk_' in r['Name']:
            print('   %-60s calls %4s  avg %.1f us' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
