#!/bin/bash
# A/B of k_w4_gemm64b switches at the cfg-2 shape under rocprofv3 (mean launch duration):  tools/w4_ab.sh <out.txt> "ENV=.. ENV=.." ...
OUT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
for envs in "$@"; do
  for rep in 1 2; do
    rm -rf /tmp/wab
    env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wab -- python3 $R/tools/w4_time.py ${ITER:-60} 128,256,8 > /tmp/wab.log 2>&1 || { echo "$envs failed"; tail -3 /tmp/wab.log; }
    KS=$(find /tmp/wab -name '*kernel_stats.csv' | head -1)
    python3 - "$KS" "$envs" >> $R/$OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_w4_gemm' in r['Name']:
        print('%-40s %-28s %7.2f us  (min %.2f, max %.2f, %s launches)' % (sys.argv[2], r['Name'][:28], float(r['AverageNs']) / 1e3,
              float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Calls']))
PY
  done
done
cat $R/$OUT
