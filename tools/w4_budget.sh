#!/bin/bash
# A measured budget of k_w4_gemm64b at the cfg-2 shape: the kernel's duration (rocprofv3 --kernel-trace, mean over the
# launches of tools/w4_time.py) with parts of it ablated (NODE_TUNE_W4_ABLATE = 16 + bits: 1 no shared component, 2 operand
# requests only (no split, no MFMA), 4 no stores, 8 no operand requests (split + MFMA on register contents)).
#   usage: tools/w4_budget.sh <out.txt> [N,C,side]
OUT=${1:-gpurun_out/w4_budget.txt}
SHAPE=${2:-128,256,8}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
export NODE_HIP_DIAG=1    # the ablations live in libnode_hip_diag.so (build.py --diag), not in the product library
: > $R/$OUT
for ab in 0 17 18 20 22 24 28 30 21 25; do
  rm -rf /tmp/wb_$ab
  NODE_TUNE_W4_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wb_$ab -- python3 $R/tools/w4_time.py 40 $SHAPE > /tmp/wb_$ab.log 2>&1 || { echo "ablate $ab failed"; tail -3 /tmp/wb_$ab.log; }
  KS=$(find /tmp/wb_$ab -name '*kernel_stats.csv' | head -1)
  python3 - "$KS" $ab >> $R/$OUT <<'PY'
import csv, sys
names = {0: 'full kernel', 17: 'no shared component', 18: 'operand requests only (no split / MFMA)', 20: 'no stores',
         22: 'operand requests only, no stores', 24: 'no operand requests (split + MFMA + stores)', 28: 'split + MFMA only (no requests, no stores)',
         30: 'bits 2|4|8: bit 8 wins in w4b_run (kernels_w4.hip) -- split + MFMA only again, i.e. the same as 28, NOT launch + prologue', 21: 'no shared component, no stores',
         25: 'no shared component, no requests'}
ab = int(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_w4_gemm64b' in r['Name']:
        print('ablate %2d  %-75s %7.2f us  (min %.2f, max %.2f, %s launches)' % (ab, names.get(ab, '?'), float(r['AverageNs']) / 1e3,
              float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Calls']))
PY
done
cat $R/$OUT
