#!/bin/bash
# Timing experiment (results wrong): k_w4_gemm64b with its operand blocks pulled out of their power-of-two spacing
# (NODE_TUNE_W4_PAD = "v,u" KB per block), full kernel (ablate 16) and requests only (18), at the cfg-2 shape.
#   usage: tools/w4_pad.sh <out.txt>
OUT=${1:-gpurun_out/w4_pad.txt}
SHAPE=${2:-128,256,8}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
export NODE_HIP_DIAG=1    # the ablations live in libnode_hip_diag.so (build.py --diag), not in the product library
: > $R/$OUT
for pad in ${PADS:-0,0 1,0 0,1 1,1 2,3 3,5 5,7 7,11}; do
  for ab in ${ABS:-16 18}; do
    rm -rf /tmp/wp
    NODE_TUNE_W4_PAD=$pad NODE_TUNE_W4_ABLATE=$ab rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp -- python3 $R/tools/w4_time.py 40 $SHAPE > /tmp/wp.log 2>&1 || { echo "pad $pad ablate $ab failed"; tail -3 /tmp/wp.log; }
    KS=$(find /tmp/wp -name '*kernel_stats.csv' | head -1)
    python3 - "$KS" $ab $pad >> $R/$OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_w4_gemm64b' in r['Name']:
        print('pad %-5s ablate %s  %7.2f us  (min %.2f, max %.2f, %s launches)' % (sys.argv[3], sys.argv[2], float(r['AverageNs']) / 1e3,
              float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Calls']))
PY
  done
done
cat $R/$OUT
