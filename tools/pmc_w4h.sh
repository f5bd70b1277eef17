#!/bin/bash
# Counters of the fp16-pair component GEMM (k_w4_gemm64h) at the cfg-2 shape, product library: one counter group per pass.
#   usage: tools/pmc_w4h.sh <out.txt> [kernel substring] [N,C,side]
OUT=${1:-gpurun_out/pmc_w4h.txt}
KERN=${2:-k_w4_gemm64h}
SHAPE=${3:-128,256,8}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "TA_BUSY_avr" "TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/ph_$i
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/ph_$i -- python3 $R/tools/w4_time.py 6 $SHAPE > /tmp/ph_$i.log 2>&1 || echo "group '$grp' failed: $(grep -iE 'error|invalid|not' /tmp/ph_$i.log | head -1)" >> $R/$OUT
  python3 - /tmp/ph_$i $KERN >> $R/$OUT <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            a = acc[r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
for k, (s, n) in acc.items():
    print('%-36s %16.1f per launch (%d launches)' % (k, s / n, n))
PY
  i=$((i + 1))
done
cat $R/$OUT
