#!/bin/bash
# Review item 1b (round 5): what bounds k_w4_gemm64b?  Texture-path / L1 / L2-request counters beside the MFMA duty counter, for
# the full kernel and for the two ablations that separate the operand requests from the matrix work (diagnostics library:
# NODE_TUNE_W4_ABLATE = 16 full, 18 requests only, 24 split + MFMA + stores without requests).  One counter group per pass.
#   usage: tools/pmc_w4_limiter.sh <out.txt> [N,C,side]
OUT=${1:-gpurun_out/pmc_w4_limiter.txt}
SHAPE=${2:-128,256,8}
R=$(cd "$(dirname "$0")/.." && pwd)
export NODE_HIP_DIAG=1
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_TA_BUSY|TA_BUSY|TCP_PENDING_STALL_CYCLES|TCP_TCC_READ_REQ|TCP_TA_TCP_STATE_READ|TA_ADDR_STALLED_BY_TC_CYCLES|TA_DATA_STALLED_BY_TC_CYCLES|TCP_GATE_EN1|TCP_GATE_EN2|TCP_TCC_READ_REQ_LATENCY|TCP_READ_TAGCONFLICT_STALL_CYCLES|TCC_REQ|TCC_EA0_RDREQ|TCP_TD_TCP_STALL_CYCLES|TD_TD_BUSY|TCP_TCR_TCP_STALL_CYCLES)[A-Za-z_0-9\[\]]*" | sort -u | tr '\n' ' ' >> $R/$OUT
echo >> $R/$OUT
for ab in 16 18 24; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum" "TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum" "TCP_GATE_EN2_sum" "TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCR_TCP_STALL_CYCLES_sum" "TD_TD_BUSY_sum"; do
    rm -rf /tmp/pl_$i
    NODE_TUNE_W4_ABLATE=$ab timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pl_$i -- python3 $R/tools/w4_time.py 6 $SHAPE > /tmp/pl_$i.log 2>&1 || echo "ablate $ab group '$grp' failed: $(grep -iE 'error|invalid|not' /tmp/pl_$i.log | head -1)" >> $R/$OUT
    python3 - /tmp/pl_$i $ab >> $R/$OUT <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_w4_gemm64b' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
for k, (s, n) in acc.items():
    print('ablate %s  %-36s %16.1f per launch (%d launches)' % (sys.argv[2], k, s / n, n))
PY
    i=$((i + 1))
  done
done
cat $R/$OUT
