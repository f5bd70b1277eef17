#!/bin/bash
# What bounds the GroupNorm / transform passes (k_w4s_pass)?  Texture-path, L1 and wave-state counters per pass instance over augmented
# evaluations at the cfg-2 shape (tools/prof_eval.py).  One counter group per rocprofv3 pass.
#   usage: tools/pmc_pass_limiter.sh <out.txt> [N,C,H,W]
OUT=${1:-gpurun_out/pmc_pass_limiter.txt}
SHAPE=${2:-128,256,8,8}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
export NODE_TUNE_WINO4=2
: > $R/$OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE" "TA_BUSY_avr" "TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_sum" "TCP_TCR_TCP_STALL_CYCLES_sum" "TD_TD_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pp_$i
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/pp_$i -- python3 $R/tools/prof_eval.py --shape $SHAPE --iters 4 > /tmp/pp_$i.log 2>&1 || echo "group '$grp' failed: $(grep -iE 'error|invalid' /tmp/pp_$i.log | head -1)" >> $R/$OUT
  python3 - /tmp/pp_$i >> $R/$OUT <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_w4s_pass' in k or 'k_w4_gemm64b' in k or 'k_w4_wgrad' in k:
            a = acc[(k, r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print('%-34s %-36s %16.1f per launch (%d launches)' % (k.replace('node::', ''), c, s / n, n))
PY
  i=$((i + 1))
done
cat $R/$OUT
