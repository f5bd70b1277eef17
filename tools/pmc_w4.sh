#!/bin/bash
# PMC passes over tools/w4_time.py (the F(4x4,3x3) diagnostic convolution; default: the cfg-2 shape): MFMA duty and clock
# of the component GEMM, its HBM / L2 traffic, LDS bank conflicts.  One counter group per run.
#   usage: tools/pmc_w4.sh <out.json> [N,C,side]          (cfg 5: 64,1024,16)
OUT=${1:-gpurun_out/pmc_w4.json}
SHAPE=${2:-128,256,8}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
dirs=""
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  rm -rf /tmp/pw_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pw_$i -- python3 $R/tools/w4_time.py 10 $SHAPE > /tmp/pw_$i.log 2>&1 || { echo "group '$grp' failed"; tail -3 /tmp/pw_$i.log; }
  dirs="$dirs /tmp/pw_$i"
  i=$((i + 1))
done
cd $R
python3 tools/pmc_agg.py $OUT $dirs
python3 -c "
import json
d = json.load(open('$OUT'))
for k, v in d.items():
    print(k, {a: round(b, 1) for a, b in v.items()})
"
