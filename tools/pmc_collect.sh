#!/bin/bash
# Collect the per-kernel PMC counters bench.py's `roofline.traffic` is taken from: separate rocprofv3 --pmc passes
# (one counter group per run, never combined with a trace domain) over tools/prof_eval.py at the cfg-2 state
# shape, aggregated per kernel by tools/pmc_agg.py.      usage: tools/pmc_collect.sh <out.json> [shape] [iters]
set -e
OUT=${1:-gpurun_out/pmc_eval.json}
SHAPE=${2:-128,256,8,8}
ITERS=${3:-10}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -- python3 $R/tools/prof_eval.py --shape $SHAPE --iters $ITERS > /tmp/pmc_$i.log 2>&1 || { tail -5 /tmp/pmc_$i.log; exit 1; }
  i=$((i + 1))
done
cd $R
python3 tools/pmc_agg.py $OUT /tmp/pmc_0 /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3
