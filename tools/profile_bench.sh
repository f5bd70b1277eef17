#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py + steady-state / idle-gap summaries.
#   usage: tools/profile_bench.sh <out_prefix> [bench args...]      (run from the repo root on the GPU box)
# Writes <out_prefix>_kernel_stats.csv, _steady.txt, _gaps.txt, _bench.json next to each other.
set -e
OUT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
STEPS=12; WARM=3
mkdir -p "$(dirname "$OUT")"
D=/tmp/prof_$$
rm -rf $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline "$@" > $D.log 2>&1 || { tail -20 $D.log; exit 1; }
cd $R
grep '"metric"' $D.log > ${OUT}_bench.json || true
KS=$(find $D -name '*kernel_stats.csv' | head -1)
KT=$(find $D -name '*kernel_trace.csv' | head -1)
cp $KS ${OUT}_kernel_stats.csv
# bench runs WARM + STEPS + min(STEPS,5) profiled-repeat steps
TOTAL=$((STEPS + WARM + 5))
python3 tools/steady_profile.py $KT 5 $TOTAL > ${OUT}_steady.txt
python3 tools/gap_profile.py $KT 5 $TOTAL > ${OUT}_gaps.txt
head -30 ${OUT}_steady.txt
head -16 ${OUT}_gaps.txt
