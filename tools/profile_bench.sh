#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py + steady-state / idle-gap summaries.
#   usage: tools/profile_bench.sh <out_prefix> [bench args...]      (run from the repo root on the GPU box)
# Writes <out_prefix>_kernel_stats.csv, _steady.txt, _gaps.txt, _bench.json next to each other.
set -e
OUT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
STEPS=${STEPS:-12}; WARM=${WARM:-3}
mkdir -p "$(dirname "$OUT")"
D=/tmp/prof_$$
rm -rf $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-other-configs --no-latency "$@" > $D.log 2>&1 || { tail -20 $D.log; exit 1; }
cd $R
grep '"metric"' $D.log > ${OUT}_bench.json || true
# (bench.py starts child processes -- the held-clock measurement on the diagnostics library -- and rocprofv3 writes one set of files per
#  process: the bench's own is the largest)
KS=$(ls -S $(find $D -name '*kernel_stats.csv') | head -1)
KT=$(ls -S $(find $D -name '*kernel_trace.csv') | head -1)
cp $KS ${OUT}_kernel_stats.csv
# bench runs WARM + settle_steps + STEPS (x timed_region, if a region was measured again) + 2 + STEPS (drop-in region) + 1 + min(STEPS,5) roofline-repeat steps
REGION=$(python3 -c "import json,sys; print(json.loads(open('${OUT}_bench.json').read().strip().splitlines()[-1])['config'].get('timed_region', 1))")
SETTLE=$(python3 -c "import json,sys; print(json.loads(open('${OUT}_bench.json').read().strip().splitlines()[-1])['config'].get('settle_steps', 0))")
python3 tools/step_profile.py $KT --warmup $((WARM + SETTLE + (REGION - 1) * STEPS)) --steps $STEPS --repeat 5 --gaps --top ${TOP:-60} ${DUMP_WINDOW:+--dump ${OUT}_window.csv} > ${OUT}_steps.txt
cat ${OUT}_steps.txt
