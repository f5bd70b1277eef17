#!/bin/bash
# A/B of environment switches on one box at the training-step level: tools/ab_env.sh <out.txt> <config> <rounds> "ENV=.. ENV=.." ...
# ("-" = no switch).  Alternating rounds of bench.py (no roofline pass); images/s and ms/step per run.
OUT=$1; CFG=$2; ROUNDS=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
: > $R/$OUT
for r in $(seq $ROUNDS); do
  for envs in "$@"; do
    e=$envs; [ "$e" = "-" ] && e="NODE_AB_NOTHING=1"
    env $e python $R/bench.py --config $CFG --steps 30 --warmup 10 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %9.1f images/s  %.3f ms/step  retries %s  dead steps per step %s  fresh %.0f' % ('$envs', d['value'], d['ms_per_step'], d['config']['retries'], d['config']['dead_steps_per_step'], (d.get('fresh_batches') or {}).get('value', 0)))" >> $R/$OUT
  done
done
cat $R/$OUT
