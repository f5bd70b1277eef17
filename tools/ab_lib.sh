#!/bin/bash
# A/B of two BUILDS of the library on one box: tools/ab_lib.sh <out.txt> <config> libA.so libB.so [rounds]  (paths relative to csrc/).
# Each round copies a build over libnode_hip.so and runs bench.py (no roofline pass); the installed build is restored at the end.
OUT=$1; CFG=$2; A=$3; B=$4; ROUNDS=${5:-3}
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/neural-ode-features_amd/csrc
cp $L/libnode_hip.so /tmp/libnode_hip.keep
: > $R/$OUT
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    cp $L/$v $L/libnode_hip.so
    python $R/bench.py --config $CFG --steps 30 --warmup 10 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %9.1f images/s  %.3f ms/step  retries %s' % ('$v', d['value'], d['ms_per_step'], d['config']['retries']))" >> $R/$OUT
  done
done
cp /tmp/libnode_hip.keep $L/libnode_hip.so
cat $R/$OUT
