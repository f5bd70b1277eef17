#!/bin/bash
# Kernel trace of the bs = 1 latency solve (tools/latency_bs1.py): per-kernel mean durations and the launch sequence of one solve.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-gpurun_out/latency_trace.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -- python3 $R/tools/latency_bs1.py --tols 1e-3 --iters 20 > /tmp/lt.log 2>&1
KT=$(find /tmp/lt -name '*kernel_trace.csv' | head -1)
python3 - "$KT" > $R/$OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last solve: the dispatches after the last k_set_ctrl-like start marker: take the last 140 dispatches
tail = rows[-130:]
t0 = int(tail[0]['Start_Timestamp'])
prev_end = None
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%8.2f us  dur %6.2f  gap %5.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'][:60]))
    prev_end = e
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print()
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-60s n %5d  mean %6.2f us  total %8.1f us' % (k, len(v), sum(v) / len(v), sum(v)))
PY
tail -40 $R/$OUT
