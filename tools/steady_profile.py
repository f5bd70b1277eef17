#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state tail of a bench.py run
(skips MIOpen's first-use search kernels that pollute --stats on a fresh box).

    python tools/steady_profile.py <kernel_trace.csv> [n_steps_tail=5] [steps_total=12]
"""
import csv, sys, collections
f = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 5
total = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
conv = [i for i, r in enumerate(rows) if 'k_conv3x3' in r['Kernel_Name']]
# one k_sgd_multi launch per training step (older traces: fall back to the step count given)
nsgd = sum(1 for r in rows if 'k_sgd_multi' in r['Kernel_Name'])
per_step = len(conv) // (nsgd if nsgd else total)
start = conv[len(conv) - tail * per_step]
# back up to the beginning of that step: first kernel after the previous step's last node:: kernel
win = rows[start:]
agg = collections.OrderedDict()
for r in win:
    n = r['Kernel_Name']
    n = n.split('(')[0][:70]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += d
tot = sum(v[1] for v in agg.values())
span = int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])
print('steady-state window: %d steps, %d dispatches, kernel time %.3f ms/step, wall span %.3f ms/step (GPU busy %.1f%%)'
      % (tail, len(win), tot / 1e6 / tail, span / 1e6 / tail, 100.0 * tot / span))
print('%-72s %8s %10s %9s %6s' % ('kernel', 'calls/st', 'avg_us', 'ms/step', '%'))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print('%-72s %8.1f %10.2f %9.3f %6.2f' % (n, c / tail, d / c / 1e3, d / 1e6 / tail, 100.0 * d / tot))
