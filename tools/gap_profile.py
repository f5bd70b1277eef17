#!/usr/bin/env python
"""Where the GPU idles: gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV over the
steady-state tail of a bench.py run, attributed to the (previous kernel -> next kernel) pair.

    python tools/gap_profile.py <kernel_trace.csv> [n_steps_tail=5] [steps_total=15]
"""
import csv, sys, collections
f = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 5
total = int(sys.argv[3]) if len(sys.argv) > 3 else 15
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
conv = [i for i, r in enumerate(rows) if 'k_conv3x3' in r['Kernel_Name']]
# one k_sgd_multi launch per training step (older traces: fall back to the step count given)
nsgd = sum(1 for r in rows if 'k_sgd_multi' in r['Kernel_Name'])
per_step = len(conv) // (nsgd if nsgd else total)
win = rows[conv[len(conv) - tail * per_step]:]
short = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
gaps = collections.defaultdict(lambda: [0, 0])
idle = 0
end = int(win[0]['End_Timestamp'])
for a, b in zip(win, win[1:]):
    g = int(b['Start_Timestamp']) - end
    end = max(end, int(b['End_Timestamp']))
    if g > 0:
        idle += g
        k = gaps[(short(a), short(b))]
        k[0] += 1
        k[1] += g
span = int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])
print('window: %d steps, span %.3f ms/step, idle %.3f ms/step (%.1f%%)' % (tail, span / 1e6 / tail, idle / 1e6 / tail, 100.0 * idle / span))
print('%-46s %-46s %8s %9s %9s' % ('after', 'before', 'gaps/st', 'avg_us', 'ms/step'))
for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%-46s %-46s %8.1f %9.2f %9.3f' % (a, b, c / tail, g / c / 1e3, g / 1e6 / tail))
