#!/usr/bin/env python
"""Per-step summary of a rocprofv3 --kernel-trace CSV of a `bench.py` run, cut into training steps at the
`k_sgd_multi` launches (one per step) so that each phase of the run is reported on its own:

  * the TIMED window -- steps [warmup, warmup + steps): deferred completion, what `value` is measured on;
  * the ROOFLINE window -- the last `repeat` steps: a read-back per solve, the launches `roofline.avg_launch_us`
    averages with HIP events (bench.py repeats the steps in that mode so that no launch is an early-exit one).

Under deferred completion the host enqueues a guessed number of solver steps; launches behind the solve's end
return at their first instruction (`if (ctrl->done) return;`: 1 - 5 us to dispatch and retire).  They are counted apart ("dead") and left
out of the live average, which is the figure to hold against `roofline.avg_launch_us`.

    python tools/step_profile.py <kernel_trace.csv> --warmup 3 --steps 12 --repeat 5 [--gaps]
"""
import argparse
import collections
import csv
import statistics


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        rows.append((s, e, r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:64]))
    rows.sort()
    return rows


def cut_steps(rows):
    """[[row, ...] per training step]: a step ends with its k_sgd_multi launch."""
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if 'k_sgd_multi' in r[2]:
            steps.append(cur)
            cur = []
    return steps, cur


def dead_threshold(rows):
    """name -> duration under which a launch of an early-exit kernel counts as dead: 0.35 x the median of the launches
    above 2 us (a dead launch of a 256-workgroup kernel still takes 2 - 5 us to dispatch and retire); only for node::
    kernels whose live launches are long enough to tell apart."""
    by = collections.defaultdict(list)
    for s, e, n in rows:
        if n.startswith('node::'):
            by[n].append(e - s)
    thr = {}
    for n, d in by.items():
        live = [x for x in d if x > 2000]
        if live and statistics.median(live) > 8000:
            thr[n] = max(2000, int(0.35 * statistics.median(live)))
    return thr


def report(title, steps, thr, show_gaps):
    if not steps:
        print('%s: no steps' % title)
        return
    k = len(steps)
    flat = [r for st in steps for r in st]
    agg = collections.OrderedDict()
    for s, e, n in flat:
        a = agg.setdefault(n, [0, 0, 0])      # live calls, live ns, dead calls
        if n in thr and e - s < thr[n]:
            a[2] += 1
        else:
            a[0] += 1
            a[1] += e - s
    busy = 0
    idle = 0
    gaps = collections.defaultdict(lambda: [0, 0])
    end = flat[0][1]
    busy += flat[0][1] - flat[0][0]
    for a, b in zip(flat, flat[1:]):
        g = b[0] - end
        if g > 0:
            idle += g
            q = gaps[(a[2][:40], b[2][:40])]
            q[0] += 1
            q[1] += g
        end = max(end, b[1])
    span = end - flat[0][0]
    tot = sum(v[1] for v in agg.values())
    print('%s: %d steps, %.1f dispatches/step, span %.3f ms/step, kernel time %.3f ms/step, idle %.3f ms/step (GPU busy %.1f%%)'
          % (title, k, len(flat) / k, span / 1e6 / k, tot / 1e6 / k, idle / 1e6 / k, 100.0 * (span - idle) / span))
    print('  %-64s %8s %8s %10s %9s %6s' % ('kernel', 'live/st', 'dead/st', 'live avg us', 'ms/step', '%'))
    for n, (c, d, dead) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:TOP]:
        if c:
            print('  %-64s %8.1f %8.1f %10.2f %9.3f %6.2f' % (n, c / k, dead / k, d / c / 1e3, d / 1e6 / k, 100.0 * d / tot))
    edges = [2, 5, 10, 20, 40, 60]
    for name in ('node::k_conv3x3_w2', 'node::k_w4_wgrad<8>', 'node::k_w4_wgrad128b', 'node::k_w4s_pass<1, 0, 1>', 'node::k_w4s_pass<1, 0, 4>'):
        d = [(e - s) / 1e3 for s, e, n in flat if n == name]
        if d:
            hist = [sum(1 for x in d if lo <= x < hi) for lo, hi in zip([0] + edges, edges + [1e9])]
            print('  %-24s launches/step by duration [<2, 2-5, 5-10, 10-20, 20-40, 40-60, >60 us]: %s'
                  % (name, ' '.join('%.1f' % (h / k) for h in hist)))
    if show_gaps:
        print('  idle by (after -> before) pair:')
        for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
            print('    %-40s -> %-40s %7.1f/st %8.2f us %8.3f ms/step' % (a, b, c / k, g / c / 1e3, g / 1e6 / k))


TOP = 22


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--repeat', type=int, default=5)
    ap.add_argument('--gaps', action='store_true')
    ap.add_argument('--top', type=int, default=22, help='kernels listed per window')
    ap.add_argument('--dump', default=None, help='write the timed window as a compact CSV (start_ns,dur_ns,name)')
    a = ap.parse_args()
    global TOP
    TOP = a.top
    rows = load(a.trace)
    steps, rest = cut_steps(rows)
    thr = dead_threshold(rows)
    print('%d dispatches, %d training steps (k_sgd_multi launches), %d dispatches behind the last one'
          % (len(rows), len(steps), len(rest)))
    report('TIMED window (deferred completion)', steps[a.warmup:a.warmup + a.steps], thr, a.gaps)
    if a.dump:
        win = [r for st in steps[a.warmup:a.warmup + a.steps] for r in st]
        t0 = win[0][0]
        with open(a.dump, 'w') as f:
            for s, e, n in win:
                f.write('%d,%d,%s\n' % (s - t0, e - s, n))
    print()
    report('ROOFLINE window (read-back per solve)', steps[-a.repeat:] if a.repeat else [], thr, a.gaps)


if __name__ == '__main__':
    main()
