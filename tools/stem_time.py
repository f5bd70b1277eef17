#!/usr/bin/env python
"""Per-launch durations of the residual stem's kernels (forward + backward at the cfg-2 batch), in launch order.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 tools/stem_time.py --run
    python3 tools/stem_time.py --report /tmp/st        (prints the last iteration's launches)
"""
import argparse
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n, filters, iters):
    import torch
    import neural_ode_features_amd as nof
    torch.manual_seed(0)
    net = nof.ODENet(3, out=10, n_filters=filters, downsample='residual', adjoint=True).cuda()
    stem = net.downsample.module
    x = torch.randn(n, 3, 32, 32, device='cuda')
    for _ in range(iters):
        out = stem(x)
        out.square().mean().backward()
    torch.cuda.synchronize()


def report(d, per_iter):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
    rows.sort()
    rows = [r for r in rows if 'node::' in r[2]]
    idx = max(i for i, r in enumerate(rows) if 'k_stem_prep' in r[2])     # the last iteration starts at its preparation launch
    last = rows[idx:]
    tot = 0.0
    for s, e, name in last:
        short = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('node::', '')
        print('%-28s %8.1f us' % (short, (e - s) / 1e3))
        tot += (e - s) / 1e3
    print('total %.1f us over %d launches; span %.1f us' % (tot, len(last), (last[-1][1] - last[0][0]) / 1e3))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--run', action='store_true')
    ap.add_argument('--report', default=None)
    ap.add_argument('--n', type=int, default=128)
    ap.add_argument('--filters', type=int, default=256)
    ap.add_argument('--iters', type=int, default=6)
    ap.add_argument('--per-iter', type=int, default=36)
    a = ap.parse_args()
    if a.run:
        run(a.n, a.filters, a.iters)
    if a.report:
        report(a.report, a.per_iter)
