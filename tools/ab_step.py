#!/usr/bin/env python
"""Same-process A/B of training-step variants at cfg 2 (box-to-box and run-to-run spread exceeds most single
changes, so variants are timed in alternating blocks inside one process; medians of the per-step HIP-event times).

    python tools/ab_step.py eager graph_head graph_stem graph_both
"""
import copy
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402


def build(variant, x):
    cfg = dict(bench.CONFIGS[2])
    model = bench.build_model(torch.device('cuda', 0), cfg, 'dopri5')
    model.train()
    if variant == 'graph_head':
        nof.graphs.capture_static_parts(model, x, stem=False, head=True)
    elif variant == 'graph_stem':
        nof.graphs.capture_static_parts(model, x, stem=True, head=False)
    elif variant == 'graph_both':
        nof.graphs.capture_static_parts(model, x, stem=True, head=True)
    opt = nof.FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    return model, opt


def main():
    variants = sys.argv[1:] or ['eager', 'graph_head', 'graph_stem', 'graph_both']
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(128, 3, 32, 32, generator=gen).cuda()
    y = torch.randint(0, 10, (128,), generator=gen).cuda()
    built = {v: build(v, x) for v in variants}
    times = {v: [] for v in variants}
    for v in variants:
        for _ in range(5):
            bench.train_step(*built[v], x, y)
    for rnd in range(4):
        for v in variants:
            model, opt = built[v]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
            torch.cuda.synchronize()
            ev[0].record()
            for i in range(15):
                bench.train_step(model, opt, x, y)
                ev[i + 1].record()
            torch.cuda.synchronize()
            times[v] += [ev[i].elapsed_time(ev[i + 1]) for i in range(15)]
    for v in variants:
        t = times[v]
        print('%-12s median %.3f ms  min %.3f  mean %.3f  (%d steps)' % (v, statistics.median(t), min(t), statistics.mean(t), len(t)))


if __name__ == '__main__':
    main()
