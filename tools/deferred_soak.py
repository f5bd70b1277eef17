#!/usr/bin/env python
"""How often does deferred completion miss when the data changes every step?  N optimizer steps of the cfg-2 / cfg-3 model on a
FRESH synthetic batch per step (class-dependent means, so the loss falls and the weights move), with deferred
completion (integrate.DeferredLoop: a missed batch is repeated, never skipped) and with a read-back per solve; reports
throughput, miss events, batches repeated, dead steps and the loss trajectory -- every one of the N updates is committed in
both modes, so the two trajectories are the same.

    python tools/deferred_soak.py [--steps 300] [--lr 0.05]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402
from neural_ode_features_amd import integrate  # noqa: E402


def run(steps, lr, deferred_on, seed=0, config=2):
    dev = torch.device('cuda', 0)
    cfg = dict(bench.CONFIGS[config])
    model = bench.build_model(dev, cfg, 'dopri5')
    model.train()
    opt = nof.FusedSGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator(device='cuda').manual_seed(seed)
    means = torch.randn(10, 3, 1, 1, device=dev, generator=gen)
    d = integrate.Deferred(dev) if deferred_on else None
    loop = integrate.DeferredLoop(d, opt, lambda xx, yy: bench.train_step(model, opt, xx, yy)) if d is not None else None
    losses = []

    def batch():
        y = torch.randint(0, 10, (128,), device=dev, generator=gen)
        return torch.randn(128, 3, 32, 32, device=dev, generator=gen) + means[y], y

    def step(x, y):
        done = loop.step(x, y) if loop is not None else [bench.train_step(model, opt, x, y)]
        losses.extend(r[0].detach() for r in done)

    for i in range(10):              # first-use warm-up
        step(*batch())
    if loop is not None:
        losses.extend(r[0].detach() for r in loop.flush())
    torch.cuda.synchronize()
    losses.clear()
    c0 = (loop.retries, loop.miss_events, d.dead_steps, d.blind_solves) if loop is not None else (0, 0, 0, 0)
    t0 = time.perf_counter()
    for i in range(steps):
        step(*batch())
    if loop is not None:
        losses.extend(r[0].detach() for r in loop.flush())
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c1 = (loop.retries, loop.miss_events, d.dead_steps, d.blind_solves) if loop is not None else (0, 0, 0, 0)
    ls = torch.stack(losses).float().cpu()
    finite = torch.isfinite(ls)
    return dict(mode='deferred (DeferredLoop: misses are repeated)' if deferred_on else 'read-back', steps=steps, updates=len(losses),
                images_per_s=steps * 128 / wall, miss_events=c1[1] - c0[1], batches_repeated=c1[0] - c0[0],
                dead_steps_per_step=(c1[2] - c0[2]) / steps, blind=c1[3] - c0[3],
                loss_first10=float(ls[:10][finite[:10]].mean()), loss_last10=float(ls[-10:][finite[-10:]].mean()),
                nonfinite_losses=int((~finite).sum()))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--lr', type=float, default=0.05)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3), help='BASELINE.json config: 2 = tol 1e-3, 3 = tol 1e-5')
    a = ap.parse_args()
    for mode in (False, True):
        print(run(a.steps, a.lr, mode, config=a.config), flush=True)
