#!/usr/bin/env python
"""How often does deferred completion miss when the data changes every step?  N optimizer steps of the cfg-2 model on a
FRESH synthetic batch per step (class-dependent means, so the loss falls and the weights move), with deferred
completion and with a read-back per solve; reports throughput, misses (skipped updates) and the loss trajectory.

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
    if d is not None:
        opt.use_deferred(d)
    losses = []
    import contextlib
    with (d if d is not None else contextlib.nullcontext()):
        for i in range(10):              # MIOpen's first-use searches, allocator warm-up
            y = torch.randint(0, 10, (128,), device=dev, generator=gen)
            x = torch.randn(128, 3, 32, 32, device=dev, generator=gen) + means[y]
            bench.train_step(model, opt, x, y)
        torch.cuda.synchronize()
        m0 = d.resolve() if d is not None else 0
        t0 = time.perf_counter()
        for i in range(steps):
            y = torch.randint(0, 10, (128,), device=dev, generator=gen)
            x = torch.randn(128, 3, 32, 32, device=dev, generator=gen) + means[y]
            loss, nf, nb = bench.train_step(model, opt, x, y)
            losses.append(loss.detach())
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    misses = (d.resolve() - m0) if d is not None else 0
    ls = torch.stack(losses).float().cpu()
    finite = torch.isfinite(ls)
    return dict(mode='deferred' if deferred_on else 'read-back', steps=steps, images_per_s=steps * 128 / wall,
                misses=misses, blind=d.blind_solves if d is not None else 0,
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
