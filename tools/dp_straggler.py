#!/usr/bin/env python
"""What an 8-GPU data-parallel run of configs[3] (bs 1024 = 8 x 128, local-norm mode: every rank adapts its own steps,
SURVEY.md 8e) will look like, measured on ONE GPU: the eight shards of every global batch are integrated one after the
other (gradients summed, one optimizer step per global batch -- the arithmetic of the 8-rank run), and per shard the
solver's step counts and the device time of its forward + backward are recorded.

  * stragglers: all ranks meet at the gradient all-reduce, so a step costs the SLOWEST shard's time:
        predicted scaling efficiency = mean_step(mean over shards) / mean_step(max over shards)
  * misses under deferred completion: every rank predicts its own step counts (integrate.Deferred's policy, replayed
    here on each rank's own count sequence); a miss on ANY rank voids the update on ALL ranks (the flag rides in the
    all-reduce), so the per-rank rate compounds.

    python tools/dp_straggler.py [--steps 50] [--config 2] [--ranks 8]
"""
import argparse
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402
from neural_ode_features_amd import integrate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3))
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--lr', type=float, default=0.05)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = dict(bench.CONFIGS[a.config])
    model = bench.build_model(dev, cfg, 'dopri5')
    model.train()
    opt = nof.FusedSGD(model.parameters(), lr=a.lr, momentum=0.9, weight_decay=1e-4)
    opt.grad_scale = 1.0 / a.ranks
    gen = torch.Generator(device='cuda').manual_seed(0)
    means = torch.randn(10, 3, 1, 1, device=dev, generator=gen)
    bs = cfg['batch']
    func = model.odeblock.odefunc
    # one predictor per rank and kind of solve: integrate.Deferred's own policy, fed with that rank's counts
    pred = [integrate.Deferred(dev) for _ in range(a.ranks)]
    times, counts, misses_rank, miss_any, dead = [], [], [0] * a.ranks, 0, 0
    for step in range(a.steps + 5):
        y = torch.randint(0, 10, (a.ranks * bs,), device=dev, generator=gen)
        x = torch.randn(a.ranks * bs, 3, cfg['image'], cfg['image'], device=dev, generator=gen) + means[y]
        row_t, row_c, any_miss = [], [], False
        for r in range(a.ranks):
            xs, ys = x[r * bs:(r + 1) * bs], y[r * bs:(r + 1) * bs]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            loss = F.cross_entropy(model(xs), ys)
            loss.backward()                       # gradients ACCUMULATE over the shards (the all-reduce's sum)
            e1.record()
            torch.cuda.synchronize()
            fs, bs_ = func.last_forward_stats, func.last_backward_stats
            cf, cb = fs['accepted'] + fs['rejected'], bs_['accepted'] + bs_['rejected']
            row_t.append(e0.elapsed_time(e1))
            row_c.append((cf, cb))
            for kind, c, st in (('fwd', cf, fs), ('bwd', cb, bs_)):
                key = (kind, 0, (0.0, 1.0))
                d = pred[r]
                if d.guess.get(key):
                    enq = d._enqueue(key)
                    if c > enq:
                        misses_rank[r] += step >= 5
                        any_miss = True
                        d.hist[key] = []
                        d.learned(key, c)
                        continue
                    dead += (enq - c) * (step >= 5)
                    end = 1.0 if kind == 'fwd' else 0.0
                    d.fragile[key] = bool(st['last_dt'] > 0 and (st['t_final'] - end) < d.FRAGILE * st['last_dt'])
                    d._observe(key, c)
                else:
                    d.learned(key, c)
        opt.step()
        opt.zero_grad()
        if step >= 5:
            times.append(row_t)
            counts.append(row_c)
            miss_any += any_miss
    n = len(times)
    mean_of_mean = statistics.mean(statistics.mean(r) for r in times)
    mean_of_max = statistics.mean(max(r) for r in times)
    print('config %d, %d optimizer steps of a %d x %d batch (fresh data every step), one GPU, shards one after the other'
          % (a.config, n, a.ranks, bs))
    print('per-shard device time (forward + adjoint backward, ms): mean %.3f, mean of the per-step MAX %.3f, worst %.3f'
          % (mean_of_mean, mean_of_max, max(max(r) for r in times)))
    print('predicted scaling efficiency from stragglers alone (mean / max): %.3f  -> %.2fx at %d ranks'
          % (mean_of_mean / mean_of_max, a.ranks * mean_of_mean / mean_of_max, a.ranks))
    for kind, idx in (('forward', 0), ('backward', 1)):
        flat = [c[idx] for row in counts for c in row]
        spread = [max(c[idx] for c in row) - min(c[idx] for c in row) for row in counts]
        hist = {v: flat.count(v) for v in sorted(set(flat))}
        print('%s step counts over all shards: %s; steps where the shards disagree: %d of %d (largest spread %d)'
              % (kind, hist, sum(1 for s in spread if s > 0), n, max(spread)))
    blind = 2 * n * a.ranks
    print("deferred completion, every rank predicting its own counts (integrate.Deferred's policy): %d misses in %d blind "
          'solves (per rank: %s); optimizer steps voided on ALL ranks: %d of %d (%.1f %%); dead steps enqueued: %.2f per rank and step'
          % (sum(misses_rank), blind, misses_rank, miss_any, n, 100.0 * miss_any / n, dead / (n * a.ranks)))


if __name__ == '__main__':
    main()
