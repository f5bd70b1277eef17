#!/usr/bin/env python
"""Host time to ENQUEUE one training step of cfg 2 (deferred completion, blind solves), against the GPU time of that step:
is the loop GPU-bound or host-bound?   python tools/host_time.py"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402
from neural_ode_features_amd import integrate  # noqa: E402

dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS[2])
model = bench.build_model(dev, cfg, 'dopri5')
model.train()
opt = nof.FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
d = integrate.Deferred(dev)
loop = integrate.DeferredLoop(d, opt, lambda xx, yy: bench.train_step(model, opt, xx, yy))
x = torch.randn(128, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (128,), device=dev)
for _ in range(30):
    loop.step(x, y)
loop.flush()
torch.cuda.synchronize()
host, gpu = [], []
for _ in range(20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    loop.step(x, y)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    host.append((t1 - t0) * 1e3)
    gpu.append(e0.elapsed_time(e1))
print('host enqueue time per step: median %.3f ms (min %.3f); GPU time of the same step: median %.3f ms' %
      (statistics.median(host), min(host), statistics.median(gpu)))
