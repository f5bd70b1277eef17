#!/usr/bin/env python
"""cProfile of the DROP-IN training step (cfg 2, read-back per solve): which Python functions the host spends its time in -- the
two C calls that contain the solves hold the GPU's time, everything else is what the GPU waits for behind a read-back."""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402

dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS[2])
model = bench.build_model(dev, cfg, 'dopri5')
model.train()
opt = nof.FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
x = torch.randn(128, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (128,), device=dev)


def step():
    p = model(x)
    loss = nof.cross_entropy(p, y)
    loss.backward()
    opt.step()
    opt.zero_grad()


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40)
print(s.getvalue()[:9000])
