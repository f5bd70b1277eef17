#!/usr/bin/env python
"""Where does a DROP-IN training step (read-back per solve: the reference's own call shape) spend its
wall time?  cfg 2, fixed batch.  Host time of every phase of the step (time.perf_counter around it; the two solves split into
"enqueue" = the C call up to its stream synchronisation and "wait" = inside that synchronisation, measured by the library-free
proxy: a torch.cuda.synchronize() right behind the call costs nothing once the call has synchronised), wall time per step, and
the same step with deferred completion for reference.     python tools/dropin_time.py [steps]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402
from neural_ode_features_amd import integrate  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS[2])
model = bench.build_model(dev, cfg, 'dopri5')
model.train()
opt = nof.FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
x = torch.randn(128, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (128,), device=dev)

marks = {}
_sf, _sa = integrate.solve_forward, integrate.solve_adjoint


def timed(name, fn):
    def wrapped(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        marks.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return r
    return wrapped


integrate.solve_forward = timed('solve_forward (enqueue + read-back wait)', _sf)
integrate.solve_adjoint = timed('solve_adjoint (enqueue + read-back wait)', _sa)


def step():
    t = [time.perf_counter()]
    p = model(x); t.append(time.perf_counter())
    loss = nof.cross_entropy(p, y); t.append(time.perf_counter())
    model.nfe(reset=True)
    loss.backward(); t.append(time.perf_counter())
    model.nfe(reset=True)
    opt.step(); opt.zero_grad(); t.append(time.perf_counter())
    for name, a, b in (('model(x)  [stem + forward solve + head]', 0, 1), ('cross_entropy', 1, 2), ('loss.backward() [head + adjoint solve + stem]', 2, 3),
                       ('optimizer', 3, 4)):
        marks.setdefault(name, []).append((t[b] - t[a]) * 1e3)


for _ in range(8):
    step()
torch.cuda.synchronize()
marks.clear()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
print('drop-in (read-back per solve): %.3f ms per step wall, %.0f images/s' % (wall, 128e3 / wall))
for name, v in marks.items():
    print('  host %-52s median %.3f ms  (min %.3f, max %.3f)' % (name, statistics.median(v), min(v), max(v)))

# the GPU time of the same work: the step with deferred completion (host far ahead)
integrate.solve_forward, integrate.solve_adjoint = _sf, _sa
d = integrate.Deferred(dev)
loop = integrate.DeferredLoop(d, opt, lambda xx, yy: bench.train_step(model, opt, xx, yy))
for _ in range(30):
    loop.step(x, y)
loop.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loop.step(x, y)
loop.flush()
torch.cuda.synchronize()
wd = (time.perf_counter() - t0) / steps * 1e3
print('deferred completion:           %.3f ms per step wall, %.0f images/s   (drop-in / deferred = %.3f)' % (wd, 128e3 / wd, wd / wall))
# host-only cost of enqueueing: one step's enqueue time with nothing to wait for
host = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.step(x, y)
    host.append((time.perf_counter() - t0) * 1e3)
loop.flush()
print('host time to enqueue one deferred step: median %.3f ms (min %.3f)' % (statistics.median(host), min(host)))
