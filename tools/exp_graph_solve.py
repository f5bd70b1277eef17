#!/usr/bin/env python
"""Does replaying a forward solve from a hipGraph shorten its GPU time?  A deferred-completion solve has no host
synchronisation in it (kernels + one device-to-device copy), so it can be captured; eager enqueue (host far ahead of
the GPU) against graph replay, cfg-2 state."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof  # noqa: E402
from neural_ode_features_amd import integrate  # noqa: E402

torch.manual_seed(0)
dev = torch.device('cuda', 0)
f = nof.ODEfunc(256).to(dev)
rec = integrate.Recognised(f)
y = torch.randn(128, 256, 8, 8, device=dev)
times = [0.0, 1.0]
out, st = integrate.solve_forward(rec, rec.params, y, times, 1e-3, 1e-3, 0, None)
steps = st['accepted'] + st['rejected']
print('steps', steps)
record = torch.zeros(64, dtype=torch.uint8, device=dev)
flag = torch.zeros(1, device=dev)


def blind():
    return integrate.solve_forward(rec, rec.params, y, times, 1e-3, 1e-3, 0, None, blind=(steps, record, flag))[0]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('eager blind solve     %.3f ms' % timeit(blind))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    blind()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    static_out = blind()
print('graph replay          %.3f ms' % timeit(g.replay))
ref = blind()
g.replay()
torch.cuda.synchronize()
print('same result', bool(torch.equal(ref, static_out)), 'miss flag', float(flag))
print('max abs diff', float((ref - static_out).abs().max()), 'ref max', float(ref.abs().max()))
ref2 = blind()
torch.cuda.synchronize()
print('eager vs eager equal', bool(torch.equal(ref, ref2)))
g.replay(); torch.cuda.synchronize(); a = static_out.clone(); g.replay(); torch.cuda.synchronize()
print('replay vs replay equal', bool(torch.equal(a, static_out)))
