#!/usr/bin/env python
"""Experiment: does the chip run two half-size training steps (bs 64 + bs 64, two HIP streams, two host threads) faster than one
bs-128 step?  The step is a chain of ~290 dependent 10 - 50 us launches that each fill and drain the chip; two independent chains
overlap one's drain with the other's fill.  (An upper bound for splitting ONE solve's evaluations over two streams: GroupNorm is
per sample, so the halves of a batch only meet at the error norm.)    python tools/two_stream.py [seconds]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import neural_ode_features_amd as nof  # noqa: E402
from neural_ode_features_amd import integrate  # noqa: E402

dev = torch.device('cuda', 0)
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
CFG = int(os.environ.get('CFG', '2'))


def worker(bs, stream, out, idx, barrier):
    cfg = dict(bench.CONFIGS[CFG])
    with torch.cuda.stream(stream):
        torch.manual_seed(0)
        model = bench.build_model(dev, cfg, 'dopri5')
        model.train()
        opt = nof.FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        d = integrate.Deferred(dev)
        loop = integrate.DeferredLoop(d, opt, lambda xx, yy: bench.train_step(model, opt, xx, yy))
        x = torch.randn(bs, 3, 32, 32, device=dev)
        y = torch.randint(0, 10, (bs,), device=dev)
        for _ in range(30):
            loop.step(x, y)
        loop.flush()
        stream.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < SECS:
            loop.step(x, y)
            n += 1
        loop.flush()
        stream.synchronize()
        out[idx] = (n * bs, time.perf_counter() - t0, loop.retries)


def run(sizes):
    out = [None] * len(sizes)
    barrier = threading.Barrier(len(sizes))
    ths = [threading.Thread(target=worker, args=(bs, torch.cuda.Stream(dev), out, i, barrier)) for i, bs in enumerate(sizes)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    total = sum(o[0] / o[1] for o in out)
    print('batches %-12s  %8.0f images/s   (%s)' % (sizes, total, ', '.join('%d in %.2fs, %d retries' % o for o in out)), flush=True)


for sizes in ([128], [64, 64], [128], [64, 64], [128, 128], [64]):
    run(sizes)
