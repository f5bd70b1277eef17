#!/usr/bin/env python
"""Unprofiled GPU time per phase of the cfg-2 training step (CUDA events on the current stream):
stem forward | ODE forward | head + loss | head backward | ODE adjoint | stem backward | optimizer."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof

torch.manual_seed(23)
if os.environ.get('BENCHMARK'): torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
model = nof.ODENet(3, out=10, n_filters=256, downsample='residual', method='dopri5', tol=1e-3, adjoint=True, t1=1, dropout=0.5).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=bool(os.environ.get('FUSED_SGD')))
gen = torch.Generator().manual_seed(1234)
x = torch.randn(128, 3, 32, 32, generator=gen).to(dev).requires_grad_(bool(os.environ.get('GRAPH_STEM')))
y = torch.randint(0, 10, (128,), generator=gen).to(dev)
if os.environ.get('CHANNELS_LAST'):
    model.downsample = model.downsample.to(memory_format=torch.channels_last)
    x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(x.requires_grad)
if os.environ.get('GRAPH_STEM'):
    model.downsample = torch.cuda.make_graphed_callables(model.downsample, (torch.randn(128, 3, 32, 32, device=dev, requires_grad=True),))
if os.environ.get('GRAPH_HEAD'):
    sample = torch.randn(128, 256, 8, 8, device=dev, requires_grad=True)
    model.classifier = torch.cuda.make_graphed_callables(model.classifier, (sample,))
ev = lambda: torch.cuda.Event(enable_timing=True)
names = ['stem fwd', 'ode fwd', 'head+loss', 'head bwd', 'ode adjoint', 'stem bwd', 'optimizer']
tot = [0.0] * len(names)
wall = 0.0
steps, warm = 20, 5
for it in range(steps + warm):
    e = [ev() for _ in range(8)]
    marks = {}
    def mark(i):
        e[i].record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mark(0)
    h = model.downsample(x)
    mark(1)
    h.register_hook(lambda g: mark(5))           # grad wrt stem output arrives: adjoint done
    z = model.odeblock(h)
    mark(2)
    z.register_hook(lambda g: mark(4))           # grad wrt ODE output arrives: head backward done
    p = model.classifier(z)
    loss = F.cross_entropy(p, y)
    mark(3)
    loss.backward()
    mark(6)
    opt.step(); opt.zero_grad()
    mark(7)
    torch.cuda.synchronize()
    if it >= warm:
        wall += time.perf_counter() - t0
        for k in range(7):
            tot[k] += e[k].elapsed_time(e[k + 1])
    model.nfe(reset=True)
print('wall %.3f ms/step' % (wall / steps * 1e3))
for n, t in zip(names, tot):
    print('%-12s %7.3f ms' % (n, t / steps))
