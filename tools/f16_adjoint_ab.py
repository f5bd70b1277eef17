#!/usr/bin/env python
"""Whole adjoint solves with the fp16-pair operands (NODE_TUNE_W4_F16=1, default) against the bf16-triple ones (=0) on the same
inputs: outputs, gradients, step histories.  GPU, no oracle: the two pipelines differ in operand format only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_ode_features_amd as nof
from tests.helpers import make_func, rel_err


def run(shape, tol, seed, kink_free, gscale):
    N, C, H, W = shape
    f, _ = make_func(C, seed=seed, device='cuda', kink_free=kink_free)
    gen = torch.Generator().manual_seed(seed + 1)
    y = torch.randn(N, C, H, W, generator=gen)
    wgt = torch.randn(2, N, C, H, W, generator=gen) / (C * H * W) ** 0.5 * gscale
    t = torch.tensor([0.0, 1.0]).cuda()
    res = {}
    for f16 in ('1', '0'):
        os.environ['NODE_TUNE_W4_F16'] = f16
        yh = y.cuda().requires_grad_(True)
        for p in f.parameters():
            p.grad = None
        out = nof.odeint_adjoint(f, yh, t, rtol=tol, atol=tol, method='dopri5')
        (out * wgt.cuda()).sum().backward()
        gp = torch.cat([p.grad.reshape(-1) for p in f.parameters()])
        res[f16] = (out.detach(), yh.grad.clone(), gp.clone(), dict(f.last_forward_stats), dict(f.last_backward_stats))
    del os.environ['NODE_TUNE_W4_F16']
    a, b = res['1'], res['0']
    print(shape, 'tol', tol, 'kink_free', kink_free, 'grad scale', gscale)
    print('  fwd', {k: a[3][k] for k in ('accepted', 'rejected', 'nfe')}, {k: b[3][k] for k in ('accepted', 'rejected', 'nfe')})
    print('  bwd', {k: a[4][k] for k in ('accepted', 'rejected', 'nfe')}, {k: b[4][k] for k in ('accepted', 'rejected', 'nfe')})
    print('  out rel %.3e   grad_y rel %.3e   grad_theta rel %.3e' % (rel_err(a[0], b[0]), rel_err(a[1], b[1]), rel_err(a[2], b[2])), flush=True)
    assert torch.isfinite(a[1]).all() and torch.isfinite(a[2]).all()


for gs in (1.0, 1e-6, 1e4):
    run((128, 256, 8, 8), 1e-3, 51, True, gs)
run((128, 256, 8, 8), 1e-5, 51, True, 1.0)
run((128, 256, 8, 8), 1e-3, 53, False, 1.0)
run((32, 128, 8, 8), 1e-3, 7, False, 1.0)
