#!/usr/bin/env python
"""Kernel-iteration driver: N augmented dynamics evaluations (node_odefunc_vjp) at a
given state shape, timed per kernel class with the library's HIP events.  Run it
plain for a quick table, or under `rocprofv3 --kernel-trace --stats` for the
per-kernel breakdown.

    python tools/prof_eval.py --shape 128,256,8,8 --iters 20
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='128,256,8,8')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--fwd-only', action='store_true')
    ap.add_argument('--solve', type=float, default=0.0, metavar='TOL',
                    help='whole adaptive solves (forward + adjoint, dopri5 at this tolerance) instead of single evaluations: the kernels a '
                         'training step launches, fp16-pair component GEMMs and weight gradient included')
    args = ap.parse_args()
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    N, C, H, W = [int(v) for v in args.shape.split(',')]
    torch.manual_seed(0)
    f = nof.ODEfunc(C).cuda()
    y = torch.randn(N, C, H, W, device='cuda')
    cot = torch.randn(N, C, H, W, device='cuda')

    tt = torch.tensor([0.0, 1.0], device='cuda')
    wgt = torch.randn(N, C, H, W, device='cuda') / (C * H * W) ** 0.5

    def once():
        if args.solve > 0.0:
            yy = y.clone().requires_grad_(True)
            out = nof.odeint_adjoint(f, yy, tt, rtol=args.solve, atol=args.solve, method='dopri5')[-1]
            (out * wgt).sum().backward()
            return out
        if args.fwd_only:
            return nof.odefunc_forward(f, 0.3, y)
        return nof.odefunc_vjp(f, 0.3, y, cot)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        once()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters
    integrate.profile_begin()
    for _ in range(args.iters):
        once()
    torch.cuda.synchronize()
    prof = integrate.profile_end()
    out = {'shape': [N, C, H, W], 'wall_us_per_eval': wall * 1e6}
    for k, v in prof.items():
        if v['launches']:
            us = v['total_ms'] / v['launches'] * 1e3
            out[k] = {'avg_us': us, 'tflops': v['flops'] / v['launches'] / (us * 1e-6) / 1e12,
                      'launches': v['launches']}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
