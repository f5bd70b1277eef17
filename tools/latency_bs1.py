#!/usr/bin/env python
"""Latency regime of the path (evaluate.py:97-142: bs=1 loader, tol x t1 sweep, NFE per image): wall time of one
forward solve of a single image's state, per dopri5 step.

    python tools/latency_bs1.py [--shape 1,256,8,8] [--tols 1e-3,1e-5] [--iters 50]
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
    ap.add_argument('--shape', default='1,256,8,8')
    ap.add_argument('--tols', default='1e-3,1e-5')
    ap.add_argument('--iters', type=int, default=50)
    args = ap.parse_args()
    import neural_ode_features_amd as nof
    N, C, H, W = [int(v) for v in args.shape.split(',')]
    torch.manual_seed(0)
    f = nof.ODEfunc(C).cuda()
    y = torch.randn(N, C, H, W, device='cuda')
    t = torch.tensor([0.0, 1.0], device='cuda')
    out = []
    for tol in [float(v) for v in args.tols.split(',')]:
        with torch.no_grad():
            for _ in range(5):
                nof.odeint(f, y, t, rtol=tol, atol=tol, method='dopri5')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                nof.odeint(f, y, t, rtol=tol, atol=tol, method='dopri5')
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.iters
        st = f.last_forward_stats
        steps = st['accepted'] + st['rejected']
        out.append({'shape': [N, C, H, W], 'tol': tol, 'steps': steps, 'nfe': st['nfe'], 'solve_us': wall * 1e6,
                    'us_per_step': wall * 1e6 / max(1, steps), 'us_per_nfe': wall * 1e6 / st['nfe']})
        print(json.dumps(out[-1]))


if __name__ == '__main__':
    main()
