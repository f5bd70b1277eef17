#!/usr/bin/env python
"""Soak of the resident solve (csrc/kernels_tiny_solve.hip): tens of thousands of solves back to back on alternating inputs, every result compared
bit for bit with the first solve of its input.  The hand-offs inside the launch are unfenced tagged words: a stale or torn word that passed
for a valid one would show up here as a differing result (or as a deadline: NodeHipError).

    python tools/resident_soak.py [--solves 20000] [--shapes 1,256,8,8;2,256,8,8;3,128,7,7;1,64,8,8]
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
    ap.add_argument('--solves', type=int, default=20000)
    ap.add_argument('--shapes', default='1,256,8,8;2,256,8,8;3,128,7,7;1,64,8,8')
    args = ap.parse_args()
    import neural_ode_features_amd as nof
    for spec in args.shapes.split(';'):
        N, C, H, W = [int(v) for v in spec.split(',')]
        torch.manual_seed(C + N)
        f = nof.ODEfunc(C).cuda()
        ys = [torch.randn(N, C, H, W, device='cuda') * s for s in (1.0, 0.3, 2.5)]
        t = torch.tensor([0.0, 0.4, 1.0], device='cuda')
        tols = [1e-3, 1e-5, 1e-2]
        with torch.no_grad():
            refs = [nof.odeint(f, y, t, rtol=tol, atol=tol, method='dopri5').clone() for y, tol in zip(ys, tols)]
            nfes = []
            for y, tol in zip(ys, tols):
                nof.odeint(f, y, t, rtol=tol, atol=tol, method='dopri5')
                nfes.append(f.last_forward_stats['nfe'])
            bad = 0
            t0 = time.perf_counter()
            for i in range(args.solves):
                k = i % 3
                out = nof.odeint(f, ys[k], t, rtol=tols[k], atol=tols[k], method='dopri5')
                if i % 16 == 0 or i == args.solves - 1:       # (the comparison synchronises: most solves run back to back without it)
                    if not torch.equal(out, refs[k]):
                        bad += 1
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            # and every solve of a final stretch
            for i in range(600):
                k = i % 3
                if not torch.equal(nof.odeint(f, ys[k], t, rtol=tols[k], atol=tols[k], method='dopri5'), refs[k]):
                    bad += 1
        print(json.dumps({'shape': [N, C, H, W], 'solves': args.solves + 600, 'nfe_per_solve': nfes, 'differing_results': bad,
                          'us_per_solve': wall / args.solves * 1e6}))


if __name__ == '__main__':
    main()
