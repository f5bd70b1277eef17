#!/usr/bin/env python
"""The bs = 1 NFE census of the reference (evaluate.py:97-142) end to end: wall time per image of `model(x)` -- stem, ODE block, head, the
`.item()` of the prediction -- on the residual CIFAR-10 net of BASELINE.json configs[1], random weights, per tolerance.

    python tools/census_bs1.py [--filters 256] [--images 200] [--tols 1e-3,1e-1,1e1]
    NODE_TUNE_TINY_RESIDENT=0 python tools/census_bs1.py        (the launch-per-convolution latency path)
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
    ap.add_argument('--filters', type=int, default=256)
    ap.add_argument('--images', type=int, default=200)
    ap.add_argument('--tols', default='1e-3,1e-1,1e1')
    ap.add_argument('--graphs', action='store_true', help='stem and head as inference hipGraphs (graphs.capture_inference)')
    args = ap.parse_args()
    import neural_ode_features_amd as nof
    torch.manual_seed(0)
    model = nof.ODENet(3, out=10, n_filters=args.filters, downsample='residual', method='dopri5', tol=1e-3).cuda().eval()
    x = torch.randn(args.images, 3, 32, 32, device='cuda')
    if args.graphs:
        from neural_ode_features_amd import graphs
        graphs.capture_inference(model, x[:1])
    for tol in [float(v) for v in args.tols.split(',')]:
        model.odeblock.tol = tol
        with torch.no_grad():
            for i in range(10):
                model(x[i:i + 1]).argmax(dim=1).item()
            model.nfe(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.images):
                model(x[i:i + 1]).argmax(dim=1).item()
            wall = (time.perf_counter() - t0) / args.images
            nfe = model.nfe(reset=True) / args.images
            # the block alone on the same states
            h = model.downsample(x[:1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.images):
                model.odeblock(h)
            torch.cuda.synchronize()
            block = (time.perf_counter() - t0) / args.images
        print(json.dumps({'filters': args.filters, 'tol': tol, 'nfe_per_image': nfe, 'us_per_image': wall * 1e6, 'images_per_s': 1.0 / wall,
                          'ode_block_us': block * 1e6, 'resident': os.environ.get('NODE_TUNE_TINY_RESIDENT', '1'), 'graphs': bool(args.graphs)}))


if __name__ == '__main__':
    main()
