#!/usr/bin/env python
"""The two evaluations of the reference that drive the ODE block hardest, on the HIP backend
(`/root/reference/evaluate.py:24-142`; the second-heaviest user of the path):

  features   `model.to_features_extractor()`, `odeblock.t1 = [0, .05, ..., 1]`, `odeblock.tol` swept: one dense-output
             solve per batch and tolerance, the head's pooling per time slice  -> features [tols, T, N, C]
             (evaluate.py:24-94; written as .npz -- h5py is not in this image)
  nfe        batch size 1, `tol x t1` sweep, `model.nfe(reset=True)` per image -> nfe.csv.gz with the reference's
             columns y_true, y_pred, nfe, t1, tol (evaluate.py:97-142): the latency regime

Runs on a run directory written by `neural_ode_features_amd.train` (or any `{'params', 'model'}` checkpoint with the
reference's state_dict keys).  Test data: `--data file.pt` (`x_test`, `y_test`) or the synthetic set of that run.

    python -m neural_ode_features_amd.evaluate features runs_cifar10/odenet --t1 0 0.5 1 --tol 1e-3 1e-1
    python -m neural_ode_features_amd.evaluate nfe runs_cifar10/odenet --limit 100
"""
from __future__ import annotations

import argparse
import itertools
import os
import sys
import types

import numpy as np
import torch


def load_run(run_dir, which='best'):
    """`utils.load_model` stand-in (utils.py:248-270): rebuild the net from the run's params, load its weights."""
    import neural_ode_features_amd as nof
    path = os.path.join(run_dir, which + '.pth')
    if not os.path.exists(path):
        path = os.path.join(run_dir, 'last.pth')
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    p = types.SimpleNamespace(**ckpt['params'])
    from .train import SHAPES, load_data
    if getattr(p, 'data', None):
        blob = torch.load(p.data, map_location='cpu')
        xte, yte = blob['x_test'], blob['y_test']
        in_ch, out = xte.shape[1], int(blob['y_train'].max()) + 1
    else:
        _, _, xte, yte, in_ch, out = load_data(p)
    model = nof.ODENet(in_ch, out=out, n_filters=p.filters, downsample=p.downsample, method=p.method, tol=p.tol,
                       adjoint=p.adjoint, dropout=p.dropout, norm=p.norm)
    model.load_state_dict(ckpt['model'])
    return model, p, xte, yte


def features(args):
    """evaluate.py:24-94."""
    model, p, xte, yte = load_run(args.run)
    if args.limit:
        xte, yte = xte[:args.limit], yte[:args.limit]
    model = model.to(args.device).eval()
    model.to_features_extractor()
    model.odeblock.t1 = list(args.t1)
    if 'ode' in p.downsample:
        model.downsample.odeblock.t1 = list(args.t1)
    feats = []
    with torch.no_grad():
        for tol in args.tol:
            model.odeblock.tol = tol
            f = [model(xte[i:i + p.batch_size].to(args.device)).cpu().numpy() for i in range(0, xte.shape[0], p.batch_size)]
            feats.append(np.concatenate(f, -2))       # concat along the batch dimension
    out = os.path.join(args.run, 'features.npz')
    np.savez(out, features=np.stack(feats), y_true=yte.numpy(), tols=np.array(args.tol), t1s=np.array(args.t1))
    print('features', np.stack(feats).shape, '->', out)
    return out


def nfe(args):
    """evaluate.py:97-142: per-image function evaluations, batch size 1."""
    import pandas as pd
    model, p, xte, yte = load_run(args.run)
    if args.limit:
        xte, yte = xte[:args.limit], yte[:args.limit]
    model = model.to(args.device).eval()
    rows = []
    with torch.no_grad():
        for tol, t1 in itertools.product(args.tol, args.t1):
            # (t1 = 0 is the identity block, model.py:363-364: rows with nfe = 0, as the reference writes them)
            model.odeblock.t1 = t1
            model.odeblock.tol = tol
            model.nfe(reset=True)
            for i in range(xte.shape[0]):
                pred = model(xte[i:i + 1].to(args.device)).argmax(dim=1).item()
                rows.append({'y_true': int(yte[i]), 'y_pred': pred, 'nfe': model.nfe(reset=True), 't1': t1, 'tol': tol})
    out = os.path.join(args.run, 'nfe.csv.gz')
    df = pd.DataFrame(rows)
    df.to_csv(out, index=False)
    print(df.groupby(['tol', 't1']).nfe.mean())
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description='features / nfe evaluations of the reference on the HIP backend')
    ap.add_argument('mode', choices=('features', 'nfe'))
    ap.add_argument('run')
    ap.add_argument('--t1', type=float, nargs='+', default=np.arange(0, 1.05, .05).tolist())      # evaluate.py:424
    ap.add_argument('--tol', type=float, nargs='+', default=[1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2])      # evaluate.py:423
    ap.add_argument('--limit', type=int, default=0, help='only the first N test images')
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('neural_ode_features_amd.evaluate needs a HIP device: the ODE block has no CPU path')
    args.device = torch.device('cuda')
    return {'features': features, 'nfe': nfe}[args.mode](args)


if __name__ == '__main__':
    main()
