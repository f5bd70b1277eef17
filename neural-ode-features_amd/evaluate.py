#!/usr/bin/env python
"""The evaluations of the reference that drive the ODE block, on the HIP backend
(`/root/reference/evaluate.py:24-305`; the second-heaviest user of the path):

  features   `model.to_features_extractor()`, `odeblock.t1 = [0, .05, ..., 1]`, `odeblock.tol` swept: one dense-output
             solve per batch and tolerance, the head's pooling per time slice  -> features [tols, T, N, C]
             (evaluate.py:24-94; written as .npz -- h5py is not in this image)
  nfe        batch size 1, `tol x t1` sweep, `model.nfe(reset=True)` per image -> nfe.csv.gz with the reference's
             columns y_true, y_pred, nfe, t1, tol (evaluate.py:97-142): the latency regime
  tradeoff   `tol x t1` sweep at the run's batch size, `odeblock.t1` / `.tol` mutated between forwards: test loss, accuracy and
             mean NFE per batch -> tradeoff.csv with the reference's columns t1, test_loss, test_acc, test_nfe, test_tol
             (evaluate.py:145-204)
  accuracy   ONE dense-output solve per batch and tolerance at t1 = [.05, ..., 1] (`return_last_only = False`; the ODE stem of an
             `ode2` run alike, `apply_conv = True`): loss and accuracy of the classifier at EVERY time slice -> `results` (csv)
             with the reference's columns t1, test_loss, test_acc, test_nfe, test_tol (evaluate.py:207-305)

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


def _test_batches(xte, yte, bs, device):
    for i in range(0, xte.shape[0], bs):
        yield xte[i:i + bs].to(device), yte[i:i + bs].to(device)


def tradeoff(args):
    """evaluate.py:145-204: accuracy / NFE trade-off over `tol x t1`, the live block mutated between forwards."""
    import pandas as pd
    import torch.nn.functional as F
    model, p, xte, yte = load_run(args.run)
    if args.limit:
        xte, yte = xte[:args.limit], yte[:args.limit]
    model = model.to(args.device).eval()
    rows = []
    with torch.no_grad():
        for tol, t1 in itertools.product(args.tol, args.t1):
            model.odeblock.t1 = t1
            model.odeblock.tol = tol
            model.nfe(reset=True)
            n_correct = n_processed = n_batches = nfe_forward = 0
            loss = None
            for x, y in _test_batches(xte, yte, p.batch_size, args.device):
                pr = model(x)
                nfe_forward += model.nfe(reset=True)
                loss = F.cross_entropy(pr, y)
                n_correct += int((y == pr.argmax(dim=1)).sum())
                n_processed += y.shape[0]
                n_batches += 1
            # (the reference reports the LAST batch's mean loss over the images processed so far, evaluate.py:181: kept as it is)
            rows.append({'t1': t1, 'test_loss': float(loss) / n_processed, 'test_acc': n_correct / n_processed,
                         'test_nfe': nfe_forward / n_batches, 'test_tol': tol})
    out = os.path.join(args.run, 'tradeoff.csv')
    df = pd.DataFrame(rows)
    df.to_csv(out, index=False)
    print(df)
    return out


def accuracy(args):
    """evaluate.py:207-305: the classifier's loss / accuracy at every time slice of ONE dense-output solve per batch."""
    import pandas as pd
    import torch.nn.functional as F
    model, p, xte, yte = load_run(args.run)
    if args.limit:
        xte, yte = xte[:args.limit], yte[:args.limit]
    model = model.to(args.device).eval()
    t1 = torch.arange(0, 1.05, .05) if args.t1 is None or len(args.t1) < 2 else torch.tensor([0.0] + [t for t in args.t1 if t > 0])
    model.odeblock.t1 = t1[1:].tolist()           # 0 is implicit (evaluate.py:231)
    model.odeblock.return_last_only = False
    if p.downsample == 'ode2':
        model.downsample.odeblock.t1 = t1[1:].tolist()
        model.downsample.odeblock.return_last_only = False
        model.downsample.odeblock.apply_conv = True
        t1 = torch.cat((t1, t1))
    T = len(t1)
    frames = []
    with torch.no_grad():
        for tol in args.tol:
            model.odeblock.tol = tol
            if 'ode' in p.downsample:
                model.downsample.odeblock.tol = tol
            model.nfe(reset=True)
            n_correct, tot_losses = torch.zeros(T), torch.zeros(T)
            n_processed = n_batches = nfe_forward = 0
            for x, y in _test_batches(xte, yte, p.batch_size, args.device):
                pr = model(x)                                      # timestamps (T) x batch (N) x classes (C)
                nfe_forward += model.nfe(reset=True)
                losses = F.cross_entropy(pr.permute(1, 2, 0), y.unsqueeze(1).expand(-1, T), reduction='none')      # N x T
                tot_losses += losses.sum(0).cpu()
                n_correct += (y.unsqueeze(0).expand(T, -1) == pr.argmax(dim=-1)).sum(-1).float().cpu()
                n_processed += y.shape[0]
                n_batches += 1
            frames.append(pd.DataFrame({'t1': t1.numpy(), 'test_loss': (tot_losses / n_processed).numpy(),
                                        'test_acc': (n_correct / n_processed).numpy(), 'test_nfe': [nfe_forward / n_batches] * T,
                                        'test_tol': [tol] * T}))
    out = os.path.join(args.run, 'results')
    df = pd.concat(frames, ignore_index=True)
    df.to_csv(out, index=False)
    print(df)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description='features / nfe / tradeoff / accuracy evaluations of the reference on the HIP backend')
    ap.add_argument('mode', choices=('features', 'nfe', 'tradeoff', 'accuracy'))
    ap.add_argument('run')
    ap.add_argument('--t1', type=float, nargs='+', default=np.arange(0, 1.05, .05).tolist())      # evaluate.py:424
    ap.add_argument('--tol', type=float, nargs='+', default=[1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2])      # evaluate.py:423
    ap.add_argument('--limit', type=int, default=0, help='only the first N test images')
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('neural_ode_features_amd.evaluate needs a HIP device: the ODE block has no CPU path')
    args.device = torch.device('cuda')
    return {'features': features, 'nfe': nfe, 'tradeoff': tradeoff, 'accuracy': accuracy}[args.mode](args)


if __name__ == '__main__':
    main()
