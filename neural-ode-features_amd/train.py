#!/usr/bin/env python
"""Training loop of the reference (`/root/reference/train.py:26-192`) on the HIP backend: same flags, same per-epoch
metrics (loss, acc, NFE-F, NFE-B, test_loss, test_acc, test_nfe), same checkpoint dictionary
(`{'epoch', 'params', 'model', 'optim', 'metrics'}`, train.py:18-23,182-188), resume (train.py:145-152) and LR
schedules (`fixed` / `plateau` / `cosine`, train.py:158-163) -- so a run started by the reference continues here and
the reverse: the model's state_dict keys and the optimizer's state layout are the reference's.

What is NOT here (out of the hot path's scope): torchvision datasets / augmentation and the
`expman` run-directory bookkeeping.  Data comes from `--data file.pt` (a dict of tensors `x_train, y_train, x_test,
y_test`) or, by default, a synthetic set of the dataset's shape; the run directory is `--run-dir`.

    python -m neural_ode_features_amd.train --dataset cifar10 -d residual -f 256 --dropout 0.5 -a --lr 0.1 --wd 1e-4 \
        --lrschedule cosine --lrcycle 250 -e 250 --run-dir runs_cifar10/odenet
"""
from __future__ import annotations

import argparse
import csv
import os
import shutil
import sys

import torch
import torch.nn.functional as F
from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR, ReduceLROnPlateau

SHAPES = {'mnist': (1, 28, 10), 'cifar10': (3, 32, 10), 'cifar100': (3, 32, 100), 'tiny-imagenet-200': (3, 64, 200)}


def load_data(args):
    """`utils.load_dataset` stand-in: (x_train, y_train, x_test, y_test, in_ch, out)."""
    in_ch, side, out = SHAPES[args.dataset]
    if args.data:
        blob = torch.load(args.data, map_location='cpu')
        return blob['x_train'], blob['y_train'], blob['x_test'], blob['y_test'], blob['x_train'].shape[1], int(blob['y_train'].max()) + 1
    gen = torch.Generator().manual_seed(args.seed)
    n_tr, n_te = args.synthetic_size, max(args.batch_size, args.synthetic_size // 4)
    # class-dependent means so that there is something to learn
    means = torch.randn(out, in_ch, 1, 1, generator=gen)
    ytr, yte = torch.randint(0, out, (n_tr,), generator=gen), torch.randint(0, out, (n_te,), generator=gen)
    xtr = torch.randn(n_tr, in_ch, side, side, generator=gen) + means[ytr]
    xte = torch.randn(n_te, in_ch, side, side, generator=gen) + means[yte]
    return xtr, ytr, xte, yte, in_ch, out


def batches(x, y, bs, shuffle, gen):
    idx = torch.randperm(x.shape[0], generator=gen) if shuffle else torch.arange(x.shape[0])
    for i in range(0, x.shape[0], bs):
        j = idx[i:i + bs]
        yield x[j], y[j]


def train(data, model, optimizer, args, gen):
    """train.py:26-72."""
    model.train()
    optimizer.zero_grad()
    nfe_forward = nfe_backward = 0
    n_correct = n_processed = n_batch = 0
    total_loss = 0.0
    for x, y in batches(data[0], data[1], args.batch_size, True, gen):
        x, y = x.to(args.device), y.to(args.device)
        p = model(x)
        loss = F.cross_entropy(p, y)
        total_loss += loss.item()
        n_correct += (y == p.argmax(dim=1)).sum().item()
        n_processed += y.shape[0]
        nfe_forward += model.nfe(reset=True)
        loss.backward()
        nfe_backward += model.nfe(reset=True)
        n_batch += 1
        if n_batch % args.batch_accumulation == 0:
            optimizer.step()
            optimizer.zero_grad()
    return {'loss': total_loss / n_batch, 'acc': n_correct / n_processed, 'nfe-f': nfe_forward / n_batch,
            'nfe-b': nfe_backward / n_batch}


def evaluate(data, model, args):
    """train.py:75-108 (like the reference: no `no_grad`, the solver does not record a graph anyway)."""
    model.eval()
    nfe_forward = n_correct = n_batches = n_processed = 0
    total_loss = 0.0
    with torch.no_grad():
        for x, y in batches(data[0], data[1], args.batch_size, False, None):
            x, y = x.to(args.device), y.to(args.device)
            p = model(x)
            nfe_forward += model.nfe(reset=True)
            total_loss += F.cross_entropy(p, y, reduction='sum').item()
            n_correct += (y == p.argmax(dim=1)).sum().item()
            n_processed += y.shape[0]
            n_batches += 1
    return {'test_loss': total_loss / n_processed, 'test_acc': n_correct / n_processed, 'test_nfe': nfe_forward / n_batches}


def read_log(path):
    if not os.path.exists(path):
        return []
    with open(path) as fh:
        return list(csv.DictReader(fh))


def push_log(path, metrics):
    rows = read_log(path)
    rows.append({k: metrics[k] for k in metrics})
    keys = list(rows[0].keys())
    for r in rows:
        for k in r:
            if k not in keys:
                keys.append(k)
    with open(path, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=keys)
        w.writeheader()
        w.writerows(rows)


def main(argv=None):
    parser = argparse.ArgumentParser(description='ODENet training on the HIP backend (flags of the reference train.py:196-225)')
    parser.add_argument('--dataset', type=str, choices=tuple(SHAPES), default='mnist')
    parser.add_argument('-d', '--downsample', type=str, choices=('ode2', 'ode', 'residual', 'convolution', 'minimal', 'one-shot'),
                        default='residual')
    parser.add_argument('-n', '--norm', type=str, choices=('group',), default='group')
    parser.add_argument('-f', '--filters', type=int, default=64)
    parser.add_argument('--dropout', type=float, default=0)
    parser.add_argument('-e', '--epochs', type=int, default=100)
    parser.add_argument('-b', '--batch-size', type=int, default=128)
    parser.add_argument('--batch-accumulation', type=int, default=1)
    parser.add_argument('-o', '--optim', type=str, choices=('sgd', 'adam'), default='sgd')
    parser.add_argument('--lr', type=float, default=0.1)
    parser.add_argument('--lrschedule', type=str, choices=('fixed', 'plateau', 'cosine'), default='plateau')
    parser.add_argument('--lrcycle', type=int, default=0)
    parser.add_argument('-p', '--patience', type=int, default=10)
    parser.add_argument('--wd', type=float, default=0, help='weight decay')
    parser.add_argument('--method', default='dopri5', choices=('dopri5', 'rk4'))
    parser.add_argument('-t', '--tol', type=float, default=1e-3)
    parser.add_argument('-a', '--adjoint', default=False, action='store_true')
    parser.add_argument('-r', '--resume', action='store_true', default=False)
    parser.add_argument('-s', '--seed', type=int, default=23)
    # stand-ins for the reference's torchvision datasets / expman run directory
    parser.add_argument('--data', type=str, default=None, help='.pt file with x_train, y_train, x_test, y_test')
    parser.add_argument('--synthetic-size', type=int, default=512)
    parser.add_argument('--run-dir', type=str, default=None)
    args = parser.parse_args(argv)

    torch.manual_seed(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit('neural_ode_features_amd.train needs a HIP device: the ODE block has no CPU path')
    torch.cuda.manual_seed_all(args.seed)
    args.device = torch.device('cuda')
    run_dir = args.run_dir or os.path.join('runs_' + args.dataset, 'odenet_%s_f%d_%s_tol%g%s' % (
        args.downsample, args.filters, args.method, args.tol, '_adjoint' if args.adjoint else ''))
    os.makedirs(run_dir, exist_ok=True)
    log_path, last, best = os.path.join(run_dir, 'log.csv'), os.path.join(run_dir, 'last.pth'), os.path.join(run_dir, 'best.pth')
    if os.path.exists(log_path) and not args.resume:
        print('Skipping ...')                       # train.py:116-118
        return 0

    import neural_ode_features_amd as nof
    xtr, ytr, xte, yte, in_ch, out = load_data(args)
    model = nof.ODENet(in_ch, out=out, n_filters=args.filters, downsample=args.downsample, method=args.method, tol=args.tol,
                       adjoint=args.adjoint, dropout=args.dropout, norm=args.norm).to(args.device)
    if args.optim == 'sgd':
        optimizer = nof.FusedSGD(model.parameters(), lr=args.lr, momentum=0.9, weight_decay=args.wd)   # train.py:136
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)           # train.py:138

    if args.resume:
        ckpt = torch.load(last, map_location=args.device, weights_only=False)
        model.load_state_dict(ckpt['model'])
        optimizer.load_state_dict(ckpt['optim'])
        start_epoch = ckpt['epoch'] + 1
        best_accuracy = max(float(r['test_acc']) for r in read_log(log_path))
        print('Resuming from epoch {}: {}'.format(start_epoch, run_dir))
    else:
        best_accuracy = evaluate((xte, yte), model, args)['test_acc']
        start_epoch = 1

    if args.lrschedule == 'fixed':
        scheduler = LambdaLR(optimizer, lr_lambda=lambda x: 1)
    elif args.lrschedule == 'plateau':
        scheduler = ReduceLROnPlateau(optimizer, mode='max', patience=args.patience)
    else:
        scheduler = CosineAnnealingLR(optimizer, args.lrcycle, last_epoch=start_epoch - 2)

    gen = torch.Generator().manual_seed(args.seed + start_epoch)
    for epoch in range(start_epoch, args.epochs + 1):
        metrics = {'epoch': epoch}
        metrics.update(train((xtr, ytr), model, optimizer, args, gen))
        metrics.update(evaluate((xte, yte), model, args))
        is_best = metrics['test_acc'] > best_accuracy
        best_accuracy = max(metrics['test_acc'], best_accuracy)
        params = {k: (str(v) if isinstance(v, torch.device) else v) for k, v in vars(args).items()}
        torch.save({'epoch': epoch, 'params': params, 'model': model.state_dict(), 'optim': optimizer.state_dict(),
                    'metrics': metrics}, last)
        if is_best:
            shutil.copyfile(last, best)
        push_log(log_path, metrics)
        scheduler.step(metrics['test_acc'] if args.lrschedule == 'plateau' else None)
        print('epoch %d: loss %.3f acc %.2f%% NFE-F %.1f NFE-B %.1f | test loss %.3f acc %.2f%% nfe %.1f | lr %g'
              % (epoch, metrics['loss'], 100 * metrics['acc'], metrics['nfe-f'], metrics['nfe-b'], metrics['test_loss'],
                 100 * metrics['test_acc'], metrics['test_nfe'], optimizer.param_groups[0]['lr']), flush=True)
    return 0


if __name__ == '__main__':
    sys.exit(main())
