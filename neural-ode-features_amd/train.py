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
from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR, ReduceLROnPlateau

from .head import cross_entropy      # F.cross_entropy (train.py:43,93) on the library's kernels; PyTorch's for CPU tensors

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


class Tally:
    """Running sums of one pass over the data (the reference keeps them as locals, train.py:30-37, 78-84)."""

    def __init__(self):
        self.loss_sum = 0.0
        self.hits = self.seen = self.batches = 0
        self.nfe = [0, 0]                      # forward, backward

    def count(self, logits, target, loss):
        """train.py:44,46 / :94-96.  The library's loss kernel leaves {loss, correct predictions} of the batch in device
        memory (`loss.node_stat`): ONE read-back instead of `loss.item()` + an arg-max / compare / sum chain + `.item()`."""
        stat = getattr(loss, 'node_stat', None)
        if stat is not None:
            value, hits = stat.tolist()
            self.loss_sum += value
            self.hits += int(round(hits))
        else:
            self.loss_sum += loss.item()
            self.hits += int((logits.argmax(dim=1) == target).sum())
        self.seen += target.shape[0]

    def count_host(self, value, hits, n):
        self.loss_sum += value
        self.hits += int(round(hits))
        self.seen += n


def train(data, model, optimizer, args, gen, loop=None):
    """One epoch; semantics of train.py:26-72 (loss read per batch, NFE counter read and reset after the forward and
    after the backward, optimizer stepped every `batch_accumulation` batches).  With `loop` (--deferred:
    integrate.DeferredLoop) the solves run with deferred completion: nothing is read back per batch, a solve that
    missed its step count is REPEATED (never skipped), and the per-batch numbers are tallied when their update is
    known to be committed -- the same sums, one iteration late."""
    model.train()
    optimizer.zero_grad()
    tally = Tally()
    if loop is not None:
        def tally_done(results):
            for host, event, n, nf, nb in results:
                event.synchronize()              # (a batch is handed back one iteration late: its numbers arrived long ago)
                tally.count_host(float(host[0]), float(host[1]), n)
                tally.nfe[0] += nf
                tally.nfe[1] += nb
                tally.batches += 1
        for images, target in batches(data[0], data[1], args.batch_size, True, gen):
            tally_done(loop.step(images.to(args.device), target.to(args.device)))
        tally_done(loop.flush())
        nb = tally.batches
        return {'loss': tally.loss_sum / nb, 'acc': tally.hits / tally.seen, 'nfe-f': tally.nfe[0] / nb, 'nfe-b': tally.nfe[1] / nb}
    for images, target in batches(data[0], data[1], args.batch_size, True, gen):
        images, target = images.to(args.device), target.to(args.device)
        logits = model(images)
        loss = cross_entropy(logits, target)
        tally.count(logits, target, loss)
        tally.nfe[0] += model.nfe(reset=True)
        loss.backward()
        tally.nfe[1] += model.nfe(reset=True)
        tally.batches += 1
        if tally.batches % args.batch_accumulation == 0:
            optimizer.step()
            optimizer.zero_grad()
    nb = tally.batches
    return {'loss': tally.loss_sum / nb, 'acc': tally.hits / tally.seen, 'nfe-f': tally.nfe[0] / nb, 'nfe-b': tally.nfe[1] / nb}


def deferred_loop(model, optimizer, args):
    """--deferred: the training step as a closure for integrate.DeferredLoop (needs FusedSGD: its launch is the one
    commit point the device flag predicates; one optimizer step per batch)."""
    from . import integrate
    from .optim import FusedSGD
    if not isinstance(optimizer, FusedSGD) or args.batch_accumulation != 1 or not args.adjoint or args.method != 'dopri5':
        raise SystemExit('--deferred needs -o sgd, --batch-accumulation 1, --adjoint and --method dopri5')

    def step(images, target):
        logits = model(images)
        loss = cross_entropy(logits, target)
        nf = model.nfe(reset=True)
        loss.backward()
        nb = model.nfe(reset=True)
        optimizer.step()
        optimizer.zero_grad()
        # the batch's {loss, correct predictions} travel to pinned host memory behind the step: nothing waits for them
        host = torch.empty(2, dtype=torch.float32).pin_memory()
        stat = getattr(loss, 'node_stat', None)
        if stat is None:      # (`cross_entropy` went to PyTorch: > 1024 classes, class-probability targets, non-fp32 logits)
            with torch.no_grad():
                stat = torch.stack([loss.detach().float(), (logits.argmax(dim=1) == target).sum().float()])
        host.copy_(stat, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        return host, event, target.shape[0], nf, nb

    return integrate.DeferredLoop(integrate.Deferred(args.device), optimizer, step)


def evaluate(data, model, args):
    """Test pass; semantics of train.py:75-108 (summed loss per image, NFE per batch)."""
    model.eval()
    tally = Tally()
    with torch.no_grad():
        for images, target in batches(data[0], data[1], args.batch_size, False, None):
            images, target = images.to(args.device), target.to(args.device)
            logits = model(images)
            tally.nfe[0] += model.nfe(reset=True)
            tally.count(logits, target, cross_entropy(logits, target, reduction='sum'))
            tally.batches += 1
    return {'test_loss': tally.loss_sum / tally.seen, 'test_acc': tally.hits / tally.seen, 'test_nfe': tally.nfe[0] / tally.batches}


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
    stems = ('ode2', 'ode', 'residual', 'convolution', 'minimal', 'one-shot')
    flags = [   # (names, keyword arguments) -- names, defaults and choices are the reference's
        (('--dataset',), dict(type=str, choices=tuple(SHAPES), default='mnist')),
        (('-d', '--downsample'), dict(type=str, choices=stems, default='residual')),
        (('-n', '--norm'), dict(type=str, choices=('group',), default='group')),
        (('-f', '--filters'), dict(type=int, default=64)),
        (('--dropout',), dict(type=float, default=0)),
        (('-e', '--epochs'), dict(type=int, default=100)),
        (('-b', '--batch-size'), dict(type=int, default=128)),
        (('--batch-accumulation',), dict(type=int, default=1)),
        (('-o', '--optim'), dict(type=str, choices=('sgd', 'adam'), default='sgd')),
        (('--lr',), dict(type=float, default=0.1)),
        (('--lrschedule',), dict(type=str, choices=('fixed', 'plateau', 'cosine'), default='plateau')),
        (('--lrcycle',), dict(type=int, default=0)),
        (('-p', '--patience'), dict(type=int, default=10)),
        (('--wd',), dict(type=float, default=0, help='weight decay')),
        (('--method',), dict(default='dopri5', choices=('dopri5', 'rk4'))),
        (('-t', '--tol'), dict(type=float, default=1e-3)),
        (('-a', '--adjoint'), dict(default=False, action='store_true')),
        (('-r', '--resume'), dict(action='store_true', default=False)),
        (('-s', '--seed'), dict(type=int, default=23)),
        # stand-ins for the reference's torchvision datasets / expman run directory
        (('--data',), dict(type=str, default=None, help='.pt file with x_train, y_train, x_test, y_test')),
        (('--synthetic-size',), dict(type=int, default=512)),
        (('--run-dir',), dict(type=str, default=None)),
        # not in the reference: solves without a read-back per batch; missed step counts are repeated, never skipped
        (('--deferred',), dict(action='store_true', default=False)),
    ]
    for names, kw in flags:
        parser.add_argument(*names, **kw)
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
    loop = deferred_loop(model, optimizer, args) if args.deferred else None
    for epoch in range(start_epoch, args.epochs + 1):
        metrics = {'epoch': epoch}
        metrics.update(train((xtr, ytr), model, optimizer, args, gen, loop))
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
