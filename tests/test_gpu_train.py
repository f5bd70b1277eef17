"""GPU: the training loop mirror (neural_ode_features_amd/train.py <-> reference train.py:26-192): metrics of record,
checkpoint dictionary, resume, LR schedule, and the optimizer state in torch.optim.SGD's layout."""
import csv
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_resume_and_checkpoint_layout(tmp_path):
    from neural_ode_features_amd import train as T
    run = str(tmp_path / 'run')
    common = ['--dataset', 'mnist', '-f', '16', '-b', '32', '--synthetic-size', '96', '-a', '--lr', '0.05', '--wd', '1e-4',
              '--lrschedule', 'cosine', '--lrcycle', '4', '--run-dir', run]
    assert T.main(common + ['-e', '2']) == 0
    rows = list(csv.DictReader(open(os.path.join(run, 'log.csv'))))
    assert [int(r['epoch']) for r in rows] == [1, 2]
    for k in ('loss', 'acc', 'nfe-f', 'nfe-b', 'test_loss', 'test_acc', 'test_nfe'):     # train.py:72,108
        assert k in rows[0]
    assert float(rows[0]['nfe-f']) >= 14 and float(rows[0]['nfe-b']) >= 15              # 2 + 6 k, 3 + 6 k
    ck = torch.load(os.path.join(run, 'last.pth'), map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'params', 'model', 'optim', 'metrics'} and ck['epoch'] == 2   # train.py:182-188
    assert 'odeblock.odefunc.conv1._layer.weight' in ck['model']
    # torch.optim.SGD's state layout: a reference checkpoint's optimizer state loads, and this one loads there
    st = ck['optim']['state']
    assert all('momentum_buffer' in v for v in st.values()) and ck['optim']['param_groups'][0]['momentum'] == 0.9
    import neural_ode_features_amd as nof
    net = nof.ODENet(1, out=10, n_filters=16, adjoint=True)
    ref_opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    ref_opt.load_state_dict(ck['optim'])
    # a run that already has a log is skipped unless --resume (train.py:116-118)
    assert T.main(common + ['-e', '3']) == 0
    assert len(list(csv.DictReader(open(os.path.join(run, 'log.csv'))))) == 2
    assert T.main(common + ['-e', '3', '--resume']) == 0
    rows = list(csv.DictReader(open(os.path.join(run, 'log.csv'))))
    assert [int(r['epoch']) for r in rows] == [1, 2, 3]
    ck3 = torch.load(os.path.join(run, 'last.pth'), map_location='cpu', weights_only=False)
    # cosine schedule after a resume: whatever torch's CosineAnnealingLR(optimizer, T, last_epoch=start_epoch - 2)
    # (train.py:163) does to a torch.optim.SGD restored from the epoch-2 checkpoint, it must do the same here
    import warnings
    from torch.optim.lr_scheduler import CosineAnnealingLR
    lin = torch.nn.Linear(2, 2)
    o1 = torch.optim.SGD(lin.parameters(), lr=0.05, momentum=0.9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        s1 = CosineAnnealingLR(o1, 4)
        s1.step()                                   # end of epoch 1 -> the state the epoch-2 checkpoint holds
        saved = o1.state_dict()
        o2 = torch.optim.SGD(lin.parameters(), lr=0.05, momentum=0.9)
        o2.load_state_dict(saved)
        CosineAnnealingLR(o2, 4, last_epoch=3 - 2)
    assert abs(ck3['optim']['param_groups'][0]['lr'] - o2.param_groups[0]['lr']) < 1e-12
    assert abs(ck['optim']['param_groups'][0]['lr'] - saved['param_groups'][0]['lr']) < 1e-12


def test_train_with_deferred_completion_matches_the_read_back_loop(tmp_path):
    """`--deferred` (integrate.DeferredLoop): the same epochs without a read-back per batch.  Every batch is tallied
    once (a missed solve is repeated, not skipped), the NFE sums are the synchronous loop's, and loss / accuracy
    agree with it to the noise of the stem's library convolutions."""
    from neural_ode_features_amd import train as T
    common = ['--dataset', 'mnist', '-f', '16', '-b', '32', '--synthetic-size', '160', '-a', '--lr', '0.05', '--wd', '1e-4',
              '--lrschedule', 'fixed', '-e', '2']
    assert T.main(common + ['--run-dir', str(tmp_path / 'sync')]) == 0
    assert T.main(common + ['--run-dir', str(tmp_path / 'blind'), '--deferred']) == 0
    a = list(csv.DictReader(open(os.path.join(str(tmp_path / 'sync'), 'log.csv'))))
    b = list(csv.DictReader(open(os.path.join(str(tmp_path / 'blind'), 'log.csv'))))
    assert len(a) == len(b) == 2
    for ra, rb in zip(a, b):
        assert abs(float(ra['loss']) - float(rb['loss'])) < 2e-2 * max(1.0, abs(float(ra['loss'])))
        assert abs(float(ra['nfe-f']) - float(rb['nfe-f'])) <= 6.0 and abs(float(ra['nfe-b']) - float(rb['nfe-b'])) <= 6.0
        assert abs(float(ra['test_acc']) - float(rb['test_acc'])) <= 0.1


def test_features_and_nfe_evaluations_on_a_trained_run(tmp_path):
    """evaluate.py:24-142 on the HIP backend: dense-output feature extraction over a tolerance sweep, and the bs=1
    NFE census, from a run directory written by the training loop."""
    import numpy as np
    import pandas as pd
    from neural_ode_features_amd import evaluate as E
    from neural_ode_features_amd import train as T
    run = str(tmp_path / 'run')
    assert T.main(['--dataset', 'mnist', '-f', '16', '-b', '32', '--synthetic-size', '64', '-a', '--lr', '0.05', '-e', '1',
                   '--run-dir', run]) == 0
    out = E.main(['features', run, '--t1', '0', '0.25', '0.5', '1', '--tol', '1e-3', '1e-1', '--limit', '24'])
    z = np.load(out)
    assert z['features'].shape == (2, 4, 24, 16) and z['y_true'].shape == (24,)          # [tols, T, N, C] (evaluate.py:84-88)
    assert np.array_equal(z['features'][0, 0], z['features'][1, 0])                      # t = 0: the stem's output, no solve
    assert not np.allclose(z['features'][0, -1], z['features'][0, 0])
    assert np.abs(z['features'][0] - z['features'][1]).max() < 0.5                       # the two tolerances agree roughly
    out = E.main(['nfe', run, '--t1', '0.5', '1', '--tol', '1e-3', '1e-1', '--limit', '6'])
    df = pd.read_csv(out)
    assert list(df.columns) == ['y_true', 'y_pred', 'nfe', 't1', 'tol'] and len(df) == 2 * 2 * 6   # evaluate.py:121-140
    assert ((df.nfe - 2) % 6 == 0).all() and (df.nfe >= 8).all()                         # show.py:199: NFE = 2 + 6 steps
    assert df[df.tol == 1e-1].nfe.mean() <= df[df.tol == 1e-3].nfe.mean()
    # evaluate.py:145-204: tol x t1 trade-off at the run's batch size, the live block mutated between forwards
    out = E.main(['tradeoff', run, '--t1', '0.5', '1', '--tol', '1e-3', '1e-1', '--limit', '64'])
    df = pd.read_csv(out)
    assert list(df.columns) == ['t1', 'test_loss', 'test_acc', 'test_nfe', 'test_tol'] and len(df) == 4
    assert ((df.test_nfe - 2) % 6 == 0).all() and (df.test_acc >= 0).all() and (df.test_acc <= 1).all()
    assert float(df[(df.test_tol == 1e-1)].test_nfe.mean()) <= float(df[(df.test_tol == 1e-3)].test_nfe.mean())
    # evaluate.py:207-305: loss / accuracy of the classifier at every time slice of ONE dense-output solve per batch and tolerance
    out = E.main(['accuracy', run, '--tol', '1e-3', '1e-1', '--limit', '64'])
    df = pd.read_csv(out)
    assert list(df.columns) == ['t1', 'test_loss', 'test_acc', 'test_nfe', 'test_tol'] and len(df) == 2 * 21      # t1 = 0, .05, ..., 1
    assert np.allclose(df.t1.values[:21], np.arange(0, 1.05, .05), atol=1e-6)
    a, b = df[df.test_tol == 1e-3], df[df.test_tol == 1e-1]
    assert abs(float(a.test_loss.values[0]) - float(b.test_loss.values[0])) < 1e-6        # the slice at t = 0 is the stem's output: no solve in it
    assert np.abs(a.test_loss.values - b.test_loss.values).max() < 0.2                     # ... and the two tolerances agree roughly everywhere
    # the slice at t1 = 1 of the dense-output solve is what the trade-off run computed at t1 = 1, tol 1e-3 (same solve, same end point)
    tr = pd.read_csv(os.path.join(run, 'tradeoff.csv'))
    assert abs(float(a.test_acc.values[-1]) - float(tr[(tr.t1 == 1.0) & (tr.test_tol == 1e-3)].test_acc.values[0])) < 1e-6
