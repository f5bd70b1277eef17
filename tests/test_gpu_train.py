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
