"""GPU: the fused SGD step (`node_sgd_step` through the C ABI, optim.FusedSGD) against torch.optim.SGD -- the
reference's optimizer (train.py:136), stepped and zeroed per iteration (train.py:56-58)."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_fused_sgd_matches_torch_sgd_on_random_gradients():
    import neural_ode_features_amd as nof
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3), torch.nn.GroupNorm(7, 7), torch.nn.Flatten(), torch.nn.Linear(7 * 36, 5)).cuda()
    ref = copy.deepcopy(net)
    opt = nof.FusedSGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator(device='cuda').manual_seed(4)
    big = torch.empty(4096, device='cuda')
    for step in range(5):
        off = 1                                   # gradients at odd offsets of a shared buffer: the unaligned path
        for i, (p, q) in enumerate(zip(net.parameters(), ref.parameters())):
            g = torch.randn(p.shape, generator=gen, device='cuda')
            if step % 2 and p.numel() < 512:
                view = big[off:off + p.numel()].view(p.shape)
                view.copy_(g)
                p.grad = view
                off += p.numel() + 1
            else:
                p.grad = g.clone()
            q.grad = g.clone()
        if step == 3:
            list(net.parameters())[1].grad = None          # a parameter without gradient is skipped, like torch.optim.SGD
            list(ref.parameters())[1].grad = None
        opt.step()
        opt.zero_grad()
        ropt.step()
        ropt.zero_grad()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), step
            assert p.grad is None


def test_fused_sgd_on_a_real_model_matches_torch_sgd():
    """Three training steps of a small ODE-Net (HIP solver): the gradients of every step are handed to BOTH optimizers
    (FusedSGD on the trained net, torch.optim.SGD on a twin that never runs backward -- MIOpen's weight-gradient
    kernels accumulate with atomics, so two backward passes of identical nets differ in the last bits and three
    momentum steps amplify that; the optimizer is what is under test here)."""
    import bench
    import neural_ode_features_amd as nof
    torch.manual_seed(5)
    a = nof.ODENet(3, out=10, n_filters=32, downsample='residual', method='rk4', tol=1e-3, adjoint=True, dropout=0).cuda()
    b = copy.deepcopy(a)
    oa = nof.FusedSGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    ob = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(8, 3, 32, 32, generator=gen).cuda()
    y = torch.randint(0, 10, (8,), generator=gen).cuda()
    for step in range(3):
        F.cross_entropy(a(x), y).backward()
        for p, q in zip(a.parameters(), b.parameters()):
            q.grad = p.grad.detach().clone()
        oa.step()
        oa.zero_grad()
        ob.step()
        ob.zero_grad()
        for (k, p), q in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (step, k)
    # momentum state in torch.optim.SGD's layout, interchangeable both ways
    ob.load_state_dict(oa.state_dict())
    oa.load_state_dict(ob.state_dict())
    # learning-rate schedulers drive it through param_groups like a torch optimizer (train.py:158-163)
    oa.param_groups[0]['lr'] = 0.0
    before = [p.detach().clone() for p in a.parameters()]
    bench.train_step(a, oa, x, y)
    for p, q in zip(a.parameters(), before):
        assert torch.equal(p, q)


def test_fused_sgd_without_momentum_keeps_no_buffer_and_rejects_unsupported_modes():
    import neural_ode_features_amd as nof
    torch.manual_seed(4)
    ps = [torch.randn(257, device='cuda', requires_grad=True), torch.randn(8, 16, device='cuda', requires_grad=True)]
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    for p, r in zip(ps, ref):
        g = torch.randn_like(p)
        p.grad, r.grad = g.clone(), g.clone()
    a = nof.FusedSGD(ps, lr=0.05, momentum=0.0, weight_decay=1e-3)
    b = torch.optim.SGD(ref, lr=0.05, momentum=0.0, weight_decay=1e-3)
    a.step(); b.step()
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r, rtol=1e-6, atol=1e-7)
        assert torch.equal(p.grad, r.grad)                              # the gradients are left as autograd wrote them
    assert all('momentum_buffer' not in a.state[p] for p in ps)         # torch.optim.SGD keeps none either
    for bad in (dict(nesterov=True), dict(dampening=0.1), dict(maximize=True)):
        a.param_groups[0].update(bad)
        with pytest.raises(ValueError):
            a.step()
        a.param_groups[0].update(dict(nesterov=False, dampening=0, maximize=False))


def test_sgd_step_argument_checks():
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    t = torch.zeros(64, device='cuda')
    row = _lib.NodeSgdTensor(t.data_ptr(), t.data_ptr(), t.data_ptr(), 64)
    assert lib.node_sgd_step(None, 1, 0.1, 0.9, 0.0, 1.0, None, None) == -1
    assert lib.node_sgd_step((_lib.NodeSgdTensor * 1)(_lib.NodeSgdTensor(t.data_ptr(), None, t.data_ptr(), 64)), 1, 0.1, 0.9, 0.0, 1.0, None, None) == -1
    assert lib.node_sgd_step((_lib.NodeSgdTensor * 1)(_lib.NodeSgdTensor(t.data_ptr() + 2, t.data_ptr(), t.data_ptr(), 8)), 1, 0.1, 0.9, 0.0, 1.0, None, None) == -9
    assert lib.node_sgd_step((_lib.NodeSgdTensor * 1)(row), 1, -0.1, 0.9, 0.0, 1.0, None, None) == -9
    assert lib.node_sgd_step((_lib.NodeSgdTensor * 1)(row), 0, 0.1, 0.9, 0.0, 1.0, None, None) == 0


@pytest.mark.timeout(900)
def test_bench_two_ranks_through_self_launcher_on_one_gpu():
    """The N-rank path of bench.py end to end on a 1-GPU box: `--gpus 2` spawns two ranks itself (they share cuda:0,
    collectives over gloo because RCCL refuses two ranks on one device): head captured as hipGraphs, reducer packing
    the gradients into bucket buffers, all-reduce, FusedSGD reading them in place with 1/world folded in.  The JSON
    line must say n_gpus = 2; its throughput is not a measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--share-gpu',
                        '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--no-roofline', '--batch', '32'],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 64 and d['value'] > 0
    assert [r['rank'] for r in d['config']['per_rank']] == [0, 1] and all(r['ms_per_step_own'] > 0 for r in d['config']['per_rank'])
    assert 'SHARE' in d['config']['parallelism'] and d['config']['step_norm'].startswith('local')
    # ... and with GLOBAL-NORM solves (dp.enable_global_norm): deferred completion keeps working with the hook inside the solves, and both
    # ranks report the SAME step counts for every block (different synthetic shards: with local norms they need not)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--share-gpu', '--global-norm',
                        '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--no-roofline', '--batch', '32'],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['config']['step_norm'].startswith('global')
    pr = d['config']['per_rank']
    assert pr[0]['last_steps_fwd_bwd_per_block'] == pr[1]['last_steps_fwd_bwd_per_block'], pr


@pytest.mark.timeout(900)
def test_bench_one_rank_through_rccl():
    """What a 1-GPU box can prove about the multi-GPU path on the REAL backend: `bench.py --force-dist` runs a world of
    one rank through init_process_group('nccl', device_id=...), parameter broadcast, the reducer's bucket packing and
    asynchronous all-reduces (RCCL, sum over one rank), the deferred-completion flag slot riding in the last bucket, and
    FusedSGD reading the reduced gradients in place.  The rate must match the plain single-GPU path within a few percent
    (the collectives are local copies): it is printed, not asserted, because boxes differ."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '6', '--warmup', '3',
                        '--no-cpu-baseline', '--no-roofline', '--no-dropin', '--no-other-configs', '--no-latency'], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    print('one rank through RCCL:', d['value'], 'images/s;', d['config']['collectives'], d['config']['solver_completion'])
    assert d['n_gpus'] == 1 and 'RCCL' in d['config']['collectives'] and d['value'] > 0
    assert 'deferred' in d['config']['solver_completion'] and d['config']['retries'] == 0 and d['config']['miss_events'] == 0
    assert d['config']['per_rank'][0]['rank'] == 0 and d['config']['per_rank'][0]['ms_per_step_own'] > 0


@pytest.mark.timeout(900)
def test_bench_line_carries_the_contract_keys():
    """`python bench.py` (N = 1): ONE JSON line with the driver's keys, the roofline object of the dominant kernel and
    the CPU baseline; the roofline fraction is a utilisation (<= 1), the algorithmic figure lives under its own key."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    # (the default line also carries short side runs of configs 3 and 5 -- a minute of child processes; tools/r05_bench_lines.sh has them)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '3', '--no-other-configs'],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'step_ms'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['warmup'] == 3 and d['unit'] == 'images/sec' and d['dtype'] == 'f32'
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic'):
        assert k in rf, k
    # the dominant kernel of configs[1] is the component GEMM of the F(4x4,3x3) pipeline on the bf16 matrix pipe
    # (k_w4_gemm64b): left of the bf16 ridge, so the HBM side bounds it; the matrix-pipe view rides along under `mfma`
    assert rf['bound'] == 'hbm' and rf['unit'] == 'TB/s' and rf['peak'] == 8.0 and 0.1 < rf['frac'] < 1.0
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert 0.05 < rf['mfma']['frac_of_bf16_peak'] < 1.0 and rf['mfma']['fp32_equivalent_tflops'] > 50
    # the HBM-bound passes (both fractions), the drop-in rate of the same run, the settle accounting
    assert rf['hbm']['bound'] == 'hbm' and 0.05 < rf['hbm']['frac'] < 1.0 and rf['hbm']['all_passes']['ms_per_step'] > 0
    assert d['dropin']['value'] > 0 and d['dropin']['value'] < 1.05 * d['value']
    assert d['config']['settle_steps'] >= 0 and d['config']['dead_steps_per_step'] >= 0 and d['config']['retries'] >= 0
    assert d['fresh_batches']['value'] > 0 and d['fresh_batches']['retries'] >= 0      # a new batch every step, misses repeated
    assert len(d['config']['per_rank']) == 1 and d['config']['per_rank'][0]['last_steps_fwd_bwd_per_block'][0][0] >= 1
    # the direct-convolution count behind the launch lives under its own key; a rate against it is quoted only where
    # one launch IS the whole convolution (the fused kernels), not for the component GEMMs of the F(4x4,3x3) pipeline
    assert rf['algorithmic']['flops_per_launch'] > rf['flops_per_launch']
    assert 'achieved' not in rf['algorithmic'] or rf['algorithmic']['achieved'] > rf['achieved']
    assert rf['traffic'] is None or rf['traffic'] > 1e7          # counted by rocprofv3 --pmc child passes in this run, or null
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['value'] > 0 and d['value'] > 10 * cb['value']
    assert abs(d['value'] - 128 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    # the side block of the other regime of the path (bs = 1 census, evaluate.py:97-142): cfg 2's state is one resident launch per solve
    lat = d['latency_bs1']
    assert 'error' not in lat, lat
    assert lat['state'] == [1, 256, 8, 8] and lat['one_launch_per_solve'] in (True, False)
    assert all(row['nfe'] >= 14 and row['us_per_evaluation'] > 0.0 for row in lat['solves']), lat     # (no wall-clock bound: the box may be shared)
