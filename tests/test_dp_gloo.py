"""Data-parallel path on CPU: world_size-2 `gloo` processes.  The DP wiring (sharding,
bucketed overlapped all-reduce, averaging) is host logic; the ODE solve inside each rank
is stood in for by the oracle (tests may), exactly as bench.py's cpu_baseline does."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(method='rk4'):
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    torch.manual_seed(23)
    net = nof.ODENet(1, out=10, n_filters=8, downsample='residual', method=method, tol=1e-3, adjoint=True, dropout=0)
    net.odeblock.odeint = tdq.odeint_adjoint
    return net


def _data(n):
    gen = torch.Generator().manual_seed(7)
    return torch.rand(n, 1, 28, 28, generator=gen), torch.randint(0, 10, (n,), generator=gen)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import neural_ode_features_amd as nof
    net = _build()
    if rank != 0:                       # ranks start from different weights; broadcast must fix that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.1)
    nof.dp.broadcast_parameters(net, 0)
    reducer = nof.dp.GradientReducer(net, bucket_bytes=1 << 16)
    x, y = _data(4 * world)
    xs, ys = nof.dp.shard_batch(x, rank, world), nof.dp.shard_batch(y, rank, world)
    loss = F.cross_entropy(net(xs), ys)
    loss.backward()
    order = list(reducer.launch_order)
    owners = [('classifier' if any(p is q for q in net.classifier.parameters()) else
               'odeblock' if any(p is q for q in net.odeblock.parameters()) else 'downsample')
              for p in [b.params[0] for b in reducer.buckets]]
    reducer.finish()
    grads = {k: v.grad.clone() for k, v in net.named_parameters()}
    torch.save({'grads': grads, 'order': order, 'owners': owners, 'nfe': net.nfe()},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_matches_full_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'), weights_only=False)
    for k in r0['grads']:
        assert torch.equal(r0['grads'][k], r1['grads'][k]), k        # every rank holds the same averaged gradient
    # single-process full batch (rk4: fixed steps, so sharding does not change the arithmetic)
    net = _build()
    x, y = _data(4 * world)
    F.cross_entropy(net(x), y).backward()
    for k, v in net.named_parameters():
        assert torch.allclose(r0['grads'][k], v.grad, rtol=2e-4, atol=2e-6), k
    # overlap order: buckets are launched head first, then the ODE block, then the stem
    assert r0['order'] == sorted(r0['order'])
    launched = [r0['owners'][i] for i in r0['order']]
    assert launched[0] == 'classifier' and launched[-1] == 'downsample' and 'odeblock' in launched
    assert launched.index('odeblock') < launched.index('downsample')


def test_shard_batch_and_single_process_reducer():
    import neural_ode_features_amd as nof
    x = torch.arange(12).view(12, 1)
    assert nof.dp.shard_batch(x, 1, 4).flatten().tolist() == [3, 4, 5]
    with pytest.raises(ValueError):
        nof.dp.shard_batch(x, 0, 5)
    lin = torch.nn.Linear(3, 2)
    red = nof.dp.GradientReducer(lin)          # world size 1: hooks fire, nothing is communicated
    lin(torch.ones(1, 3)).sum().backward()
    g = lin.weight.grad.clone()
    red.finish()
    assert torch.equal(lin.weight.grad, g)
    red.remove()


def _single_process_reference(steps, world):
    import bench
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    torch.manual_seed(23)
    net = nof.ODENet(1, out=10, n_filters=8, downsample='residual', method='rk4', tol=1e-3, adjoint=True, dropout=0)
    net.odeblock.odeint = tdq.odeint_adjoint
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x, y = _data(4 * world)
    for _ in range(steps):
        bench.train_step(net, opt, x, y)
    return net


@pytest.mark.timeout(600)
@pytest.mark.parametrize('accumulate', [0, 1])
def test_bench_train_step_through_self_launcher(tmp_path, accumulate):
    """bench.py's launcher (`--gpus N` without torchrun) + its train step + the in-place reducer, world size 2 over
    gloo: after 3 optimizer steps every rank holds the parameters a single process gets from the full batch
    (rk4: fixed steps, so sharding / micro-batching does not change the arithmetic beyond summation order)."""
    import bench
    world, steps = 2, 3
    rc = bench.spawn_ranks(world, [sys.executable, os.path.join(ROOT, 'tests', 'dp_child.py'), str(tmp_path), str(steps),
                                   str(accumulate)], timeout=500)
    assert rc == 0
    r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'), weights_only=False)
    for k in r0['params']:
        assert torch.equal(r0['params'][k], r1['params'][k]), k
    ref = _single_process_reference(steps, world)
    for k, v in ref.named_parameters():
        assert torch.allclose(r0['params'][k], v.detach(), rtol=5e-4, atol=5e-6), k


def _check_global_norm_histories(tmp_path, mode):
    import bench
    rc = bench.spawn_ranks(2, [sys.executable, os.path.join(ROOT, 'tests', 'dp_gnorm_child.py'), str(tmp_path), mode], timeout=700)
    assert rc == 0
    r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'), weights_only=False)
    g0, g1, full = r0['global'], r1['global'], r0['full']
    # (1) every rank takes the SAME steps, forward and backward: sizes bit-identical, decisions identical
    assert g0['fwd'] == g1['fwd'] and g0['bwd'] == g1['bwd'], (g0['fwd'], g1['fwd'], g0['bwd'], g1['bwd'])
    # (2) with local norms the ranks do NOT agree on this batch (rank 0's samples are three times larger): the mode changes something
    assert r0['local']['fwd'] != r1['local']['fwd'] or r0['local']['bwd'] != r1['local']['bwd']
    # (3) the forward solve has one sharded segment: its global norm IS the norm of the unsharded batch -- the history of one process
    # integrating the whole batch: same decisions; the step sizes agree as far as an error ESTIMATE does between two batch shapes of the
    # same convolutions (the estimate is a difference of nearly equal terms: rounding moves it by ~1e-3, the step by its fifth root)
    assert [a for _, a in g0['fwd']] == [a for _, a in full['fwd']]
    for (d, _), (e, _) in zip(g0['fwd'], full['fwd']):
        assert abs(d - e) <= 2e-3 * abs(e), (d, e)
    per = g0['out'].shape[0]
    assert float((g0['out'] - full['out'][:per]).abs().max()) <= 2e-4 * float(full['out'].abs().max())     # (both within tol = 1e-4 of the solution)
    # (4) the adjoint solve: y and adj_y enter with the unsharded batch's norm, adj_params / adj_t (each rank's own partial sums) with
    # the mean of the ranks' ratios -- identical on all ranks by construction, within a decision or two of the one-process solve
    assert abs(len(g0['bwd']) - len(full['bwd'])) <= 2
    print(mode, 'global-norm histories: forward', len(g0['fwd']), 'steps, backward', len(g0['bwd']), '(one process, whole batch:',
          len(full['fwd']), '/', len(full['bwd']), '; local norms:', len(r0['local']['bwd']), '/', len(r1['local']['bwd']), ')')


@pytest.mark.timeout(900)
def test_global_norm_mode_definition_on_the_oracle(tmp_path):
    """SURVEY.md 8e, collective (2): two gloo ranks, each integrating its shard with the oracle's `norm_reduce` option (the CPU
    statement of the mode the HIP library implements behind node_solve_opts::norm_reduce): identical step histories on both ranks,
    equal to the single-process history of the whole batch for the forward solve."""
    _check_global_norm_histories(tmp_path, 'cpu')


def test_bench_refuses_smaller_world():
    """`python bench.py --gpus 2` must never measure one GPU: no devices here -> non-zero exit, no JSON line;
    a launcher environment that disagrees with --gpus is refused too."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'refusing' in (r.stderr + r.stdout) and '"metric"' not in r.stdout
    env['WORLD_SIZE'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_reducer_rejects_unannounced_second_backward():
    import neural_ode_features_amd as nof
    lin = torch.nn.Linear(3, 2)
    red = nof.dp.GradientReducer(lin)
    lin(torch.ones(1, 3)).sum().backward()
    with pytest.raises(RuntimeError, match='accumulate'):
        lin(torch.ones(1, 3)).sum().backward()
    red.finish()
    red.zero_grad()
    assert lin.weight.grad is None
    with red.accumulate():
        lin(torch.ones(1, 3)).sum().backward()
    lin(torch.ones(1, 3)).sum().backward()
    red.finish()
    assert torch.equal(lin.weight.grad, 2 * torch.ones(2, 3))
    red.remove()
