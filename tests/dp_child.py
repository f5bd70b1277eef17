"""One rank of the world-size-N CPU data-parallel test (started by bench.spawn_ranks from tests/test_dp_gloo.py).

Drives bench.py's OWN train step + GradientReducer over `gloo`; the HIP solve inside the ODE block is stood in for
by the oracle (test infrastructure -- the product path has no CPU solver)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, steps, accumulate = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    torch.manual_seed(23 + rank)                     # ranks start from different weights; broadcast must fix that
    net = nof.ODENet(1, out=10, n_filters=8, downsample='residual', method='rk4', tol=1e-3, adjoint=True, dropout=0)
    net.odeblock.odeint = tdq.odeint_adjoint
    nof.dp.broadcast_parameters(net, 0)
    reducer = nof.dp.GradientReducer(net, bucket_bytes=1 << 16)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(4 * world, 1, 28, 28, generator=gen)
    y = torch.randint(0, 10, (4 * world,), generator=gen)
    xs, ys = nof.dp.shard_batch(x, rank, world), nof.dp.shard_batch(y, rank, world)
    losses = []
    for _ in range(steps):
        if accumulate:                               # two micro-batches per optimizer step (train.py:56-58)
            import torch.nn.functional as F
            half = xs.shape[0] // 2
            with reducer.accumulate():
                (F.cross_entropy(net(xs[:half]), ys[:half]) / 2).backward()
            loss = F.cross_entropy(net(xs[half:]), ys[half:]) / 2
            loss.backward()
            reducer.finish()
            opt.step()
            opt.zero_grad()
        else:
            loss, _, _ = bench.train_step(net, opt, xs, ys, reducer)
        losses.append(float(loss))
    # one more backward + reduce, not stepped: the reduced gradients are read where the all-reduce left them
    import torch.nn.functional as F
    F.cross_entropy(net(xs), ys).backward()
    reducer.finish()
    for b in reducer.buckets:
        for p, v in zip(b.params, b.views):
            assert p.grad.data_ptr() == v.data_ptr()
    torch.save({'params': {k: v.detach().clone() for k, v in net.named_parameters()}, 'losses': losses},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
