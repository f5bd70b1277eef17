"""One rank of the two-rank deferred-completion test on ONE GPU (started by bench.spawn_ranks from
tests/test_gpu_deferred.py): the ranks share cuda:0, collectives run over gloo.  A MISS is forced on rank 1 ONLY;
the flag rides in the reducer's last bucket (dp.GradientReducer.carry_flag), so BOTH ranks must skip that update and
their parameters must stay bit-identical -- before, at and after the skipped step."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch.nn.functional as F
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    torch.manual_seed(23 + rank)                  # different initial weights: the broadcast must fix that
    net = nof.ODENet(3, out=10, n_filters=32, downsample='residual', method='dopri5', tol=1e-3, adjoint=True, dropout=0).to(dev)
    nof.dp.broadcast_parameters(net, 0)
    reducer = nof.dp.GradientReducer(net, average=False)
    opt = nof.FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.grad_scale = 1.0 / world
    d = integrate.Deferred(dev)
    opt.use_deferred(d, reducer)
    gen = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(8, 3, 32, 32, generator=gen).to(dev)
    y = torch.randint(0, 10, (8,), generator=gen).to(dev)

    def step(xx):
        loss = F.cross_entropy(net(xx), y)
        loss.backward()
        reducer.finish()
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()

    def same_on_all_ranks(flat):
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))

    with d:
        step(x)                                    # read-back: learns the step counts
        p1 = step(x)                               # blind, exact
        p2 = step(x)
        assert d.blind_solves >= 4 and d.resolve() == 0, (rank, d.blind_solves, d.misses)
        assert not torch.equal(p1, p2) and same_on_all_ranks(p2)
        d.resolve()
        if rank == 1:                              # a miss on THIS rank only: one step, no spare, on a much stiffer input
            d.force_counts(1)
        p3 = step(x * 40.0 if rank == 1 else x)
        assert torch.equal(p2, p3), 'rank %d committed an update although rank 1 missed' % rank
        assert same_on_all_ranks(p3)
        assert float(d.miss_flag) == 0.0           # reset behind the (skipped) optimizer launch
        p4 = step(x)                               # rank 1 finds its miss, re-learns with a read-back; both commit
        misses = torch.tensor([float(d.resolve())], device=dev)
        dist.all_reduce(misses)
        assert not torch.equal(p3, p4) and same_on_all_ranks(p4)
        assert int(misses.item()) >= 1 and (d.misses >= 1) == (rank == 1), (rank, d.misses)
        p5 = step(x)
        assert not torch.equal(p4, p5) and same_on_all_ranks(p5) and bool(torch.isfinite(p5).all())
    print('rank %d ok: blind solves %d, own misses %d' % (rank, d.blind_solves, d.misses), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
