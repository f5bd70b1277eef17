"""One rank of the GLOBAL-NORM tests (started by bench.spawn_ranks): every rank integrates its shard of a batch, the sums behind
every step decision are added over the ranks, so all ranks must take IDENTICAL steps -- and, for the forward solve (one sharded
segment), exactly the steps of one process that integrates the whole batch.

    dp_gnorm_child.py <out_dir> cpu     the oracle on the CPU with its `norm_reduce` option (the checker of the mode's definition)
    dp_gnorm_child.py <out_dir> hip     the HIP path: `options={'global_norm': True}`; the ranks share cuda:0, collectives over gloo
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    from tests.helpers import make_func
    C, per = (16, 2) if mode == 'cpu' else (64, 8)
    dev = 'cpu' if mode == 'cpu' else 'cuda'
    if mode == 'hip':
        torch.cuda.set_device(0)
    f, twin = make_func(C, seed=61, device=dev, kink_free=True)
    gen = torch.Generator().manual_seed(62)
    y = torch.randn(per * world, C, 8, 8, generator=gen)
    wgt = torch.randn(2, per * world, C, 8, 8, generator=gen) / (C * 64) ** 0.5
    # samples of very different size: the ranks' LOCAL norms would disagree about every step
    y[:per] *= 3.0
    ys, ws = nof.dp.shard_batch(y, rank, world), torch.stack([nof.dp.shard_batch(wgt[0], rank, world), nof.dp.shard_batch(wgt[1], rank, world)])
    t = torch.tensor([0.0, 1.0])
    tol = 1e-4

    def reduce_sum(v):
        v = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return v

    res = {}
    if mode == 'cpu':
        for name, opts, yy, ww in (('global', {'norm_reduce': (reduce_sum, world)}, ys, ws), ('local', {}, ys, ws)) + \
                ((('full', {}, y, wgt),) if rank == 0 else ()):
            fs, bs = tdq.SolverStats(), tdq.SolverStats()
            y0 = yy.clone().requires_grad_(True)
            out = tdq.odeint_adjoint(twin, y0, t, rtol=tol, atol=tol, method='dopri5', options=dict(opts), fwd_stats=fs, bwd_stats=bs)
            (out * ww).sum().backward()
            res[name] = dict(fwd=list(zip(fs.dts, fs.accepts)), bwd=list(zip(bs.dts, bs.accepts)), out=out.detach()[-1].clone(), gy=y0.grad.clone())
            for p in twin.parameters():
                p.grad = None
    else:
        for name, opts, yy, ww in (('global', {'global_norm': True, 'record_dt': 256}, ys, ws), ('local', {'record_dt': 256}, ys, ws)) + \
                ((('full', {'record_dt': 256}, y, wgt),) if rank == 0 else ()):
            y0 = yy.cuda().requires_grad_(True)
            out = nof.odeint_adjoint(f, y0, t.cuda(), rtol=tol, atol=tol, method='dopri5', options=dict(opts))
            (out * ww.cuda()).sum().backward()
            torch.cuda.synchronize()
            fs, bs = f.last_forward_stats, f.last_backward_stats
            res[name] = dict(fwd=list(zip(fs['dts'], fs['accepts'])), bwd=list(zip(bs['dts'], bs['accepts'])), out=out.detach()[-1].cpu(),
                             gy=y0.grad.cpu())
            for p in f.parameters():
                p.grad = None
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
