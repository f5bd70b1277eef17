"""Batch data-parallel training of the ODE-Net: one process per GPU, gradients
all-reduced over RCCL (backend "nccl" on ROCm) across the xGMI mesh.

The reference is single-process (`/root/reference/train.py:230`); sharding the
batch is new work.  The path shards cleanly: `ODEfunc` has
no cross-sample op (GroupNorm is per-sample), so each rank integrates its own
shard with its own adaptive steps ("local-norm" mode -- exactly what wrapping the
reference in DDP would do) and the only exchange step is the gradient sum.

Layout: every bucket owns ONE persistent flat fp32 buffer.  When the last gradient of a bucket lands, the bucket's
gradients (which autograd left wherever it allocated them) are packed into that buffer with one launch, each
parameter's `.grad` is re-pointed at its slice (host only), and the all-reduce runs in place -- nothing is copied
back: the optimizer reads the reduced gradients from the bucket (`optim.FusedSGD` takes raw pointers).

Overlap: parameters are grouped into buckets in *reverse* registration order
(head first -- its gradients are ready before the adjoint solve starts).  A
post-accumulate-grad hook launches the bucket's all-reduce asynchronously as
soon as its last gradient lands, so the head's bucket travels over xGMI while
the adjoint ODE solve runs, and the ODE block's bucket (only final once the
reverse solve reaches t0) travels during the stem's backward.  On a fully
connected 8-GPU xGMI node each all-reduce is per-link bound; the ODE bucket
(4.75 MB at C=256) is kept whole so RCCL can split it over all 7 links.

    reducer = GradientReducer(model)          # after dist.init_process_group
    loss.backward()                           # hooks pack + launch async all-reduces
    reducer.finish()                          # wait (and average, unless the optimizer folds 1/world in)
    optimizer.step(); optimizer.zero_grad()

GLOBAL-NORM mode (opt-in, SURVEY.md 8e collective 2): `enable_global_norm(model)` makes every ODE block of the model add, over the
ranks, the sums each step decision is taken from (one 32-byte all-reduce per step, two per initial step) -- all ranks then take
IDENTICAL steps (no straggler rank holding the gradient all-reduce up) and the sharded state segments are controlled by exactly the
mixed norm of the unsharded batch, which is what a single process at the global batch size does (`train.py:40-58` at bs = 1024).

One backward per `finish()`; micro-batches that accumulate (`train.py:56-58`) run
under `with reducer.accumulate():` for all but the last backward.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn


def enable_global_norm(model: nn.Module, process_group=True) -> int:
    """Switch every ODE block of `model` (the package's `ODEBlock`: anything with a `global_norm` attribute and an `odefunc`) to
    GLOBAL-NORM solves over `process_group` (True = the default group; None / False switches the mode off again).  Returns the
    number of blocks touched.  The C side: `node_solve_opts::norm_reduce` (include/node_hip.h)."""
    n = 0
    for m in model.modules():
        if hasattr(m, 'global_norm') and hasattr(m, 'odefunc'):
            m.global_norm = process_group if process_group else None
            n += 1
    return n


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard `rank` of `world` along the batch dimension."""
    n = x.shape[0]
    if n % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (n, world))
    per = n // world
    return x[rank * per:(rank + 1) * per]


class _Bucket:
    def __init__(self, index: int, params: List[nn.Parameter]):
        self.index = index
        self.params = params
        self.flat: Optional[torch.Tensor] = None      # persistent [sum numel] buffer; .grad of every param views it
        self.views: List[torch.Tensor] = []
        self.fired = 0            # hooks seen since the last finish() (outside accumulate())
        self.work = None
        self.launched = False
        self.extra = 0            # trailing slots behind the gradients (the deferred-completion miss flag)
        self.payload: Optional[torch.Tensor] = None   # flat[:sum numel]

    def ensure_flat(self):
        if self.flat is not None and self.flat.device == self.params[0].device:
            return
        p0 = self.params[0]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total + self.extra, dtype=p0.dtype, device=p0.device)
        self.payload = self.flat[:total]
        self.views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n

    def reset(self):
        self.fired = 0
        self.work = None
        self.launched = False


class GradientReducer:
    """Bucketed, overlapped gradient reduction without a copy-back (see the module docstring)."""

    def __init__(self, model: nn.Module, bucket_bytes: int = 32 << 20, process_group=None,
                 boundaries: Optional[Iterable[nn.Module]] = None, average: bool = True, limits=None,
                 collectives_at_world_1: bool = False):
        """`average=False` leaves the SUM in the buckets for an optimizer that folds 1/world into its step
        (`optim.FusedSGD.grad_scale`).  `limits`: {sub-module: bucket bytes} overriding `bucket_bytes` for the
        parameters of that sub-module (the stem's gradients arrive one layer at a time and are worth sending in
        pieces while its backward runs; the ODE block's arrive all at once and travel best as one bucket)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # run the packing + all-reduce machinery even with ONE rank (a sum over one rank): lets a 1-GPU box execute the
        # N-rank code path on the real RCCL backend (bench.py --force-dist, tests/test_gpu_optim.py)
        self.always = bool(collectives_at_world_1) and dist.is_initialized()
        self.average = average
        params = [p for p in model.parameters() if p.requires_grad]
        # cut buckets at sub-module boundaries (head | ode block | stem) first, then by size and dtype/device
        owner = {}
        if boundaries is None:
            boundaries = [m for _, m in model.named_children()]
        boundaries = list(boundaries)
        limit_of = {}
        for bi, m in enumerate(boundaries):
            for p in m.parameters():
                owner[id(p)] = bi
            for lm, nbytes_limit in (limits or {}).items():
                if lm is m:
                    limit_of[bi] = int(nbytes_limit)
        self.buckets: List[_Bucket] = []
        cur, cur_bytes, cur_key = [], 0, None
        for p in reversed(params):
            key = (owner.get(id(p), -1), p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (key != cur_key or cur_bytes + nbytes > limit_of.get(key[0], bucket_bytes)):
                self.buckets.append(_Bucket(len(self.buckets), cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_key = key
        if cur:
            self.buckets.append(_Bucket(len(self.buckets), cur))
        self._slot = {}
        self._hooks = []
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._slot[id(p)] = (b, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.launch_order: List[int] = []   # bucket indices in the order their all-reduce was issued
        self._accumulating = False
        self._flag: Optional[torch.Tensor] = None

    def carry_flag(self, flag: torch.Tensor) -> torch.Tensor:
        """Append a one-float slot to the LAST bucket (the one launched last in a backward pass), filled with `flag`
        (a device float, e.g. `integrate.Deferred.miss_flag`) right before that bucket's all-reduce.  Returns the
        slot: after `finish()` it holds the SUM of the ranks' flags -- non-zero on every rank if any rank's was."""
        b = self.buckets[-1]
        b.extra = 1
        b.flat = None
        b.ensure_flat()
        self._flag = flag
        return b.flat[-1:]

    def zero_grad(self):
        """`optimizer.zero_grad()` (set to None) for the reducer's parameters: no kernel; the next backward's
        gradients land in fresh tensors and are packed into the bucket buffers again."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None

    @contextlib.contextmanager
    def accumulate(self):
        """Backward passes inside this context only accumulate locally (micro-batching, train.py:56-58)."""
        prev, self._accumulating = self._accumulating, True
        try:
            yield
        finally:
            self._accumulating = prev

    # -- hooks ---------------------------------------------------------------
    def _on_grad(self, p: nn.Parameter):
        if self._accumulating:
            return
        b, _ = self._slot[id(p)]
        if b.launched:
            raise RuntimeError('GradientReducer: a gradient of bucket %d arrived after its all-reduce was launched -- '
                               'two backward() calls per finish(); wrap all but the last in `reducer.accumulate()`'
                               % b.index)
        b.fired += 1
        if b.fired == len(b.params):
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.launched = True
        if self.world == 1 and not self.always:
            return
        b.ensure_flat()
        # pack: gradients that already live in their slice (a caller that zeroes in place instead of dropping them)
        # stay; the rest go in with ONE launch when none is in place, one copy each otherwise
        loose = [(p, v) for p, v in zip(b.params, b.views) if p.grad is None or p.grad.data_ptr() != v.data_ptr()]
        if len(loose) == len(b.params) and all(p.grad is not None for p in b.params):
            torch.cat([p.grad.reshape(-1) for p in b.params], out=b.payload)
        else:
            for p, v in loose:
                if p.grad is None:
                    v.zero_()            # parameter unused this step
                else:
                    v.copy_(p.grad)
        for p, v in zip(b.params, b.views):
            p.grad = v                   # host only: the reduced gradient is read where the all-reduce leaves it
        if b.extra and self._flag is not None:
            b.flat[-1:].copy_(self._flag)    # this rank's miss flag rides behind the last bucket's gradients
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launch_order.append(b.index)

    def finish(self):
        """Wait for every in-flight all-reduce (and average in place unless `average=False`).  Buckets whose hooks
        did not all fire (parameters unused this step) are launched here, in bucket order -- the same order on
        every rank."""
        if self.world > 1 or self.always:
            for b in self.buckets:
                if not b.launched:
                    self._launch(b)
            for b in self.buckets:
                b.work.wait()
                if self.average:
                    b.payload.div_(self.world)
        for b in self.buckets:
            b.reset()
        self.launch_order = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(model: nn.Module, src: int = 0, process_group=None):
    """Replicate rank `src`'s parameters (ranks must start from identical weights)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for p in model.parameters():
        dist.broadcast(p.data, src=src, group=process_group)
    for bname, buf in model.named_buffers():
        dist.broadcast(buf.data, src=src, group=process_group)
