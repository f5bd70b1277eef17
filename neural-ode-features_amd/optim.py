"""The optimizer step of the training loop as ONE HIP launch.

The reference trains with `torch.optim.SGD(model.parameters(), lr, momentum=0.9, weight_decay=wd)`
(`/root/reference/train.py:136`), stepped and zeroed once per iteration (`train.py:56-58`).  `FusedSGD` does
the same arithmetic for every parameter tensor in one `node_sgd_step` launch per parameter group
(csrc/kernels_optim.hip): a table of (parameter, gradient, momentum) device pointers travels in the kernel
arguments, so gradients are read where autograd -- or the data-parallel reducer's all-reduce bucket
(`dp.GradientReducer`) -- left them.

It IS a `torch.optim.Optimizer`: the reference's LR schedulers (`LambdaLR`, `ReduceLROnPlateau`,
`CosineAnnealingLR`, train.py:158-163) drive it through `param_groups`, and its `state_dict()` has
torch.optim.SGD's layout (`state[i]['momentum_buffer']`), so `optimizer.load_state_dict(ckpt['optim'])`
(train.py:147) resumes a checkpoint written by the reference, and the reverse.

No CPU path: parameters must live on a HIP device and the step raises if libnode_hip.so is missing.
"""
from __future__ import annotations

import torch

from . import _lib


class FusedSGD(torch.optim.Optimizer):
    """`torch.optim.SGD(params, lr, momentum, weight_decay)` with dampening 0 and no Nesterov (train.py:136).

        opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        loss.backward(); opt.step(); opt.zero_grad()
    """

    def __init__(self, params, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError('lr, momentum and weight_decay must be non-negative')
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False,
                        maximize=False, foreach=None, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.grad_scale = 1.0        # dp.GradientReducer(average=False) leaves a SUM: set 1/world here
        # device float (1 element) or None: the step leaves everything untouched when it holds a non-zero value --
        # the commit point of a training step whose solves ran with deferred completion (integrate.Deferred)
        self.skip_flag = None
        self.flags_to_reset = []     # device tensors zeroed behind the step (the per-step miss flags): a step that forgot
                                     # to reset them could otherwise skip every later update silently

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            if group.get('nesterov') or group.get('dampening') or group.get('maximize'):
                raise ValueError('FusedSGD implements plain SGD with momentum and weight decay (train.py:136): nesterov, '
                                 'dampening and maximize are not supported (got %r)'
                                 % {k: group.get(k) for k in ('nesterov', 'dampening', 'maximize')})
            rows = []
            keep = []
            dev = None
            for p in group['params']:
                g = p.grad
                if g is None:                 # torch.optim.SGD skips parameters without a gradient
                    continue
                if not p.is_cuda:
                    raise RuntimeError('FusedSGD has no CPU path: parameters must live on a HIP device')
                if p.dtype != torch.float32 or g.dtype != torch.float32 or g.device != p.device:
                    raise TypeError('FusedSGD needs float32 parameters and gradients on one device')
                if not p.is_contiguous():
                    raise RuntimeError('FusedSGD needs contiguous parameters')
                dev = dev or p.device
                if p.device != dev:
                    raise RuntimeError('one parameter group must live on one device')
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                buf = st.get('momentum_buffer')
                if group['momentum'] == 0:
                    buf = None                # torch.optim.SGD keeps no buffer then: same state_dict, half the memory
                elif buf is None:             # torch's first step sets buf = grad: zero + one fused step does the same
                    buf = st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                keep.append(g)
                rows.append((p.data_ptr(), g.data_ptr(), buf.data_ptr() if buf is not None else None, p.numel()))
            if not rows:
                continue
            table = (_lib.NodeSgdTensor * len(rows))(*[_lib.NodeSgdTensor(*r) for r in rows])
            with torch.cuda.device(dev):
                _lib.check(lib.node_sgd_step(table, len(rows), float(group['lr']), float(group['momentum']),
                                             float(group['weight_decay']), float(self.grad_scale),
                                             self.skip_flag.data_ptr() if self.skip_flag is not None else None,
                                             torch.cuda.current_stream(dev).cuda_stream))
            del keep
        for f in self.flags_to_reset:
            f.zero_()
        return loss

    def use_deferred(self, deferred, reducer=None):
        """Predicate the update on `deferred`'s miss flag (see integrate.Deferred).  Under data parallelism the flag
        travels in the reducer's last bucket, so that every rank skips an update any rank missed."""
        if reducer is not None and (reducer.world > 1 or getattr(reducer, 'always', False)):
            self.skip_flag = reducer.carry_flag(deferred.miss_flag)
        else:
            self.skip_flag = deferred.miss_flag
        self.flags_to_reset = [deferred.miss_flag]
        deferred.armed = True
