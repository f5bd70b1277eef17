"""The optimizer step of the training loop as ONE HIP launch.

The reference trains with `torch.optim.SGD(model.parameters(), lr, momentum=0.9, weight_decay=wd)`
(`/root/reference/train.py:136`), stepped and zeroed once per iteration (`train.py:56-58`).  `FusedSGD` does
the same arithmetic for every parameter tensor in one `node_sgd_step` launch (csrc/kernels_optim.hip): a table
of (parameter, gradient, momentum) device pointers travels in the kernel arguments, so gradients are read where
autograd -- or the data-parallel reducer's all-reduce bucket (`dp.GradientReducer`) -- left them.  `zero_grad()`
drops the gradient tensors (what `torch.optim.Optimizer.zero_grad()` does by default): no kernel at all.

No CPU path: parameters must live on a HIP device and the step raises if libnode_hip.so is missing.
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from . import _lib


class FusedSGD:
    """`torch.optim.SGD(params, lr, momentum, weight_decay)` (dampening 0, no Nesterov: train.py:136).

        opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        loss.backward(); opt.step(); opt.zero_grad()
    """

    def __init__(self, params, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError('lr, momentum and weight_decay must be non-negative')
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError('FusedSGD needs float32 parameters')
        total = sum(p.numel() for p in self.params)
        self._momentum_flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self.momentum_bufs: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            self.momentum_bufs.append(self._momentum_flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        # same surface as torch optimizers where the reference touches it (LR schedulers read / write
        # `param_groups[0]['lr']`, train.py:158-163; checkpoints store `state_dict()`, train.py:18-23)
        self.param_groups = [{'params': self.params, 'lr': float(lr), 'initial_lr': float(lr), 'momentum': float(momentum),
                              'weight_decay': float(weight_decay), 'dampening': 0, 'nesterov': False}]
        self.grad_scale = 1.0        # dp.GradientReducer(average=False) leaves a SUM: set 1/world here

    def step(self):
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FusedSGD has no CPU path: parameters must live on a HIP device')
        if self._momentum_flat.device != dev:      # the model moved after the optimizer was built
            self._momentum_flat = self._momentum_flat.to(dev)
            off = 0
            for i, p in enumerate(self.params):
                self.momentum_bufs[i] = self._momentum_flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
        rows = []
        keep = []
        for p, m in zip(self.params, self.momentum_bufs):
            g = p.grad
            if g is None:                 # torch.optim.SGD skips parameters without a gradient
                continue
            if g.dtype != torch.float32 or g.device != dev:
                raise TypeError('gradients must be float32 on the parameters\' device')
            if not g.is_contiguous():
                g = g.contiguous()
            if not p.is_contiguous():
                raise RuntimeError('FusedSGD needs contiguous parameters')
            keep.append(g)
            rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), p.numel()))
        if not rows:
            return
        table = (_lib.NodeSgdTensor * len(rows))(*[_lib.NodeSgdTensor(*r) for r in rows])
        g = self.param_groups[0]
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.node_sgd_step(table, len(rows), float(g['lr']), float(g['momentum']), float(g['weight_decay']),
                                         float(self.grad_scale), torch.cuda.current_stream(dev).cuda_stream))
        del keep

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        return {'momentum_buf': self._momentum_flat,
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, sd):
        self._momentum_flat.copy_(sd['momentum_buf'])
        self.param_groups[0].update(sd['param_groups'][0])
