"""Host-side mirror of the reference's ODE-block interface.

Same names, constructor arguments, mutable attributes and state_dict keys as
`/root/reference/model.py:313-403` so that `load_state_dict` of a reference
checkpoint works (`utils.py:267-268`) and callers that poke `tol`, `t1`,
`method`, `return_last_only`, `nfe` at run time (`evaluate.py:62,80,116-117`,
`model.py:50-51,58-62`) keep working:

    ConcatConv2d  model.py:313-323     ODEfunc  model.py:326-348
    ODEBlock      model.py:351-403     normalization('group')  model.py:268-271

`ODEBlock.forward` hands the solve to `integrate.odeint[_adjoint]`, i.e. to the
HIP library.  `ODEfunc.forward` / `ConcatConv2d.forward` are kept as ordinary
PyTorch modules (they define the dynamics for anyone who calls them directly,
e.g. a checker), but the product solve path never executes them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import integrate


def normalization(norm: str = 'group'):
    """model.py:268-281.  'group' dynamics run on the fused kernels; 'batch' (model.py:274: BatchNorm2d without running
    statistics -- it couples the samples of a batch, so no fused kernel takes it) runs the generic solver (generic.py)."""
    if norm == 'group':
        return lambda dim: nn.GroupNorm(min(32, dim), dim)
    if norm == 'batch':
        return lambda dim: nn.BatchNorm2d(dim, track_running_stats=False)
    raise NotImplementedError('Normalization layer not implemented: {}'.format(norm))


class ConcatConv2d(nn.Module):
    """Conv2d over [t, x]: a constant time plane is prepended as channel 0."""

    def __init__(self, dim_in, dim_out, transpose=False, **kwargs):
        super().__init__()
        if transpose:
            raise NotImplementedError('transpose=True is never used by the reference ODEfunc')
        self._layer = nn.Conv2d(dim_in + 1, dim_out, **kwargs)

    def forward(self, t, x):
        plane = x.new_ones(x.shape[0], 1, x.shape[2], x.shape[3]) * t
        return self._layer(torch.cat((plane, x), dim=1))


class ODEfunc(nn.Module):
    """f(t, x) = GN3(conv2(t, relu(GN2(conv1(t, relu(GN1(x)))))))."""

    def __init__(self, dim, norm='group'):
        super().__init__()
        make_norm = normalization(norm)
        self.norm1 = make_norm(dim)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = ConcatConv2d(dim, dim, kernel_size=3, stride=1, padding=1)
        self.norm2 = make_norm(dim)
        self.conv2 = ConcatConv2d(dim, dim, kernel_size=3, stride=1, padding=1)
        self.norm3 = make_norm(dim)
        self.nfe = 0

    def forward(self, t, x):
        self.nfe += 1
        h = F.relu(self.norm1(x))
        h = F.relu(self.norm2(self.conv1(t, h)))
        return self.norm3(self.conv2(t, h))


class ODEBlock(nn.Module):
    """x(t1) = x(0) + int_0^t1 ODEfunc(t, x) dt, solved on the GPU by libnode_hip."""

    def __init__(self, n_filters=64, tol=1e-3, method='dopri5', adjoint=False, t1=1, norm='group'):
        super().__init__()
        self.odefunc = ODEfunc(n_filters, norm=norm)
        self.t1 = t1
        self.tol = tol
        self.method = method
        self.odeint = integrate.odeint_adjoint if adjoint else integrate.odeint
        self.return_last_only = True
        # data parallel, opt-in (not in the reference, which is single-device): True or a process group = GLOBAL-NORM solves -- the ranks
        # add the sums every step decision is taken from, so that all of them take identical steps (integrate._opts_struct)
        self.global_norm = None

    def forward(self, x):
        if self.integration_time is None:      # t1 == 0: identity (model.py:363-364)
            return x
        t = self.integration_time
        if t.device != x.device or t.dtype != x.dtype:
            # model.py:366.  The host copy of the grid that the `t1` setter attached travels with it, so the
            # solve never reads the device tensor back (a grid assigned to `integration_time` directly carries
            # no tag and is read back once per tensor object, integrate._host_times)
            moved = t.type_as(x)
            tag = getattr(t, integrate.HOST_TIMES_ATTR, None)
            if tag is not None and tag[0] == t._version:
                integrate.tag_host_times(moved, tag[1])
            self.integration_time = moved
        if self.return_last_only and (self.odeint is integrate.odeint or self.odeint is integrate.odeint_adjoint):
            # same solve, same `out[-1]`; the backward receives that slice's gradient alone (integrate.solve_last)
            return integrate.solve_last(self.odefunc, x, self.integration_time, self.tol, self.tol, self.method,
                                        adjoint=self.odeint is integrate.odeint_adjoint,
                                        options={'global_norm': self.global_norm} if getattr(self, 'global_norm', None) else None)
        if getattr(self, 'global_norm', None) and (self.odeint is integrate.odeint or self.odeint is integrate.odeint_adjoint):
            out = self.odeint(self.odefunc, x, self.integration_time, method=self.method, rtol=self.tol, atol=self.tol,
                              options={'global_norm': self.global_norm})
            return out[-1] if self.return_last_only else out
        out = self.odeint(self.odefunc, x, self.integration_time,
                          method=self.method, rtol=self.tol, atol=self.tol)
        return out[-1] if self.return_last_only else out

    # --- counters / knobs the reference's callers mutate -------------------
    @property
    def nfe(self):
        return self.odefunc.nfe

    @nfe.setter
    def nfe(self, value):
        self.odefunc.nfe = value

    @staticmethod
    def _grid(points):
        t = torch.tensor(points, dtype=torch.float32)
        return integrate.tag_host_times(t, t.tolist())      # fp32-rounded: what the device copy will hold

    @property
    def t1(self):
        return self.integration_time[1]

    @t1.setter
    def t1(self, value):
        if isinstance(value, (int, float)):
            self.integration_time = None if value == 0 else self._grid([0, value])
            return
        if isinstance(value, (list, tuple, torch.Tensor)):
            points = value.tolist() if isinstance(value, torch.Tensor) else list(value)
            if points[0] != 0:
                print(points[0])               # the reference prints the first point it prepends 0 to
                points = [0] + points
            self.integration_time = self._grid(points)
            return
        raise ValueError('Argument must be a scalar, a list, or a tensor')
