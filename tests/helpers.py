"""Shared helpers for the parity tests (tests/ may import oracle/)."""
import copy

import torch

from oracle.dynamics import OracleODEfunc, PARAM_ORDER, odefunc_vjp as oracle_vjp


def make_func(C, seed=0, device='cpu', kink_free=False):
    """An ODEfunc (package class) with non-trivial parameters + an oracle twin on CPU.

    kink_free=True shifts the GroupNorm biases in front of the two ReLUs to +8, so every
    pre-activation is positive and the ReLU derivative has no discontinuity anywhere near the data.
    Gradient parity through a whole solve can then be asserted tightly: with ordinary parameters one
    or two of the ~10^6 pre-activations of a solve land within fp32 rounding (~1e-6) of zero, two
    correct fp32 implementations then disagree on that element's ReLU mask, and the gradient changes
    by O(1) around that pixel (measured: the oracle against either GPU kernel generation, 4 of 6 seeds
    at [2, 256, 8, 8]).  The mask logic itself is covered by the single-evaluation VJP tests."""
    import neural_ode_features_amd as nof
    torch.manual_seed(seed)
    f = nof.ODEfunc(C)
    gen = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for name, p in f.named_parameters():
            if 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.25 * torch.randn(p.shape, generator=gen))
            elif 'norm' in name and name.endswith('bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
                if kink_free and not name.startswith('norm3'):
                    p.add_(8.0)
    twin = OracleODEfunc(C)
    twin.load_state_dict(f.state_dict())
    return f.to(device), twin


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def robust_grad_err(a, b):
    """(relative L2 error, fraction of elements off by more than 1e-3 x max|b|) -- for gradients that
    went through ReLU kinks, where a max-norm comparison is ill-posed (see make_func)."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    frac = float(((a - b).abs() > 1e-3 * b.abs().max()).double().mean())
    return l2, frac


class ProbedODEfunc(OracleODEfunc):
    """The oracle dynamics, additionally recording per SAMPLE the smallest |pre-activation| either ReLU saw over
    every evaluation made through it (forward solve, and the recomputed forwards of the adjoint solve).  A sample
    whose record stays above the fp32 disagreement of two correct implementations (~1e-6) cannot have had a ReLU
    mask flip: its gradient must agree tightly.  Same ops in the same order as `oracle.dynamics.odefunc_forward`."""

    def __init__(self, dim):
        super().__init__(dim)
        self.min_abs = None

    def _note(self, z):
        m = z.detach().abs().flatten(1).amin(dim=1)
        self.min_abs = m if self.min_abs is None or self.min_abs.shape != m.shape else torch.minimum(self.min_abs, m)

    def forward(self, t, x):
        import torch.nn.functional as F
        from oracle.dynamics import concat_conv2d, n_groups
        self.nfe += 1
        p = dict(self.named_parameters())
        g = n_groups(x.shape[1])
        z1 = F.group_norm(x, g, p['norm1.weight'], p['norm1.bias'], 1e-5)
        self._note(z1)
        out = concat_conv2d(t, F.relu(z1), p['conv1._layer.weight'], p['conv1._layer.bias'])
        z2 = F.group_norm(out, g, p['norm2.weight'], p['norm2.bias'], 1e-5)
        self._note(z2)
        out = concat_conv2d(t, F.relu(z2), p['conv2._layer.weight'], p['conv2._layer.bias'])
        return F.group_norm(out, g, p['norm3.weight'], p['norm3.bias'], 1e-5)


def per_sample_err(a, b):
    """max |a - b| of every sample, relative to the largest |b| of the whole tensor."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return (a - b).abs().flatten(1).amax(dim=1) / (b.abs().max() + 1e-30)
