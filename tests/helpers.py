"""Shared helpers for the parity tests (tests/ may import oracle/)."""
import copy

import torch

from oracle.dynamics import OracleODEfunc, PARAM_ORDER, odefunc_vjp as oracle_vjp


def make_func(C, seed=0, device='cpu'):
    """An ODEfunc (package class) with non-trivial parameters + an oracle twin on CPU."""
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
    twin = OracleODEfunc(C)
    twin.load_state_dict(f.state_dict())
    return f.to(device), twin


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
