"""GPU: the generic solver (neural-ode-features_amd/generic.py, node_flat_*) -- dynamics the fused kernels do not take run as
the caller's PyTorch function under the library's own device-resident step controller (SURVEY.md 8b "Fallback";
model.py:367 accepts any nn.Module, train.py:202 offers norm='batch').  Checked against the oracle on the CPU: outputs within
10 x atol (BASELINE.json north_star), step histories, the reference's NFE counter, adjoint gradients."""
import copy

import pytest
import torch
from torch import nn

from oracle import torchdiffeq_restated as tdq
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _same_nfe(hip, ref, intervals=1):
    """Equal counters -- or one accept / reject decision apart per solve (six evaluations; the backward runs one solve per time
    interval): two fp32 implementations may land on different sides of an error ratio of 1."""
    (hf, hb), (rf, rb) = hip, ref
    df, db = abs(hf - rf), abs((hb - hf) - (rb - rf))
    return df % 6 == 0 and df <= 6 and db % 6 == 0 and db <= 6 * intervals


def _grads_close(hip, ref, bound):
    """Every parameter gradient against the oracle's, each relative to max(its own scale, 1e-4 x the largest gradient of the
    function): a conv bias in front of a normalisation layer, and a parameter the function never uses, have exactly-zero
    gradients -- both sides then hold rounding noise only (or None on the HIP side, like upstream's allow_unused)."""
    gmax = max(float(r.abs().max()) for r in ref)
    for g, r in zip(hip, ref):
        g = torch.zeros_like(r) if g is None else g
        scale = max(float(r.abs().max()), 1e-4 * gmax)
        assert float((g - r).abs().max()) / scale < bound, (float((g - r).abs().max()), scale)


def _both(func, y, tpts, tol, method='dopri5', adjoint=True, weight_seed=7):
    """(HIP-side result, oracle result): outputs, grad_y0, parameter gradients, nfe after forward / backward."""
    import neural_ode_features_amd as nof
    twin = copy.deepcopy(func).cpu()
    f = func.cuda()
    t = torch.tensor(tpts)
    wgt = torch.randn((len(tpts),) + tuple(y.shape), generator=torch.Generator().manual_seed(weight_seed)) / y[0].numel() ** 0.5
    res = []
    for mod, dev, solve in ((f, 'cuda', nof.odeint_adjoint if adjoint else nof.odeint), (twin, 'cpu', tdq.odeint_adjoint)):
        if hasattr(mod, 'nfe'):
            mod.nfe = 0
        for prm in mod.parameters():
            prm.grad = None
        y0 = y.detach().clone().to(dev).requires_grad_(True)
        out = solve(mod, y0, t.to(dev), rtol=tol, atol=tol, method=method)
        nfe_f = getattr(mod, 'nfe', None)
        (out * wgt.to(dev)).sum().backward()
        nfe_b = getattr(mod, 'nfe', None)
        res.append(dict(out=out.detach().cpu(), gy=y0.grad.cpu(), gp=[None if p.grad is None else p.grad.cpu() for p in mod.parameters()],
                        nfe=(nfe_f, nfe_b)))
    return res


@pytest.mark.parametrize('tol,tpts', [(1e-3, (0.0, 1.0)), (1e-3, (0.0, 0.3, 0.55, 1.0)), (1e-4, (0.0, 1.0))])
def test_batchnorm_dynamics_through_odeint_adjoint(tol, tpts):
    """The reference's `ODEfunc(dim, norm='batch')` (model.py:274, train.py:202): BatchNorm couples the samples, so no fused
    kernel takes it -- the solve runs the generic path and agrees with the oracle."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    torch.manual_seed(3)
    func = nof.ODEfunc(16, norm='batch')
    with pytest.raises(NotImplementedError):
        integrate.Recognised(func)
    y = torch.randn(4, 16, 6, 6, generator=torch.Generator().manual_seed(4))
    hip, ref = _both(func, y, tpts, tol)
    assert torch.equal(hip['out'][0], y)
    assert float((hip['out'] - ref['out']).abs().max()) <= 10 * tol
    assert _same_nfe(hip['nfe'], ref['nfe'], len(tpts) - 1), (hip['nfe'], ref['nfe'])      # same steps tried, same evaluations counted (model.py:340)
    assert rel_err(hip['gy'], ref['gy']) < 2e-2
    _grads_close(hip['gp'], ref['gp'], 2e-2)


@pytest.mark.parametrize('shape', [(2, 24, 9, 9), (2, 6, 9, 9), (1, 8, 20, 20), (3, 12, 33, 5)])
def test_geometries_outside_the_fused_tiling(shape):
    """GroupNorm dynamics on states the fused kernels may refuse (channels not a multiple of four, 20x20 and 33x5 images):
    `nof.odeint_adjoint` takes whichever path serves the shape; the result is the oracle's within 10 x atol either way."""
    import neural_ode_features_amd as nof
    N, Cc, H, W = shape
    torch.manual_seed(11)
    func = nof.ODEfunc(Cc)
    with torch.no_grad():
        for name, p in func.named_parameters():
            if 'norm' in name:
                p.add_(0.2 * torch.randn(p.shape))
    y = torch.randn(*shape, generator=torch.Generator().manual_seed(12))
    for tol in (1e-3,):
        hip, ref = _both(func, y, (0.0, 1.0), tol)
        assert float((hip['out'] - ref['out']).abs().max()) <= 10 * tol, (shape, tol)
        assert _same_nfe(hip['nfe'], ref['nfe']), (shape, tol, hip['nfe'], ref['nfe'])
        assert rel_err(hip['gy'], ref['gy']) < 2e-2
        _grads_close(hip['gp'], ref['gp'], 5e-2)
    hip, ref = _both(func, y, (0.0, 0.5, 1.0), 1e-3, method='rk4')
    assert float((hip['out'] - ref['out']).abs().max()) <= 1e-4
    assert rel_err(hip['gy'], ref['gy']) < 1e-3


class _Mlp(nn.Module):
    """Foreign dynamics on a rank-2 state with an explicit time dependence and a parameter the output does not depend on."""

    def __init__(self, dim):
        super().__init__()
        self.a = nn.Linear(dim, 2 * dim)
        self.b = nn.Linear(2 * dim, dim)
        self.unused = nn.Parameter(torch.ones(3))
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        return self.b(torch.tanh(self.a(y) * (1.0 + 0.5 * torch.sin(3.0 * t)))) - 0.1 * y


@pytest.mark.parametrize('tpts', [(0.0, 1.0), (0.0, 0.25, 0.5, 2.0), (1.0, 0.4, 0.0)])
def test_foreign_module_on_a_rank_two_state(tpts):
    """Any nn.Module taking (t, y) (model.py:367): an MLP with time-dependent dynamics on a [N, D] state, increasing,
    multi-point and DECREASING time grids; the parameter `func` never uses gets a zero / None gradient like upstream."""
    torch.manual_seed(21)
    func = _Mlp(32)
    y = torch.randn(50, 32, generator=torch.Generator().manual_seed(22))
    for tol in (1e-3, 1e-5):     # (fp32 states: below ~1e-5 both sides sit on their rounding floor)
        hip, ref = _both(func, y, tpts, tol)
        assert float((hip['out'] - ref['out']).abs().max()) <= 10 * tol, tol
        assert _same_nfe(hip['nfe'], ref['nfe'], len(tpts) - 1), (hip['nfe'], ref['nfe'])
        assert rel_err(hip['gy'], ref['gy']) < max(2e-3, 50 * tol)
        k = [name for name, _ in func.named_parameters()].index('unused')
        assert hip['gp'][k] is None or float(hip['gp'][k].abs().max()) == 0.0        # `unused`: no gradient, like upstream
        _grads_close(hip['gp'], ref['gp'], max(2e-3, 50 * tol))


def test_plain_odeint_and_the_error_surface():
    """`odeint` (adjoint=False) on the generic path: same forward values; its backward is the continuous adjoint.  With
    the fallback switched off the round-4 behaviour is back (refusal, not emulation); CPU / fp64 states always raise."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    from neural_ode_features_amd._lib import NodeHipError
    torch.manual_seed(31)
    func = _Mlp(8)
    y = torch.randn(4, 8)
    hip, ref = _both(func, y, (0.0, 1.0), 1e-4, adjoint=False)
    assert float((hip['out'] - ref['out']).abs().max()) <= 1e-3 and rel_err(hip['gy'], ref['gy']) < 1e-2
    f = func.cuda()
    with pytest.raises(RuntimeError, match='no CPU path'):
        nof.odeint(f, y, torch.tensor([0.0, 1.0]))
    with pytest.raises(TypeError):
        nof.odeint(f, y.cuda().double(), torch.tensor([0.0, 1.0]).cuda())
    with pytest.raises(NodeHipError, match='MAX_STEPS'):
        nof.odeint(f, y.cuda(), torch.tensor([0.0, 1.0]).cuda(), rtol=1e-9, atol=1e-9, options={'max_num_steps': 2})
    integrate.GENERIC_FALLBACK = False
    try:
        with pytest.raises(ValueError):
            nof.odeint(f, y.cuda(), torch.tensor([0.0, 1.0]).cuda())              # rank-2 state
        with pytest.raises(NotImplementedError):
            nof.odeint(nn.Conv2d(8, 8, 3, 1, 1).cuda(), torch.randn(1, 8, 4, 4).cuda(), torch.tensor([0.0, 1.0]).cuda())
        g = nof.ODEfunc(8).cuda()
        with pytest.raises(NodeHipError, match='UNSUPPORTED'):
            nof.odeint(g, torch.randn(1, 8, 20, 20).cuda(), torch.tensor([0.0, 1.0]).cuda())
    finally:
        integrate.GENERIC_FALLBACK = True


def test_reference_style_odeblock_with_batch_norm_trains():
    """`ODENet(..., norm='batch')` (train.py:202): the whole net steps through stem (module sequence: the fused stem is
    GroupNorm's) -> ODE block on the generic solver -> head, forward and backward, finite gradients everywhere."""
    import neural_ode_features_amd as nof
    torch.manual_seed(5)
    net = nof.ODENet(3, out=10, n_filters=16, downsample='residual', method='dopri5', tol=1e-3, adjoint=True, norm='batch').cuda().train()
    x = torch.randn(4, 3, 32, 32, device='cuda')
    yv = torch.randint(0, 10, (4,), device='cuda')
    loss = nof.cross_entropy(net(x), yv)
    nfe_f = net.nfe(reset=True)
    loss.backward()
    nfe_b = net.nfe(reset=True)
    assert nfe_f >= 8 and (nfe_f - 2) % 6 == 0 and nfe_b >= 9 and (nfe_b - 3) % 6 == 0      # show.py:199 cost model
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
