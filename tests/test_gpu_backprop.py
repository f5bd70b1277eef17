"""GPU: gradient of the NON-adjoint `odeint` (ODEBlock's default, model.py:7,359: adjoint=False) -- backpropagation
through the accepted solver steps (`node_solve_backprop`) against autograd through the oracle solver on the CPU."""
import pytest
import torch

from oracle import torchdiffeq_restated as tdq
from tests.helpers import make_func, rel_err

pytestmark = pytest.mark.gpu


def _both(shape, tol, method, tpts, seed, options=None, kink_free=True):
    import neural_ode_features_amd as nof
    N, C, H, W = shape
    f, twin = make_func(C, seed=seed, device='cuda', kink_free=kink_free)
    gen = torch.Generator().manual_seed(seed + 1)
    y = torch.randn(N, C, H, W, generator=gen)
    wgt = torch.randn(len(tpts), N, C, H, W, generator=gen) / (C * H * W) ** 0.5
    t = torch.tensor(tpts)
    yo = y.clone().requires_grad_(True)
    st = tdq.SolverStats()
    out_o = tdq.odeint(twin, yo, t, rtol=tol, atol=tol, method=method, options=options, stats=st)
    (out_o * wgt).sum().backward()                      # plain autograd through the solver's operations
    gp_o = torch.cat([p.grad.reshape(-1) for p in twin.parameters()])
    yh = y.cuda().requires_grad_(True)
    f.nfe = 0
    out_h = nof.odeint(f, yh, t.cuda(), rtol=tol, atol=tol, method=method, options=options)
    nfe_f = f.nfe
    (out_h * wgt.cuda()).sum().backward()
    assert f.nfe == nfe_f                               # upstream's autograd backward never calls func.forward
    gp_h = torch.cat([p.grad.reshape(-1) for p in f.parameters()])
    fs = f.last_forward_stats
    same = (fs['accepted'], fs['rejected']) == (st.accepted, st.rejected)
    return dict(out=rel_err(out_h, out_o), gy=rel_err(yh.grad, yo.grad), gp=rel_err(gp_h, gp_o), same=same,
                steps=(fs['accepted'], fs['rejected']))


@pytest.mark.parametrize('tpts', [[0.0, 1.0], [0.0, 0.25, 0.7, 1.0]])
def test_backprop_rk4(tpts):
    r = _both((4, 64, 7, 7), 1e-3, 'rk4', tpts, seed=101)
    print('rk4 backprop', tpts, r)
    assert r['out'] < 1e-5 and r['gy'] < 5e-5 and r['gp'] < 5e-5


@pytest.mark.parametrize('shape,tpts,dts', [((3, 64, 8, 8), [0.0, 1.0], [0.1, 0.2, 0.3, 0.45]),
                                            ((2, 32, 8, 8), [0.0, 0.15, 0.5, 0.6, 1.0], [0.2, 0.2, 0.3, 0.35]),
                                            ((2, 32, 16, 16), [0.0, 1.0], [0.3, 0.3, 0.4])])
def test_backprop_dopri5_replay_tight(shape, tpts, dts):
    """Forced step sizes: constants on both sides, so the two gradients are the same discrete quantity; several
    output times inside one step and a step end that coincides with an output exercise the transposed dense output."""
    r = _both(shape, 1e-3, 'dopri5', tpts, seed=102, options={'forced_dts': dts})
    print('dopri5 replay backprop', shape, tpts, r)
    assert r['same'] and r['out'] < 1e-5 and r['gy'] < 1e-4 and r['gp'] < 1e-4


@pytest.mark.parametrize('shape,tol', [((4, 64, 7, 7), 1e-3), ((2, 256, 8, 8), 1e-3), ((3, 16, 6, 6), 1e-5),
                                       ((8, 64, 8, 8), 1e-3)])     # (the last one: forward AND replay on the F(4x4,3x3) pipeline)
def test_backprop_dopri5_free_running(shape, tol):
    """Free-running: upstream's 2019 step-size controller is itself differentiable, so its autograd gradient carries
    an O(local error) sensitivity to the step sizes that this implementation (step sizes held constant) leaves out."""
    r = _both(shape, tol, 'dopri5', [0.0, 1.0], seed=103)
    print('dopri5 free-running backprop', shape, tol, r)
    assert r['out'] < 2e-4 or not r['same']
    assert r['gy'] < 5e-2 and r['gp'] < 5e-2


def test_odeblock_default_is_non_adjoint_and_trains():
    """ODEBlock(adjoint=False) -- the reference's constructor default -- end to end through autograd."""
    import neural_ode_features_amd as nof
    torch.manual_seed(7)
    blk = nof.ODEBlock(n_filters=32, tol=1e-3, method='dopri5').cuda()
    assert blk.odeint is nof.integrate.odeint
    x = torch.randn(2, 32, 8, 8, device='cuda', requires_grad=True)
    out = blk(x)
    nfe_f = blk.nfe
    out.square().mean().backward()
    assert blk.nfe == nfe_f and x.grad is not None and all(p.grad is not None for p in blk.parameters())
    assert float(x.grad.abs().max()) > 0
