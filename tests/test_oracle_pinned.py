"""Pins the oracle (oracle/torchdiffeq_restated.py) without the upstream source:
exact tableau identities, closed-form ODEs, convergence order, the reference's NFE
cost model (show.py:199), scipy cross-checks, adjoint-vs-autograd and finite differences."""
from fractions import Fraction as Fr
import math

import numpy as np
import pytest
import torch

from oracle import torchdiffeq_restated as tdq

F = Fr
ALPHA = [F(1, 5), F(3, 10), F(4, 5), F(8, 9), F(1), F(1)]
BETA = [
    [F(1, 5)],
    [F(3, 40), F(9, 40)],
    [F(44, 45), F(-56, 15), F(32, 9)],
    [F(19372, 6561), F(-25360, 2187), F(64448, 6561), F(-212, 729)],
    [F(9017, 3168), F(-355, 33), F(46732, 5247), F(49, 176), F(-5103, 18656)],
    [F(35, 384), F(0), F(500, 1113), F(125, 192), F(-2187, 6784), F(11, 84)],
]
B5 = [F(35, 384), F(0), F(500, 1113), F(125, 192), F(-2187, 6784), F(11, 84), F(0)]
B4 = [F(1951, 21600), F(0), F(22642, 50085), F(451, 720), F(-12231, 42400), F(649, 6300), F(1, 60)]
CMID = [F(6025192743, 30085553152) / 2, F(0), F(51252292925, 65400821598) / 2, F(-2691868925, 45128329728) / 2,
        F(187940372067, 1594534317056) / 2, F(-1776094331, 19743644256) / 2, F(11237099, 235043384) / 2]
C_NODES = [F(0)] + ALPHA


def test_tableau_floats_match_rationals():
    for row_f, row_q in zip(tdq.DP_BETA, BETA):
        assert [float(q) for q in row_q] == pytest.approx(row_f, rel=1e-15)
    assert [float(a) for a in ALPHA] == pytest.approx(tdq.DP_ALPHA, rel=1e-15)
    assert [float(q) for q in B5] == pytest.approx(tdq.DP_C_SOL, rel=1e-15)
    assert [float(a - b) for a, b in zip(B5, B4)] == pytest.approx(tdq.DP_C_ERROR, rel=1e-12, abs=1e-18)
    assert [float(q) for q in CMID] == pytest.approx(tdq.DP_C_MID, rel=1e-15)


def test_row_sums_and_fsal():
    for a, row in zip(ALPHA, BETA):
        assert sum(row) == a                       # consistency: c_i = sum_j a_ij
    assert BETA[-1] == B5[:-1] and B5[-1] == 0     # FSAL: y1 is the last stage point


def _order_conditions(b, upto):
    c = C_NODES
    A = [[F(0)] * 7 for _ in range(7)]
    for i, row in enumerate(BETA):
        for j, v in enumerate(row):
            A[i + 1][j] = v
    Ac = [sum(A[i][j] * c[j] for j in range(7)) for i in range(7)]
    Ac2 = [sum(A[i][j] * c[j] ** 2 for j in range(7)) for i in range(7)]
    AAc = [sum(A[i][j] * Ac[j] for j in range(7)) for i in range(7)]
    conds = {1: [(sum(b), F(1))],
             2: [(sum(bi * ci for bi, ci in zip(b, c)), F(1, 2))],
             3: [(sum(bi * ci ** 2 for bi, ci in zip(b, c)), F(1, 3)), (sum(bi * x for bi, x in zip(b, Ac)), F(1, 6))],
             4: [(sum(bi * ci ** 3 for bi, ci in zip(b, c)), F(1, 4)),
                 (sum(bi * ci * x for bi, ci, x in zip(b, c, Ac)), F(1, 8)),
                 (sum(bi * x for bi, x in zip(b, Ac2)), F(1, 12)), (sum(bi * x for bi, x in zip(b, AAc)), F(1, 24))],
             5: [(sum(bi * ci ** 4 for bi, ci in zip(b, c)), F(1, 5))]}
    for order in range(1, upto + 1):
        for got, want in conds[order]:
            assert got == want, (order, got, want)


def test_order_conditions_exact():
    _order_conditions(B5, 5)     # 5th-order solution weights
    _order_conditions(B4, 4)     # Shampine's embedded 4th-order weights
    assert sum(a - b for a, b in zip(B5, B4)) == 0


def test_midpoint_weights():
    for p in range(1, 5):        # sum c_mid * c^(p-1) = (1/2)^p / p
        assert sum(m * c ** (p - 1) for m, c in zip(CMID, C_NODES)) == F(1, 2) ** p / p


def test_tableau_vs_scipy():
    rk = pytest.importorskip('scipy.integrate._ivp.rk')
    A = np.zeros((7, 7))
    for i, row in enumerate(tdq.DP_BETA):
        A[i + 1, :len(row)] = row
    assert np.allclose(A[:6, :5], rk.RK45.A[:6, :5], rtol=1e-14, atol=0)
    assert np.allclose(tdq.DP_C_SOL[:6], rk.RK45.B, rtol=1e-14)
    # scipy's E uses the classic embedded pair; Shampine's error weights are -2/3 of it
    assert np.allclose(np.array(tdq.DP_C_ERROR), -2.0 / 3.0 * rk.RK45.E, rtol=1e-10, atol=1e-16)


@pytest.mark.parametrize('method', ['dopri5', 'rk4'])
def test_closed_form_odes(method):
    t = torch.linspace(0, 1, 11 if method == 'rk4' else 3, dtype=torch.float64)
    y = tdq.odeint(lambda t, y: -y, torch.tensor([1.0, 2.0], dtype=torch.float64), t, rtol=1e-9, atol=1e-11, method=method)
    assert torch.allclose(y[-1], torch.tensor([1.0, 2.0], dtype=torch.float64) * math.exp(-1), atol=1e-5)
    y = tdq.odeint(lambda t, y: t * y, torch.tensor([1.0], dtype=torch.float64), t, rtol=1e-9, atol=1e-11, method=method)
    assert abs(float(y[-1]) - math.exp(0.5)) < 1e-5          # time-dependent: exercises t + alpha*dt
    Amat = torch.tensor([[0.0, 1.0], [-1.0, 0.0]], dtype=torch.float64)
    y = tdq.odeint(lambda t, y: Amat @ y, torch.tensor([1.0, 0.0], dtype=torch.float64), t, rtol=1e-9, atol=1e-11, method=method)
    assert torch.allclose(y[-1], torch.tensor([math.cos(1.0), -math.sin(1.0)], dtype=torch.float64), atol=1e-5)


def test_decreasing_time_and_tuple_state():
    t = torch.tensor([1.0, 0.0], dtype=torch.float64)
    y = tdq.odeint(lambda t, y: -y, torch.tensor([math.exp(-1.0)], dtype=torch.float64), t, rtol=1e-9, atol=1e-11, method='dopri5')
    assert abs(float(y[-1]) - 1.0) < 1e-6
    out = tdq.odeint(lambda t, y: (-y[0], 2 * y[1]), (torch.ones(2, dtype=torch.float64), torch.ones(3, dtype=torch.float64)),
                     torch.tensor([0.0, 0.5], dtype=torch.float64), rtol=1e-9, atol=1e-11, method='dopri5')
    assert torch.allclose(out[0][-1], torch.full((2,), math.exp(-0.5), dtype=torch.float64), atol=1e-7)
    assert torch.allclose(out[1][-1], torch.full((3,), math.exp(1.0), dtype=torch.float64), atol=1e-6)


def _global_error(method, n):
    t = torch.linspace(0, 1, n + 1, dtype=torch.float64)
    f = lambda t, y: torch.cos(t) * y   # noqa: E731   y = exp(sin t)
    y0 = torch.tensor([1.0], dtype=torch.float64)
    if method == 'rk4':
        y = tdq.odeint(f, y0, t, method='rk4')
    else:   # fixed-h dopri5 through replay mode
        y = tdq.odeint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), rtol=1e-3, atol=1e-3, method='dopri5',
                       options={'forced_dts': [1.0 / n] * n})
    return abs(float(y[-1]) - math.exp(math.sin(1.0)))


def test_convergence_orders():
    e1, e2 = _global_error('rk4', 8), _global_error('rk4', 16)
    assert 3.6 < math.log2(e1 / e2) < 4.4            # 3/8-rule RK4: h^4
    e1, e2 = _global_error('dopri5', 4), _global_error('dopri5', 8)
    assert 4.6 < math.log2(e1 / e2) < 5.6            # 5th-order propagated solution: h^5


def test_nfe_cost_model_show_py_199():
    """show.py:199: steps = (NFE - 2) / 6."""
    st = tdq.SolverStats()
    tdq.odeint(lambda t, y: torch.sin(5 * t) * y, torch.ones(4), torch.tensor([0.0, 1.0]), rtol=1e-5, atol=1e-5,
               method='dopri5', stats=st)
    assert st.nfe == 2 + 6 * (st.accepted + st.rejected)
    st = tdq.SolverStats()
    tdq.odeint(lambda t, y: -y, torch.ones(4), torch.tensor([0.0, 1.0]), method='rk4', stats=st)
    assert st.nfe == 4


def test_one_step_and_dense_output_vs_scipy_rk45():
    """An independent implementation of the same pair: scipy's RK45 (Dormand-Prince 5(4), Shampine's dense output).  One
    step of a fixed size from the same state: the propagated solution, the FSAL derivative, the embedded error estimate
    (scipy's E is -3/2 of Shampine's weights, see above) and the continuous extension at interior points -- the quartic the
    restated `_interp_fit_dopri5` builds from (y0, y1, y_mid, f0, f1) must be scipy's polynomial."""
    rk = pytest.importorskip('scipy.integrate._ivp.rk')

    def f_np(t, y):
        return np.array([np.sin(3 * t) * y[1] - 0.5 * y[0] ** 2, np.cos(y[0]) + t * y[2], -y[2] * y[1] + 0.3])

    def f_t(t, y):
        (y,) = y
        return (torch.stack([torch.sin(3 * t) * y[1] - 0.5 * y[0] ** 2, torch.cos(y[0]) + t * y[2], -y[2] * y[1] + 0.3]),)

    t0, h = 0.2, 0.37
    y0 = np.array([0.7, -1.1, 0.4])
    K = np.empty((7, 3))
    y1_s, f1_s = rk.rk_step(f_np, t0, y0, f_np(t0, y0), h, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
    err_s = K.T @ rk.RK45.E * h
    Q = K.T @ rk.RK45.P
    y0_t = (torch.tensor(y0, dtype=torch.float64),)
    f0_t = f_t(torch.tensor(t0, dtype=torch.float64), y0_t)
    y1_o, f1_o, err_o, k = tdq._runge_kutta_step(f_t, y0_t, f0_t, t0, h)
    assert np.allclose(y1_o[0].numpy(), y1_s, rtol=1e-13, atol=1e-15)
    assert np.allclose(f1_o[0].numpy(), f1_s, rtol=1e-13, atol=1e-15)
    assert np.allclose(err_o[0].numpy(), -2.0 / 3.0 * err_s, rtol=1e-9, atol=1e-16)
    coef = tdq._interp_fit_dopri5(y0_t, y1_o, k, torch.tensor(h, dtype=torch.float64))
    for x in (0.0, 0.1, 0.25, 0.5, 0.77, 1.0):
        got = tdq._interp_evaluate(coef, t0, t0 + h, t0 + x * h)[0].numpy()
        want = y0 + h * (Q @ np.cumprod(np.full(4, x)))
        assert np.allclose(got, want, rtol=1e-11, atol=1e-13), (x, got, want)


def test_initial_step_vs_scipy():
    """Hairer's starting step is the one piece of the step-size logic the 2019 torchdiffeq shares with scipy unchanged
    (`select_initial_step`, error-estimator order 4, root-mean-square norm): same number from both, on a stiff-ish and on
    a mild problem, wherever scipy's extra clamps (interval length, max_step) do not bind."""
    common = pytest.importorskip('scipy.integrate._ivp.common')
    for scale_y, rate in ((1.0, 3.0), (40.0, 0.05), (1e-3, 200.0)):
        def f_np(t, y, r=rate):
            return np.array([-r * y[0] + np.sin(t), r * y[0] * y[1] - y[2], np.cos(3 * t) * y[2] ** 2 + 0.1])

        def f_t(t, y, r=rate):
            (y,) = y
            return (torch.stack([-r * y[0] + torch.sin(t), r * y[0] * y[1] - y[2], torch.cos(3 * t) * y[2] ** 2 + 0.1]),)

        y0 = scale_y * np.array([0.9, -0.4, 1.3])
        for tol in (1e-3, 1e-6):
            want = common.select_initial_step(f_np, 0.1, y0, 1e9, np.inf, f_np(0.1, y0), 1.0, 4, tol, tol)
            y0_t = (torch.tensor(y0, dtype=torch.float64),)
            t0 = torch.tensor(0.1, dtype=torch.float64)
            got = float(tdq._select_initial_step(f_t, t0, y0_t, 4, tol, tol, f_t(t0, y0_t)))
            assert abs(got - want) <= 1e-12 * abs(want), (scale_y, rate, tol, got, want)


def test_dense_output_against_tight_solve():
    f = lambda t, y: torch.stack([y[1], -y[0]])   # noqa: E731
    y0 = torch.tensor([0.0, 1.0], dtype=torch.float64)
    t = torch.linspace(0, 2, 21, dtype=torch.float64)
    y = tdq.odeint(f, y0, t, rtol=1e-6, atol=1e-8, method='dopri5')
    assert torch.allclose(y[:, 0], torch.sin(t), atol=2e-5)


class _Lin(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.A = torch.nn.Parameter(0.5 * torch.randn(3, 3, dtype=torch.float64))
        self.b = torch.nn.Parameter(0.1 * torch.randn(3, dtype=torch.float64))

    def forward(self, t, y):
        return torch.tanh(y @ self.A.t() + self.b * t)


def test_adjoint_gradients_vs_autograd_and_fd():
    func = _Lin()
    y0 = torch.tensor([[0.3, -0.2, 0.5], [0.1, 0.4, -0.3]], dtype=torch.float64, requires_grad=True)
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64)
    w = torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64).view(3, 1, 1)
    out = tdq.odeint_adjoint(func, y0, t, rtol=1e-9, atol=1e-10, method='dopri5')
    (out * w).sum().backward()
    g_adj = (y0.grad.clone(), func.A.grad.clone(), func.b.grad.clone())
    y0.grad = None
    func.zero_grad()
    out2 = tdq.odeint(func, y0, t, rtol=1e-9, atol=1e-10, method='dopri5')   # autograd through the unrolled solver
    (out2 * w).sum().backward()
    for a, b in zip(g_adj, (y0.grad, func.A.grad, func.b.grad)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # finite difference on one parameter entry
    eps = 1e-6
    with torch.no_grad():
        func.A[0, 1] += eps
        lp = float((tdq.odeint(func, y0.detach(), t, rtol=1e-10, atol=1e-11, method='dopri5') * w).sum())
        func.A[0, 1] -= 2 * eps
        lm = float((tdq.odeint(func, y0.detach(), t, rtol=1e-10, atol=1e-11, method='dopri5') * w).sum())
        func.A[0, 1] += eps
    assert abs((lp - lm) / (2 * eps) - float(g_adj[1][0, 1])) < 1e-5


def test_fp32_tracks_fp64_truth():
    # local-error control: the global error of the fp32 solve is a small multiple of tol
    # (12e-3 at tol 1e-3, 7e-5 at tol 1e-5 here) and shrinks with it
    torch.manual_seed(0)
    A = 0.7 * torch.randn(6, 6, dtype=torch.float64)
    f64 = lambda t, y: torch.tanh(y @ A.t())          # noqa: E731
    A32 = A.float()
    f32 = lambda t, y: torch.tanh(y @ A32.t())        # noqa: E731
    y0 = torch.randn(4, 6, dtype=torch.float64)
    truth = tdq.odeint(f64, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), rtol=1e-10, atol=1e-10)[-1]
    for tol in (1e-3, 1e-5):
        got = tdq.odeint(f32, y0.float(), torch.tensor([0.0, 1.0]), rtol=tol, atol=tol, method='dopri5')[-1]
        assert float((got.double() - truth).abs().max()) <= 30 * tol


def test_unsupported_method_and_bad_inputs():
    with pytest.raises(NotImplementedError):
        tdq.odeint(lambda t, y: y, torch.ones(1), torch.tensor([0.0, 1.0]), method='adams')
    with pytest.raises(TypeError):
        tdq.odeint(lambda t, y: y, torch.ones(1, dtype=torch.int64), torch.tensor([0.0, 1.0]))
    with pytest.raises(ValueError):
        tdq.odeint_adjoint(lambda t, y: y, torch.ones(1), torch.tensor([0.0, 1.0]))
