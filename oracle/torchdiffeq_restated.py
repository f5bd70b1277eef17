"""CPU restatement of the `torchdiffeq` solver path used by the reference.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is shipped or measured as
the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker.

What it restates
----------------
The reference calls one third-party function on its hot path
(``/root/reference/model.py:3`` import, ``:359`` select, ``:367`` call)::

    out = self.odeint(self.odefunc, x, self.integration_time,
                      method=self.method, rtol=self.tol, atol=self.tol)

``torchdiffeq`` is an *empty, un-pinned git submodule* in the reference
(``/root/reference/.gitmodules:1-3``) and is not installable here, so the
algorithm below is a restatement of the published 2019-era (pre-0.1.0)
torchdiffeq algorithm -- the era is fixed by ``train.py:219`` which offers
``--method adams``, a solver name that only exists in that generation.
The normative spec is SURVEY.md section 8(c).

PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
this path and the upstream source is absent, so this
oracle cannot be checked against the reference's own numbers.  It is pinned
instead by (tests/test_oracle_*.py):
  * exact-rational identities of the Dormand-Prince/Shampine tableau
    (order conditions 1..5 for b, 1..4 for b-hat, mid-point conditions);
  * closed-form ODEs (y'=-y, y'=Ay, y'=t*y);
  * an independent implementation of the same pair, scipy's RK45: tableau, and -- one step of a
    fixed size from the same state -- propagated solution, FSAL derivative, embedded error estimate
    and the continuous extension at interior points (the quartic of `_interp_fit_dopri5` IS
    scipy's dense-output polynomial, to 1e-11); Hairer's starting step equals scipy's
    `select_initial_step` (order 4, rms norm) to 1e-12;
  * convergence order (h^5 dopri5 fixed-h, h^4 rk4 3/8 rule);
  * the reference's own cost model NFE = 2 + 6*steps (``show.py:199``);
  * adjoint gradients vs autograd through the unrolled solver and vs fp64
    finite differences;
  * the dynamics ``f(t, y)`` and its VJPs against the *imported reference*
    ``ODEfunc`` (``model.py:326-348``) -- see tests/golden/.

Everything is tuple-native like the original: a tensor ``y0`` is wrapped into a
1-tuple, the adjoint integrates the 4-tuple ``(y, a, adj_t, adj_params)``.
The adaptive solver tracks ``t``/``dt`` in float64 (upstream's
``AdaptiveStepsizeODESolver.integrate`` casts the time grid to float64) and
rounds them to the state dtype for every stage evaluation; the fixed-grid
solver keeps them in the state dtype.  (SURVEY.md 8c says "fp32"; the two
choices differ by ~1e-7 relative in dt, four orders below the parity bound.)
"""
from __future__ import annotations

import math
from typing import Callable, List, Sequence, Tuple

import torch
from torch import nn

# --------------------------------------------------------------------------
# Dormand-Prince 5(4) tableau with Shampine's embedded weights + mid-point row
# (SURVEY.md 8c "Tableau").  Kept as python floats built from exact rationals.
# --------------------------------------------------------------------------
DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
DP_C_ERROR = [
    35 / 384 - 1951 / 21600,
    0,
    500 / 1113 - 22642 / 50085,
    125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400,
    11 / 84 - 649 / 6300,
    -1.0 / 60.0,
]
DP_C_MID = [
    6025192743 / 30085553152 / 2,
    0,
    51252292925 / 65400821598 / 2,
    -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2,
    -1776094331 / 19743644256 / 2,
    11237099 / 235043384 / 2,
]

SAFETY = 0.9
IFACTOR = 10.0
DFACTOR = 0.2
MAX_NUM_STEPS = 2 ** 31 - 1


class SolverStats:
    """Telemetry the reference only exposes as NFE (`model.py:337,340`)."""

    def __init__(self):
        self.nfe = 0
        self.accepted = 0
        self.rejected = 0
        self.dts: List[float] = []       # dt tried at every step (accepted or not)
        self.accepts: List[bool] = []
        self.first_step = None

    def as_dict(self):
        return dict(nfe=self.nfe, accepted=self.accepted, rejected=self.rejected,
                    dts=list(self.dts), accepts=list(self.accepts), first_step=self.first_step)


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _scaled_dot_product(scale, xs, ys):
    """sum(scale * x * y) -- evaluation order as upstream (python `sum`)."""
    return sum(scale * x * y for x, y in zip(xs, ys))


def _dot_product(xs, ys):
    return sum(x * y for x, y in zip(xs, ys))


def _rms(x: torch.Tensor) -> torch.Tensor:
    return x.norm() / math.sqrt(x.numel()) if x.numel() > 0 else x.new_zeros(())


def _decreasing(t: torch.Tensor) -> bool:
    return bool((t[1:] < t[:-1]).all())


def _check_inputs(func, y0, t):
    tensor_input = False
    if torch.is_tensor(y0):
        tensor_input = True
        y0 = (y0,)
        base = func
        func = lambda t, y: (base(t, y[0]),)  # noqa: E731
    assert isinstance(y0, tuple), 'y0 must be either a torch.Tensor or a tuple'
    for y0_ in y0:
        assert torch.is_tensor(y0_), 'each element must be a torch.Tensor'
    if _decreasing(t):
        t = -t
        base_rev = func
        func = lambda t, y: tuple(-f_ for f_ in base_rev(-t, y))  # noqa: E731
    for y0_ in y0:
        if not torch.is_floating_point(y0_):
            raise TypeError('`y0` must be a floating point Tensor but is a {}'.format(y0_.type()))
    if not torch.is_floating_point(t):
        raise TypeError('`t` must be a floating point Tensor but is a {}'.format(t.type()))
    return tensor_input, func, y0, t


def _global_means(squares, norm_reduce):
    """NOT upstream behaviour -- the checker of this package's GLOBAL-NORM mode (include/node_hip.h, node_solve_opts::norm_reduce):
    every rank of a data-parallel solve holds a shard; `norm_reduce = (fn, world)`, fn adding a 1-D tensor over the ranks.  Mean of
    each tensor of `squares` over ALL ranks' elements: sum over ranks of the local sums / (local count x world)."""
    fn, world = norm_reduce
    sums = fn(torch.stack([q.detach().sum().to(torch.float32) for q in squares]))
    return tuple(sums[i].to(q) / (q.numel() * world) for i, q in enumerate(squares))


def _select_initial_step(fun, t0, y0, order, rtol, atol, f0, norm_reduce=None):
    """Hairer's starting step, order argument 4 (SURVEY.md 8c "Initial step")."""
    t0 = t0.to(y0[0])
    scale = tuple(atol + torch.abs(y0_) * rtol for y0_ in y0)
    if norm_reduce is not None:      # (global-norm mode: the root mean squares over the unsharded batch)
        d0 = tuple(torch.sqrt(m) for m in _global_means([(y0_ / scale_) ** 2 for y0_, scale_ in zip(y0, scale)], norm_reduce))
        d1 = tuple(torch.sqrt(m) for m in _global_means([(f0_ / scale_) ** 2 for f0_, scale_ in zip(f0, scale)], norm_reduce))
        if max(d0).item() < 1e-5 or max(d1).item() < 1e-5:
            h0 = torch.tensor(1e-6).to(t0)
        else:
            h0 = 0.01 * max(d0_ / d1_ for d0_, d1_ in zip(d0, d1))
        y1 = tuple(y0_ + h0 * f0_ for y0_, f0_ in zip(y0, f0))
        f1 = fun(t0 + h0, y1)
        d2 = tuple(torch.sqrt(m) / h0 for m in _global_means([((f1_ - f0_) / scale_) ** 2 for f1_, f0_, scale_ in zip(f1, f0, scale)], norm_reduce))
        if max(d1).item() <= 1e-15 and max(d2).item() <= 1e-15:
            h1 = torch.max(torch.tensor(1e-6).to(h0), h0 * 1e-3)
        else:
            h1 = (0.01 / max(d1 + d2)) ** (1.0 / float(order + 1))
        return torch.min(100 * h0, h1)
    d0 = tuple(_rms(y0_ / scale_) for y0_, scale_ in zip(y0, scale))
    d1 = tuple(_rms(f0_ / scale_) for f0_, scale_ in zip(f0, scale))
    if max(d0).item() < 1e-5 or max(d1).item() < 1e-5:
        h0 = torch.tensor(1e-6).to(t0)
    else:
        h0 = 0.01 * max(d0_ / d1_ for d0_, d1_ in zip(d0, d1))
    y1 = tuple(y0_ + h0 * f0_ for y0_, f0_ in zip(y0, f0))
    f1 = fun(t0 + h0, y1)
    d2 = tuple(_rms((f1_ - f0_) / scale_) / h0 for f1_, f0_, scale_ in zip(f1, f0, scale))
    if max(d1).item() <= 1e-15 and max(d2).item() <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6).to(h0), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1 + d2)) ** (1.0 / float(order + 1))
    return torch.min(100 * h0, h1)


def _compute_error_ratio(error_estimate, rtol, atol, y0, y1, norm_reduce=None):
    error_tol = tuple(atol + rtol * torch.max(torch.abs(y0_), torch.abs(y1_)) for y0_, y1_ in zip(y0, y1))
    error_ratio = tuple(e / tol for e, tol in zip(error_estimate, error_tol))
    if norm_reduce is not None:
        return _global_means([r * r for r in error_ratio], norm_reduce)
    return tuple(torch.mean(r * r) for r in error_ratio)


def _optimal_step_size(last_step, mean_error_ratio, safety=SAFETY, ifactor=IFACTOR, dfactor=DFACTOR, order=5):
    """dt <- dt / clamp(ratio^(1/(2*order)) / safety, 1/ifactor, 1/dfactor)."""
    mean_error_ratio = max(mean_error_ratio)  # highest ratio over the tuple
    if mean_error_ratio == 0:
        return last_step * ifactor
    if mean_error_ratio < 1:
        dfactor = 1.0
    error_ratio = torch.sqrt(mean_error_ratio).to(last_step)
    exponent = torch.tensor(1.0 / order).to(last_step)
    factor = torch.max(torch.tensor(1.0 / ifactor).to(last_step),
                       torch.min(error_ratio ** exponent / safety, torch.tensor(1.0 / dfactor).to(last_step)))
    return last_step / factor


def _runge_kutta_step(func, y0, f0, t0, dt):
    dtype, device = y0[0].dtype, y0[0].device
    t0 = torch.as_tensor(t0, dtype=dtype, device=device)
    dt = torch.as_tensor(dt, dtype=dtype, device=device)
    k = tuple([f0_] for f0_ in f0)
    yi = y0
    for alpha_i, beta_i in zip(DP_ALPHA, DP_BETA):
        ti = t0 + alpha_i * dt
        yi = tuple(y0_ + _scaled_dot_product(dt, beta_i, k_) for y0_, k_ in zip(y0, k))
        for k_, f_ in zip(k, func(ti, yi)):
            k_.append(f_)
    # c_sol[:-1] == beta[-1] and c_sol[-1] == 0 (FSAL): y1 is the 6th stage point.
    y1 = yi
    f1 = tuple(k_[-1] for k_ in k)
    y1_error = tuple(_scaled_dot_product(dt, DP_C_ERROR, k_) for k_ in k)
    return y1, f1, y1_error, k


def _interp_fit(y0, y1, y_mid, f0, f1, dt):
    a = tuple(_dot_product([-2 * dt, 2 * dt, -8, -8, 16], [f0_, f1_, y0_, y1_, ym_])
              for f0_, f1_, y0_, y1_, ym_ in zip(f0, f1, y0, y1, y_mid))
    b = tuple(_dot_product([5 * dt, -3 * dt, 18, 14, -32], [f0_, f1_, y0_, y1_, ym_])
              for f0_, f1_, y0_, y1_, ym_ in zip(f0, f1, y0, y1, y_mid))
    c = tuple(_dot_product([-4 * dt, dt, -11, -5, 16], [f0_, f1_, y0_, y1_, ym_])
              for f0_, f1_, y0_, y1_, ym_ in zip(f0, f1, y0, y1, y_mid))
    d = tuple(dt * f0_ for f0_ in f0)
    e = y0
    return [a, b, c, d, e]


def _interp_fit_dopri5(y0, y1, k, dt):
    dt = dt.type_as(y0[0])
    y_mid = tuple(y0_ + _scaled_dot_product(dt, DP_C_MID, k_) for y0_, k_ in zip(y0, k))
    f0 = tuple(k_[0] for k_ in k)
    f1 = tuple(k_[-1] for k_ in k)
    return _interp_fit(y0, y1, y_mid, f0, f1, dt)


def _interp_evaluate(coefficients, t0, t1, t):
    dtype, device = coefficients[0][0].dtype, coefficients[0][0].device
    t0 = torch.as_tensor(t0, dtype=dtype, device=device)
    t1 = torch.as_tensor(t1, dtype=dtype, device=device)
    t = torch.as_tensor(t, dtype=dtype, device=device)
    assert (t0 <= t) & (t <= t1), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(t0, t, t1)
    x = ((t - t0) / (t1 - t0)).type(dtype).to(device)
    xs = [torch.tensor(1).type(dtype).to(device), x]
    for _ in range(2, len(coefficients)):
        xs.append(xs[-1] * x)
    return tuple(_dot_product(coefficients_, reversed(xs)) for coefficients_ in zip(*coefficients))


# --------------------------------------------------------------------------
# solvers
# --------------------------------------------------------------------------
class _Dopri5:
    def __init__(self, func, y0, rtol, atol, stats: SolverStats, forced_dts=None, norm_reduce=None):
        self.func, self.y0, self.rtol, self.atol = func, y0, rtol, atol
        self.stats = stats
        self.norm_reduce = norm_reduce          # (fn, world) or None: see _global_means
        # replay mode (test aid): a forced sequence of step sizes, every one accepted
        self.forced_dts = list(forced_dts) if forced_dts is not None else None

    def before_integrate(self, t):
        f0 = self.func(t[0].type_as(self.y0[0]), self.y0)
        if self.forced_dts is None:
            first_step = _select_initial_step(self.func, t[0], self.y0, 4, self.rtol, self.atol, f0=f0, norm_reduce=self.norm_reduce).to(t)
        else:
            first_step = torch.tensor(self.forced_dts.pop(0)).to(t)
        self.stats.first_step = float(first_step.detach())
        # (y, f, t0, t1, dt, interp)
        self.state = (self.y0, f0, t[0], t[0], first_step, [self.y0] * 5)

    def advance(self, next_t):
        n_steps = 0
        while next_t > self.state[3]:
            assert n_steps < MAX_NUM_STEPS, 'max_num_steps exceeded'
            self.state = self._step(self.state)
            n_steps += 1
        return _interp_evaluate(self.state[5], self.state[2], self.state[3], next_t)

    def _step(self, state):
        y0, f0, _, t0, dt, interp_coeff = state
        assert t0 + dt > t0, 'underflow in dt {}'.format(dt.item())
        for y0_ in y0:
            assert torch.isfinite(y0_).all(), 'non-finite values in state `y`'
        y1, f1, y1_error, k = _runge_kutta_step(self.func, y0, f0, t0, dt)
        ratio = _compute_error_ratio(y1_error, self.rtol, self.atol, y0, y1, self.norm_reduce)
        if self.forced_dts is None:
            accept = bool((torch.stack([r.detach() for r in ratio]) <= 1).all())
        else:
            accept = True
        self.stats.dts.append(float(dt))
        self.stats.accepts.append(accept)
        if accept:
            self.stats.accepted += 1
        else:
            self.stats.rejected += 1
        y_next = y1 if accept else y0
        f_next = f1 if accept else f0
        t_next = t0 + dt if accept else t0
        interp_coeff = _interp_fit_dopri5(y0, y1, k, dt) if accept else interp_coeff
        if self.forced_dts is None:
            dt_next = _optimal_step_size(dt, ratio)
        else:
            dt_next = torch.tensor(self.forced_dts.pop(0) if self.forced_dts else float(dt)).to(dt)
        return (y_next, f_next, t0, t_next, dt_next, interp_coeff)

    def integrate(self, t):
        # upstream's adaptive solvers track time and step size in float64 on the
        # state's device; stage times are rounded to the state dtype per eval.
        t = t.to(self.y0[0].device, torch.float64)
        solution = [self.y0]
        self.before_integrate(t)
        for i in range(1, len(t)):
            solution.append(self.advance(t[i]))
        return tuple(torch.stack(s) for s in zip(*solution))


def _rk4_alt_step(func, t, dt, y):
    """3/8-rule RK4 (upstream `rk4_alt_step_func`); returns the increment."""
    k1 = func(t, y)
    k2 = func(t + dt / 3, tuple(y_ + dt * k1_ / 3 for y_, k1_ in zip(y, k1)))
    k3 = func(t + dt * 2 / 3, tuple(y_ + dt * (k1_ / -3 + k2_) for y_, k1_, k2_ in zip(y, k1, k2)))
    k4 = func(t + dt, tuple(y_ + dt * (k1_ - k2_ + k3_) for y_, k1_, k2_, k3_ in zip(y, k1, k2, k3)))
    return tuple((k1_ + 3 * k2_ + 3 * k3_ + k4_) * (dt / 8) for k1_, k2_, k3_, k4_ in zip(k1, k2, k3, k4))


class _RK4:
    """Fixed grid = the requested time points (no `step_size` option ever
    reaches the solver from `model.py:367`)."""

    def __init__(self, func, y0, stats: SolverStats, **unused):
        self.func, self.y0, self.stats = func, y0, stats

    def integrate(self, t):
        t = t.type_as(self.y0[0])
        solution = [self.y0]
        y0 = self.y0
        for t0, t1 in zip(t[:-1], t[1:]):
            dy = _rk4_alt_step(self.func, t0, t1 - t0, y0)
            y1 = tuple(y0_ + dy_ for y0_, dy_ in zip(y0, dy))
            self.stats.accepted += 1
            self.stats.dts.append(float(t1 - t0))
            self.stats.accepts.append(True)
            solution.append(y1)  # grid == t, so the linear interpolation is the identity
            y0 = y1
        return tuple(torch.stack(s) for s in zip(*solution))


def odeint(func, y0, t, rtol=1e-7, atol=1e-12, method=None, options=None, stats: SolverStats = None):
    """`torchdiffeq.odeint` restated.  Returns `[len(t), *y0.shape]` (or a tuple)."""
    tensor_input, func, y0, t = _check_inputs(func, y0, t)
    options = dict(options or {})
    if method is None:
        method = 'dopri5'
    stats = stats if stats is not None else SolverStats()

    def counted(tt, yy, _f=func):
        stats.nfe += 1
        return _f(tt, yy)

    if method == 'dopri5':
        solver = _Dopri5(counted, y0, rtol, atol, stats, forced_dts=options.get('forced_dts'), norm_reduce=options.get('norm_reduce'))
    elif method == 'rk4':
        solver = _RK4(counted, y0, stats)
    else:
        raise NotImplementedError('oracle restates dopri5 and rk4 only (got {!r})'.format(method))
    solution = solver.integrate(t)
    return solution[0] if tensor_input else solution


# --------------------------------------------------------------------------
# adjoint
# --------------------------------------------------------------------------
def _flatten(sequence):
    flat = [p.contiguous().view(-1) for p in sequence]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


def _flatten_convert_none_to_zeros(sequence, like_sequence):
    flat = [p.contiguous().view(-1) if p is not None else torch.zeros_like(q).view(-1)
            for p, q in zip(sequence, like_sequence)]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


class _OdeintAdjointMethod(torch.autograd.Function):
    @staticmethod
    def forward(ctx, func, rtol, atol, method, options, fwd_stats, bwd_stats, t, flat_params, *y0):
        # replay (test aid): 'forced_dts' drives the forward solve, 'forced_dts_bwd' every backward interval
        fopts = dict(options or {})
        bopts = dict(options or {})
        fopts.pop('forced_dts_bwd', None)
        bopts.pop('forced_dts', None)
        if 'forced_dts_bwd' in bopts:
            bopts['forced_dts'] = bopts.pop('forced_dts_bwd')
        ctx.func, ctx.rtol, ctx.atol, ctx.method, ctx.options = func, rtol, atol, method, bopts
        ctx.bwd_stats = bwd_stats
        with torch.no_grad():
            ans = odeint(func, tuple(y0), t, rtol=rtol, atol=atol, method=method, options=fopts, stats=fwd_stats)
        ctx.save_for_backward(t, flat_params, *ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_output):
        t, flat_params, *ans = ctx.saved_tensors
        ans = tuple(ans)
        func, rtol, atol, method, options = ctx.func, ctx.rtol, ctx.atol, ctx.method, ctx.options
        stats = ctx.bwd_stats if ctx.bwd_stats is not None else SolverStats()
        n_tensors = len(ans)
        f_params = tuple(func.parameters())

        def augmented_dynamics(t, y_aug):
            # (y, a, adj_t, adj_params): the last two are integrated, never read.
            y, adj_y = y_aug[:n_tensors], y_aug[n_tensors:2 * n_tensors]
            with torch.set_grad_enabled(True):
                t = t.to(y[0].device).detach().requires_grad_(True)
                y = tuple(y_.detach().requires_grad_(True) for y_ in y)
                func_eval = func(t, y)
                vjp_t, *vjp_y_and_params = torch.autograd.grad(
                    func_eval, (t,) + y + f_params,
                    tuple(-adj_y_ for adj_y_ in adj_y), allow_unused=True, retain_graph=True)
            vjp_y = vjp_y_and_params[:n_tensors]
            vjp_params = vjp_y_and_params[n_tensors:]
            vjp_t = torch.zeros_like(t) if vjp_t is None else vjp_t
            vjp_y = tuple(torch.zeros_like(y_) if vjp_y_ is None else vjp_y_ for vjp_y_, y_ in zip(vjp_y, y))
            vjp_params = _flatten_convert_none_to_zeros(vjp_params, f_params)
            if len(f_params) == 0:
                vjp_params = torch.tensor(0.).to(vjp_y[0])
            return (*func_eval, *vjp_y, vjp_t, vjp_params)

        T = ans[0].shape[0]
        with torch.no_grad():
            adj_y = tuple(g[-1] for g in grad_output)
            adj_params = torch.zeros_like(flat_params)
            adj_time = torch.tensor(0.).to(t)
            time_vjps = []
            for i in range(T - 1, 0, -1):
                ans_i = tuple(ans_[i] for ans_ in ans)
                grad_output_i = tuple(g[i] for g in grad_output)
                stats.nfe += 1
                func_i = func(t[i], ans_i)
                dLd_cur_t = sum(torch.dot(f_.reshape(-1), g_.reshape(-1)).reshape(1)
                                for f_, g_ in zip(func_i, grad_output_i))
                adj_time = adj_time - dLd_cur_t
                time_vjps.append(dLd_cur_t)
                if adj_params.numel() == 0:
                    adj_params = torch.tensor(0.).to(adj_y[0])
                aug_y0 = (*ans_i, *adj_y, adj_time, adj_params)
                aug_ans = odeint(augmented_dynamics, aug_y0, torch.stack([t[i], t[i - 1]]).to(t),
                                 rtol=rtol, atol=atol, method=method, options=options, stats=stats)
                adj_y = aug_ans[n_tensors:2 * n_tensors]
                adj_time = aug_ans[2 * n_tensors]
                adj_params = aug_ans[2 * n_tensors + 1]
                adj_y = tuple(a_[1] if len(a_) > 0 else a_ for a_ in adj_y)
                if len(adj_time) > 0:
                    adj_time = adj_time[1]
                if len(adj_params) > 0:
                    adj_params = adj_params[1]
                adj_y = tuple(a_ + g[i - 1] for a_, g in zip(adj_y, grad_output))
            time_vjps.append(adj_time)
            time_vjps = torch.cat(time_vjps[::-1])
            return (None, None, None, None, None, None, None, time_vjps, adj_params, *adj_y)


class _TupleFunc(nn.Module):
    def __init__(self, base):
        super().__init__()
        self.base_func = base

    def forward(self, t, y):
        return (self.base_func(t, y[0]),)


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None,
                   fwd_stats: SolverStats = None, bwd_stats: SolverStats = None):
    """`torchdiffeq.odeint_adjoint` restated (continuous adjoint, O(1) memory)."""
    if not isinstance(func, nn.Module):
        raise ValueError('func is required to be an instance of nn.Module.')
    tensor_input = False
    if torch.is_tensor(y0):
        tensor_input = True
        y0 = (y0,)
        func = _TupleFunc(func)
    flat_params = _flatten(func.parameters())
    ys = _OdeintAdjointMethod.apply(func, rtol, atol, method, options, fwd_stats, bwd_stats, t, flat_params, *y0)
    if tensor_input:
        ys = ys[0]
    return ys
