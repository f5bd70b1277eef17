"""Generic solver: `odeint` / `odeint_adjoint` for dynamics the fused kernels do not take.

`torchdiffeq.odeint[_adjoint]` accepts ANY `nn.Module` as `func` (`/root/reference/model.py:367`), and the reference
itself offers dynamics outside the fused path: `--norm batch` (`train.py:202`, `model.py:274`), channel counts that are
not multiples of four, images beyond 32x32.  For a HIP fp32 state those solves run HERE: the caller's `func` is evaluated
as ordinary PyTorch operations on the current stream, and everything else -- stage states, stage times, the Hairer
initial step, the mixed error norm per tensor, accept / reject, the next step size, quartic dense output, FSAL -- is
done on the device by the same controller kernels the fused solves use (`node_flat_*`, csrc/node_api.hip:
`k_lincomb`, `k_init_norms`, `k_error_norm`, `k_step_controller`, `k_emit_flat`, `k_commit`).  The host takes no
decision: it enqueues as many steps as the previous solve of the same problem needed and reads the controller back once.

Nothing here imports `oracle/`; CPU and non-fp32 tensors still raise (integrate._check_state).

Algorithm = the fused path's: dopri5 with torchdiffeq's 2019 controller, rk4 3/8 rule on the `t` grid,
continuous adjoint on the augmented state (y, a, adj_t, adj_params) with one `torch.autograd.grad` per evaluation.

`func` must be PURE in the sense upstream's solvers assume as well: the solver enqueues as many steps as the previous solve of the
same problem took before it reads the controller back, so steps past the end of the interval still CALL `func` (on stage states that are
then discarded; only `func.nfe` is corrected afterwards).  Modules with side effects -- BatchNorm running statistics, dropout RNG,
counters, hooks -- see those extra calls; `BatchNorm2d(track_running_stats=False)`, what the reference's `norm='batch'` is
(model.py:268-271), has none.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib

_GUESS: Dict[tuple, int] = {}       # steps the last solve of the same problem took (enqueued before the first read-back)


def _aligned(nbytes, device):
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return buf, (buf.data_ptr() + 255) & ~255


class FlatSolve:
    """One flat state of 1..3 tensor segments (+ an optional scalar kept in the controller) and its buffers."""

    def __init__(self, numels: Sequence[int], has_scalar: bool, rtol: float, atol: float, device, n_targets: int):
        self.lib = _lib.load()
        self.device = device
        self.nseg = len(numels)
        assert 1 <= self.nseg <= 3
        self.y = [torch.empty(n, dtype=torch.float32, device=device) for n in numels]
        self.y1 = [torch.empty(n, dtype=torch.float32, device=device) for n in numels]
        self.k = [[torch.zeros(n, dtype=torch.float32, device=device) for _ in range(7)] for n in numels]
        self.stage = [torch.empty(n, dtype=torch.float32, device=device) for n in numels]
        self.t_stage = torch.zeros(1, dtype=torch.float32, device=device)
        self.n_targets = 0
        self.has_scalar, self.rtol, self.atol = bool(has_scalar), float(rtol), float(atol)
        self._ws = None
        self.s = _lib.NodeFlatSolve()
        self.s.nseg, self.s.has_scalar = self.nseg, 1 if has_scalar else 0
        self.s.rtol, self.s.atol, self.s.tsign = float(rtol), float(atol), 1.0
        for i, n in enumerate(numels):
            sg = self.s.seg[i]
            sg.y, sg.y1, sg.n = self.y[i].data_ptr(), self.y1[i].data_ptr(), n
            for j in range(7):
                sg.k[j] = self.k[i][j].data_ptr()
        self._reserve(n_targets)

    def _reserve(self, n_targets):
        if self._ws is None or n_targets > self.n_targets:
            nbytes = self.lib.node_flat_workspace_bytes(int(n_targets))
            self._ws, base = _aligned(nbytes, self.device)
            self.s.ws, self.s.ws_bytes = base, nbytes
        self.n_targets = n_targets
        self.s.n_targets = n_targets

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def begin(self, t0: float, targets: Sequence[float], tsign: float, first_dt: float = 0.0, new_solve: bool = True):
        self._reserve(len(targets))
        self.s.tsign = float(tsign)
        arr = (C.c_double * len(targets))(*[float(v) for v in targets])
        _lib.check(self.lib.node_flat_begin(C.byref(self.s), float(t0), arr, float(first_dt), 1 if new_solve else 0, self.stream))

    def stage_state(self, method_id: int, stage: int, into_y1: bool = False):
        """Stage states -> self.stage (or self.y1), stage time -> self.t_stage.  Returns the list of state tensors."""
        if stage == _lib.FLAT_F0:
            _lib.check(self.lib.node_flat_stage(C.byref(self.s), method_id, stage, None, self.t_stage.data_ptr(), self.stream))
            return self.y
        dst = self.y1 if into_y1 else self.stage
        ptrs = (C.c_void_p * self.nseg)(*[t.data_ptr() for t in dst])
        _lib.check(self.lib.node_flat_stage(C.byref(self.s), method_id, stage, ptrs, self.t_stage.data_ptr(), self.stream))
        return dst

    def scalar(self, which: int, src: torch.Tensor, scale: float, accumulate: bool = False):
        _lib.check(self.lib.node_flat_scalar(C.byref(self.s), which, src.data_ptr(), float(scale), 1 if accumulate else 0, self.stream))

    def initial_step(self, phase: int):
        _lib.check(self.lib.node_flat_initial_step(C.byref(self.s), phase, self.stream))

    def finish_step(self, method_id: int, y_out: Optional[torch.Tensor] = None):
        _lib.check(self.lib.node_flat_finish_step(C.byref(self.s), method_id, None if y_out is None else y_out.data_ptr(), self.stream))

    def status(self) -> _lib.NodeFlatStatus:
        st = _lib.NodeFlatStatus()
        _lib.check(self.lib.node_flat_status_read(C.byref(self.s), C.byref(st), self.stream))
        return st


def _raise_for(status: int, dt: float):
    if status == 0:
        return
    msg = {-5: 'max_num_steps exceeded', -6: 'non-finite error norm / state', -7: 'underflow in dt %g' % dt}.get(status, 'solver status %d' % status)
    raise _lib.NodeHipError(status, msg)


def _drive_dopri5(fs: FlatSolve, evaluate, key, max_steps: int, y_out=None):
    """The dopri5 steps of the current interval.  `evaluate(kslot, states)` evaluates the dynamics at fs.t_stage / `states`
    and stores tsign * derivative into the stage-derivative buffers `kslot`.  Returns (controller status, steps enqueued)."""
    D = _lib.METHOD_DOPRI5
    batch = max(1, _GUESS.get(key, 1))
    enq = 0
    while True:
        batch = min(batch, max_steps - enq)
        for _ in range(batch):
            for s in range(6):
                evaluate(s + 1, fs.stage_state(D, s, into_y1=(s == 5)))
            fs.finish_step(D, y_out)
        enq += batch
        st = fs.status()
        _raise_for(st.status, st.dt)
        if st.done:
            _GUESS[key] = st.steps
            if len(_GUESS) > 256:
                _GUESS.pop(next(iter(_GUESS)))
            return st, enq
        if enq >= max_steps:
            _raise_for(-5, st.dt)
        batch = 2


def _first_step(fs: FlatSolve, evaluate):
    """f0 (the FSAL seed) and the Hairer initial step (one probe evaluation, upstream: + 1 NFE)."""
    D = _lib.METHOD_DOPRI5
    evaluate(0, fs.stage_state(D, _lib.FLAT_F0))
    fs.initial_step(0)
    evaluate(1, fs.stage_state(D, _lib.FLAT_PROBE))
    fs.initial_step(1)


def _fix_nfe(func, extra_evals: int):
    """Steps enqueued past the end of an interval evaluated `func` on the device although the solver ignored the
    results: the reference's counter (model.py:340, bumped inside func.forward) must not see them."""
    if extra_evals and hasattr(func, 'nfe'):
        try:
            func.nfe -= extra_evals
        except Exception:
            pass


def solve_forward(func, y0: torch.Tensor, times: List[float], rtol: float, atol: float, method_id: int,
                  max_steps: int = 2 ** 31 - 1):
    """[len(times), *y0.shape] with out[0] == y0; func evaluated under no_grad (what `odeint_adjoint`'s forward does)."""
    dev, shape, n = y0.device, tuple(y0.shape), y0.numel()
    tsign = -1.0 if times[1] < times[0] else 1.0
    ts = [tsign * float(t) for t in times]
    out = torch.empty((len(times),) + shape, dtype=torch.float32, device=dev)
    out[0].copy_(y0)
    with torch.cuda.device(dev), torch.no_grad():
        fs = FlatSolve([n], False, rtol, atol, dev, len(times) - 1)
        fs.y[0].copy_(y0.reshape(-1))

        def evaluate(kslot, states):
            f = func(fs.t_stage[0], states[0].view(shape))
            torch.mul(f.reshape(-1), tsign, out=fs.k[0][kslot])

        stats = {'accepted': 0, 'rejected': 0, 'status': 0}
        if method_id == _lib.METHOD_RK4:
            R = _lib.METHOD_RK4
            for j in range(1, len(ts)):
                t0f, t1f = _f32(ts[j - 1]), _f32(ts[j])          # upstream keeps the fixed grid in the state dtype
                fs.begin(t0f, [t1f], tsign, first_dt=_f32(t1f - t0f), new_solve=(j == 1))
                evaluate(0, fs.stage_state(R, _lib.FLAT_F0))
                for s in (1, 2, 3):
                    evaluate(s, fs.stage_state(R, s))
                fs.finish_step(R)
                out[j].copy_(fs.y[0].view(shape))
                stats['accepted'] += 1
            return out, stats
        fs.begin(ts[0], ts[1:], tsign, new_solve=True)
        _first_step(fs, evaluate)
        key = ('fwd', id(type(func)), shape, rtol, atol, tuple(times))
        st, enq = _drive_dopri5(fs, evaluate, key, max_steps, y_out=out[1:])
        _fix_nfe(func, 6 * (enq - st.steps))
        stats.update(accepted=st.accepted, rejected=st.rejected, first_dt=st.first_dt, t_final=st.t, last_dt=st.dt)
    return out, stats


def _f32(v: float) -> float:
    return float(torch.tensor(v, dtype=torch.float32))


def solve_adjoint(func, y_traj: torch.Tensor, grad_out: torch.Tensor, times: List[float], rtol: float, atol: float,
                  method_id: int, max_steps: int = 2 ** 31 - 1):
    """Continuous adjoint (what `odeint_adjoint`'s backward does upstream): returns (grad_y0, [grad per parameter])."""
    dev, shape, n = y_traj.device, tuple(y_traj.shape[1:]), y_traj[0].numel()
    params = [p for p in func.parameters() if p.requires_grad]
    sizes = [p.numel() for p in params]
    P = sum(sizes)
    numels = [n, n] + ([P] if P > 0 else [])
    T = len(times)
    with torch.cuda.device(dev):
        fs = FlatSolve(numels, True, rtol, atol, dev, 1)
        fs.y[1].copy_(grad_out[-1].reshape(-1))           # adj_y = grad_output[-1]
        if P > 0:
            fs.y[2].zero_()                               # adj_params = 0
        tsign_box = [1.0]
        zero1 = torch.zeros(1, dtype=torch.float32, device=dev)

        def evaluate(kslot, states):
            tsign = tsign_box[0]
            with torch.enable_grad():
                t_ = fs.t_stage[0].detach().clone().requires_grad_(True)
                y_ = states[0].detach().view(shape).requires_grad_(True)
                f = func(t_, y_)
                a = states[1].detach().view(shape)
                grads = torch.autograd.grad(f, (t_, y_) + tuple(params), -a, allow_unused=True)
            with torch.no_grad():
                torch.mul(f.detach().reshape(-1), tsign, out=fs.k[0][kslot])
                if grads[1] is None:
                    fs.k[1][kslot].zero_()
                else:
                    torch.mul(grads[1].reshape(-1), tsign, out=fs.k[1][kslot])
                fs.scalar(kslot, grads[0].reshape(1) if grads[0] is not None else zero1, tsign)
                off = 0
                for g, m in zip(grads[2:], sizes):
                    dst = fs.k[2][kslot][off:off + m]
                    if g is None:
                        dst.zero_()
                    else:
                        torch.mul(g.reshape(-1), tsign, out=dst)
                    off += m

        accepted = rejected = 0
        for i in range(T - 1, 0, -1):
            decreasing = times[i - 1] < times[i]
            tsign = -1.0 if decreasing else 1.0
            tsign_box[0] = tsign
            s0, s1 = tsign * float(times[i]), tsign * float(times[i - 1])
            with torch.no_grad():
                fs.y[0].copy_(y_traj[i].reshape(-1))
                # func_i = f(t_i, y_i); adj_time -= <func_i, grad_output_i>  (upstream evaluates f here and again as the
                # first stage of the augmented solve: both calls happen, like upstream -- the reference's NFE counter sees both)
                ti = torch.tensor(float(times[i]), dtype=torch.float32, device=dev)
                fi = func(ti, y_traj[i])
                dot = (fi * grad_out[i]).sum().reshape(1)
            if method_id == _lib.METHOD_RK4:
                R = _lib.METHOD_RK4
                t0f, t1f = _f32(s0), _f32(s1)
                fs.begin(t0f, [t1f], tsign, first_dt=_f32(t1f - t0f), new_solve=(i == T - 1))
                fs.scalar(-1, dot, -1.0, accumulate=True)
                evaluate(0, fs.stage_state(R, _lib.FLAT_F0))
                for s in (1, 2, 3):
                    evaluate(s, fs.stage_state(R, s))
                fs.finish_step(R)
                accepted += 1
            else:
                fs.begin(s0, [s1], tsign, new_solve=(i == T - 1))
                fs.scalar(-1, dot, -1.0, accumulate=True)
                _first_step(fs, evaluate)
                key = ('bwd', id(type(func)), shape, rtol, atol, float(times[i]), float(times[i - 1]))
                st, enq = _drive_dopri5(fs, evaluate, key, max_steps)
                _fix_nfe(func, 6 * (enq - st.steps))
                accepted, rejected = st.accepted, st.rejected          # cumulative over the intervals
            with torch.no_grad():
                fs.y[1].add_(grad_out[i - 1].reshape(-1))             # adj_y += grad_output[i - 1]
        grad_y0 = fs.y[1].view(shape).clone()
        grads, off = [], 0
        for p, m in zip(params, sizes):
            grads.append(fs.y[2][off:off + m].view_as(p).clone())
            off += m
    return grad_y0, params, grads, {'accepted': accepted, 'rejected': rejected, 'status': 0}


class _GenericOdeint(torch.autograd.Function):
    """Forward = the generic solve under no_grad; backward = the generic continuous adjoint (for `odeint` too: upstream
    differentiates through the solver's own operations there -- same gradient up to O(tolerance))."""

    @staticmethod
    def forward(ctx, func, times, rtol, atol, method_id, options, y0, *params):
        max_steps = int((options or {}).get('max_num_steps', 0) or 0) or 2 ** 31 - 1
        out, st = solve_forward(func, y0.detach().contiguous(), times, rtol, atol, method_id, max_steps)
        ctx.func, ctx.times, ctx.rtol, ctx.atol, ctx.method_id, ctx.max_steps = func, times, rtol, atol, method_id, max_steps
        ctx.n_params = len(params)
        ctx.save_for_backward(out)
        try:
            func.last_forward_stats = st
        except Exception:
            pass
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        gy0, params, grads, st = solve_adjoint(ctx.func, out, grad_out.contiguous(), ctx.times, ctx.rtol, ctx.atol,
                                               ctx.method_id, ctx.max_steps)
        try:
            ctx.func.last_backward_stats = st
        except Exception:
            pass
        by_id = {id(p): g for p, g in zip(params, grads)}
        pg = [by_id.get(id(p)) for p in ctx.func.parameters()][:ctx.n_params]
        return (None, None, None, None, None, None, gy0, *pg)


def odeint_generic(func, y0, times, rtol, atol, method_id, options=None):
    params = tuple(func.parameters()) if isinstance(func, torch.nn.Module) else ()
    return _GenericOdeint.apply(func, times, float(rtol), float(atol), method_id, options, y0, *params)
