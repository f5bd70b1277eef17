"""`odeint` / `odeint_adjoint` -- the torchdiffeq surface the reference imports
(`/root/reference/model.py:3`) and calls (`model.py:367`), backed by the HIP
library.  Host logic only: argument checking, recognising the conv `ODEfunc`
(`model.py:326-348`), handing raw device pointers to the C ABI, and wiring the
adjoint into autograd.  All arithmetic happens in libnode_hip.so.

No CPU path, no path through `oracle/`: CPU tensors and non-fp32 tensors raise.
Dynamics that are not the reference's Conv-GroupNorm-ReLU `ODEfunc` (any other
nn.Module, `norm='batch'`) and geometries the fused kernels do not tile run the
GENERIC solver (generic.py): the caller's function evaluated as PyTorch
operations, the library's own device-resident step controller around it.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib

PARAM_ATTRS = [
    ('norm1', 'weight'), ('norm1', 'bias'), ('conv1', 'weight'), ('conv1', 'bias'),
    ('norm2', 'weight'), ('norm2', 'bias'), ('conv2', 'weight'), ('conv2', 'bias'),
    ('norm3', 'weight'), ('norm3', 'bias'),
]


class Recognised:
    """The ten parameters + GroupNorm config of a conv ODEfunc (duck-typed so the
    reference's own `model.ODEfunc` instance is accepted unchanged)."""

    def __init__(self, func: nn.Module):
        def conv_of(m):
            layer = getattr(m, '_layer', None)
            if not isinstance(layer, nn.Conv2d):
                raise NotImplementedError('ConcatConv2d with a plain nn.Conv2d `_layer` expected')
            if layer.kernel_size != (3, 3) or layer.stride != (1, 1) or layer.padding != (1, 1) \
                    or layer.dilation != (1, 1) or layer.groups != 1 or layer.bias is None:
                raise NotImplementedError('only Conv2d(dim+1, dim, 3, 1, 1, bias=True) dynamics are accelerated')
            return layer

        for name in ('norm1', 'conv1', 'norm2', 'conv2', 'norm3'):
            if not hasattr(func, name):
                raise NotImplementedError(
                    'neural-ode-features_amd accelerates the reference ODEfunc (model.py:326-348) only; '
                    '%s has no attribute %r' % (type(func).__name__, name))
        norms = [func.norm1, func.norm2, func.norm3]
        for nrm in norms:
            if not isinstance(nrm, nn.GroupNorm):
                raise NotImplementedError("only norm='group' dynamics are accelerated (BatchNorm couples samples)")
            if not nrm.affine:
                raise NotImplementedError('GroupNorm must be affine')
        c1, c2 = conv_of(func.conv1), conv_of(func.conv2)
        dim = norms[0].num_channels
        for nrm in norms:
            if nrm.num_channels != dim or nrm.num_groups != norms[0].num_groups or nrm.eps != norms[0].eps:
                raise NotImplementedError('the three GroupNorms must share channels / groups / eps')
        for cv in (c1, c2):
            if cv.in_channels != dim + 1 or cv.out_channels != dim:
                raise NotImplementedError('ConcatConv2d(dim, dim) expected')
        self.dim, self.groups, self.eps = dim, norms[0].num_groups, float(norms[0].eps)
        self.params: List[torch.Tensor] = [
            func.norm1.weight, func.norm1.bias, c1.weight, c1.bias,
            func.norm2.weight, func.norm2.bias, c2.weight, c2.bias,
            func.norm3.weight, func.norm3.bias]
        # must equal func.parameters() order: defines the flat-gradient layout
        own = list(func.parameters())
        if len(own) != 10 or any(a is not b for a, b in zip(own, self.params)):
            raise NotImplementedError('ODEfunc.parameters() is not the expected ten tensors in reference order')


# Recognised(func) walks the module tree (named_modules, parameters): ~25 us per call, once per solve.  The result is remembered per
# func OBJECT (weakly) together with the identities it was built from -- the five sub-modules, the two conv layers, the ten parameter
# objects, the counts of func's own entries -- and rebuilt when any of them is another object (a replaced layer or parameter, a
# `load_state_dict` keeps the objects and needs nothing).
_REC_SEEN: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _rec_signature(func):
    try:      # (straight through the modules' own dicts: nn.Module.__getattr__ costs ~0.4 us per lookup, this runs once per solve)
        m = func._modules
        n1, c1, n2, c2, n3 = m['norm1'], m['conv1'], m['norm2'], m['conv2'], m['norm3']
        l1, l2 = c1._modules['_layer'], c2._modules['_layer']
        p1, q1, p2, q2, p3 = n1._parameters, l1._parameters, n2._parameters, l2._parameters, n3._parameters
        return (id(n1), id(c1), id(n2), id(c2), id(n3), id(l1), id(l2),
                id(p1['weight']), id(p1['bias']), id(q1['weight']), id(q1['bias']), id(p2['weight']), id(p2['bias']),
                id(q2['weight']), id(q2['bias']), id(p3['weight']), id(p3['bias']), n1.num_groups, n1.eps, len(m), len(func._parameters))
    except (AttributeError, KeyError):
        return None


def recognised(func: nn.Module) -> Recognised:
    """Recognised(func), remembered per object; raises NotImplementedError like the constructor."""
    sig = _rec_signature(func)
    if sig is not None:
        try:
            hit = _REC_SEEN.get(func)
        except TypeError:       # (an unhashable / non-weakly-referenceable object: no cache)
            hit, sig = None, None
        if hit is not None and hit[0] == sig:
            return hit[1]
    rec = Recognised(func)
    if sig is not None:
        _REC_SEEN[func] = (sig, rec)
    return rec


def _method_id(method) -> int:
    if method is None:
        method = 'dopri5'
    if method not in _lib.METHODS:
        # train.py:219 also offers 'adams'; it is not on any graded config
        raise NotImplementedError("method %r is not implemented (have: 'dopri5', 'rk4')" % (method,))
    return _lib.METHODS[method]


HOST_TIMES_ATTR = '_node_host_times'


def tag_host_times(t: torch.Tensor, times) -> torch.Tensor:
    """Attach the host copy of a time grid to the tensor that carries it (`ODEBlock.t1`'s setter builds
    the grid from host values, so the solve never has to read the device tensor back)."""
    setattr(t, HOST_TIMES_ATTR, (t._version, [float(v) for v in times]))
    return t


# Untagged device tensors (the reference's own `ODEBlock` run on top of this package, model.py:366): read back
# once per tensor OBJECT.  The entry holds a weak reference and the tensor's version counter; identity of the
# object is what is compared, never its address -- a freed-and-reallocated grid can share `data_ptr()` with a
# dead one (a `t1` sweep, evaluate.py:116-117, does exactly that), but not a live Python object.
_T_SEEN: Dict[int, tuple] = {}


def _host_times(t: torch.Tensor) -> List[float]:
    """Time grid as host floats."""
    if not torch.is_tensor(t):
        raise TypeError('t must be a tensor')
    if not torch.is_floating_point(t):
        raise TypeError('`t` must be a floating point Tensor but is a {}'.format(t.type()))
    if t.dim() != 1 or t.numel() < 2:
        raise ValueError('t must be one-dimensional with at least two points')
    tag = getattr(t, HOST_TIMES_ATTR, None)
    if tag is not None and tag[0] == t._version and len(tag[1]) == t.numel():
        return list(tag[1])
    if t.device.type == 'cpu':
        return [float(v) for v in t.detach().to(torch.float32).tolist()]
    hit = _T_SEEN.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return list(hit[2])
    vals = [float(v) for v in t.detach().to(torch.float32).cpu().tolist()]     # one read-back per tensor object
    if len(_T_SEEN) > 64:
        for k in [k for k, v in _T_SEEN.items() if v[0]() is None]:
            del _T_SEEN[k]
        if len(_T_SEEN) > 64:
            _T_SEEN.clear()
    _T_SEEN[id(t)] = (weakref.ref(t), t._version, vals)
    return list(vals)


_WS: Dict[Tuple[int, int], torch.Tensor] = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Caller-owned device workspace (the library never allocates device memory)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _aligned_ptr(buf: torch.Tensor) -> int:
    return (buf.data_ptr() + 255) & ~255


def _check_state(y0: torch.Tensor, any_rank: bool = False):
    if not torch.is_tensor(y0):
        raise NotImplementedError('tuple states are not supported: the reference passes a single tensor (model.py:367)')
    if not y0.is_cuda:
        raise RuntimeError('neural-ode-features_amd has no CPU path: y0 must live on a HIP device '
                           '(got %s). The CPU restatement lives in oracle/ and is test-only.' % y0.device)
    if y0.dtype != torch.float32:
        raise TypeError('y0 must be float32 (got %s)' % y0.dtype)
    if y0.dim() != 4 and not any_rank:
        raise ValueError('y0 must be [N, C, H, W]')


# Foreign dynamics / geometries the fused kernels do not tile: the generic solver (generic.py).  False restores the
# round-4 behaviour (NotImplementedError / NODE_ERR_UNSUPPORTED) -- tests of the error surface use it.
GENERIC_FALLBACK = True


def _fused_plan(func, y0, method_id, adjoint, n_t):
    """Recognised(func) when the fused kernels take this (func, state); None -> the generic solver."""
    try:
        rec = recognised(func)
    except NotImplementedError:
        if not GENERIC_FALLBACK:
            raise
        return None
    if y0.dim() != 4:
        if not GENERIC_FALLBACK:
            raise ValueError('y0 must be [N, C, H, W]')
        return None
    if not GENERIC_FALLBACK:
        return rec
    n, c, h, w = y0.shape
    if c != rec.dim:
        raise ValueError('state has %d channels but the ODEfunc was built for %d' % (c, rec.dim))
    shape = _lib.NodeShape(n, c, h, w, rec.groups, rec.eps)
    return rec if _lib.load().node_workspace_bytes(C.byref(shape), method_id, 1 if adjoint else 0, n_t) != 0 else None


def _shape_struct(y0: torch.Tensor, rec: Recognised) -> _lib.NodeShape:
    n, c, h, w = y0.shape
    if c != rec.dim:
        raise ValueError('state has %d channels but the ODEfunc was built for %d' % (c, rec.dim))
    return _lib.NodeShape(n, c, h, w, rec.groups, rec.eps)


def _params_struct(params: List[torch.Tensor], device) -> Tuple[_lib.NodeParams, List[torch.Tensor]]:
    keep = []
    ptrs = []
    for p in params:
        q = p.detach()
        if q.device != device or q.dtype != torch.float32:
            raise RuntimeError('ODEfunc parameters must be float32 on %s' % (device,))
        if not q.is_contiguous():
            q = q.contiguous()
        keep.append(q)
        ptrs.append(q.data_ptr())
    return _lib.NodeParams(*ptrs), keep


def _global_norm_group(options: Optional[dict]):
    """The process group of a GLOBAL-NORM solve (`options['global_norm']`: True = the default group, or a group), or None:
    no option, torch.distributed not initialised, or a world of one."""
    g = (options or {}).get('global_norm')
    if g is None or g is False:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = None if g is True else g
    return (group,) if dist.get_world_size(group) > 1 else None


def _plain(options: Optional[dict]) -> bool:
    """no option that deferred completion cannot serve (replay lists, dt logs, step limits); `global_norm` can ride along"""
    return not options or all(k == 'global_norm' for k in options)


def _opts_struct(options: Optional[dict], key: str, blind: Optional[tuple] = None, grad_last_only: bool = False, device=None):
    """(struct-or-None, keepalive).  blind = (steps, record device tensor, miss flag device tensor) for a solve
    with deferred completion."""
    options = options or {}
    forced = options.get(key)
    max_steps = int(options.get('max_num_steps', 0) or 0)
    record = int(options.get('record_dt', 0) or 0)
    gn = _global_norm_group(options) if device is not None else None
    if forced is None and max_steps == 0 and record == 0 and blind is None and not grad_last_only and gn is None:
        return None, None
    o = _lib.NodeSolveOpts()
    keep = {}
    if gn is not None:
        # GLOBAL-NORM mode (include/node_hip.h; SURVEY.md 8e, collective 2): the library packs this rank's sums of every step decision
        # into `buf` on the solve's stream and calls back; the all-reduce is enqueued behind them (RCCL orders its work against the
        # current stream, which IS the solve's; gloo -- the CPU tests -- completes before it returns)
        import torch.distributed as dist
        buf = torch.zeros(8, dtype=torch.float32, device=device)
        group = gn[0]

        def _reduce(_ctx, _ptr, _n, _stream):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)

        cb = _lib.NORM_REDUCE_FN(_reduce)
        keep['norm_buf'], keep['norm_cb'] = buf, cb
        o.norm_reduce = C.cast(cb, C.c_void_p)
        o.norm_buf = buf.data_ptr()
        o.norm_world = dist.get_world_size(group)
    o.max_num_steps = max_steps
    o.grad_last_only = 1 if grad_last_only else 0
    if blind is not None:
        o.blind_steps = int(blind[0])
        o.record = blind[1].data_ptr()
        o.miss_flag = blind[2].data_ptr() if blind[2] is not None else None
    if forced is not None:
        arr = (C.c_double * len(forced))(*[float(v) for v in forced])
        keep['forced'] = arr
        o.n_forced_dt = len(forced)
        o.forced_dt = C.cast(arr, C.POINTER(C.c_double))
    if record > 0:
        log = (C.c_double * record)()
        cnt = C.c_int32(0)
        keep['log'] = log
        keep['cnt'] = cnt
        o.record_dt = record
        o.dt_log = C.cast(log, C.POINTER(C.c_double))
        o.n_dt_log = C.pointer(cnt)
    return o, keep


def _stats_dict(stats: _lib.NodeStats, keep) -> dict:
    out = stats.as_dict()
    if keep and 'log' in keep:
        n = keep['cnt'].value
        vals = list(keep['log'][:n])
        out['dts'] = [abs(v) for v in vals]
        out['accepts'] = [v > 0 for v in vals]
    return out


def solve_forward(rec: Recognised, params: List[torch.Tensor], y0: torch.Tensor, times: List[float],
                  rtol: float, atol: float, method_id: int, options: Optional[dict], blind: Optional[tuple] = None):
    lib = _lib.load()
    y0c = y0.detach().contiguous()
    shape = _shape_struct(y0c, rec)
    pstruct, keep_p = _params_struct(params, y0c.device)
    n_t = len(times)
    with torch.cuda.device(y0c.device):
        ws_bytes = lib.node_workspace_bytes(C.byref(shape), method_id, 0, n_t)
        if ws_bytes == 0:
            raise _lib.NodeHipError(-3, lib.node_last_error().decode())
        ws = _workspace(y0c.device, ws_bytes)
        out = torch.empty((n_t,) + tuple(y0c.shape), dtype=torch.float32, device=y0c.device)
        tarr = (C.c_float * n_t)(*times)
        stats = _lib.NodeStats()
        opts, keep_o = _opts_struct(options, 'forced_dts', blind, device=y0c.device)
        rc = lib.node_solve_fwd(C.byref(shape), C.byref(pstruct), y0c.data_ptr(), tarr, n_t,
                                float(rtol), float(atol), method_id,
                                C.byref(opts) if opts is not None else None,
                                out.data_ptr(), C.byref(stats), _aligned_ptr(ws), ws_bytes,
                                torch.cuda.current_stream(y0c.device).cuda_stream)
    _lib.check(rc)
    del keep_p
    return out, _stats_dict(stats, keep_o)


def solve_adjoint(rec: Recognised, params: List[torch.Tensor], y_traj: torch.Tensor, grad_out: torch.Tensor,
                  times: List[float], rtol: float, atol: float, method_id: int, options: Optional[dict],
                  want_grad_t: bool = False, blind: Optional[tuple] = None, grad_last_only: bool = False):
    """`grad_last_only`: `grad_out` is dL/d(y_traj[-1]) alone, shape [N, C, H, W]; all other slices are zero."""
    lib = _lib.load()
    y_traj = y_traj.detach().contiguous()
    grad_out = grad_out.detach().contiguous()
    dev = y_traj.device
    shape = _shape_struct(y_traj[0], rec)
    pstruct, keep_p = _params_struct(params, dev)
    n_t = len(times)
    with torch.cuda.device(dev):
        ws_bytes = lib.node_workspace_bytes(C.byref(shape), method_id, 1, n_t)
        if ws_bytes == 0:
            raise _lib.NodeHipError(-3, lib.node_last_error().decode())
        ws = _workspace(dev, ws_bytes)
        P = lib.node_param_count(C.byref(shape))
        grad_y0 = torch.empty_like(y_traj[0])
        grad_p = torch.empty(P, dtype=torch.float32, device=dev)
        grad_t = torch.empty(n_t, dtype=torch.float32, device=dev) if want_grad_t else None
        tarr = (C.c_float * n_t)(*times)
        stats = _lib.NodeStats()
        opts, keep_o = _opts_struct(options, 'forced_dts_bwd', blind, grad_last_only, device=dev)
        rc = lib.node_solve_adjoint(C.byref(shape), C.byref(pstruct), y_traj.data_ptr(), grad_out.data_ptr(),
                                    tarr, n_t, float(rtol), float(atol), method_id,
                                    C.byref(opts) if opts is not None else None,
                                    grad_y0.data_ptr(), grad_p.data_ptr(),
                                    grad_t.data_ptr() if grad_t is not None else None,
                                    C.byref(stats), _aligned_ptr(ws), ws_bytes,
                                    torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc)
    del keep_p
    return grad_y0, grad_p, grad_t, _stats_dict(stats, keep_o)


def solve_backprop(rec: Recognised, params: List[torch.Tensor], y0: torch.Tensor, grad_out: torch.Tensor,
                   times: List[float], step_dts: List[float], method_id: int, rtol: float = 0.0, atol: float = 0.0):
    """Backward of the non-adjoint `odeint`: backpropagation through the forward solve's accepted steps.  `rtol`,
    `atol`: the forward solve's (they select the convolution kernels the replay runs, like the forward solve did)."""
    lib = _lib.load()
    y0 = y0.detach().contiguous()
    grad_out = grad_out.detach().contiguous()
    dev = y0.device
    shape = _shape_struct(y0, rec)
    pstruct, keep_p = _params_struct(params, dev)
    n_t = len(times)
    n_steps = len(step_dts) if method_id == _lib.METHOD_DOPRI5 else n_t - 1
    with torch.cuda.device(dev):
        ws_bytes = lib.node_backprop_workspace_bytes(C.byref(shape), method_id, n_t, n_steps)
        if ws_bytes == 0:
            raise _lib.NodeHipError(-3, lib.node_last_error().decode())
        ws = _workspace(dev, ws_bytes)
        grad_y0 = torch.empty_like(y0)
        grad_p = torch.empty(lib.node_param_count(C.byref(shape)), dtype=torch.float32, device=dev)
        tarr = (C.c_float * n_t)(*times)
        darr = (C.c_double * max(1, len(step_dts)))(*step_dts)
        rc = lib.node_solve_backprop(C.byref(shape), C.byref(pstruct), y0.data_ptr(), tarr, n_t, darr, n_steps,
                                     float(rtol), float(atol), method_id,
                                     grad_out.data_ptr(), grad_y0.data_ptr(), grad_p.data_ptr(), _aligned_ptr(ws), ws_bytes,
                                     torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc)
    del keep_p
    return grad_y0, grad_p


class Deferred:
    """Deferred completion of the solves of a training step: keep the queue fed across the solver.

    A dopri5 solve normally ends with one read-back of the device controller (did the steps enqueued finish the
    interval?), and the host dispatches everything behind it only afterwards -- the GPU idles while PyTorch launches
    the head, and again while it launches the stem's backward.  With deferred completion enabled, a solve whose step
    count is known from the previous iteration enqueues exactly that many steps and returns at once; whether they
    were the steps needed is written by the device into a record, and a MISS (the interval unfinished after them, or an
    error status; finishing EARLY is no miss -- the steps past the end return at once) bumps the device flag
    `miss_flag`.  The output of a missed solve is its y0 (the trajectory buffer is pre-filled), and every number
    computed from it -- loss, accuracy -- belongs to a step whose update was skipped: `step_verdicts()` lists which.  Nothing that commits results may run unconditionally:
    `optim.FusedSGD.skip_flag = deferred.miss_flag` predicates the parameter update on the device, so a step with a
    miss changes nothing (under data parallelism the flag rides in the reducer's last bucket, so every rank skips
    together).  The host looks at a record one iteration later, when it is long complete: the true step count in it
    becomes the next guess; a miss makes the next solve of that kind run with a read-back again.  While a solve's
    count is still moving one spare step is enqueued, so only a count that jumps by two or more is a miss.

        deferred = integrate.Deferred(device)        # opt-in; the drop-in API is unaffected while none is active
        opt.use_deferred(deferred)                   # FusedSGD: the update is predicated on the flag and zeroes it
        with deferred:                               # (nothing runs blind in a scope no optimizer was armed with)
            for x, y in loader:
                loss = F.cross_entropy(model(x), y); loss.backward(); opt.step(); opt.zero_grad()
        deferred.resolve()                           # -> number of steps whose update was skipped

    `func.nfe` advances by the predicted count at once and is corrected when the record is read (sums over an
    epoch are exact); `last_*_stats` of a deferred solve hold the prediction."""

    active = None      # the instance whose `with` block is open
    # How many steps to enqueue for a solve whose true count nobody knows yet.  A step past the end of the interval returns
    # at its first instruction in every kernel, but its ~40 launches still cost ~0.1 ms; a miss costs the iteration and
    # (DeferredLoop) the repetition of two batches: ~150x more.  So: enqueue the LARGEST count of the last HIST solves of
    # this kind (a count that wobbles by one between iterations then wastes half a step on average; a count that DROPS is
    # followed after HIST iterations), plus ONE spare step whenever a larger count is not unlikely:
    #   * the history is short (CALM);
    #   * the last solve overshot the end of its interval by less than FRAGILE of its last step (the record carries the
    #     last step: node_step_record.t_prev / dt_used) -- step sizes that come out a few percent smaller on the next batch
    #     then need one step more;
    #   * a count above everything in the history was seen within the last QUIET solves.
    # Margins: FRAGILE 0.15 / QUIET 16 in round 4; the measured sweep on the cfg-3 bench loop (profiles/r04_deferred_policy_cfg3.txt)
    # has 0.05 / 8 at +1.3 % images/s with 0 misses in 200 steps, soaked again in round 5 (profiles/r05_deferred_soak_cfg3.txt).
    # Round 3 enqueued last count + 1 until eight exact predictions in a row -- at tol 1e-5 that never happened, and every
    # solve carried one or two dead steps (profiles/r03_r_cfg3_steps.txt: 29 dead component GEMMs per step).
    HIST = int(os.environ.get('NODE_DEFERRED_HIST', 8))
    CALM = 4
    FRAGILE = float(os.environ.get('NODE_DEFERRED_FRAGILE', 0.05))      # (environment: A/B measurements, tools/deferred_soak.py)
    QUIET = int(os.environ.get('NODE_DEFERRED_QUIET', 8))

    def __init__(self, device):
        self.device = torch.device(device)
        self.miss_flag = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.guess: Dict[tuple, Optional[int]] = {}
        self.calm: Dict[tuple, int] = {}          # solves of a key since its history was last reset
        self.hist: Dict[tuple, List[int]] = {}    # true step counts of the last HIST solves of a key
        self.fragile: Dict[tuple, bool] = {}      # the last solve of a key barely reached the end of its interval
        self.quiet: Dict[tuple, int] = {}         # solves of a key since one needed more steps than any in its history
        self.pending: Dict[tuple, tuple] = {}
        self._seq: Dict[tuple, int] = {}
        self.records: Dict[tuple, tuple] = {}
        self.misses = 0
        self.blind_solves = 0
        self.dead_steps = 0        # steps enqueued past the end of their interval (read from the records)
        self.verdicts: List[tuple] = []   # (blind solve index, key kind, missed?) in the order the records were read
        self.armed = False         # set by FusedSGD.use_deferred: without a predicated commit point nothing runs blind
        self._last_enqueued: Dict[tuple, int] = {}

    def __enter__(self):
        Deferred.active = self
        return self

    def __exit__(self, *exc):
        Deferred.active = None
        return False

    def _buffers(self, key):
        b = self.records.get(key)
        if b is None:
            n = C.sizeof(_lib.NodeStepRecord)
            b = (torch.zeros(n, dtype=torch.uint8, device=self.device), torch.zeros(n, dtype=torch.uint8).pin_memory(),
                 torch.cuda.Event())
            self.records[key] = b
        return b

    def plan(self, key, func=None):
        """Steps to enqueue blind for `key`, or None for a solve with a read-back.  Looks first at the record the
        previous blind solve of this key left (one iteration old: complete): its true step count becomes the new
        guess (and corrects `func.nfe`, which was advanced by the guess), a miss sends the next solve back to a
        read-back.  While a key's count is still moving, ONE spare step is enqueued (a step past the end of the
        interval returns at once on the device: ~0.1 ms of launches) so that a count that grows by one is no miss."""
        pend = self.pending.pop(key, None)
        if pend is not None:
            guessed, fref, enqueued = pend
            if func is None and fref is not None:
                func = fref()
            dev, host, event = self._buffers(key)
            event.synchronize()
            r = _lib.NodeStepRecord.from_buffer_copy(bytes(host.numpy().tobytes()))
            self.verdicts.append((self._seq.pop(key, -1), key[0], bool(r.miss)))
            if len(self.verdicts) > 4096:
                del self.verdicts[:2048]
            if r.miss:
                self.misses += 1
                self.guess[key] = None
                self.calm[key] = 0
                self.hist[key] = []
            else:
                if func is not None and r.steps != guessed:
                    func.nfe = getattr(func, 'nfe', 0) + 6 * (r.steps - guessed)    # late, but sums stay exact
                self.dead_steps += max(0, enqueued - int(r.steps))
                # solver time runs upwards in both directions (a solve towards smaller t integrates -t): how far past the
                # end of the interval did the last step land, in units of that step?
                end = key[-1][-1] if key[0] == 'fwd' else -key[-1][0]
                self.fragile[key] = bool(r.dt_used > 0.0 and (r.t - end) < self.FRAGILE * r.dt_used)
                self._observe(key, int(r.steps))
        g = self.guess.get(key)
        if not g or not self.armed:
            return None
        return g, self._enqueue(key)

    def _observe(self, key, steps):
        h = self.hist.setdefault(key, [])
        self.quiet[key] = 0 if (h and int(steps) > max(h)) else self.quiet.get(key, self.QUIET) + 1
        h.append(int(steps))
        del h[:-self.HIST]
        self.calm[key] = self.calm.get(key, 0) + 1
        self.guess[key] = int(steps)

    def _spare(self, key):
        """One spare step while a count above the history's maximum is not unlikely (see the class constants)."""
        h = self.hist.get(key) or []
        return len(h) < self.CALM or self.fragile.get(key, False) or self.quiet.get(key, self.QUIET) < self.QUIET

    def _enqueue(self, key):
        h = self.hist.get(key) or [self.guess[key]]
        return max(h) + (1 if self._spare(key) else 0)

    def blind_args(self, key, enqueue):
        dev, _, _ = self._buffers(key)
        self._last_enqueued[key] = int(enqueue)
        return (enqueue, dev, self.miss_flag)

    def launched(self, key, guessed, func=None):
        dev, host, event = self._buffers(key)
        host.copy_(dev, non_blocking=True)
        event.record(torch.cuda.current_stream(self.device))
        self.pending[key] = (guessed, weakref.ref(func) if func is not None else None, self._last_enqueued.pop(key, guessed))
        self._seq[key] = self.blind_solves
        self.blind_solves += 1

    def learned(self, key, steps):
        """A solve of `key` ran with a read-back: its count starts a new history."""
        self.hist[key] = []
        self.calm[key] = 0
        self.fragile[key] = False
        self.quiet[key] = 0
        self._observe(key, int(steps))

    def forget(self):
        """Drop every step-count guess (the next solve of each kind runs with a read-back and learns its count anew)."""
        for k in list(self.guess):
            self.guess[k] = None
            self.calm[k] = 0
            self.hist[k] = []

    def force_counts(self, counts):
        """Tests: pretend every kind of solve (or those in the {key: n} mapping) has needed exactly n steps for a long
        time -- the next blind solve enqueues n steps and no spare one."""
        for k in list(self.guess):
            n = counts.get(k) if isinstance(counts, dict) else counts
            if n is None:
                continue
            self.guess[k] = int(n)
            self.hist[k] = [int(n)] * self.CALM
            self.calm[k] = self.CALM
            self.fragile[k] = False
            self.quiet[k] = self.QUIET

    def step_verdicts(self):
        """[(blind solve index, 'fwd' | 'bwd', missed)] for the records read so far (one iteration late): a caller that
        logs losses or drives an LR schedule from them drops the iterations whose solves missed."""
        return list(self.verdicts)

    def settled(self):
        """True once every kind of solve seen so far runs blind with exactly its last count (no spare step, no larger count
        in the history window):
        from then on a training step enqueues no launch that returns at once."""
        return bool(self.guess) and all(g and not self._spare(k) and max(self.hist[k]) == self.hist[k][-1]
                                        for k, g in self.guess.items())

    def resolve(self):
        """Wait for every outstanding record (a synchronisation point) and count the misses."""
        for key in list(self.pending):
            self.plan(key)
        return self.misses


class DeferredLoop:
    """Training steps under deferred completion that never lose an update: a miss costs time, not training data.

    `Deferred` alone SKIPS the optimizer step of an iteration one of whose solves missed.  Here the device flag is
    STICKY -- nothing zeroes it behind the optimizer step, so once a solve has missed, that update and every later one
    is skipped on the device -- and the host keeps each batch until its verdict is in.  `lag` iterations later (the copy
    of the flag as the optimizer step saw it is long complete by then: no stall) the host reads the verdict; on a miss
    it drains the queue, zeroes the flag, forgets the step-count guesses, and runs the voided batches again IN ORDER
    with a read-back per solve, each under the random-generator state its first attempt started from.  The sequence
    of committed updates is therefore exactly the synchronous run's (tests/test_gpu_deferred.py: parameters
    bit-identical after a forced miss).  Under data parallelism the flag every rank reads is the all-reduced one
    (`dp.GradientReducer.carry_flag`), so all ranks void and repeat the same batches together.

        loop = integrate.DeferredLoop(deferred, opt, step_fn, reducer)     # step_fn(*batch) -> anything
        for x, y in loader:
            for result in loop.step(x, y):      # results of batches whose update is now known to be committed
                log(result)
        for result in loop.flush(): log(result)
    """

    def __init__(self, deferred: 'Deferred', opt, step_fn, reducer=None, lag: int = 1):
        self.d, self.opt, self.step_fn, self.lag = deferred, opt, step_fn, max(0, int(lag))
        opt.use_deferred(deferred, reducer)
        opt.flags_to_reset = []            # sticky: only the host clears the flag, after it has seen the miss
        self.queue: List[list] = []        # [batch, result, rng state, pinned flag copy, event]
        self.retries = 0                   # batches run a second time
        self.miss_events = 0
        self.steps = 0
        self._pool: List[tuple] = []       # (pinned float, event) pairs of settled entries, reused

    def _rng(self):
        return torch.cuda.get_rng_state(self.d.device), torch.get_rng_state()

    def _set_rng(self, st):
        torch.cuda.set_rng_state(st[0], self.d.device)
        torch.set_rng_state(st[1])

    def _run(self, batch):
        with self.d:
            return self.step_fn(*batch)

    def _settle(self, keep: int):
        """Pop committed entries until `keep` remain; on a miss repeat everything queued.  Returns committed results."""
        out = []
        while len(self.queue) > keep:
            batch, result, rng, host, event = self.queue[0]
            event.synchronize()
            if float(host[0]) == 0.0:
                self.queue.pop(0)
                self._pool.append((host, event))
                out.append(result)
                continue
            # a solve of this iteration (on some rank) missed: its update and every later one were skipped on the device
            self.miss_events += 1
            torch.cuda.synchronize(self.d.device)
            self.d.resolve()
            self.d.miss_flag.zero_()
            self.d.forget()
            voided, self.queue = self.queue, []
            flag, self.opt.skip_flag = self.opt.skip_flag, None
            armed, self.d.armed = self.d.armed, False
            try:
                for b, _, r, _, _ in voided:
                    self._set_rng(r)      # the generator state the first attempt started from: after the last repeat
                    out.append(self._run(b))   # it stands where the synchronous run's stands
                    self.retries += 1
            finally:
                self.opt.skip_flag = flag
                self.d.armed = armed
        return out

    def step(self, *batch):
        done = self._settle(self.lag)
        rng = self._rng()
        result = self._run(batch)
        host, event = self._pool.pop() if self._pool else (torch.zeros(1, dtype=torch.float32).pin_memory(), torch.cuda.Event())
        src = self.opt.skip_flag if self.opt.skip_flag is not None else self.d.miss_flag
        host.copy_(src, non_blocking=True)      # the flag as this iteration's optimizer step saw it
        event.record(torch.cuda.current_stream(self.d.device))
        self.queue.append([batch, result, rng, host, event])
        self.steps += 1
        return done

    def flush(self):
        return self._settle(0)


_TOKENS = itertools.count(1)
_TOKEN_OF: 'weakref.WeakKeyDictionary' = weakref.WeakKeyDictionary()


def _func_token(func):
    """A key for `func` that is never reused and never shared: id() can be handed to a new object after the old one
    was collected (a new dynamics function would inherit a stale step-count guess), and an attribute on the module
    would travel with `copy.deepcopy(model)` (an EMA / evaluation copy would share its original's pending record).
    Tokens therefore live in a weak-keyed table beside the modules, not on them."""
    try:
        tok = _TOKEN_OF.get(func)
        if tok is None:
            tok = _TOKEN_OF[func] = next(_TOKENS)
        return tok
    except TypeError:          # not weak-referenceable / unhashable: fall back to identity
        return id(func)


BACKPROP_LOG = 4096      # step sizes the forward solve can record for the non-adjoint backward (csrc: STEP_LIST_CAP)


class _HipOdeint(torch.autograd.Function):
    """Forward = node_solve_fwd under no_grad; backward = node_solve_adjoint
    (continuous adjoint, what `odeint_adjoint` does upstream)."""

    @staticmethod
    def forward(ctx, func, rec, times, rtol, atol, method_id, options, adjoint, wants_grad, last_only, y0, *params):
        if not adjoint and wants_grad:
            options = dict(options or {})          # the backward replays the accepted steps: log their sizes
            options['record_dt'] = max(int(options.get('record_dt', 0) or 0), BACKPROP_LOG)
        # deferred completion only where a predicated commit point follows: training solves (a gradient is wanted)
        # inside an armed `Deferred` scope; inference, sweeps and the drop-in API always return finished solves
        d = Deferred.active if (adjoint and wants_grad and method_id == _lib.METHOD_DOPRI5 and len(times) == 2
                                and _plain(options)) else None
        if d is not None and d.device != y0.device:
            d = None
        dkey = ('fwd', _func_token(func), tuple(y0.shape), rtol, atol, tuple(times)) if d is not None else None
        steps = d.plan(dkey, func) if d is not None else None
        if steps:
            out, st = solve_forward(rec, list(params), y0, times, rtol, atol, method_id, options, blind=d.blind_args(dkey, steps[1]))
            st['nfe'] -= 6 * (steps[1] - steps[0])      # advance the counter by the guess; the record corrects it
            st['accepted'] = steps[0]
            d.launched(dkey, steps[0], func)
        else:
            out, st = solve_forward(rec, list(params), y0, times, rtol, atol, method_id, options)
            if d is not None:
                d.learned(dkey, st['accepted'] + st['rejected'])
        func.nfe = getattr(func, 'nfe', 0) + st['nfe']          # model.py:340 convention
        func.last_forward_stats = st
        ctx.func, ctx.rec, ctx.times = func, rec, times
        ctx.rtol, ctx.atol, ctx.method_id, ctx.options = rtol, atol, method_id, options
        ctx.adjoint = adjoint
        ctx.step_dts = None
        if not adjoint and 'dts' in st:
            if st['accepted'] + st['rejected'] > len(st['dts']):
                raise RuntimeError('the forward solve took more than %d steps: too many for the non-adjoint backward; '
                                   'use odeint_adjoint' % BACKPROP_LOG)
            ctx.step_dts = [d for d, a in zip(st['dts'], st['accepts']) if a]
        ctx.save_for_backward(out, *params)
        ctx.last_only = bool(last_only)
        # last_only: the caller keeps y(t[-1]) alone (ODEBlock.return_last_only); its gradient then arrives as one
        # slice -- no [T, N, C, H, W] tensor of zeros is built by autograd, none is read by the adjoint
        return out[-1] if last_only else out

    @staticmethod
    def backward(ctx, grad_out):
        out, *params = ctx.saved_tensors
        if ctx.adjoint:
            d = Deferred.active if (ctx.method_id == _lib.METHOD_DOPRI5 and len(ctx.times) == 2 and _plain(ctx.options)) else None
            if d is not None and d.device != out.device:
                d = None
            dkey = ('bwd', _func_token(ctx.func), tuple(out.shape[1:]), ctx.rtol, ctx.atol, tuple(ctx.times)) if d is not None else None
            steps = d.plan(dkey, ctx.func) if d is not None else None
            if steps:
                gy0, gp, _, st = solve_adjoint(ctx.rec, params, out, grad_out, ctx.times, ctx.rtol, ctx.atol,
                                               ctx.method_id, ctx.options, blind=d.blind_args(dkey, steps[1]),
                                               grad_last_only=ctx.last_only)
                st['nfe'] -= 6 * (steps[1] - steps[0])
                st['accepted'] = steps[0]
                d.launched(dkey, steps[0], ctx.func)
            else:
                gy0, gp, _, st = solve_adjoint(ctx.rec, params, out, grad_out, ctx.times, ctx.rtol, ctx.atol,
                                               ctx.method_id, ctx.options, grad_last_only=ctx.last_only)
                if d is not None:
                    d.learned(dkey, st['accepted'] + st['rejected'])
            ctx.func.nfe = getattr(ctx.func, 'nfe', 0) + st['nfe']
            ctx.func.last_backward_stats = st
        else:
            # upstream's non-adjoint backward is plain autograd through the solver's operations: it never calls
            # func.forward, so the reference's NFE counter does not move (model.py:340)
            if ctx.last_only:        # the tape walk wants the cotangent of every output slice
                full = torch.zeros_like(out)
                full[-1] = grad_out
                grad_out = full
            gy0, gp = solve_backprop(ctx.rec, params, out[0], grad_out, ctx.times, ctx.step_dts or [], ctx.method_id,
                                     ctx.rtol, ctx.atol)
            ctx.func.last_backward_stats = {'nfe': 0, 'accepted': len(ctx.step_dts or []), 'rejected': 0, 'status': 0}
        grads = []
        off = 0
        for p in params:
            n = p.numel()
            grads.append(gp[off:off + n].view_as(p))
            off += n
        return (None, None, None, None, None, None, None, None, None, None, gy0, *grads)


def _odeint_impl(func, y0, t, rtol, atol, method, options, adjoint=True, last_only=False):
    _check_state(y0, any_rank=GENERIC_FALLBACK)
    if not isinstance(func, nn.Module):
        raise ValueError('func is required to be an instance of nn.Module.')
    method_id = _method_id(method)
    times = _host_times(t)
    inc = all(b > a for a, b in zip(times[:-1], times[1:]))
    dec = all(b < a for a, b in zip(times[:-1], times[1:]))
    if not (inc or dec):
        raise ValueError('t must be strictly increasing or strictly decreasing')
    rec = _fused_plan(func, y0, method_id, adjoint, len(times))
    if rec is None:
        # any other nn.Module, or a geometry outside the fused kernels' tiling: the caller's function under the library's
        # device-resident step controller (generic.py).  Replay lists / dt logs are features of the fused solves.
        if options and any(k in options for k in ('forced_dts', 'forced_dts_bwd', 'record_dt')):
            raise NotImplementedError('replay / dt-log options are not available on the generic solver')
        from . import generic
        out = generic.odeint_generic(func, y0, times, rtol, atol, method_id, options)
        return out[-1] if last_only else out
    # (grad mode is off inside autograd.Function.forward: whether a gradient will be wanted is decided here)
    wants_grad = torch.is_grad_enabled() and (y0.requires_grad or any(p.requires_grad for p in rec.params))
    return _HipOdeint.apply(func, rec, times, float(rtol), float(atol), method_id, options, adjoint, wants_grad, last_only, y0,
                            *rec.params)


def solve_last(func, y0, t, rtol, atol, method, adjoint, options=None):
    """y(t[-1]) alone, [N, C, H, W]: what `ODEBlock.forward` returns with `return_last_only` (model.py:368-369),
    without materialising the gradient of the slices nobody keeps."""
    return _odeint_impl(func, y0, t, rtol, atol, method, options, adjoint=adjoint, last_only=True)


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None):
    """Drop-in for `torchdiffeq.odeint_adjoint` on the reference's call
    (model.py:359,367).  Returns `[len(t), *y0.shape]`, `out[0] == y0`;
    gradients flow to `y0` and `func.parameters()` through the HIP adjoint solve."""
    return _odeint_impl(func, y0, t, rtol, atol, method, options)


def odeint(func, y0, t, rtol=1e-7, atol=1e-12, method=None, options=None):
    """Drop-in for `torchdiffeq.odeint` on the reference's call (model.py:359,367).

    Forward values are identical to `odeint_adjoint`.  The gradient is what upstream's
    autograd produces by differentiating through the solver: backpropagation through
    the accepted steps (`node_solve_backprop`: stage derivatives on a tape, one VJP of
    the dynamics per stage evaluation), with the step sizes held constant; like upstream's
    backward it adds nothing to `func.nfe`."""
    return _odeint_impl(func, y0, t, rtol, atol, method, options, adjoint=False)


# ---------------------------------------------------------------------------
# thin wrappers over the two single-eval entry points (tests, smoke, profiling)
# ---------------------------------------------------------------------------
def odefunc_forward(func, t: float, y: torch.Tensor) -> torch.Tensor:
    _check_state(y)
    lib = _lib.load()
    rec = Recognised(func)
    yc = y.detach().contiguous()
    shape = _shape_struct(yc, rec)
    pstruct, keep = _params_struct(rec.params, yc.device)
    with torch.cuda.device(yc.device):
        ws_bytes = lib.node_workspace_bytes(C.byref(shape), 0, 0, 2)
        ws = _workspace(yc.device, ws_bytes)
        out = torch.empty_like(yc)
        rc = lib.node_odefunc_fwd(C.byref(shape), C.byref(pstruct), float(t), yc.data_ptr(), out.data_ptr(),
                                  _aligned_ptr(ws), ws_bytes, torch.cuda.current_stream(yc.device).cuda_stream)
    _lib.check(rc)
    del keep
    return out


def odefunc_vjp(func, t: float, y: torch.Tensor, cot: torch.Tensor):
    """(f, vjp_y, vjp_t, vjp_params_flat) == autograd.grad(f, (t, y, *params), cot)."""
    _check_state(y)
    lib = _lib.load()
    rec = Recognised(func)
    yc, cc = y.detach().contiguous(), cot.detach().contiguous()
    shape = _shape_struct(yc, rec)
    pstruct, keep = _params_struct(rec.params, yc.device)
    with torch.cuda.device(yc.device):
        ws_bytes = lib.node_workspace_bytes(C.byref(shape), 0, 1, 2)
        ws = _workspace(yc.device, ws_bytes)
        f = torch.empty_like(yc)
        vy = torch.empty_like(yc)
        vt = torch.empty(1, dtype=torch.float32, device=yc.device)
        vp = torch.empty(lib.node_param_count(C.byref(shape)), dtype=torch.float32, device=yc.device)
        rc = lib.node_odefunc_vjp(C.byref(shape), C.byref(pstruct), float(t), yc.data_ptr(), cc.data_ptr(),
                                  f.data_ptr(), vy.data_ptr(), vt.data_ptr(), vp.data_ptr(),
                                  _aligned_ptr(ws), ws_bytes, torch.cuda.current_stream(yc.device).cuda_stream)
    _lib.check(rc)
    del keep
    return f, vy, vt, vp


def profile_begin():
    _lib.check(_lib.load().node_profile_begin())


def profile_end() -> dict:
    prof = _lib.NodeProfile()
    _lib.check(_lib.load().node_profile_end(C.byref(prof)))
    # classes 3..8: the GroupNorm / transform passes of the F(4x4,3x3) pipeline, `flops` = algorithmic BYTES there
    names = ['conv3x3_implicit_gemm', 'wgrad_gemm', 'w4_component_gemm', 'w4s_pass<0,1> combine', 'w4s_pass<1,0> forward',
             'w4s_pass<1,1> forward + next combine', 'w4s_pass<1,2> forward + backward top', 'w4s_pass<2,0> backward',
             'w4s_pass<2,1> backward + next combine']
    return {names[i]: {'launches': int(prof.launches[i]), 'total_ms': float(prof.total_ms[i]),
                       'flops': float(prof.flops[i])} for i in range(len(names))}
