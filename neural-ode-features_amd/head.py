"""Fused classifier head: GroupNorm -> ReLU -> global average pool -> [Dropout] -> Flatten
(`/root/reference/model.py:231-250`, FCClassifier without its Linear layer) as ONE HIP launch forward
and one backward (`node_head_fwd / node_head_bwd`, csrc/kernels_head.hip) instead of ~15 launch-bound
PyTorch kernels.  The dropout mask is drawn by PyTorch's own generator (so seeds reproduce what
`nn.Dropout` would have drawn on the pooled `[N, C, 1, 1]` tensor) and handed to the kernel as a
per-(sample, channel) scale.  No CPU or PyTorch fallback inside: callers route non-CUDA / non-fp32
inputs to the plain module sequence themselves (`FCClassifier.forward`)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _HeadPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, scale, groups, eps):
        lib = _lib.load()
        z = z.contiguous()
        n, c, h, w = z.shape
        shape = _lib.NodeShape(n, c, h, w, groups, eps)
        pooled = torch.empty(n, c, device=z.device, dtype=torch.float32)
        stats = torch.empty(n, groups, 2, device=z.device, dtype=torch.float32)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        stream = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(lib.node_head_fwd(shape, _ptr(z), _ptr(g), _ptr(b), _ptr(scale), _ptr(pooled), _ptr(stats), stream))
        ctx.save_for_backward(z, g, b, scale, stats)
        ctx.shape = (n, c, h, w, groups, eps)
        return pooled

    @staticmethod
    def backward(ctx, g_pooled):
        lib = _lib.load()
        z, g, b, scale, stats = ctx.saved_tensors
        n, c, h, w, groups, eps = ctx.shape
        shape = _lib.NodeShape(n, c, h, w, groups, eps)
        g_pooled = g_pooled.contiguous()
        dz = torch.empty_like(z)
        gpart = torch.empty(n + 1, 2, c, device=z.device, dtype=torch.float32)     # per-sample partials, then their sums
        dgamma, dbeta = gpart[n].unbind(0)
        stream = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(lib.node_head_bwd(shape, _ptr(z), _ptr(g), _ptr(b), _ptr(scale), _ptr(stats), _ptr(g_pooled),
                                     _ptr(dz), _ptr(gpart), _ptr(dgamma), stream))
        return dz, dgamma, dbeta, None, None, None


def head_pool(z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
              p: float = 0.0, training: bool = False) -> torch.Tensor:
    """pooled[n, c] = dropout(mean_px relu(GroupNorm(z)))   -- model.py:239-247."""
    scale = None
    if training and p > 0.0:
        # exactly the numbers nn.Dropout would multiply the pooled [N, C, 1, 1] tensor by
        scale = F.dropout(torch.ones(z.shape[0], z.shape[1], 1, 1, device=z.device, dtype=torch.float32), p, True)
        scale = scale.reshape(z.shape[0], z.shape[1]).contiguous()
    return _HeadPool.apply(z, gamma, beta, scale, groups, eps)


class _GnRelu(torch.autograd.Function):
    """out = [relu](GroupNorm(z)) -- the stem's `relu(norm(x))` pairs (model.py:304-307), one launch each way."""

    @staticmethod
    def forward(ctx, z, gamma, beta, groups, eps, relu):
        lib = _lib.load()
        z = z.contiguous()
        n, c, h, w = z.shape
        shape = _lib.NodeShape(n, c, h, w, groups, eps)
        out = torch.empty_like(z)
        stats = torch.empty(n, groups, 2, device=z.device, dtype=torch.float32)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        stream = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(lib.node_gn_relu_fwd(shape, _ptr(z), _ptr(g), _ptr(b), int(relu), _ptr(out), _ptr(stats), stream))
        ctx.save_for_backward(z, g, b, stats)
        ctx.shape = (n, c, h, w, groups, eps, int(relu))
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        z, g, b, stats = ctx.saved_tensors
        n, c, h, w, groups, eps, relu = ctx.shape
        shape = _lib.NodeShape(n, c, h, w, groups, eps)
        g_out = g_out.contiguous()
        dz = torch.empty_like(z)
        gpart = torch.empty(n + 1, 2, c, device=z.device, dtype=torch.float32)     # per-sample partials, then their sums
        dgamma, dbeta = gpart[n].unbind(0)
        stream = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(lib.node_gn_relu_bwd(shape, _ptr(z), _ptr(g), _ptr(b), _ptr(stats), relu, _ptr(g_out), _ptr(dz),
                                        _ptr(gpart), _ptr(dgamma), stream))
        return dz, dgamma, dbeta, None, None, None


def gn_relu(z: torch.Tensor, norm: torch.nn.GroupNorm, relu: bool = True) -> torch.Tensor:
    """`relu(norm(z))` for CUDA fp32 4-D inputs through the fused HIP kernels; the plain modules otherwise."""
    if z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and isinstance(norm, torch.nn.GroupNorm) and norm.affine:
        return _GnRelu.apply(z, norm.weight, norm.bias, norm.num_groups, norm.eps, relu)
    out = norm(z)
    return F.relu(out) if relu else out


# ---------------------------------------------------------------------------------------------------------------
# Linear + cross-entropy (model.py:244-250 `nn.Linear(in_ch, out)`, train.py:43 `F.cross_entropy(p, y)`): one launch each
# way (node_head_loss_fwd / node_head_loss_bwd, csrc/kernels_loss.hip) instead of 3 hipBLASLt + ~10 ATen launches.
# ---------------------------------------------------------------------------------------------------------------
import ctypes as C

_LOSS_SCRATCH = {}
_LINEAR_ATTR = '_node_linear_inputs'         # on a logits tensor: the (pooled, weight, bias) `linear` computed it from


def _scratch(device, n):
    """Partials + arrival counter of the loss reduction: zero before the first use, left at zero by every launch; one per
    (device, stream)."""
    lib = _lib.load()
    need = lib.node_head_loss_scratch_bytes(int(n)) // 4
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _LOSS_SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.zeros(max(need, 1024), dtype=torch.float32, device=device)
        _LOSS_SCRATCH[key] = buf
    return buf


def _loss_struct(n, c, classes, reduction, pooled=None, weight=None, bias=None, target=None, logits=None, loss=None,
                 stat=None, scratch=None):
    return _lib.NodeHeadLoss(n, c, classes, reduction, _ptr(pooled), _ptr(weight), _ptr(bias), _ptr(target), _ptr(logits),
                             _ptr(loss), _ptr(stat), _ptr(scratch))


def _fusable_linear(x, weight, bias):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and weight.is_cuda and weight.dtype == torch.float32
            and weight.dim() == 2 and weight.shape[1] == x.shape[1] and weight.shape[0] <= 1024
            and (bias is None or (bias.is_cuda and bias.dtype == torch.float32)))


class _Linear(torch.autograd.Function):
    """logits = pooled @ W^T + b (one launch); backward from a given dL/dlogits (one launch)."""

    @staticmethod
    def forward(ctx, pooled, weight, bias):
        lib = _lib.load()
        p, w = pooled.detach().contiguous(), weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        n, c = p.shape
        o = w.shape[0]
        logits = torch.empty(n, o, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(lib.node_head_loss_fwd(C.byref(_loss_struct(n, c, o, 0, pooled=p, weight=w, bias=b, logits=logits)),
                                              torch.cuda.current_stream(p.device).cuda_stream))
        ctx.save_for_backward(p, w)
        ctx.has_bias = bias is not None
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        lib = _lib.load()
        p, w = ctx.saved_tensors
        n, c = p.shape
        o = w.shape[0]
        g = g_logits.contiguous()
        dp, dw = torch.empty_like(p), torch.empty_like(w)
        db = torch.empty(o, dtype=torch.float32, device=p.device) if ctx.has_bias else None
        grads = _lib.NodeHeadLossGrad(None, _ptr(g), None, _ptr(dp), _ptr(dw), _ptr(db))
        with torch.cuda.device(p.device):
            _lib.check(lib.node_head_loss_bwd(C.byref(_loss_struct(n, c, o, 0, pooled=p, weight=w)), C.byref(grads),
                                              torch.cuda.current_stream(p.device).cuda_stream))
        return dp, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """`F.linear` for the classifier's last layer: the library's kernel for [N, C] fp32 inputs on a HIP device (at most
    1024 outputs), PyTorch's otherwise.  The result remembers what it was computed from, so that `cross_entropy` on it
    can run Linear's and the loss's backward as one launch."""
    if not _fusable_linear(x, weight, bias):
        return F.linear(x, weight, bias)
    out = _Linear.apply(x, weight, bias)
    if torch.is_grad_enabled() and out.requires_grad:
        # (with the version of the result: an in-place edit of the logits between `linear` and `cross_entropy` -- `logits.div_(T)`,
        #  `masked_fill_` -- must take the ordinary autograd edge, not the fused backward that assumes logits = x W^T + b)
        setattr(out, _LINEAR_ATTR, (x, weight, bias, out._version))
    return out


class _CrossEntropy(torch.autograd.Function):
    """loss = CE(logits, target) for given logits: one launch forward, one backward (d_logits)."""

    @staticmethod
    def forward(ctx, logits, target, reduction):
        lib = _lib.load()
        lg = logits.detach().contiguous()
        n, o = lg.shape
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        stat = torch.empty(2, dtype=torch.float32, device=lg.device)       # {loss, correct predictions}
        with torch.cuda.device(lg.device):
            h = _loss_struct(n, 0, o, reduction, target=target, logits=lg, loss=loss, stat=stat, scratch=_scratch(lg.device, n))
            _lib.check(lib.node_head_loss_fwd(C.byref(h), torch.cuda.current_stream(lg.device).cuda_stream))
        ctx.save_for_backward(lg, target)
        ctx.reduction = reduction
        ctx.mark_non_differentiable(stat)
        return loss, stat

    @staticmethod
    def backward(ctx, g_loss, _g_stat):
        lib = _lib.load()
        lg, target = ctx.saved_tensors
        n, o = lg.shape
        dl = torch.empty_like(lg)
        g = g_loss.contiguous().reshape(1)
        grads = _lib.NodeHeadLossGrad(_ptr(g), None, _ptr(dl), None, None, None)
        with torch.cuda.device(lg.device):
            _lib.check(lib.node_head_loss_bwd(C.byref(_loss_struct(n, 0, o, ctx.reduction, target=target, logits=lg)), C.byref(grads),
                                              torch.cuda.current_stream(lg.device).cuda_stream))
        return dl, None, None


class _LinearCrossEntropy(torch.autograd.Function):
    """loss = CE(pooled @ W^T + b, target).  `logits` given: only the loss is computed forward (the Linear launch has
    happened: `linear`); None: Linear and loss in ONE forward launch.  Backward: ONE launch for d_pooled, d_weight, d_bias."""

    @staticmethod
    def forward(ctx, pooled, weight, bias, target, reduction, logits):
        lib = _lib.load()
        p, w = pooled.detach().contiguous(), weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        n, c = p.shape
        o = w.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        stat = torch.empty(2, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            scratch = _scratch(p.device, n)
            if logits is None:
                lg = torch.empty(n, o, dtype=torch.float32, device=p.device)
                h = _loss_struct(n, c, o, reduction, pooled=p, weight=w, bias=b, target=target, logits=lg, loss=loss,
                                 stat=stat, scratch=scratch)
            else:
                lg = logits.detach().contiguous()
                h = _loss_struct(n, 0, o, reduction, target=target, logits=lg, loss=loss, stat=stat, scratch=scratch)
            _lib.check(lib.node_head_loss_fwd(C.byref(h), torch.cuda.current_stream(p.device).cuda_stream))
        ctx.save_for_backward(p, w, lg, target)
        ctx.reduction, ctx.has_bias = reduction, bias is not None
        ctx.mark_non_differentiable(stat, lg)
        return loss, stat, lg

    @staticmethod
    def backward(ctx, g_loss, _g_stat, _g_logits):
        lib = _lib.load()
        p, w, lg, target = ctx.saved_tensors
        n, c = p.shape
        o = w.shape[0]
        dp, dw = torch.empty_like(p), torch.empty_like(w)
        db = torch.empty(o, dtype=torch.float32, device=p.device) if ctx.has_bias else None
        g = g_loss.contiguous().reshape(1)
        grads = _lib.NodeHeadLossGrad(_ptr(g), None, None, _ptr(dp), _ptr(dw), _ptr(db))
        with torch.cuda.device(p.device):
            h = _loss_struct(n, c, o, ctx.reduction, pooled=p, weight=w, target=target, logits=lg)
            _lib.check(lib.node_head_loss_bwd(C.byref(h), C.byref(grads), torch.cuda.current_stream(p.device).cuda_stream))
        return dp, dw, db, None, None, None


_REDUCTIONS = {'mean': _lib.REDUCE_MEAN, 'sum': _lib.REDUCE_SUM}


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, reduction: str = 'mean') -> torch.Tensor:
    """`F.cross_entropy(logits, target, reduction=...)` (train.py:43, :93) on the library's kernels: one launch forward,
    one backward; when `logits` came from `linear` (the classifier's last layer), that backward launch also produces
    the Linear layer's gradients.  The returned 0-d tensor carries `.node_stat` = device tensor {loss, correct
    predictions} for loops that read their running sums once per logging interval (train.py:44,46 read them per batch).
    CPU / non-fp32 / >1024-class inputs, class-probability targets and 'none' reduction go to PyTorch."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[1] <= 1024
            and target.is_cuda and target.dtype == torch.int64 and target.dim() == 1 and target.shape[0] == logits.shape[0]
            and reduction in _REDUCTIONS):
        return F.cross_entropy(logits, target, reduction=reduction)
    target = target.contiguous()
    src = getattr(logits, _LINEAR_ATTR, None) if torch.is_grad_enabled() else None
    if src is not None and (logits._version != src[3] or logits._backward_hooks or logits.retains_grad):
        src = None      # edited in place since `linear`, or somebody wants to see its gradient: the plain edge through `_Linear`
    if src is not None:
        # (the logits go in DETACHED: as an autograd input of the fused node they would pull `linear`'s own backward into the graph --
        #  called with a materialised zero gradient: a second, useless launch per step, seen in profiles/r05_cfg2_steps.txt's first cut)
        loss, stat, _ = _LinearCrossEntropy.apply(src[0], src[1], src[2], target, _REDUCTIONS[reduction], logits.detach())
    else:
        loss, stat = _CrossEntropy.apply(logits, target, _REDUCTIONS[reduction])
    loss.node_stat = stat
    return loss


def linear_cross_entropy(pooled, weight, bias, target, reduction: str = 'mean'):
    """(loss, logits) with Linear AND loss in one forward launch -- for loops that hand the targets to the model
    (`ODENet.loss`)."""
    if not (_fusable_linear(pooled, weight, bias) and target.is_cuda and target.dtype == torch.int64 and reduction in _REDUCTIONS):
        logits = F.linear(pooled, weight, bias)
        return F.cross_entropy(logits, target, reduction=reduction), logits
    loss, stat, logits = _LinearCrossEntropy.apply(pooled, weight, bias, target.contiguous(), _REDUCTIONS[reduction], None)
    loss.node_stat = stat
    return loss, logits
