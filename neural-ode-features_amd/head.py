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
        gpart = torch.empty(n, 2, c, device=z.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(lib.node_head_bwd(shape, _ptr(z), _ptr(g), _ptr(b), _ptr(scale), _ptr(stats), _ptr(g_pooled),
                                     _ptr(dz), _ptr(gpart), stream))
        gsum = gpart.sum(0)
        return dz, gsum[0], gsum[1], None, None, None


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
        gpart = torch.empty(n, 2, c, device=z.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(z.device).cuda_stream
        with torch.cuda.device(z.device):
            _lib.check(lib.node_gn_relu_bwd(shape, _ptr(z), _ptr(g), _ptr(b), _ptr(stats), relu, _ptr(g_out), _ptr(dz),
                                        _ptr(gpart), stream))
        gsum = gpart.sum(0)
        return dz, gsum[0], gsum[1], None, None, None


def gn_relu(z: torch.Tensor, norm: torch.nn.GroupNorm, relu: bool = True) -> torch.Tensor:
    """`relu(norm(z))` for CUDA fp32 4-D inputs through the fused HIP kernels; the plain modules otherwise."""
    if z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and isinstance(norm, torch.nn.GroupNorm) and norm.affine:
        return _GnRelu.apply(z, norm.weight, norm.bias, norm.num_groups, norm.eps, relu)
    out = norm(z)
    return F.relu(out) if relu else out
