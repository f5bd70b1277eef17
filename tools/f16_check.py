#!/usr/bin/env python
"""fp16-pair component GEMMs (k_w4_gemm64h) against the bf16-triple ones (k_w4_gemm64b) and an fp64 convolution, through the
diagnostic F(4x4,3x3) convolution: error of both at several input / filter magnitudes (the power-of-two scales must make the
result independent of them), and subnormal handling.  GPU."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from neural_ode_features_amd import _lib

lib = _lib.load()


def conv(x, w, dgrad=0):
    N, Cc, H, W = x.shape
    shape = _lib.NodeShape(N, Cc, H, W, min(32, Cc), 1e-5)
    nbytes = lib.node_conv3x3_w4_workspace_bytes(C.byref(shape))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
    base = (ws.data_ptr() + 255) & ~255
    y = torch.empty_like(x)
    _lib.check(lib.node_conv3x3_w4(C.byref(shape), w.data_ptr(), int(dgrad), x.data_ptr(), y.data_ptr(), base, nbytes,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return y


worst = 0.0
for shape in ((128, 256, 8, 8), (16, 64, 8, 8), (4, 256, 16, 16), (16, 128, 8, 8), (8, 1024, 16, 16), (32, 512, 8, 8)):
    for xs, wsc in ((1.0, 1.0), (1e-4, 1.0), (1e3, 1.0), (1.0, 1e-3), (1.0, 30.0), (3e-7, 1e-2)):
        for dgrad in (0, 1):
            N, Cc, H, W = shape
            gen = torch.Generator().manual_seed(5)
            x = (torch.randn(N, Cc, H, W, generator=gen).relu() * xs).cuda()
            w = (((torch.rand(Cc, Cc + 1, 3, 3, generator=gen) * 2 - 1) / (9 * Cc) ** 0.5) * wsc).cuda()
            wd = w[:, 1:].double()
            ref = F.conv_transpose2d(x.double(), wd, padding=1) if dgrad else F.conv2d(x.double(), wd, padding=1)
            errs = {}
            for f16 in ('1', '0'):
                os.environ['NODE_TUNE_W4_F16'] = f16
                got = conv(x, w, dgrad)
                errs[f16] = float((got.double() - ref).abs().max() / ref.abs().max())
            del os.environ['NODE_TUNE_W4_F16']
            print('%-18s x*%-7g w*%-6g %s  fp16 pairs %.2e   bf16 triples %.2e' % (shape, xs, wsc, 'dgrad' if dgrad else 'fwd  ', errs['1'], errs['0']), flush=True)
            worst = max(worst, errs['1'] / max(errs['0'], 1e-7))
print('worst ratio pairs / triples: %.2f' % worst)
assert worst < 1.6, worst
