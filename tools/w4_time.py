#!/usr/bin/env python
"""Run the F(4x4,3x3) diagnostic convolution a few times at the cfg-2 shape (under rocprofv3 --kernel-trace --stats:
per-kernel durations of k_w4_gemm and the stand-alone transforms)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_ode_features_amd import _lib

lib = _lib.load()
N, Cc = 128, 256
shape = _lib.NodeShape(N, Cc, 8, 8, 32, 1e-5)
x = torch.randn(N, Cc, 8, 8, device='cuda')
w = torch.randn(Cc, Cc + 1, 3, 3, device='cuda') / 48
nbytes = lib.node_conv3x3_w4_workspace_bytes(C.byref(shape))
ws = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
base = (ws.data_ptr() + 255) & ~255
y = torch.empty_like(x)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    _lib.check(lib.node_conv3x3_w4(C.byref(shape), w.data_ptr(), 0, x.data_ptr(), y.data_ptr(), base, nbytes,
                                   torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print('ok', float(y.abs().max()))
