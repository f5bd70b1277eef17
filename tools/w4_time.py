#!/usr/bin/env python
"""Run the F(4x4,3x3) diagnostic convolution a few times (under rocprofv3 --kernel-trace --stats: per-kernel durations of
the component GEMM and the stand-alone transforms):  w4_time.py [iterations] [N,C,side]   (default: the cfg-2 shape)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_ode_features_amd import _lib

if os.environ.get('NODE_HIP_LIB_AB'):      # A/B of two builds of the library on one box
    _lib.LIB_PATH = os.environ['NODE_HIP_LIB_AB']
lib = _lib.load()
N, Cc, side = (int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else (128, 256, 8)))    # cfg 5: 64,1024,16
shape = _lib.NodeShape(N, Cc, side, side, 32, 1e-5)
x = torch.randn(N, Cc, side, side, device='cuda')
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
