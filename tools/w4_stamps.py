#!/usr/bin/env python
"""Where the time of one k_w4_gemm64b launch goes: wall-clock stamps (100 MHz) written by every wave (NODE_TUNE_W4_STAMPS), relative
to the first wave's start: [0] wave started, [1] operand ring requested, [2] the first ring turn (four K steps) arrived and
multiplied, [3] K loop done, [4] own component's stores issued, [5] shared component multiplied, [6] all stores issued,
[7] this wave's stores drained.     python tools/w4_stamps.py [N,C,side]"""
import ctypes as C
import os
import sys

os.environ['NODE_HIP_DIAG'] = '1'     # the stamps live in libnode_hip_diag.so (build.py --diag)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_ode_features_amd import _lib

lib = _lib.load()
N, Cc, side = (int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else (128, 256, 8)))
shape = _lib.NodeShape(N, Cc, side, side, 32, 1e-5)
x = torch.randn(N, Cc, side, side, device='cuda')
w = torch.randn(Cc, Cc + 1, 3, 3, device='cuda') / 48
nbytes = lib.node_conv3x3_w4_workspace_bytes(C.byref(shape))
ws = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
base = (ws.data_ptr() + 255) & ~255
y = torch.empty_like(x)
grid = (N // 16) * (Cc // 64) * 8
stamps = torch.zeros(grid * 4, 16, dtype=torch.int64, device='cuda')


def run():
    _lib.check(lib.node_conv3x3_w4(C.byref(shape), w.data_ptr(), 0, x.data_ptr(), y.data_ptr(), base, nbytes,
                                   torch.cuda.current_stream().cuda_stream))


for _ in range(10):
    run()
torch.cuda.synchronize()
os.environ['NODE_TUNE_W4_STAMPS'] = hex(stamps.data_ptr())
names = ['wave started', 'ring requested', 'first ring turn done', 'K loop done', 'own stores issued', 'shared comp multiplied',
         'all stores issued', 'stores drained']
acc = torch.zeros(8, 5, dtype=torch.float64)
clk = torch.zeros(3, dtype=torch.float64)
REP = 10
for _ in range(REP):
    run()
    torch.cuda.synchronize()
    raw = stamps.cpu().double()
    s = raw[:, :8]
    cyc = raw[:, 8:]
    mhz = (cyc[:, 3] - cyc[:, 1]) / ((s[:, 3] - s[:, 1]) / 100.0)     # shader-clock ticks per us over the K loop
    clk += torch.tensor([mhz.min(), mhz.median(), mhz.max()], dtype=torch.float64)
    s = (s - s[:, 0].min()) / 100.0          # us since the first wave started
    for k in range(8):
        v = s[:, k]
        acc[k] += torch.tensor([v.min(), v.quantile(0.5), v.mean(), v.quantile(0.95), v.max()], dtype=torch.float64)
acc /= REP
clk /= REP
print('component GEMM at %s, %d waves, mean of %d launches; us since the first wave started' % ((N, Cc, side), grid * 4, REP))
print('%-26s %8s %8s %8s %8s %8s' % ('stamp', 'min', 'median', 'mean', 'p95', 'max'))
for k in range(8):
    print('%-26s %8.2f %8.2f %8.2f %8.2f %8.2f' % ((names[k],) + tuple(acc[k].tolist())))
print('clock64() ticks per us over the K loop (min / median / max over waves): %.0f / %.0f / %.0f' % tuple(clk.tolist()))
