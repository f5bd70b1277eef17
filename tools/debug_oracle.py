"""Adjoint solve: HIP, with the Winograd (1) and the direct (0) conv kernel, vs the CPU oracle over several seeds.
Debug aid for the ReLU-kink sensitivity of solve-level gradients described in DESIGN.md section 2."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof
from neural_ode_features_amd import _lib
from oracle import torchdiffeq_restated as tdq
from tests.helpers import make_func, rel_err

lib = _lib.load()
gv = ctypes.c_int.in_dll(lib, '_ZN4node11g_conv_winoE')
N, C, H, W = 2, 256, 8, 8
tol = 1e-3
for seed in (21, 31, 41, 51, 61, 71):
    line = 'seed %d:' % seed
    for v in (0, 1):
        gv.value = v
        f, twin = make_func(C, seed=seed, device='cuda')
        gen = torch.Generator().manual_seed(seed + 1)
        y = torch.randn(N, C, H, W, generator=gen)
        wgt = torch.randn(2, N, C, H, W, generator=gen) / (N * C * H * W) ** 0.5
        t = torch.tensor([0.0, 1.0])
        if v == 0:
            yo = y.clone().requires_grad_(True)
            out_o = tdq.odeint_adjoint(twin, yo, t, rtol=tol, atol=tol, method='dopri5')
            (out_o * wgt).sum().backward()
            gy_o = yo.grad
        yh = y.cuda().requires_grad_(True)
        out = nof.odeint_adjoint(f, yh, t.cuda(), rtol=tol, atol=tol, method='dopri5')
        (out * wgt.cuda()).sum().backward()
        d = (yh.grad.cpu() - gy_o).abs()
        nbad = int((d > 1e-3 * gy_o.abs().max()).sum())
        line += '  v%d gy rel %.2e (%d elems > 1e-3)' % (v, rel_err(yh.grad, gy_o), nbad)
    print(line, flush=True)
