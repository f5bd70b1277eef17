"""Run one adjoint solve under conv variant 0 and 1 in one process and compare (debug aid)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof
from neural_ode_features_amd import _lib
from tests.helpers import make_func

lib = _lib.load()
gv = ctypes.c_int.in_dll(lib, '_ZN4node14g_conv_variantE')
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '2,256,8,8').split(','))
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
N, C, H, W = shape
SEED = int(sys.argv[4]) if len(sys.argv) > 4 else 21
res = {}
VARS = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '0,1').split(',')]
for v in VARS:
    gv.value = v
    f, twin = make_func(C, seed=SEED, device='cuda')
    gen = torch.Generator().manual_seed(SEED + 1)
    y = torch.randn(N, C, H, W, generator=gen)
    wgt = torch.randn(2, N, C, H, W, generator=gen) / (N * C * H * W) ** 0.5
    yh = y.cuda().requires_grad_(True)
    out = nof.odeint_adjoint(f, yh, torch.tensor([0.0, 1.0]).cuda(), rtol=tol, atol=tol, method='dopri5',
                             options={'record_dt': 64})
    (out * wgt.cuda()).sum().backward()
    gp = torch.cat([p.grad.reshape(-1) for p in f.parameters()])
    res[v] = dict(out=out.detach().cpu(), gy=yh.grad.cpu(), gp=gp.cpu(), fs=f.last_forward_stats, bs=f.last_backward_stats)
    print('variant', v, 'fwd dts', res[v]['fs'].get('dts'), 'bwd dts', res[v]['bs'].get('dts'))
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
A, B = VARS[-1], VARS[0]
for A in VARS[1:]:
    print('variant', A, 'vs', B, 'out rel', rel(res[A]['out'], res[B]['out']), 'gy rel', rel(res[A]['gy'], res[B]['gy']), 'gp rel', rel(res[A]['gp'], res[B]['gp']))
d = (res[A]['gy'] - res[B]['gy']).abs()
print('gy diff per sample max', d.flatten(1).max(1).values.tolist())
print('gy diff per channel-block max', [float(d[:, i:i + 64].max()) for i in range(0, C, 64)])
print('gy diff per pixel max', d.amax(dim=(0, 1)))
