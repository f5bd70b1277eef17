"""GPU parity of the 2-D Winograd conv kernel on SMALL batches.

The tiling heuristic serves small batches with 64-pixel tiles (1-D Winograd kernel), so the regular parity
shapes reach `k_conv3x3_w2` only at full size.  This test forces 128-pixel tiles (NODE_TUNE_CONV_BM=128, read
once per process, hence the child process) and checks forward and VJP against the oracle on shapes that
exercise ragged last tiles, 4x4 and 4x8 images (8 and 4 samples per tile), two K chunks and six."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from tests.helpers import make_func, rel_err
import neural_ode_features_amd as nof
from oracle.dynamics import odefunc_vjp as oracle_vjp
bad = 0
for shape in [(5, 64, 8, 8), (16, 32, 4, 4), (3, 32, 4, 8), (7, 96, 8, 8), (33, 64, 8, 8)]:
    N, C, H, W = shape
    f, twin = make_func(C, seed=C + H, device='cuda', kink_free=True)
    gen = torch.Generator().manual_seed(5)
    y = torch.randn(N, C, H, W, generator=gen)
    cot = torch.randn(N, C, H, W, generator=gen)
    fo, vy, vt, vp = nof.odefunc_vjp(f, 0.41, y.cuda(), cot.cuda())
    f_ref, vy_ref, vt_ref, vp_ref = oracle_vjp(0.41, y, dict(twin.named_parameters()), cot)
    errs = (rel_err(fo, f_ref), rel_err(vy, vy_ref), rel_err(vp, vp_ref))
    print(shape, errs)
    if not (errs[0] < 2e-5 and errs[1] < 5e-5 and errs[2] < 5e-5):
        bad += 1
sys.exit(bad)
'''


def test_2d_winograd_conv_on_small_batches():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NODE_TUNE_CONV_BM='128', NODE_TUNE_CONV_WINO='2')
    r = subprocess.run([sys.executable, '-c', CHILD, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


GRAD_CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
import torch.nn.functional as F
import neural_ode_features_amd as nof
torch.manual_seed(23)
net = nof.ODENet(3, out=10, n_filters=32, downsample='residual', method='dopri5', tol=1e-3, adjoint=True, t1=1, dropout=0).cuda()
gen = torch.Generator().manual_seed(3)
x = torch.randn(6, 3, 32, 32, generator=gen).cuda()
y = torch.randint(0, 10, (6,), generator=gen).cuda()
loss = F.cross_entropy(net(x), y)
loss.backward()
torch.save({k: v.grad.cpu() for k, v in net.named_parameters()}, sys.argv[2])
'''


def test_skipping_the_zero_weight_stage_derivative_is_bit_identical(tmp_path):
    """dopri5 stage 2 has zero weight in the solution, the error estimate and the dense output, and the
    parameter / time segments never form stage states: leaving their stage-2 derivative uncomputed
    (NODE_TUNE_SKIP_K2_THETA=1, the default) must not change a single bit of any gradient."""
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for skip in ('0', '1'):
        out = str(tmp_path / ('g%s.pt' % skip))
        env = dict(os.environ, NODE_TUNE_SKIP_K2_THETA=skip)
        r = subprocess.run([sys.executable, '-c', GRAD_CHILD, root, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(out))
    # the ODE block's gradients come from this library alone: bit-identical.  The stem's pass through
    # PyTorch-ROCm / MIOpen backward kernels, which are not run-to-run deterministic: compared to rounding.
    for k in outs[0]:
        if 'odeblock' in k:
            assert torch.equal(outs[0][k], outs[1][k]), k
        else:
            assert float((outs[0][k] - outs[1][k]).abs().max()) <= 1e-5 * float(outs[0][k].abs().max()) + 1e-9, k


SMALL_CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from tests.helpers import make_func, rel_err
import neural_ode_features_amd as nof
from oracle.dynamics import odefunc_forward as oracle_f
from oracle import torchdiffeq_restated as tdq
bad = 0
for shape in [(1, 128, 8, 8), (3, 256, 4, 4), (5, 160, 6, 6), (2, 1024, 4, 4), (1, 256, 16, 16), (9, 128, 7, 7), (4, 512, 8, 8),
              (40, 256, 8, 8)]:
    N, C, H, W = shape
    f, twin = make_func(C, seed=C + H, device='cuda')
    gen = torch.Generator().manual_seed(7)
    y = torch.randn(N, C, H, W, generator=gen)
    got = nof.odefunc_forward(f, 0.3, y.cuda())
    want = oracle_f(torch.tensor(0.3), y, dict(twin.named_parameters()))
    e = rel_err(got, want)
    t = torch.tensor([0.0, 1.0])
    with torch.no_grad():
        out = nof.odeint(f, y.cuda(), t.cuda(), rtol=1e-3, atol=1e-3, method='dopri5')
        ref = tdq.odeint(twin, y, t, rtol=1e-3, atol=1e-3, method='dopri5')
    e2 = float((out.cpu() - ref).abs().max())
    print(shape, e, e2)
    if not (e < 2e-5 and e2 <= 1e-2):
        bad += 1
sys.exit(bad)
'''


def test_small_grid_kernel_forced_on_many_shapes():
    """k_conv3x3_small is picked only for grids under eight workgroups; forced on (NODE_TUNE_SMALL=1, read once per
    process, hence the child) it must serve any C % 32 == 0, C >= 128 geometry: ragged last pixel tile (N*HW not a
    multiple of 32), tiles that straddle samples, 16x16 and odd images, 4 to 32 channel groups per wave, and a
    batch large enough for 80 pixel tiles."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NODE_TUNE_SMALL='1')
    r = subprocess.run([sys.executable, '-c', SMALL_CHILD, root], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
