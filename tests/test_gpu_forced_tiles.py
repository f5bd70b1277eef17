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
