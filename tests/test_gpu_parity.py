"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import pytest
import torch

from tests.helpers import make_func, rel_err

pytestmark = pytest.mark.gpu

SHAPES = [  # (N, C, H, W)
    (2, 8, 7, 7),       # 1 channel / group, odd image, C < one K chunk
    (3, 16, 5, 6),      # non-square
    (2, 64, 8, 8),      # MNIST-like width, 2 ch / group
    (5, 64, 7, 7),      # MNIST config state (49 px: 2 samples per 128-row tile, ragged last tile)
    (4, 96, 4, 4),      # 3 ch / group: N tile of 63 columns
    (3, 256, 8, 8),     # CIFAR width
    (2, 32, 16, 16),    # 256-row tiles (one sample per tile); Winograd conv MT=4, Winograd wgrad <16, 2>
    (2, 32, 6, 6),      # even width, odd number of column pairs: Winograd conv, direct wgrad
    (3, 64, 4, 4),      # Winograd wgrad <4, 4>, four samples per 64-pixel tile (ragged: 3)
    (2, 32, 12, 12),    # HW = 144: 256-pixel tile with one sample; wgrad falls back to the generic kernel
    (2, 16, 10, 14),    # non-square, even width 14 (7 column pairs)
    (130, 64, 8, 8),    # 130 tiles of 64 pixels, Winograd wgrad <8, 8>, split-K over 65 units per slab
]
# Shapes with >= 10^5 elements use the kink-free parameter set (tests/helpers.py:make_func): with ordinary
# parameters one pre-activation of this very input lands within fp32 rounding of a ReLU kink and the oracle and
# the GPU disagree on ONE element of vjp_y by 5 % (measured; 0 elements with kink-free parameters).
KINK_FREE = {(130, 64, 8, 8)}


@pytest.mark.parametrize('shape', SHAPES)
def test_odefunc_forward_matches_oracle(shape):
    import neural_ode_features_amd as nof
    N, C, H, W = shape
    f, twin = make_func(C, seed=C + H, device='cuda', kink_free=shape in KINK_FREE)
    gen = torch.Generator().manual_seed(1)
    y = torch.randn(N, C, H, W, generator=gen)
    t = 0.37
    got = nof.odefunc_forward(f, t, y.cuda())
    with torch.no_grad():
        want = twin(torch.tensor(t), y)
    err = rel_err(got, want)
    print('fwd', shape, err)
    assert err < 2e-5, err


@pytest.mark.parametrize('shape', SHAPES)
def test_odefunc_vjp_matches_oracle(shape):
    import neural_ode_features_amd as nof
    from oracle.dynamics import odefunc_vjp as oracle_vjp
    N, C, H, W = shape
    f, twin = make_func(C, seed=C + H, device='cuda', kink_free=shape in KINK_FREE)
    gen = torch.Generator().manual_seed(2)
    y = torch.randn(N, C, H, W, generator=gen)
    cot = torch.randn(N, C, H, W, generator=gen)
    t = -0.61
    fo, vy, vt, vp = nof.odefunc_vjp(f, t, y.cuda(), cot.cuda())
    p = dict(twin.named_parameters())
    f_ref, vy_ref, vt_ref, vp_ref = oracle_vjp(t, y, p, cot)
    errs = dict(f=rel_err(fo, f_ref), vy=rel_err(vy, vy_ref), vp=rel_err(vp, vp_ref),
                vt=abs(float(vt) - float(vt_ref)) / (abs(float(vt_ref)) + 1e-6))
    print('vjp', shape, errs)
    assert errs['f'] < 2e-5 and errs['vy'] < 5e-5 and errs['vp'] < 5e-5 and errs['vt'] < 1e-4, errs


def test_vjp_t_is_deterministic_over_repeated_launches():
    """d f / d t comes out of k_theta_finalize's last-arrival reduction (fence-free hand-off, see the kernel): 200
    back-to-back launches at the cfg-2 state must all give the SAME bits, and the arrival counter must be back at
    zero each time (a stale partial or a lost count would show as a different sum sooner or later)."""
    import neural_ode_features_amd as nof
    f, _ = make_func(256, seed=3, device='cuda')
    gen = torch.Generator().manual_seed(12)
    y = torch.randn(128, 256, 8, 8, generator=gen).cuda()
    cot = torch.randn(128, 256, 8, 8, generator=gen).cuda()
    _, vy0, vt0, vp0 = nof.odefunc_vjp(f, 0.3, y, cot)
    vt0 = float(vt0)
    for _ in range(200):
        _, vy, vt, vp = nof.odefunc_vjp(f, 0.3, y, cot)
        assert float(vt) == vt0
    assert torch.equal(vp, vp0) and torch.equal(vy, vy0)
