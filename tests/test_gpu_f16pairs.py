"""GPU parity of the fp16-PAIR operand format of the F(4x4,3x3) pipeline (csrc/wino4.h, round 6): every fp32 operand of the
component GEMMs and of the weight gradient as a power-of-two-scaled pair h + l of fp16 numbers, three MFMA products per fp32
product.  What must hold: (1) the convolution error against fp64 is that of the bf16-triple / fp32 chain whatever the magnitude
of the inputs and filters (the scales are derived from them); (2) whole adjoint solves agree with the bf16-triple pipeline to
rounding, with the SAME step histories and evaluation counts, for cotangents of any magnitude (their scale follows the data);
(3) a step whose cotangent outgrows its scale is repeated on the device and the solve ends where an undisturbed one does."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import make_func, rel_err

pytestmark = pytest.mark.gpu


def _env(**kw):
    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            os.environ.update({k: str(v) for k, v in kw.items()})

        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _E()


def _conv_w4(x, w, dgrad):
    from neural_ode_features_amd import _lib
    lib = _lib.load()
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


@pytest.mark.parametrize('shape', [(128, 256, 8, 8), (32, 512, 8, 8), (8, 1024, 16, 16)])     # k_w4_gemm64h; k_w4_gemm128h at C >= 512
@pytest.mark.parametrize('xs,wsc', [(1.0, 1.0), (1e-4, 1.0), (1e3, 1.0), (1.0, 1e-3), (1.0, 30.0), (3e-7, 1e-2)])
def test_pair_convolution_is_scale_invariant(shape, xs, wsc):
    """fp16 has five exponent bits: the power-of-two scales (k_w4_scales: from max|w| and the input's bound) must make the pair
    format's error independent of the operands' magnitudes -- at every magnitude it is the bf16-triple pipeline's (the
    transform's own rounding), forward and data-gradient filters."""
    N, Cc, H, W = shape
    for dgrad in (0, 1):
        gen = torch.Generator().manual_seed(5 + dgrad)
        x = (torch.randn(N, Cc, H, W, generator=gen).relu() * xs).cuda()
        w = (((torch.rand(Cc, Cc + 1, 3, 3, generator=gen) * 2 - 1) / (9 * Cc) ** 0.5) * wsc).cuda()
        wd = w[:, 1:].double()
        ref = F.conv_transpose2d(x.double(), wd, padding=1) if dgrad else F.conv2d(x.double(), wd, padding=1)
        errs = {}
        for f16 in ('1', '0'):
            with _env(NODE_TUNE_W4_F16=f16):
                got = _conv_w4(x, w, dgrad)
            errs[f16] = float((got.double() - ref).abs().max() / ref.abs().max())
        print(shape, 'x *', xs, 'w *', wsc, 'dgrad' if dgrad else 'fwd', 'pairs %.2e triples %.2e' % (errs['1'], errs['0']))
        assert errs['1'] < 2e-5
        assert errs['1'] <= 1.5 * errs['0'] + 1e-6, errs


@pytest.mark.parametrize('shape', [(16, 512, 16, 16), (32, 1024, 16, 16)])
def test_gemm256h_is_bit_identical_to_gemm128h(shape):
    """k_w4_gemm256h (256 x 256 tiles, one wave per SIMD with sixteen accumulators; components 0..31, components 32..35 behind it as
    k_w4_gemm128h's tail launch) forms every 32 x 32 output tile from the same products in the same order as k_w4_gemm128h; only
    components 32..35 differ in rounding (whole reductions here, four partial ones summed through LDS there).  Against fp64 and against
    NODE_TUNE_W4_H256 = 0."""
    N, Cc, H, W = shape
    for dgrad in (0, 1):
        gen = torch.Generator().manual_seed(11 + dgrad)
        x = torch.randn(N, Cc, H, W, generator=gen).relu().cuda()
        w = ((torch.rand(Cc, Cc + 1, 3, 3, generator=gen) * 2 - 1) / (9 * Cc) ** 0.5).cuda()
        wd = w[:, 1:].double()
        ref = F.conv_transpose2d(x.double(), wd, padding=1) if dgrad else F.conv2d(x.double(), wd, padding=1)
        with _env(NODE_TUNE_W4_H256='2'):      # (2: wherever the geometry has 256 x 256 tiles, not only where they fill whole rounds of the chip)
            big = _conv_w4(x, w, dgrad)
        with _env(NODE_TUNE_W4_H256='0'):
            small = _conv_w4(x, w, dgrad)
        err = float((big.double() - ref).abs().max() / ref.abs().max())
        print(shape, 'dgrad' if dgrad else 'fwd', 'error vs fp64 %.2e' % err)
        assert err < 2e-5
        diff = float((big - small).abs().max() / ref.abs().max())
        assert diff < 1e-5, diff


def _adjoint(shape, tol, seed, gscale, env):
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import _lib
    N, C_, H, W = shape
    f, _ = make_func(C_, seed=seed, device='cuda', kink_free=True)
    gen = torch.Generator().manual_seed(seed + 1)
    y = torch.randn(N, C_, H, W, generator=gen)
    wgt = torch.randn(2, N, C_, H, W, generator=gen) / (C_ * H * W) ** 0.5 * gscale
    t = torch.tensor([0.0, 1.0]).cuda()
    with _env(**env):
        yh = y.cuda().requires_grad_(True)
        out = nof.odeint_adjoint(f, yh, t, rtol=tol, atol=tol, method='dopri5')
        (out * wgt.cuda()).sum().backward()
        st = (C.c_int32 * 4)()
        _lib.check(_lib.load().node_w4_pair_stats(st))
    gp = torch.cat([p.grad.reshape(-1) for p in f.parameters()])
    return dict(out=out.detach(), gy=yh.grad.clone(), gp=gp.clone(), fwd=dict(f.last_forward_stats), bwd=dict(f.last_backward_stats),
                pair=list(st))


@pytest.mark.parametrize('shape,tol', [((32, 128, 8, 8), 1e-3), ((128, 256, 8, 8), 1e-3), ((128, 256, 8, 8), 1e-5), ((8, 128, 16, 16), 1e-3)])
@pytest.mark.parametrize('gscale', [1.0, 1e-6, 1e4])
def test_pair_and_triple_adjoint_solves_agree(shape, tol, gscale):
    """The whole adjoint solve (forward recompute, data gradients, k_w4_wgrad64h) on fp16 pairs against the same solve on bf16
    triples / fp32 MFMA: identical accept / reject histories and evaluation counts, outputs and gradients equal to rounding (kink-free
    parameters: no ReLU mask can flip), for cotangents six orders of magnitude below and four above the usual ones -- the
    cotangent-side scale follows the data."""
    if gscale != 1.0 and (shape[0] != 32 or tol != 1e-3):
        pytest.skip('cotangent magnitudes are swept at one shape')
    a = _adjoint(shape, tol, 51, gscale, dict(NODE_TUNE_W4_F16='1', NODE_TUNE_W4_STATS='1'))
    b = _adjoint(shape, tol, 51, gscale, dict(NODE_TUNE_W4_F16='0'))
    assert a['pair'][0] == 1 and a['pair'][1] == 0, a['pair']          # pairs were used; no step had to be repeated
    for k in ('accepted', 'rejected', 'nfe'):
        assert a['fwd'][k] == b['fwd'][k] and a['bwd'][k] == b['bwd'][k], (k, a['fwd'], b['fwd'], a['bwd'], b['bwd'])
    e = (rel_err(a['out'], b['out']), rel_err(a['gy'], b['gy']), rel_err(a['gp'], b['gp']))
    print(shape, tol, gscale, 'pairs vs triples: out %.2e grad_y %.2e grad_theta %.2e; cotangent exponent %d' % (e + (a['pair'][2],)))
    # (measured: 2e-6, 1.5e-5, 3.6e-5.  At 1e4 times the usual cotangent the first step's error estimate is rounding noise and the
    #  SECOND step size already differs between any two conv paths -- 0.078 / 0.083 / 0.092 for triples / pairs / F(2x2,3x3) -- which
    #  moves every gradient by a few 1e-4, the pairs no further from either than those two from each other: tools/f16_adjoint_ab.py)
    bound = 2e-3 if gscale > 1.0 else 2e-4
    assert e[0] < 2e-5 and e[1] < bound and e[2] < bound, e
    assert torch.isfinite(a['gy']).all() and torch.isfinite(a['gp']).all()


def test_overflowing_step_is_repeated_not_lost():
    """NODE_TUNE_W4_GSKEW=9 starts the interval with the cotangent exponent nine too high: the first step's passes meet values 2^9
    beyond the target, raise the overflow flag, and the device controller repeats that step at the exponent the recorded maximum
    asks for.  The repeated step is not a solver step: accepted / rejected / NFE are those of the undisturbed solve, and so is
    the result (to the rounding of a different power of two)."""
    shape, tol = (32, 128, 8, 8), 1e-3
    a = _adjoint(shape, tol, 51, 1.0, dict(NODE_TUNE_W4_F16='1', NODE_TUNE_W4_STATS='1', NODE_TUNE_W4_GSKEW='9'))
    b = _adjoint(shape, tol, 51, 1.0, dict(NODE_TUNE_W4_F16='1', NODE_TUNE_W4_STATS='1'))
    print('skewed:', a['pair'], a['bwd'], '| undisturbed:', b['pair'], b['bwd'])
    assert a['pair'][0] == 1 and a['pair'][1] >= 1 and b['pair'][1] == 0
    for k in ('accepted', 'rejected', 'nfe'):
        assert a['bwd'][k] == b['bwd'][k], (k, a['bwd'], b['bwd'])
    assert torch.isfinite(a['gy']).all() and torch.isfinite(a['gp']).all()
    assert rel_err(a['gy'], b['gy']) < 1e-5 and rel_err(a['gp'], b['gp']) < 1e-5
