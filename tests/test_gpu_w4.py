"""GPU parity of the Winograd F(4x4,3x3) pipeline (csrc/wino4.h): the convolution alone against an fp64 convolution,
then whole ODEfunc evaluations / VJPs / solves with the pipeline forced on against the same calls with it off
(NODE_TUNE_WINO4 is read per call) and against the oracle.  Bounds: the transform's own rounding is 3.2e-6 of max|y|
per convolution (tools/wino_error.py); a solve must stay within 10 x atol of the oracle (BASELINE.json north_star)."""
import contextlib
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import make_func, rel_err, oracle_vjp

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def wino4(mode):
    old = os.environ.get('NODE_TUNE_WINO4')
    os.environ['NODE_TUNE_WINO4'] = str(mode)
    try:
        yield
    finally:
        if old is None:
            del os.environ['NODE_TUNE_WINO4']
        else:
            os.environ['NODE_TUNE_WINO4'] = old


def _conv_w4(x, w, dgrad):
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    N, Cc, H, W = x.shape
    shape = _lib.NodeShape(N, Cc, H, W, min(32, Cc), 1e-5)
    nbytes = lib.node_conv3x3_w4_workspace_bytes(C.byref(shape))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
    base = (ws.data_ptr() + 255) & ~255
    y = torch.empty_like(x)
    rc = lib.node_conv3x3_w4(C.byref(shape), w.data_ptr(), int(dgrad), x.data_ptr(), y.data_ptr(), base, nbytes,
                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc)
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize('switch', ['NODE_TUNE_W4_SHAREV=0', 'NODE_TUNE_W4_SHAREV=2'])
@pytest.mark.parametrize('shape', [(128, 256, 8, 8), (8, 256, 16, 16)])
def test_w4_gemm_work_assignments_are_bit_identical(shape, switch):
    """k_w4_gemm64b's alternative assignments of a component's tiles to waves (NODE_TUNE_W4_SHAREV 0 / 1 / 2) multiply the same
    operands in the same order per output element: the convolution is bit-identical.  (The measured-and-rejected variants --
    NODE_TUNE_W4_EARLY, the LDS-DMA ring k_w4_gemm64l, the K-halves kernel k_w4_gemm64k -- live in libnode_hip_diag.so and are
    swept by tests/test_diag_w4.py under `-m diag`.)"""
    N, Cc, H, W = shape
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cc, H, W, generator=gen).cuda()
    w = ((torch.rand(Cc, Cc + 1, 3, 3, generator=gen) * 2 - 1) / (9 * Cc) ** 0.5).cuda()
    want = _conv_w4(x, w, 0)
    key, val = switch.split('=')
    os.environ[key] = val
    try:
        got = _conv_w4(x, w, 0)
    finally:
        del os.environ[key]
    assert torch.equal(got, want)


@pytest.mark.parametrize('shape', [(8, 64, 8, 8), (16, 128, 8, 8), (128, 256, 8, 8),
                                   (2, 128, 16, 16), (8, 128, 16, 16), (4, 256, 16, 16), (2, 1024, 16, 16),
                                   (16, 128, 16, 16), (8, 1024, 16, 16)])
@pytest.mark.parametrize('dgrad', [0, 1])
def test_w4_convolution_matches_fp64(shape, dgrad):
    N, Cc, H, W = shape
    gen = torch.Generator().manual_seed(3 + dgrad)
    x = torch.randn(N, Cc, H, W, generator=gen).relu().cuda()
    w = ((torch.rand(Cc, Cc + 1, 3, 3, generator=gen) * 2 - 1) / (9 * Cc) ** 0.5).cuda()
    got = _conv_w4(x, w, dgrad)
    wd = w[:, 1:].double()
    if dgrad:
        ref = F.conv_transpose2d(x.double(), wd, padding=1)
    else:
        ref = F.conv2d(x.double(), wd, padding=1)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    # the same products with the component GEMMs on the fp32 matrix instructions (k_w4_gemm64) instead of the exact
    # bf16 triples (k_w4_gemm64b): both must sit at the transform's own rounding
    os.environ['NODE_TUNE_W4_BF16X3'] = '0'
    try:
        got32 = _conv_w4(x, w, dgrad)
    finally:
        del os.environ['NODE_TUNE_W4_BF16X3']
    err32 = float((got32.double() - ref).abs().max() / ref.abs().max())
    # ... and on the LDS-tiled kernel of the long reductions (k_w4_gemm128b; by itself from C = 512), where the shape fits it:
    # the same six part products per element in the same order (only the four components whose reduction k_w4_gemm64b cuts
    # into K slices are summed in another order): the two kernels may differ by the rounding of those sums alone
    if (N * (4 if H == 16 else 1)) % 32 == 0 and Cc % 128 == 0 and ((N * (4 if H == 16 else 1) // 32) * (Cc // 128)) % 2 == 0:
        outs = {}
        for mode in ('0', '1'):
            os.environ['NODE_TUNE_W4_GEMM128'] = mode
            try:
                outs[mode] = _conv_w4(x, w, dgrad)
            finally:
                del os.environ['NODE_TUNE_W4_GEMM128']
        e128 = float((outs['1'].double() - ref).abs().max() / ref.abs().max())
        d128 = float((outs['0'] - outs['1']).abs().max() / ref.abs().max())
        print('  k_w4_gemm128b: max err / max|y| %.2e, against k_w4_gemm64b %.2e' % (e128, d128))
        assert e128 < 2e-5 and d128 < 1e-5, (e128, d128)
        assert torch.equal(outs['1' if Cc >= 512 else '0'], got)     # which of the two the library picks by itself
    print('F(4x4,3x3) conv', shape, 'dgrad' if dgrad else 'fwd', 'max err / max|y|: bf16 triples %.2e, fp32 MFMA %.2e, between them %.2e'
          % (err, err32, float((got - got32).abs().max() / ref.abs().max())))
    assert err < 2e-5 and err32 < 2e-5, (err, err32)
    assert err < 1.5 * err32 + 1e-6, (err, err32)


def _engaged(N, Cc, side=8):
    """The workspace grows by the pipeline's buffers exactly when the geometry takes it."""
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    shape = _lib.NodeShape(N, Cc, side, side, min(32, Cc), 1e-5)
    with wino4(0):
        a = lib.node_workspace_bytes(C.byref(shape), 0, 1, 2)
    with wino4(2):
        b = lib.node_workspace_bytes(C.byref(shape), 0, 1, 2)
    return b > a


@pytest.mark.parametrize('shape', [(8, 64, 8, 8), (128, 256, 8, 8)])
def test_w4_odefunc_forward_and_vjp_match_f2(shape):
    from neural_ode_features_amd import integrate
    N, Cc, H, W = shape
    assert _engaged(N, Cc)
    f, _ = make_func(Cc, seed=5, device='cuda')
    gen = torch.Generator().manual_seed(11)
    y = torch.randn(N, Cc, H, W, generator=gen).cuda()
    cot = torch.randn(N, Cc, H, W, generator=gen).cuda()
    with wino4(0):
        ref = integrate.odefunc_vjp(f, 0.3, y, cot)
        ref_f = integrate.odefunc_forward(f, 0.3, y)
    with wino4(2):
        got = integrate.odefunc_vjp(f, 0.3, y, cot)
        got_f = integrate.odefunc_forward(f, 0.3, y)
    assert rel_err(got_f, ref_f) < 5e-5
    # ORDINARY parameters: a pre-activation within rounding of zero gets a different ReLU mask on the two conv paths and
    # moves that SAMPLE's cotangent by O(1 %) (tests/helpers.py: make_func) -- seen with one sample of 128 at the
    # configs[1] shape.  So: every sample but at most two agrees to the transforms' rounding, and the parameter gradient
    # (a sum over samples) in relative L2.  The parity CLAIM of the pipeline is not this self-comparison but the
    # kink-free max-norm comparison with the oracle and the reference's own fixtures below.
    f_g, vy_g, vt_g, vp_g = got
    f_r, vy_r, vt_r, vp_r = ref
    assert rel_err(f_g, f_r) < 5e-5
    per_sample = (vy_g - vy_r).abs().flatten(1).amax(dim=1) / float(vy_r.abs().max())
    assert int((per_sample > 5e-5).sum()) <= 2, per_sample.topk(4)
    l2 = float((vp_g.double() - vp_r.double()).norm() / vp_r.double().norm())
    assert l2 < 2e-3, l2
    assert abs(float(vt_g) - float(vt_r)) < 2e-2 * max(1.0, abs(float(vt_r)))


@pytest.mark.parametrize('name', ['odefunc_c64_n8.pt', 'odefunc_c64_n8_kf.pt'])
def test_w4_pipeline_matches_reference_odefunc_fixture(golden_dir, name):
    """The F(4x4,3x3) pipeline (forced on for a single evaluation) against the REFERENCE's own ODEfunc outputs and
    autograd VJPs (tests/golden/make_golden.py, N = 8: the smallest batch the pipeline takes).  _kf: GroupNorm biases
    in front of the ReLUs at +8, no mask can flip, so every VJP is held in max-norm; the ordinary fixture holds f in
    max-norm and the VJPs in relative L2 (a flipped mask moves single entries by O(1), tests/helpers.py)."""
    import neural_ode_features_amd as nof
    g = torch.load(os.path.join(golden_dir, name), map_location='cpu', weights_only=False)
    assert _engaged(g['N'], g['C'])
    f = nof.ODEfunc(g['C'])
    f.load_state_dict(g['state_dict'])
    f = f.cuda()
    with wino4(2):
        fo, vy, vt, vp = nof.odefunc_vjp(f, float(g['t']), g['y'].cuda(), g['cotangent'].cuda())
        f_only = nof.odefunc_forward(f, float(g['t']), g['y'].cuda())
    # two F(4x4,3x3) convolutions at 3.2e-6 of max|y| each (tools/wino_error.py), GroupNorm in between
    assert rel_err(fo, g['f']) < 3e-5 and rel_err(f_only, g['f']) < 3e-5
    kf = name.endswith('_kf.pt')
    for label, got, ref in (('vjp_y', vy, g['vjp_y']), ('vjp_params', vp, g['vjp_params'])):
        l2 = float((got.cpu().double() - ref.double()).norm() / ref.double().norm())
        assert l2 < 5e-5, (label, l2)
        if kf:
            assert rel_err(got, ref) < 1e-4, (label, rel_err(got, ref))
    scale = max(1.0, abs(float(g['vjp_t'])))
    assert abs(float(vt) - float(g['vjp_t'])) < (1e-4 if kf else 2e-3) * scale


@pytest.mark.parametrize('shape', [(8, 64, 8, 8), (8, 128, 8, 8), (16, 128, 8, 8), (128, 256, 8, 8),
                                   (1, 64, 8, 8), (3, 128, 8, 8), (12, 256, 8, 8), (1, 256, 8, 8),
                                   (2, 128, 16, 16), (1, 128, 16, 16), (8, 128, 16, 16), (4, 256, 16, 16), (3, 256, 16, 16),
                                   (2, 512, 16, 16), (2, 1024, 16, 16)])
def test_w4_odefunc_forward_and_vjp_match_oracle(shape):
    """The pipeline's single evaluation and VJP against the CPU oracle (oracle/dynamics.py, pinned by the reference's
    fixtures), kink-free parameters, at the smallest and at the configs[1] shape: max-norm bounds.  C = 64 takes the
    F(2x2,3x3)-domain weight gradient behind the pipeline, C % 128 == 0 the F(4x4,3x3)-domain one (k_w4_wgrad; N = 8:
    its short operand ring).  Batches that are no multiple of 8 (the bs = 1 census, evaluate.py:97-142) run the component
    GEMMs on padding rows nobody reads; the weight gradient sees zero rows there.
    16x16 states (the one-shot stem's [n, 256, 16, 16], cfg 5's [n, 1024, 16, 16]) run as four 8x8 quadrants per image: 4, 8,
    16 and 32 channels per GroupNorm group = 4 (cpg <= 16) or 8 (cpg = 32) quadrant waves per workgroup, sums and the
    input transform's pixel ring through LDS (kernels_w4s.hip); odd batches pad the virtual-sample count there too."""
    from neural_ode_features_amd import integrate
    N, Cc, H, W = shape
    assert _engaged(N, Cc, H)
    f, twin = make_func(Cc, seed=7, device='cuda', kink_free=True)
    gen = torch.Generator().manual_seed(13)
    y = torch.randn(N, Cc, H, W, generator=gen)
    cot = torch.randn(N, Cc, H, W, generator=gen)
    ref_f, ref_vy, ref_vt, ref_vp = oracle_vjp(torch.tensor(0.3), y, dict(twin.named_parameters()), cot)
    with wino4(2):
        fo, vy, vt, vp = integrate.odefunc_vjp(f, 0.3, y.cuda(), cot.cuda())
    errs = dict(f=rel_err(fo, ref_f), vjp_y=rel_err(vy, ref_vy), vjp_params=rel_err(vp, ref_vp),
                vjp_t=abs(float(vt) - float(ref_vt)) / max(1.0, abs(float(ref_vt))))
    print('F(4x4,3x3) vs oracle at', shape, {k: '%.2e' % v for k, v in errs.items()})
    assert errs['f'] < 3e-5 and errs['vjp_y'] < 1e-4 and errs['vjp_params'] < 1e-4 and errs['vjp_t'] < 1e-4, errs


@pytest.mark.parametrize('shape', [(128, 256, 8, 8), (12, 256, 8, 8), (4, 256, 16, 16), (3, 256, 16, 16)])
def test_w4_wgrad128_matches_oracle(shape):
    """k_w4_wgrad128b (the weight gradient on bf16 triples, LDS-tiled; by itself from C = 512 -- the C = 512 / 1024 shapes of the
    test above run it) forced on at C = 256, against the oracle and against k_w4_wgrad on the same inputs."""
    from neural_ode_features_amd import integrate
    N, Cc, H, W = shape
    f, twin = make_func(Cc, seed=7, device='cuda', kink_free=True)
    gen = torch.Generator().manual_seed(13)
    y = torch.randn(N, Cc, H, W, generator=gen)
    cot = torch.randn(N, Cc, H, W, generator=gen)
    _, _, _, ref_vp = oracle_vjp(torch.tensor(0.3), y, dict(twin.named_parameters()), cot)
    got = {}
    for mode in ('0', '1'):
        os.environ['NODE_TUNE_W4_WGRAD128'] = mode
        try:
            with wino4(2):
                got[mode] = integrate.odefunc_vjp(f, 0.3, y.cuda(), cot.cuda())[3]
        finally:
            del os.environ['NODE_TUNE_W4_WGRAD128']
    e0, e1 = rel_err(got['0'], ref_vp), rel_err(got['1'], ref_vp)
    print('weight gradient vs oracle at', shape, 'k_w4_wgrad %.2e, k_w4_wgrad128b %.2e, between them %.2e' % (e0, e1, rel_err(got['1'], got['0'])))
    assert e0 < 1e-4 and e1 < 1e-4 and e1 < 2 * e0 + 1e-6


@pytest.mark.parametrize('tol,gain,side', [(1e-3, 1.0, 8), (1e-5, 1.0, 8), (1e-5, 4.0, 8), (1e-5, 12.0, 8), (1e-3, 1.0, 16), (1e-5, 4.0, 16)])
def test_w4_solve_matches_f2_and_oracle_tolerance(tol, gain, side):
    """dopri5 at tol >= 1e-5 takes the F(4x4,3x3) path by itself; with it off the same solve runs on F(2x2,3x3).
    `gain` scales the last GroupNorm's weight, i.e. |f|: the stiffer cases take many steps with rejections among them, so
    that the conv noise meets a step controller that is working (Solver::choose_w4 has the error budget: <= 5 % of tol).
    Both paths are held against each other AND against the free-running ORACLE solve of the same problem (CPU, fp32):
    step counts within one accept / reject decision, output within 10 (atol + rtol |y|), kink-free gradients to 1e-3."""
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    N, Cc = (8, 64) if side == 8 else (2, 128)      # (16x16 states: four 8x8 quadrants per image, kernels_w4s.hip)
    assert _engaged(N, Cc, side)
    f, twin = make_func(Cc, seed=2, device='cuda', kink_free=True)
    with torch.no_grad():
        f.norm3.weight.mul_(gain)
        twin.norm3.weight.mul_(gain)
    gen = torch.Generator().manual_seed(21)
    y0 = torch.randn(N, Cc, side, side, generator=gen).cuda()
    t = torch.tensor([0.0, 1.0]).cuda()
    yo = y0.cpu().clone().requires_grad_(True)
    fs_o, bs_o = tdq.SolverStats(), tdq.SolverStats()
    out_o = tdq.odeint_adjoint(twin, yo, t.cpu(), rtol=tol, atol=tol, method='dopri5', fwd_stats=fs_o, bwd_stats=bs_o)[-1]
    out_o.square().sum().backward()
    gp_o = torch.cat([p.grad.reshape(-1) for p in twin.parameters()])
    nfe_o = (2 + 6 * (fs_o.accepted + fs_o.rejected), 3 + 6 * (bs_o.accepted + bs_o.rejected))
    outs, grads, nfes = [], [], []
    for mode in (0, 1):
        with wino4(mode):
            for p in f.parameters():
                p.grad = None
            y = y0.clone().requires_grad_(True)
            f.nfe = 0
            out = nof.odeint_adjoint(f, y, t, rtol=tol, atol=tol, method='dopri5')[-1]
            nf = f.nfe
            out.square().sum().backward()
            outs.append(out.detach())
            grads.append((y.grad.detach().clone(), torch.cat([p.grad.reshape(-1) for p in f.parameters()])))
            nfes.append((nf, f.nfe - nf))
    scale = float(outs[0].abs().max())
    print('tol %g gain %g side %d: nfe F(2x2) %s F(4x4) %s, max|y| %.2f, out diff %.2e, grad rel %.2e / %.2e'
          % (tol, gain, side, nfes[0], nfes[1], scale, float((outs[0] - outs[1]).abs().max()),
             rel_err(grads[1][0], grads[0][0]), rel_err(grads[1][1], grads[0][1])))
    # the same step sequence, up to one accept/reject decision that sat within the noise (6 evaluations per step)
    assert abs(nfes[0][0] - nfes[1][0]) <= 6 and abs(nfes[0][1] - nfes[1][1]) <= 6, nfes
    assert float((outs[0] - outs[1]).abs().max()) < 10 * tol * (1 + scale)      # north star: 10 x (atol + rtol |y|)
    assert rel_err(outs[1], outs[0]) < 1e-4
    assert rel_err(grads[1][0], grads[0][0]) < 1e-3
    assert rel_err(grads[1][1], grads[0][1]) < 1e-3
    # ... and the pipeline (and F(2x2,3x3)) against the oracle's own free-running solve
    for mode in (0, 1):
        d_out = float((outs[mode].cpu() - out_o.detach()).abs().max())
        same = nfes[mode] == nfe_o
        print('  vs oracle, path %d: nfe %s oracle %s, out diff %.2e, grad rel %.2e / %.2e'
              % (mode, nfes[mode], nfe_o, d_out, rel_err(grads[mode][0], yo.grad), rel_err(grads[mode][1], gp_o)))
        assert abs(nfes[mode][0] - nfe_o[0]) <= 6 and abs(nfes[mode][1] - nfe_o[1]) <= 6, (mode, nfes[mode], nfe_o)
        assert d_out < 10 * tol * (1 + scale)
        if same:        # (a flipped accept / reject decision moves both trajectories by O(tol))
            assert rel_err(grads[mode][0], yo.grad) < 1e-3 and rel_err(grads[mode][1], gp_o) < 1e-3


@pytest.mark.parametrize('shape', [(8, 64, 8, 8), (2, 128, 16, 16), (2, 256, 16, 16)])
def test_w4_dense_output_many_time_points(shape):
    """evaluate.py:56-94: features at many interior time points from ONE solve (dense output of the accepted steps), here
    through the pipeline's own emit kernel (W4S state -> NCHW slices; 16x16: the quadrant blocking) and with a gradient
    entering at every slice.  Against the oracle solver on the CPU and against the F(2x2,3x3) path on the same inputs."""
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    N, Cc, H, W = shape
    assert _engaged(N, Cc, H)
    f, twin = make_func(Cc, seed=4, device='cuda', kink_free=True)
    gen = torch.Generator().manual_seed(31)
    y0 = torch.randn(N, Cc, H, W, generator=gen)
    t = torch.linspace(0.0, 1.0, 5)
    wgt = torch.randn(5, N, Cc, H, W, generator=gen)
    yo = y0.clone().requires_grad_(True)
    out_o = tdq.odeint_adjoint(twin, yo, t, rtol=1e-3, atol=1e-3, method='dopri5')
    (out_o * wgt).sum().backward()
    outs, gys = [], []
    for mode in (0, 1):
        with wino4(mode):
            y = y0.cuda().requires_grad_(True)
            out = nof.odeint_adjoint(f, y, t.cuda(), rtol=1e-3, atol=1e-3, method='dopri5')
            (out * wgt.cuda()).sum().backward()
            outs.append(out.detach().cpu())
            gys.append(y.grad.detach().cpu())
    assert outs[1].shape == (5, N, Cc, H, W)
    assert torch.equal(outs[1][0], y0)
    print(shape, 'dense output: vs oracle %.2e, vs F(2x2) %.2e; grad_y0 vs oracle %.2e'
          % (float((outs[1] - out_o.detach()).abs().max()), float((outs[1] - outs[0]).abs().max()), rel_err(gys[1], yo.grad)))
    assert float((outs[1] - out_o.detach()).abs().max()) <= 1e-2          # 10 x atol (BASELINE.json north_star)
    assert rel_err(outs[1], out_o.detach()) < 2e-4 and rel_err(outs[1], outs[0]) < 2e-4
    assert rel_err(gys[1], yo.grad) < 1e-3 and rel_err(gys[1], gys[0]) < 1e-3


def test_w4_split_is_exact_on_the_device():
    """The component GEMMs run on the bf16 matrix pipe "at fp32 accuracy" only if the in-register split of their row
    operands is EXACT: x = h + m + l with three bf16 values (k_w4_gemm64b / k_w4_gemm128b / k_w4_wgrad128b: w4_split8, three
    v_cvt_pk_bf16_f32 roundings and the two remainders between them).  node_w4_split3 runs that very function on
    the device: normal values over 60 binades, exact bf16 values, zeros, both signs -- the identity must hold bit for bit."""
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(17)
    n = 1 << 20
    mant = torch.rand(n, generator=gen) + 1.0
    expo = torch.randint(-30, 30, (n,), generator=gen).float()
    sign = torch.where(torch.rand(n, generator=gen) < 0.5, -1.0, 1.0)
    x = sign * mant * torch.exp2(expo)
    x[:4096] = x[:4096].bfloat16().float()          # values that ARE bf16: m = l = 0
    x[4096:8192] = 0.0
    x[8192:12288] = (x[8192:12288].bfloat16().float() + x[8192:12288].bfloat16().float() * 2.0 ** -9)   # a tie of the first rounding
    xd = x.cuda()
    out = torch.empty(n * 3, device='cuda')
    _lib.check(lib.node_w4_split3(xd.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    parts = out.view(n, 3).cpu()
    assert torch.equal(parts.bfloat16().float(), parts)                                   # every part is a bf16 value
    total = parts[:, 0].double() + parts[:, 1].double() + parts[:, 2].double()
    bad = (total != x.double()).nonzero()
    assert bad.numel() == 0, (bad[:5], x[bad[:5, 0]], parts[bad[:5, 0]])
    assert torch.equal(parts[:4096, 1:], torch.zeros(4096, 2)) and torch.equal(parts[4096:8192], torch.zeros(4096, 3))
    # each part carries what the one before left: |m| <= 2^-8 |h|, |l| <= 2^-8 |m| (half an ulp of an 8-bit mantissa)
    h, m, l = parts[:, 0].double().abs(), parts[:, 1].double().abs(), parts[:, 2].double().abs()
    assert bool((m <= h * 2.0 ** -8).all()) and bool((l <= m * 2.0 ** -8 + 0.0).all() or (l <= h * 2.0 ** -16).all())
