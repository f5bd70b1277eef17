"""GPU: the residual stem on the library's own kernels (csrc/kernels_stem.hip; reference model.py:167-178, 284-310).
Every convolution of the family against an fp64 `F.conv2d` (forward, data gradient, weight gradient, all six
64 / 256-channel geometries of the CIFAR stem incl. 30x30 / 15x15 images, stride 2 and 1x1), then the whole stem --
forward and every parameter gradient -- against the same modules run in fp64 on the CPU."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (cin, cout, image side, kernel, stride, pad): model.py:172-174 at 32x32 input
GEOMS = [(64, 64, 30, 3, 2, 1), (64, 64, 30, 1, 2, 0), (64, 64, 15, 3, 1, 1),
         (64, 256, 15, 3, 2, 1), (64, 256, 15, 1, 2, 0), (256, 256, 8, 3, 1, 1)]


def _one_conv(what, geom, n, x, w, dy):
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    cin, cout, side, k, stride, pad = geom
    g = _lib.NodeConvGeom(n, cin, cout, side, side, k, stride, pad)
    nbytes = lib.node_stem_conv_workspace_bytes(C.byref(g))
    assert nbytes > 0, lib.node_last_error()
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
    yside = (side + 2 * pad - k) // stride + 1
    shape = {0: (n, cout, yside, yside), 1: (n, cin, side, side), 2: (cout, cin, k, k)}[what]
    res = torch.full(shape, float('nan'), dtype=torch.float32, device='cuda')
    ptr = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.node_stem_conv(C.byref(g), what, ptr(x), ptr(w), ptr(dy), res.data_ptr(), (ws.data_ptr() + 255) & ~255, nbytes,
                                  torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return res.cpu()


@pytest.mark.parametrize('geom', GEOMS, ids=lambda g: 'c%d-%d_s%d_k%d_st%d' % g[:5])
@pytest.mark.parametrize('n', [3, 8])
def test_stem_convolutions_match_fp64(geom, n):
    cin, cout, side, k, stride, pad = geom
    gen = torch.Generator().manual_seed(cin + cout + side + k + n)
    x = torch.randn(n, cin, side, side, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5
    yside = (side + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, yside, yside, generator=gen)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, None, stride, pad)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (xd, wd), dy.double())
    xg, wg, dyg = x.cuda(), w.cuda(), dy.cuda()
    for what, ref, args in ((0, y_ref.detach(), (xg, wg, None)), (1, dx_ref, (None, wg, dyg)), (2, dw_ref, (xg, None, dyg))):
        got = _one_conv(what, geom, n, *args).double()
        err = float((got - ref).abs().max() / ref.abs().max())
        print('geom %s n %d what %d: max error / max|ref| = %.2e' % (geom, n, what, err))
        assert err <= 1e-5, (geom, n, what, err)


def _stem_pair(in_ch, filters, seed, kink_free=False):
    import neural_ode_features_amd as nof
    torch.manual_seed(seed)
    net = nof.ODENet(in_ch, out=10, n_filters=filters, downsample='residual', adjoint=True)
    stem = net.downsample.module
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in stem.named_parameters():
            if 'norm' in name:
                p.add_(0.2 * torch.randn(p.shape, generator=gen))
                if kink_free and name.endswith('bias'):
                    p.add_(8.0)          # every pre-activation positive: no ReLU mask can differ between fp32 and fp64
    import copy
    return stem, copy.deepcopy(stem).double()


@pytest.mark.parametrize('case', [(3, 32, 256, 6, False), (1, 28, 64, 5, False), (3, 32, 256, 8, False), (3, 32, 128, 128, True),
                                  (3, 32, 128, 128, False)],
                         ids=lambda c: 'in%d_%dpx_f%d_n%d%s' % (c[:4] + ('_kinkfree' if c[4] else '',)))
@pytest.mark.parametrize('w4', [1, 0], ids=['last_conv_w4', 'last_conv_gather'])
def test_whole_stem_forward_and_every_gradient_match_fp64(case, w4, monkeypatch):
    """Max-norm parity of the output and of all sixteen parameter gradients with the fp64 run.  At the full batch the
    stem evaluates ~12 M pre-activations: a dozen land within fp32 rounding of zero, their ReLU masks differ from the
    fp64 run's, and a single flipped element moves a GroupNorm-bias gradient (a sum of 8192 terms of random sign) by
    ~1 % -- so the full-batch case is asserted in max norm on kink-free parameters (biases in front of the ReLUs at +8) and
    in relative L2 on ordinary ones."""
    in_ch, side, filters, n, kink_free = case
    # the block's last convolution (filters -> filters on 8x8) takes the ODE block's F(4x4,3x3) pipeline where its shape
    # allows (batch % 8 == 0, filters % 128 == 0); NODE_TUNE_STEM_W4 = 0 keeps it on the stem's own gather-GEMM kernels
    takes_w4 = w4 == 1 and side == 32 and n % 8 == 0 and filters % 128 == 0
    if w4 == 0 and not (side == 32 and n % 8 == 0 and filters % 128 == 0):
        pytest.skip('this shape never takes the pipeline: covered by the other parametrization')
    monkeypatch.setenv('NODE_TUNE_STEM_W4', str(w4))
    stem, ref = _stem_pair(in_ch, filters, seed=in_ch + filters, kink_free=kink_free)
    stem = stem.cuda()
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(n, in_ch, side, side, generator=gen)
    out = stem(x.cuda())
    assert type(out.grad_fn).__name__ == '_StemFnBackward'          # the fused node, not the module sequence
    out_ref = ref(x.double())
    cot = torch.randn(out_ref.shape, generator=gen)
    out.backward(cot.cuda())
    out_ref.backward(cot.double())
    err = float((out.detach().cpu().double() - out_ref.detach()).abs().max() / out_ref.detach().abs().max())
    print('stem %s (w4 %s): output max error %.2e' % (case, takes_w4, err))
    assert err <= 2e-5
    errs, l2 = {}, {}
    for (name, p), (_, q) in zip(stem.named_parameters(), ref.named_parameters()):
        errs[name] = float((p.grad.cpu().double() - q.grad).abs().max() / q.grad.abs().max())
        l2[name] = float((p.grad.cpu().double() - q.grad).norm() / q.grad.norm())
        print('  grad %-28s max error / max|ref| = %.2e, relative L2 %.2e' % (name, errs[name], l2[name]))
    print('stem %s: worst gradient error %.2e (max norm), %.2e (L2)' % (case, max(errs.values()), max(l2.values())))
    if n >= 64 and not kink_free:
        assert max(l2.values()) <= 2e-2, l2
    else:
        assert max(errs.values()) <= (5e-5 if takes_w4 else 2e-5), errs


def test_stem_matches_the_reference_fixture(golden_dir):
    """tests/golden/stem_residual_c64.pt: the reference's own `ResDownsample(1, 64)` (model.py:167-178, imported by
    tests/golden/make_golden.py) on a [2, 1, 28, 28] input -- output and every parameter gradient.  The reference's
    state_dict loads unchanged (same keys); the library's stem kernels must reproduce its numbers."""
    import os
    import neural_ode_features_amd as nof
    g = torch.load(os.path.join(golden_dir, 'stem_residual_c64.pt'), map_location='cpu', weights_only=False)
    net = nof.ODENet(1, out=10, n_filters=64, downsample='residual', adjoint=True)
    stem = net.downsample           # (the wrapper holds the body under `.module`, like the reference's ResDownsample)
    stem.load_state_dict(g['state_dict'])
    stem = stem.cuda()
    out = stem(g['x'].cuda())
    assert type(out.grad_fn).__name__ == '_StemFnBackward'
    out.backward(g['cot'].cuda())
    err = float((out.detach().cpu() - g['out']).abs().max() / g['out'].abs().max())
    print('reference stem fixture: output max error / max|ref| = %.2e' % err)
    assert err <= 2e-5
    for name, p in stem.named_parameters():
        ref = g['grads'][name]
        e = float((p.grad.cpu() - ref).abs().max() / ref.abs().max())
        print('  grad %-28s %.2e' % (name, e))
        assert e <= 1e-4, (name, e)       # (the fixture itself is fp32: PyTorch-CPU's own rounding sits at ~1e-6)


def test_stem_runs_no_library_convolution():
    """The fused stem is ONE autograd node whose forward and backward are calls into libnode_hip.so: PyTorch dispatches no
    convolution, no GroupNorm and no layout transpose for it (its dispatcher trace of a forward + backward holds no
    operator at all besides the custom function)."""
    from torch.profiler import ProfilerActivity, profile
    stem, _ = _stem_pair(3, 256, seed=7)
    stem = stem.cuda()
    x = torch.randn(8, 3, 32, 32).cuda()
    stem(x).sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = stem(x)
        out.backward(torch.ones_like(out))
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    bad = [k for k in names if any(s in k.lower() for s in ('conv', 'miopen', 'group_norm', 'native_group_norm', 'transpose', 'relu'))]
    assert not bad, bad
    assert any('_StemFn' in k for k in names), names


def test_stem_filter_counts_the_kernels_refuse_run_the_module_sequence():
    """filters = 192 (cpg 6): the GroupNorm passes keep whole groups inside power-of-two channel blocks, so the library
    refuses the shape (check_stem_shape) and the module runs its PyTorch sequence -- output and gradients still match fp64.
    (Round-4 advisor finding: the fused path took such shapes and left channels 128..191 unwritten.)"""
    import ctypes as C_
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    for filters in (192, 384):
        shape = _lib.NodeStemShape(4, 3, 32, 32, filters, 1e-5)
        assert lib.node_stem_workspace_bytes(C_.byref(shape)) == 0
    stem, ref = _stem_pair(3, 192, seed=11)
    stem = stem.cuda()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 32, 32, generator=gen)
    out = stem(x.cuda())
    assert type(out.grad_fn).__name__ != '_StemFnBackward'
    out_ref = ref(x.double())
    cot = torch.randn(out_ref.shape, generator=gen)
    out.backward(cot.cuda())
    out_ref.backward(cot.double())
    assert float((out.detach().cpu().double() - out_ref.detach()).abs().max() / out_ref.detach().abs().max()) <= 1e-4
    for (name, p), (_, q) in zip(stem.named_parameters(), ref.named_parameters()):
        assert float((p.grad.cpu().double() - q.grad).norm() / q.grad.norm()) <= 1e-3, name


def test_stem_input_gradient_and_second_backward():
    """An input that requires a gradient (saliency maps, adversarial examples) takes the module sequence -- the fused node
    produces parameter gradients only; a second backward through one fused forward raises instead of crashing."""
    stem, ref = _stem_pair(3, 64, seed=3)
    stem = stem.cuda()
    x = torch.randn(2, 3, 32, 32)
    xg = x.cuda().requires_grad_(True)
    out = stem(xg)
    assert type(out.grad_fn).__name__ != '_StemFnBackward'
    out.sum().backward()
    xd = x.double().requires_grad_(True)
    ref(xd).sum().backward()
    assert float((xg.grad.cpu().double() - xd.grad).abs().max() / xd.grad.abs().max()) <= 1e-3
    out = stem(x.cuda())
    assert type(out.grad_fn).__name__ == '_StemFnBackward'
    out.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='second backward'):
        out.sum().backward()
