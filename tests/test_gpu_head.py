"""GPU parity of the fused classifier head (node_head_fwd / node_head_bwd) against the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

# vector kernels: 8x8 (16 lanes per channel), 4x4 (4), 16x16 (64), 2x2 (1); per-channel fallback: 7x7, 5x6, C = 1024 at 8x8
SHAPES = [(128, 256, 8, 8), (5, 64, 7, 7), (3, 96, 4, 4), (2, 8, 5, 6), (2, 1024, 4, 4), (7, 16, 2, 2), (3, 32, 16, 16),
          (2, 1024, 8, 8), (4, 64, 16, 16)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('with_scale', [False, True])
def test_head_pool_forward_backward_match_oracle(shape, with_scale):
    from neural_ode_features_amd.head import _HeadPool
    from oracle.head import head_pool as oracle_head
    N, C, H, W = shape
    G = min(32, C)
    gen = torch.Generator().manual_seed(C + H)
    z = torch.randn(N, C, H, W, generator=gen)
    gamma = 1.0 + 0.25 * torch.randn(C, generator=gen)
    beta = 0.3 * torch.randn(C, generator=gen)
    scale = (torch.rand(N, C, generator=gen) > 0.5).float() * 2.0 if with_scale else None
    cot = torch.randn(N, C, generator=gen)

    zr, gr, br = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    want = oracle_head(zr, gr, br, G, 1e-5, scale)
    want.backward(cot)

    zg, gg, bg = (t.cuda().requires_grad_(True) for t in (z, gamma, beta))
    got = _HeadPool.apply(zg, gg, bg, None if scale is None else scale.cuda(), G, 1e-5)
    got.backward(cot.cuda())
    assert rel_err(got, want) < 1e-5
    assert rel_err(zg.grad, zr.grad) < 2e-5
    assert rel_err(gg.grad, gr.grad) < 2e-5 and rel_err(bg.grad, br.grad) < 2e-5


def test_fcclassifier_fused_matches_plain_modules_and_dropout_stream():
    """The module-level switch: same logits / gradients as the plain nn.Sequential on the same device, and the
    dropout mask is the one nn.Dropout draws from the same generator state."""
    import neural_ode_features_amd as nof
    torch.manual_seed(5)
    head = nof.FCClassifier(in_ch=64, out=10, dropout=0.5).cuda().train()
    x = torch.randn(9, 64, 8, 8, device='cuda')
    for train in (True, False):
        head.train(train)
        xa = x.clone().requires_grad_(True)
        xb = x.clone().requires_grad_(True)
        torch.manual_seed(77)
        fused = head(xa)
        torch.manual_seed(77)
        plain = head.module(xb)
        assert rel_err(fused, plain) < 1e-5
        w = torch.randn_like(fused)
        head.zero_grad()
        (fused * w).sum().backward()
        gf = {k: v.grad.clone() for k, v in head.named_parameters()}
        head.zero_grad()
        (plain * w).sum().backward()
        assert rel_err(xa.grad, xb.grad) < 2e-5
        for k, v in head.named_parameters():
            assert rel_err(gf[k], v.grad) < 2e-5, k


def test_head_errors():
    from neural_ode_features_amd._lib import NodeHipError
    from neural_ode_features_amd.head import _HeadPool
    z = torch.randn(2, 12, 4, 4, device='cuda')
    with pytest.raises(NodeHipError):
        _HeadPool.apply(z, torch.ones(12, device='cuda'), torch.zeros(12, device='cuda'), None, 5, 1e-5)   # 5 does not divide 12


GN_SHAPES = [(128, 64, 30, 30), (16, 64, 15, 15), (9, 256, 8, 8), (3, 96, 5, 7), (2, 8, 3, 3), (2, 1024, 4, 4)]


@pytest.mark.parametrize('shape', GN_SHAPES)
@pytest.mark.parametrize('relu', [True, False])
def test_gn_relu_forward_backward_match_torch(shape, relu):
    """The stem's `relu(norm(x))` pairs (model.py:304-307): fused HIP op vs F.group_norm (+ F.relu) on the CPU."""
    from neural_ode_features_amd.head import _GnRelu
    N, C, H, W = shape
    G = min(32, C)
    gen = torch.Generator().manual_seed(C + H)
    z = torch.randn(N, C, H, W, generator=gen) * 1.5 + 0.3
    gamma = 1.0 + 0.25 * torch.randn(C, generator=gen)
    beta = 0.3 * torch.randn(C, generator=gen)
    cot = torch.randn(N, C, H, W, generator=gen)
    zr, gr, br = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    want = F.group_norm(zr, G, gr, br, 1e-5)
    if relu:
        want = F.relu(want)
    want.backward(cot)
    zg, gg, bg = (t.cuda().requires_grad_(True) for t in (z, gamma, beta))
    got = _GnRelu.apply(zg, gg, bg, G, 1e-5, relu)
    got.backward(cot.cuda())
    assert rel_err(got, want) < 1e-5
    assert rel_err(zg.grad, zr.grad) < 2e-5
    assert rel_err(gg.grad, gr.grad) < 2e-5 and rel_err(bg.grad, br.grad) < 2e-5


def test_resblock_fused_matches_plain_modules():
    """The fused GroupNorm(+ReLU) kernels inside a ResBlock against the plain modules, the convolutions on both sides
    being MIOpen's.  MIOpen is PINNED for this test (no algorithm search, deterministic kernels, both paths warmed once):
    on a fresh box its backward kernels for a geometry could differ between the first and the second call (search result
    vs immediate fallback; seen once in six runs in round 2: input gradients 4e-3 apart with bit-identical GroupNorm
    kernels), which had forced a 2e-2 bound that no longer caught a sub-percent defect of the fused backward.  The
    GroupNorm kernels alone are held to 2e-5 by test_gn_relu_forward_backward_match_torch at the stem's own shapes."""
    import neural_ode_features_amd as nof
    from torch import nn
    old = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic)
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    try:
        torch.manual_seed(3)
        blk = nof.ResBlock(64, 256, stride=2, downsample=nn.Conv2d(64, 256, 1, 2, bias=False)).cuda()
        x = torch.randn(8, 64, 15, 15, device='cuda')

        def fused_path(inp):
            return blk(inp)

        def plain_path(inp):
            pre = blk.relu(blk.norm1(inp))
            return blk.conv2(blk.relu(blk.norm2(blk.conv1(pre)))) + blk.downsample(pre)

        w = torch.randn(8, 256, 8, 8, device='cuda')
        for path in (fused_path, plain_path):           # warm both: every convolution geometry has run once each way
            xi = x.clone().requires_grad_(True)
            (path(xi) * w).sum().backward()
        blk.zero_grad()
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        fused, plain = fused_path(xa), plain_path(xb)
        assert rel_err(fused, plain) < 2e-5
        (fused * w).sum().backward()
        gf = {k: v.grad.clone() for k, v in blk.named_parameters()}
        blk.zero_grad()
        (plain * w).sum().backward()
        assert rel_err(xa.grad, xb.grad) < 1e-3
        for k, v in blk.named_parameters():
            assert rel_err(gf[k], v.grad) < 1e-3, k
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = old


def test_stem_and_head_on_gpu_match_the_reference_logits():
    """tests/golden/odenet_t0.pt: logits of the REFERENCE's own ODENet with the ODE block switched off (t1 = 0,
    model.py:363-364), i.e. stem + head only.  On the GPU both run through the fused GroupNorm / head kernels."""
    import os
    import neural_ode_features_amd as nof
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'odenet_t0.pt'))
    net = nof.ODENet(1, out=10, n_filters=8, downsample='residual', t1=0)
    net.load_state_dict(g['state_dict'])
    net = net.cuda().eval()
    with torch.no_grad():
        got = net(g['x'].cuda())
    assert rel_err(got, g['logits']) < 2e-5


# ---------------------------------------------------------------------------------------------------------------
# Linear + cross-entropy on the library's kernels (node_head_loss_fwd / _bwd, csrc/kernels_loss.hip)
# ---------------------------------------------------------------------------------------------------------------
# (n, in_features, classes): cfg 2 / MNIST / cfg 5 heads, a ragged batch, CIFAR-100 and tiny-imagenet class counts (the lane
# map changes at 32 and 64 classes), in_features that the channel groups do not divide
LOSS_SHAPES = [(128, 256, 10), (32, 64, 10), (64, 1024, 10), (7, 64, 10), (1, 256, 10), (33, 96, 100), (16, 256, 200),
               (5, 20, 3), (130, 64, 33), (9, 256, 64), (4, 64, 1000)]


def _loss_case(n, c, o, seed):
    gen = torch.Generator().manual_seed(seed)
    pooled = torch.randn(n, c, generator=gen).relu()
    w = torch.randn(o, c, generator=gen) / c ** 0.5
    b = 0.1 * torch.randn(o, generator=gen)
    y = torch.randint(0, o, (n,), generator=gen)
    return pooled, w, b, y


@pytest.mark.parametrize('shape', LOSS_SHAPES, ids=lambda s: 'n%d_c%d_o%d' % s)
@pytest.mark.parametrize('reduction', ['mean', 'sum'])
def test_linear_cross_entropy_matches_oracle(shape, reduction):
    """The three call shapes -- `linear` then `cross_entropy` (the reference's loop: p = model(x); loss = CE(p, y)), the
    one-launch `linear_cross_entropy`, and `cross_entropy` on foreign logits -- against oracle.head / torch.autograd on the
    CPU: logits, loss, the per-batch statistics and every gradient."""
    import neural_ode_features_amd as nof
    from oracle.head import linear_cross_entropy as oracle_lce
    n, c, o = shape
    pooled, w, b, y = _loss_case(n, c, o, seed=n + c + o)
    pr, wr, br = (t.clone().requires_grad_(True) for t in (pooled, w, b))
    want, want_logits = oracle_lce(pr, wr, br, y, reduction)
    (1.7 * want).backward()
    hits = float((want_logits.argmax(1) == y).sum())
    tol = 2e-5

    def check(loss, logits, grads, what):
        assert rel_err(logits, want_logits) < tol, what
        assert abs(float(loss) - float(want)) <= tol * max(1.0, abs(float(want))), what
        stat = loss.node_stat.tolist()
        assert abs(stat[0] - float(want)) <= tol * max(1.0, abs(float(want))) and stat[1] == hits, (what, stat, hits)
        for g, r, name in zip(grads, (pr.grad, wr.grad, br.grad), ('d_pooled', 'd_weight', 'd_bias')):
            assert rel_err(g, r) < 5e-5, (what, name)

    # 1. drop-in call shape: two forward launches, ONE backward launch (the loss node carries the Linear layer)
    pg, wg, bg = (t.cuda().requires_grad_(True) for t in (pooled, w, b))
    logits = nof.linear(pg, wg, bg)
    loss = nof.cross_entropy(logits, y.cuda(), reduction=reduction)
    assert type(loss.grad_fn).__name__ == '_LinearCrossEntropyBackward'
    (1.7 * loss).backward()
    check(loss, logits, (pg.grad, wg.grad, bg.grad), 'linear + cross_entropy')
    # 2. one forward launch
    pg, wg, bg = (t.cuda().requires_grad_(True) for t in (pooled, w, b))
    loss, logits = nof.linear_cross_entropy(pg, wg, bg, y.cuda(), reduction=reduction)
    (1.7 * loss).backward()
    check(loss, logits, (pg.grad, wg.grad, bg.grad), 'linear_cross_entropy')
    # 3. the pieces on their own: Linear's backward from a given dL/dlogits, the loss on logits that came from elsewhere
    pg, wg, bg = (t.cuda().requires_grad_(True) for t in (pooled, w, b))
    logits = nof.linear(pg, wg, bg)
    cot = torch.randn(n, o, generator=torch.Generator().manual_seed(3))
    logits.backward(cot.cuda())
    pr2, wr2, br2 = (t.clone().requires_grad_(True) for t in (pooled, w, b))
    F.linear(pr2, wr2, br2).backward(cot)
    for g, r in zip((pg.grad, wg.grad, bg.grad), (pr2.grad, wr2.grad, br2.grad)):
        assert rel_err(g, r) < 5e-5
    lg = want_logits.detach().cuda().requires_grad_(True)
    loss = nof.cross_entropy(lg, y.cuda(), reduction=reduction)
    assert type(loss.grad_fn).__name__ == '_CrossEntropyBackward'
    loss.backward()
    lr = want_logits.detach().clone().requires_grad_(True)
    F.cross_entropy(lr, y, reduction=reduction).backward()
    assert rel_err(lg.grad, lr.grad) < 5e-5


def test_loss_is_bit_reproducible_and_the_scratch_counter_resets():
    """Fixed summation order, no float atomics: 50 launches give one loss, one set of gradients; the arrival counter of the
    last-workgroup reduction is left at zero by every launch (a second batch size reuses the same scratch)."""
    import neural_ode_features_amd as nof
    pooled, w, b, y = _loss_case(128, 256, 10, seed=1)
    pg, wg, bg, yg = pooled.cuda(), w.cuda().requires_grad_(True), b.cuda(), y.cuda()
    first = None
    for it in range(50):
        wg.grad = None
        loss, _ = nof.linear_cross_entropy(pg, wg, bg, yg)
        loss.backward()
        got = (float(loss), wg.grad.clone())
        if first is None:
            first = got
        assert got[0] == first[0] and torch.equal(got[1], first[1])
    p2, w2, b2, y2 = _loss_case(37, 256, 10, seed=2)
    l2, _ = nof.linear_cross_entropy(p2.cuda(), w2.cuda(), b2.cuda(), y2.cuda())
    assert abs(float(l2) - float(F.cross_entropy(F.linear(p2, w2, b2), y2))) < 1e-5


def test_training_step_runs_no_foreign_kernel_in_the_head():
    """model(x) -> cross_entropy -> backward of the whole ODENet: PyTorch dispatches no addmm / mm / log_softmax / nll_loss
    (the head's Linear layer and the loss are the library's launches)."""
    import neural_ode_features_amd as nof
    from torch.profiler import ProfilerActivity, profile
    torch.manual_seed(0)
    net = nof.ODENet(3, out=10, n_filters=64, downsample='residual', adjoint=True, dropout=0.5).cuda().train()
    x = torch.randn(8, 3, 32, 32, device='cuda')
    y = torch.randint(0, 10, (8,), device='cuda')
    nof.cross_entropy(net(x), y).backward()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        loss = nof.cross_entropy(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    bad = [k for k in names if any(s in k.lower() for s in ('addmm', 'aten::mm', 'log_softmax', 'nll_loss', 'aten::linear', 'cross_entropy_loss'))]
    assert not bad, bad
    assert not any('_LinearBackward' in k for k in names), names        # the loss node carries the Linear layer: ONE backward launch
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
