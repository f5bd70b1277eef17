"""The resident form of the latency path (csrc/kernels_tiny_solve.hip): a whole forward dopri5 solve of a tiny state --
the bs = 1 census of the reference, evaluate.py:97-142 -- in ONE launch, against the oracle solver and against the
launch-per-convolution path (csrc/kernels_tiny.hip, NODE_TUNE_TINY_RESIDENT=0) on the same inputs."""
import contextlib
import ctypes as C
import os

import pytest
import torch

from tests.helpers import make_func, rel_err

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def resident(on):
    old = os.environ.get('NODE_TUNE_TINY_RESIDENT')
    os.environ['NODE_TUNE_TINY_RESIDENT'] = str(int(on))
    try:
        yield
    finally:
        if old is None:
            del os.environ['NODE_TUNE_TINY_RESIDENT']
        else:
            os.environ['NODE_TUNE_TINY_RESIDENT'] = old


def _is_resident(shape):
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    N, Cc, H, W = shape
    s = _lib.NodeShape(N, Cc, H, W, min(32, Cc), 1e-5)
    return lib.node_solve_is_resident(C.byref(s)) == 1


RESIDENT_SHAPES = [(1, 256, 8, 8), (1, 64, 8, 8), (2, 256, 8, 8), (3, 128, 7, 7), (4, 64, 8, 8), (1, 32, 4, 4), (2, 128, 5, 6)]


def test_which_shapes_take_the_resident_solve():
    for shape in RESIDENT_SHAPES:
        assert _is_resident(shape), shape
        with resident(False):
            assert not _is_resident(shape)
    # more workgroups than CUs, more than 64 pixels, groups that do not tile a 16-channel block, batches past the latency regime
    for shape in [(4, 256, 8, 8), (1, 256, 16, 16), (1, 96, 8, 8), (1, 512, 8, 8), (8, 64, 8, 8), (128, 256, 8, 8)]:
        assert not _is_resident(shape), shape


@pytest.mark.parametrize('tol', [1e-3, 1e-5])
@pytest.mark.parametrize('shape', RESIDENT_SHAPES)
def test_resident_solve_matches_oracle_and_launch_path(shape, tol):
    """Same problem three ways: the oracle's free-running solve (CPU, fp32), the launch-per-convolution latency path, the
    resident solve.  Dense output at interior times, step statistics, NFE accounting (show.py:199)."""
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    N, Cc, H, W = shape
    f, twin = make_func(Cc, seed=301, device='cuda', kink_free=True)
    gen = torch.Generator().manual_seed(302)
    y = torch.randn(N, Cc, H, W, generator=gen)
    t = torch.tensor([0.0, 0.25, 0.6, 1.0])
    fs_o = tdq.SolverStats()
    with torch.no_grad():
        ref = tdq.odeint(twin, y, t, rtol=tol, atol=tol, method="dopri5", stats=fs_o)
        outs, stats, nfes = [], [], []
        for on in (False, True):
            with resident(on):
                f.nfe = 0
                outs.append(nof.odeint(f, y.cuda(), t.cuda(), rtol=tol, atol=tol, method='dopri5'))
                stats.append(dict(f.last_forward_stats))
                nfes.append(f.nfe)
    steps = [s['accepted'] + s['rejected'] for s in stats]
    print(shape, tol, 'oracle acc/rej', fs_o.accepted, fs_o.rejected, 'launch path', stats[0]['accepted'], stats[0]['rejected'],
          'resident', stats[1]['accepted'], stats[1]['rejected'],
          '| resident vs launch path %.2e, vs oracle %.2e' % (rel_err(outs[1], outs[0]), rel_err(outs[1], ref)))
    assert torch.equal(outs[1][0].cpu(), y)
    assert nfes[1] == stats[1]['nfe'] == 2 + 6 * steps[1]
    scale = float(ref.abs().max())
    assert float((outs[1].cpu() - ref).abs().max()) <= 10 * tol * (1 + scale)          # north star: 10 x (atol + rtol |y|)
    assert float((outs[1] - outs[0]).abs().max()) <= 10 * tol * (1 + scale)
    if (stats[1]['accepted'], stats[1]['rejected']) == (fs_o.accepted, fs_o.rejected):
        assert rel_err(outs[1], ref) < 2e-4
    else:       # one accept / reject decision within rounding of 1.0 went the other way
        assert abs(steps[1] - (fs_o.accepted + fs_o.rejected)) <= 1
    if steps[1] == steps[0]:
        assert rel_err(outs[1], outs[0]) < 1e-4
        assert abs(stats[1]['first_dt'] - stats[0]['first_dt']) <= 1e-4 * abs(stats[0]['first_dt'])


def test_resident_solve_replays_a_step_list_and_logs_its_steps():
    """Replay mode (every step accepted, sizes from the list) and the dt log are the parity tests' instruments: the resident
    solve takes them like the step-per-launch loop -- same numbers from both, tight against the oracle's replay."""
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    shape = (1, 256, 8, 8)
    f, twin = make_func(shape[1], seed=311, device='cuda', kink_free=True)
    y = torch.randn(*shape, generator=torch.Generator().manual_seed(312))
    t = torch.tensor([0.0, 0.5, 1.0])
    forced = [0.1, 0.2, 0.3, 0.25, 0.15]
    with torch.no_grad():
        ref = tdq.odeint(twin, y, t, rtol=1e-3, atol=1e-3, method='dopri5', options={'forced_dts': forced})
        got = []
        for on in (False, True):
            with resident(on):
                got.append(nof.odeint(f, y.cuda(), t.cuda(), rtol=1e-3, atol=1e-3, method='dopri5', options={'forced_dts': forced}))
                assert f.last_forward_stats['accepted'] == 5 and f.last_forward_stats['rejected'] == 0
    print('replay: resident vs oracle %.2e, vs launch path %.2e' % (rel_err(got[1], ref), rel_err(got[1], got[0])))
    assert rel_err(got[1], ref) < 2e-5 and rel_err(got[1], got[0]) < 2e-5
    # free-running with the log on: the logged sizes reproduce the solve when replayed
    with torch.no_grad(), resident(True):
        free = nof.odeint(f, y.cuda(), t.cuda(), rtol=1e-4, atol=1e-4, method='dopri5', options={'record_dt': 64})
        st = f.last_forward_stats
        assert len(st['dts']) == st['accepted'] + st['rejected'] and sum(st['accepts']) == st['accepted']
        accepted = [d for d, ok in zip(st['dts'], st['accepts']) if ok]
        again = nof.odeint(f, y.cuda(), t.cuda(), rtol=1e-4, atol=1e-4, method='dopri5', options={'forced_dts': accepted})
    assert rel_err(again, free) < 1e-6


def test_resident_solve_backward_in_time_and_bit_reproducible():
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    shape = (2, 128, 8, 8)
    f, twin = make_func(shape[1], seed=321, device='cuda', kink_free=True)
    y = torch.randn(*shape, generator=torch.Generator().manual_seed(322))
    t = torch.tensor([1.0, 0.7, 0.0])
    with torch.no_grad(), resident(True):
        ref = tdq.odeint(twin, y, t, rtol=1e-4, atol=1e-4, method='dopri5')
        a = nof.odeint(f, y.cuda(), t.cuda(), rtol=1e-4, atol=1e-4, method='dopri5')
        b = nof.odeint(f, y.cuda(), t.cuda(), rtol=1e-4, atol=1e-4, method='dopri5')
    assert torch.equal(a, b)
    assert rel_err(a, ref) < 2e-4


def test_resident_solve_reports_status_and_recovers():
    """Upstream's assertions ('non-finite values in state', 'max_num_steps exceeded') come back as the C ABI's error codes from
    inside the one launch; the grid drains and the next solve starts clean."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd._lib import NodeHipError
    shape = (1, 64, 8, 8)
    assert _is_resident(shape)
    f, _ = make_func(shape[1], seed=331, device='cuda')
    t = torch.tensor([0.0, 1.0]).cuda()
    y = torch.randn(*shape, generator=torch.Generator().manual_seed(332)).cuda()
    with torch.no_grad(), resident(True):
        good = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3)
        bad = y.clone()
        bad[0, 3, 2, 2] = float('inf')
        with pytest.raises(NodeHipError, match='NONFINITE'):
            nof.odeint(f, bad, t, rtol=1e-3, atol=1e-3)
        for _ in range(2):
            with pytest.raises(NodeHipError, match='MAX_STEPS'):
                nof.odeint(f, y, t, rtol=1e-9, atol=1e-9, options={'max_num_steps': 3})
        again = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3)
    assert torch.equal(good, again)


def test_resident_solve_through_the_module_interface_and_deferred_record():
    """ODEBlock.forward under no_grad at bs = 1 (what evaluate.py:97-142 runs per image) lands on the resident solve; a blind
    (deferred-completion) solve of the same kind cannot miss: the kernel takes the steps it needs."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    block = nof.ODEBlock(n_filters=64, tol=1e-3).cuda()
    x = torch.randn(1, 64, 8, 8, generator=torch.Generator().manual_seed(341)).cuda()
    with torch.no_grad():
        with resident(False):
            want = block(x)
        with resident(True):
            got = block(x)
            nfe = block.nfe
    assert rel_err(got, want) < 1e-4
    rec = integrate.Recognised(block.odefunc)
    record = torch.zeros(64, dtype=torch.uint8, device='cuda')
    flag = torch.zeros(1, device='cuda')
    with resident(True):
        ref, st = integrate.solve_forward(rec, rec.params, x, [0.0, 1.0], 1e-3, 1e-3, 0, None)
        out, _ = integrate.solve_forward(rec, rec.params, x, [0.0, 1.0], 1e-3, 1e-3, 0, None, blind=(1, record, flag))   # a guess of ONE step
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and float(flag) == 0.0
    assert nfe >= 2 + 6 * (st['accepted'] + st['rejected'])


def test_a_grid_that_is_not_whole_drains_and_the_solve_falls_back():
    """The resident solve needs its whole grid on the chip.  NODE_TUNE_TINY_RESIDENT=2 launches it one workgroup short: every
    wait runs into its deadline (0.25 s of the constant clock), the grid drains, the C call repeats the solve on the
    launch-per-convolution path -- same result as that path by itself, no error; the process then leaves the resident grid alone for
    its next 64 solves and takes it again after them."""
    import time
    import neural_ode_features_amd as nof
    shape = (1, 128, 8, 8)
    f, _ = make_func(shape[1], seed=351, device='cuda')
    t = torch.tensor([0.0, 0.5, 1.0]).cuda()
    y = torch.randn(*shape, generator=torch.Generator().manual_seed(352)).cuda()
    with torch.no_grad():
        with resident(0):
            want = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3)
            stats = dict(f.last_forward_stats)
        with resident(2):
            t0 = time.perf_counter()
            got = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3)
            torch.cuda.synchronize()
            waited = time.perf_counter() - t0
            assert dict(f.last_forward_stats) == stats
        with resident(2):      # the next solves of this process do not try the resident grid again for a while: no second deadline
            t0 = time.perf_counter()
            for _ in range(64):
                again = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3)
            torch.cuda.synchronize()
            cooled = time.perf_counter() - t0
            assert torch.equal(again, want)
        with resident(1):      # ... and then it is taken again
            back = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3)
    print('fallback after %.2f s; the 64 solves behind it %.3f s' % (waited, cooled))
    assert torch.equal(got, want)
    assert 0.2 < waited < 5.0 and cooled < 1.0
    assert rel_err(back, want) < 1e-4 and not torch.equal(back, want)      # (the two paths round differently: equal bits would mean the same path)


@pytest.mark.parametrize('gain,tol,kink_free', [(1.0, 1e-4, True), (12.0, 1e-6, False)])
def test_resident_solve_with_rejected_steps_and_many_time_points(gain, tol, kink_free):
    """13 target times are more than ride in the kernel arguments (> 8: the device array).  The second case is there for the REJECT
    branch: a stiffer problem (last GroupNorm's weight x 12), ordinary ReLU-kinked parameters and a tolerance of 1e-6 -- the oracle
    takes 55 steps and rejects two.  At that tolerance fp32 rounding is a visible part of the error estimate, so a few decisions may
    go the other way in either implementation: step counts within 3, outputs within 2e-5 (1 + max|y|)."""
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    shape = (1, 64, 8, 8)
    f, twin = make_func(shape[1], seed=361, device='cuda', kink_free=kink_free)
    with torch.no_grad():
        f.norm3.weight.mul_(gain)
        twin.norm3.weight.mul_(gain)
    y = torch.randn(*shape, generator=torch.Generator().manual_seed(362))
    t = torch.linspace(0.0, 1.0, 13)
    fs_o = tdq.SolverStats()
    with torch.no_grad():
        ref = tdq.odeint(twin, y, t, rtol=tol, atol=tol, method='dopri5', stats=fs_o)
        with resident(0):
            base = nof.odeint(f, y.cuda(), t.cuda(), rtol=tol, atol=tol, method='dopri5')
            st0 = dict(f.last_forward_stats)
        with resident(1):
            f.nfe = 0
            out = nof.odeint(f, y.cuda(), t.cuda(), rtol=tol, atol=tol, method='dopri5')
            st = dict(f.last_forward_stats)
    print('gain', gain, 'tol', tol, 'oracle acc/rej', fs_o.accepted, fs_o.rejected, 'launch path', st0['accepted'], st0['rejected'],
          'resident', st['accepted'], st['rejected'], 'rel err vs oracle %.2e, vs launch path %.2e' % (rel_err(out, ref), rel_err(out, base)))
    assert f.nfe == 2 + 6 * (st['accepted'] + st['rejected'])
    slack = 1 if tol >= 1e-5 else 3
    assert abs(st['accepted'] - fs_o.accepted) <= slack and abs(st['rejected'] - fs_o.rejected) <= slack
    big = float(ref.abs().max())
    assert float((out.cpu() - ref).abs().max()) <= max(10 * tol, 2e-5) * (1 + big)
    if (st['accepted'], st['rejected']) == (fs_o.accepted, fs_o.rejected) and tol >= 1e-5:
        assert rel_err(out, ref) < 2e-4
    if gain > 1.0:
        assert fs_o.rejected >= 1 and st['rejected'] >= 1, 'this case is here for the reject branch'


def test_a_captured_solve_does_not_take_the_resident_grid():
    """A deferred-completion solve has no host synchronisation in it and can be captured into a hipGraph.
    A captured RESIDENT launch would replay its nonce -- the previous replay's words would pass for this one's -- so under capture the
    library keeps the launch-per-convolution path: replays on changing inputs equal the eager launch-path results bit for bit."""
    from neural_ode_features_amd import integrate
    import neural_ode_features_amd as nof
    shape = (1, 64, 8, 8)
    f = nof.ODEfunc(shape[1]).cuda()
    rec = integrate.Recognised(f)
    gen = torch.Generator().manual_seed(371)
    ys = [torch.randn(*shape, generator=gen).cuda() for _ in range(3)]
    y = ys[0].clone()
    record = torch.zeros(64, dtype=torch.uint8, device='cuda')
    flag = torch.zeros(1, device='cuda')
    times = [0.0, 1.0]
    with resident(0):
        wants, steps = [], 1
        for yi in ys:
            o, st = integrate.solve_forward(rec, rec.params, yi, times, 1e-3, 1e-3, 0, None)
            wants.append(o.clone())
            steps = max(steps, st['accepted'] + st['rejected'])

    def blind():
        return integrate.solve_forward(rec, rec.params, y, times, 1e-3, 1e-3, 0, None, blind=(steps + 1, record, flag))[0]

    with resident(1):
        assert _is_resident(shape)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            blind()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            static_out = blind()
        for yi, want in zip(ys, wants):
            y.copy_(yi)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(static_out, want) and float(flag) == 0.0


def test_inference_graphs_around_the_resident_solve_equal_the_eager_path():
    """`graphs.capture_inference` (the bs = 1 census, evaluate.py:97-142): stem and head as hipGraphs around the ODE block's own
    launches give the eager path's logits bit for bit, for image after image; training-mode calls, other shapes and calls that want
    gradients keep the ordinary path."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import graphs
    torch.manual_seed(5)
    model = nof.ODENet(3, out=10, n_filters=64, downsample='residual', method='dopri5', tol=1e-3).cuda().eval()
    x = torch.randn(6, 3, 32, 32, device='cuda')
    with torch.no_grad():
        want = [model(x[i:i + 1]).clone() for i in range(6)]
        batch = model(x[:4]).clone()
    graphs.capture_inference(model, x[:1])
    with torch.no_grad():
        for i in range(6):
            got = model(x[i:i + 1])
            assert torch.equal(got, want[i]), i
        assert torch.equal(model(x[:4]), batch)                  # another shape: the eager path
    assert model.nfe(reset=True) > 0
    y = model(x[:1])                                             # gradients wanted: the eager path, an autograd graph behind it
    assert y.requires_grad and torch.allclose(y.detach(), want[0], atol=1e-5)
    model.train()
    assert model(x[:2]).shape == (2, 10)
