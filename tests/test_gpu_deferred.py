"""GPU: deferred completion of the solves (integrate.Deferred): blind step counts, the device-side miss flag, and the
optimizer step predicated on it."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _block(seed=7, tol=1e-4):
    import neural_ode_features_amd as nof
    from tests.helpers import make_func
    f, _ = make_func(32, seed=seed, device='cuda', kink_free=True)
    blk = nof.ODEBlock(n_filters=32, tol=tol, method='dopri5', adjoint=True, t1=1)
    blk.odefunc.load_state_dict(f.state_dict())
    return blk.cuda()


def test_deferred_steps_equal_synchronous_steps():
    """Same block, same data: four optimizer steps with deferred completion give the parameters four steps with a
    read-back per solve give, bit for bit (same kernels, same step sequence; the solver's arithmetic is deterministic),
    and after the first (learning) iteration every solve runs blind."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    a = _block()
    b = copy.deepcopy(a)
    oa = nof.FusedSGD(a.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    ob = nof.FusedSGD(b.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(4, 32, 8, 8, generator=torch.Generator().manual_seed(1)).cuda()
    d = integrate.Deferred(x.device)
    oa.use_deferred(d)
    losses_a, losses_b = [], []
    with d:
        for _ in range(4):
            la = a(x).square().mean()
            la.backward()
            oa.step()
            oa.zero_grad()
            losses_a.append(la)
    assert integrate.Deferred.active is None
    for _ in range(4):
        lb = b(x).square().mean()
        lb.backward()
        ob.step()
        ob.zero_grad()
        losses_b.append(lb)
    assert d.resolve() == 0 and d.blind_solves == 6          # iterations 2..4: forward + adjoint each (one spare
                                                             # step enqueued each time: the counts are still young)
    for la, lb in zip(losses_a, losses_b):
        assert float(la) == float(lb)
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    assert a.nfe == b.nfe                                     # the predicted counts were the true ones


def test_deferred_miss_commits_nothing_and_relearns():
    """A solve that does not finish within the steps enqueued (the previous count + one spare while counts are young)
    is a MISS: the device flag goes up, the optimizer step of that iteration leaves parameters and momentum untouched,
    and the next solve of that kind runs with a read-back again and re-learns its step count.  Fewer steps than
    guessed is no miss: the surplus steps do nothing, the record carries the true count and corrects `nfe`."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    blk = _block()
    opt = nof.FusedSGD(blk.parameters(), lr=1e-3, momentum=0.9)
    base = torch.randn(4, 32, 8, 8, generator=torch.Generator().manual_seed(2)).cuda()
    easy, hard = base * 0.05, base * 30.0
    d = integrate.Deferred(base.device)
    opt.use_deferred(d)

    def step(x):
        loss = blk(x).square().mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in blk.parameters()]

    with d:
        step(easy)                       # learns the step counts (read-back)
        p1 = step(easy)                  # blind, exact
        assert d.blind_solves == 2 and d.resolve() == 0
        blk.nfe = 0
        p2 = step(easy)                  # blind again (records consumed by resolve(): still known good)
        steps_easy = (blk.odefunc.last_forward_stats['accepted'], blk.odefunc.last_backward_stats['accepted'])
        assert any(not torch.equal(a, b) for a, b in zip(p1, p2))
        d.resolve()                      # consume the records of the step above (they would refresh the guesses)
        d.force_counts(1)                # make the miss certain whatever the two inputs need: one step, no spare
        p3 = step(hard)                  # blind with too few steps -> miss -> nothing committed
        for a, b in zip(p2, p3):
            assert torch.equal(a, b)
        assert float(d.miss_flag) == 0.0             # the optimizer reset the flag behind its (skipped) launch
        p4 = step(hard)                  # finds the miss, solves with a read-back, learns the new counts, commits
        assert d.misses >= 1
        assert any(not torch.equal(a, b) for a, b in zip(p3, p4))
        steps_hard = (blk.odefunc.last_forward_stats['accepted'] + blk.odefunc.last_forward_stats['rejected'],
                      blk.odefunc.last_backward_stats['accepted'] + blk.odefunc.last_backward_stats['rejected'])
        assert min(steps_hard) > 1
        before = d.blind_solves
        p5 = step(hard)                  # blind with the new counts, exact
        assert d.blind_solves == before + 2 and d.resolve() == d.misses
        assert any(not torch.equal(a, b) for a, b in zip(p4, p5))
        # MORE steps enqueued than needed is no miss: the surplus does nothing on the device, the update is committed,
        # and nfe is corrected by the record to what a synchronous solve counts
        d.resolve()
        truth = {k: v for k, v in d.guess.items()}
        d.force_counts({k: truth[k] + 3 for k in truth})
        misses = d.misses
        blk.nfe = 0
        p6 = step(hard)
        d.resolve()
        assert d.misses == misses and any(not torch.equal(a, b) for a, b in zip(p5, p6))
        counts = {k[0]: v for k, v in d.guess.items()}               # from the records: steps actually tried
        assert blk.nfe == (2 + 6 * counts['fwd']) + (3 + 6 * counts['bwd'])
        assert all(abs(d.guess[k] - truth[k]) <= 1 for k in truth)    # the guesses came back down
        assert d.dead_steps >= 6                                      # ... and the surplus steps were counted as dead
    for p in p5:
        assert bool(torch.isfinite(p).all())


def test_loop_repeats_a_missed_batch_and_matches_the_synchronous_run_bit_for_bit():
    """integrate.DeferredLoop: a miss costs time, not an update.  Eight batches, the sixth enqueued with ONE step where
    several are needed (a certain miss): the loop voids that iteration and the one already enqueued
    behind it (the device flag is sticky), then repeats both in order with a read-back per solve, under the random
    generator state of their first attempt (the loss goes through a dropout mask).  Parameters, momentum and every
    loss are bit-identical to eight synchronous steps; no update is lost, two batches ran twice."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    a = _block()
    b = copy.deepcopy(a)
    oa = nof.FusedSGD(a.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    ob = nof.FusedSGD(b.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(11)
    xs = [torch.randn(4, 32, 8, 8, generator=gen).cuda() for i in range(8)]

    def make_step(blk, opt):
        def step(x):
            loss = F.dropout(blk(x), 0.5, training=True).square().mean()
            loss.backward()
            opt.step()
            opt.zero_grad()
            return loss.detach()
        return step

    torch.manual_seed(3)
    torch.cuda.manual_seed(3)
    sync_step = make_step(b, ob)
    losses_b = [sync_step(x) for x in xs]

    torch.manual_seed(3)
    torch.cuda.manual_seed(3)
    d = integrate.Deferred(xs[0].device)
    loop = integrate.DeferredLoop(d, oa, make_step(a, oa))
    losses_a = []
    for i, x in enumerate(xs):
        if i == 5:                       # make the miss certain: one step enqueued, no spare, for a solve that needs several
            d.resolve()
            d.force_counts(1)
        losses_a += loop.step(x)
    losses_a += loop.flush()
    assert len(losses_a) == 8
    assert loop.miss_events >= 1 and loop.retries >= 1 and d.misses >= 1, (loop.miss_events, loop.retries, d.misses)
    for i, (la, lb) in enumerate(zip(losses_a, losses_b)):
        assert float(la) == float(lb), i
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(oa.state[p]['momentum_buffer'], ob.state[q]['momentum_buffer'])
    assert float(d.miss_flag) == 0.0 and a.nfe > 0
    # a copy of the block (an EMA / evaluation copy) must not share the original's pending record: tokens live beside
    # the modules, not on them
    c = copy.deepcopy(a)
    assert integrate._func_token(c.odefunc) != integrate._func_token(a.odefunc)


def test_nothing_runs_blind_without_a_predicated_commit_point():
    """Inference inside a deferred scope, a scope no optimizer was armed with, and the plain drop-in API all return
    finished solves with measured statistics."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    blk = _block()
    x = torch.randn(2, 32, 8, 8).cuda()
    assert integrate.Deferred.active is None

    def finished():
        st = blk.odefunc.last_forward_stats
        return st['status'] == 0 and st['nfe'] == 2 + 6 * (st['accepted'] + st['rejected'])

    with torch.no_grad():
        blk(x)
    assert finished()
    d = integrate.Deferred(x.device)
    with d:                                   # not armed: no optimizer predicates its step on the flag
        for _ in range(3):
            blk(x.clone().requires_grad_(True)).sum().backward()
            assert finished() and blk.odefunc.last_backward_stats['status'] == 0
    opt = nof.FusedSGD(blk.parameters(), lr=1e-3)
    opt.use_deferred(d)
    with d:
        for _ in range(3):                    # armed, but inference: nothing commits, so nothing is deferred
            with torch.no_grad():
                blk(x)
            assert finished()
    assert d.blind_solves == 0


@pytest.mark.timeout(900)
def test_a_miss_on_one_rank_skips_the_update_on_every_rank():
    """Two ranks sharing cuda:0 over gloo (tests/dp_miss_child.py): a miss forced on rank 1 ONLY must skip the optimizer
    step on BOTH ranks (the flag travels in the reducer's last bucket, dp.GradientReducer.carry_flag), parameters
    bit-identical across the ranks before, at and after the skipped step."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    env_clean = {k: None for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    old = {k: os.environ.pop(k, None) for k in env_clean}
    try:
        rc = bench.spawn_ranks(2, [sys.executable, os.path.join(root, 'tests', 'dp_miss_child.py')], timeout=800)
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v
    assert rc == 0


@pytest.mark.timeout(900)
def test_global_norm_mode_two_ranks_take_identical_steps(tmp_path):
    """GLOBAL-NORM mode on the HIP path (include/node_hip.h, node_solve_opts::norm_reduce; `options={'global_norm': True}`): two ranks
    sharing cuda:0 over gloo, each integrating its shard of a batch whose two halves differ threefold in size.  Both ranks take
    bit-identical step sequences forward and backward; the forward history equals a single process integrating the whole batch; with
    local norms the ranks disagree (tests/dp_gnorm_child.py, checks shared with tests/test_dp_gloo.py)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tests.test_dp_gloo import _check_global_norm_histories
    old = {k: os.environ.pop(k, None) for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    try:
        _check_global_norm_histories(tmp_path, 'hip')
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v
