#!/usr/bin/env python
"""Headline benchmark: images/sec (forward + adjoint backward + SGD step) of the
CIFAR-10 ODE-ResNet, dopri5, bs=128 per GPU (BASELINE.json configs[1]; with --gpus N
the batch shards data-parallel, N x 128 = configs[3] at N=8).

    python bench.py                                  # cfg 2, one GPU
    python bench.py --config 3                       # tol 1e-5 (configs[2])
    python bench.py --config 5                       # 64x64 input, 1024 filters, 3 stacked blocks, 64 images / GPU
    python bench.py --gpus N --steps K --warmup W    # spawns N ranks itself (one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # ... or runs as one rank of a launcher

`--gpus N` never runs at a smaller world size: without a launcher's environment the
script starts N child ranks (before anything touches the GPU); if the box has fewer
than N devices, or the launcher's WORLD_SIZE disagrees, it exits non-zero.

A "step" is the loop body of the reference's train.py:40-58 on device-resident
synthetic tensors: p = model(x); loss = CE(p, y); loss.backward(); optimizer.step();
optimizer.zero_grad().  The ODE block(s) -- the hot path -- run in libnode_hip.so
through the C ABI; the optimizer step is one fused launch (node_sgd_step).  By default
the solves run with DEFERRED COMPLETION (integrate.Deferred): they enqueue the step
count of the previous iteration and return without a host read-back, the device
records whether that was exact, and the optimizer step is predicated on it on the
device -- a step with a miss commits nothing, and a timed region that contains one is
measured again (never reported).  `--no-deferred` gives the drop-in behaviour: one
read-back per solve.

Prints ONE JSON line (rank 0).  `value` = images of all ranks / wall time of exactly
K steps (barrier + synchronize on both sides, max over ranks); `step_ms` holds the
per-step median / min / max from HIP events on the compute stream.  A solve that misses
its enqueued step count costs TIME, not an update: integrate.DeferredLoop keeps every
batch until its verdict is in and repeats voided batches in order (`config.retries`).
`fresh_batches` is the same loop on a NEW synthetic batch per step (moving step counts),
`dropin` the same steps with a read-back per solve (the drop-in odeint / ODEBlock API).
`other_configs` (default run only) holds short side runs of BASELINE configs 3 and 5 on the same box;
`latency_bs1` is a side block, not the metric: forward solves of ONE sample of the config's
state (the reference's bs = 1 NFE census, evaluate.py:97-142), microseconds per evaluation.
`roofline` is for the dominant kernel -- at the BASELINE configs `k_w4_gemm64h`, the 36
component GEMMs of a Winograd F(4x4,3x3) convolution on fp16 MFMA at fp32 accuracy
(both operands as scaled fp16 pairs, three products), bounded by its bytes through the
fabric at C = 256 (`bound: hbm`, achieved = algorithmic bytes per launch / mean launch
duration from HIP events recorded by the library on the launch stream) and by the matrix
pipe at C = 1024 (`k_w4_gemm256h` + `k_w4_gemm128h`, `bound: mfma`); `roofline.hbm` prices the HBM-bound
GroupNorm / transform passes, `roofline.wgrad` the weight gradient.  `cpu_baseline` is
the oracle (CPU restatement of the torchdiffeq path driving PyTorch-CPU conv /
group_norm) on this box's host cores.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak
HBM_PEAK_TBS = 8.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s is what a streaming copy reaches)

# BASELINE.json configs (1-based like SURVEY.md 8d); per-GPU batch
CONFIGS = {
    2: dict(filters=256, tol=1e-3, batch=128, image=32, blocks=1,
            name='CIFAR-10 ODE-ResNet (odenet, residual stem, 256 filters), dopri5 tol=1e-3, adjoint backward'),
    3: dict(filters=256, tol=1e-5, batch=128, image=32, blocks=1,
            name='CIFAR-10 ODE-ResNet (odenet, residual stem, 256 filters), dopri5 tol=1e-5, adjoint backward'),
    5: dict(filters=1024, tol=1e-3, batch=64, image=64, blocks=3,
            name='synthetic 64x64x3 input, 4x-widened ODE-ResNet (1024 filters), 3 stacked ODE blocks, dopri5 tol=1e-3, '
                 'adjoint backward'),
}


# ---------------------------------------------------------------------------
# self-launcher: one child process per GPU, started before any GPU call
# ---------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(world, argv, env_extra=None, timeout=None):
    """Start `world` copies of `argv` as ranks 0..world-1 of one node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in the environment, rendezvous on 127.0.0.1), wait for all of them, return the worst exit code.  Children are
    fresh processes: the parent must not have initialised the GPU (it only counts devices)."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(list(argv), env=env))
    rc = 0
    deadline = None if timeout is None else time.time() + timeout
    try:
        for p in procs:
            left = None if deadline is None else max(1.0, deadline - time.time())
            code = p.wait(timeout=left)
            rc = rc or code
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def resolve_world(args):
    """(world, rank, local_rank) from the launcher's environment, or None after this process acted as the launcher."""
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is not None:
        world = int(env_world)
        if world != args.gpus:
            raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
        return world, int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus == 1:
        return 1, 0, 0
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < args.gpus and not (args.share_gpu and have >= 1):
        raise SystemExit('bench.py: --gpus %d requested but this node exposes %d HIP device(s); refusing to run at a '
                         'smaller world size' % (args.gpus, have))
    rc = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    raise SystemExit(rc)


# ---------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------
def build_model(device, cfg, method):
    import torch
    import neural_ode_features_amd as nof
    torch.manual_seed(23)       # train.py:224,227
    if cfg['blocks'] == 1:
        model = nof.ODENet(3, out=10, n_filters=cfg['filters'], downsample='residual', method=method, tol=cfg['tol'],
                           adjoint=True, t1=1, dropout=0.5)
    else:
        model = nof.StackedODENet(3, out=10, n_filters=cfg['filters'], n_blocks=cfg['blocks'], downsample='residual',
                                  method=method, tol=cfg['tol'], adjoint=True, t1=1, dropout=0.5)
    return model.to(device)


def train_step(model, opt, x, y, reducer=None):
    # train.py:40-58.  `cross_entropy` is the package's F.cross_entropy (one launch forward; its backward launch also
    # carries the classifier's Linear layer: csrc/kernels_loss.hip); on CPU tensors (cpu_baseline) it IS F.cross_entropy
    from neural_ode_features_amd import cross_entropy
    p = model(x)
    loss = cross_entropy(p, y)
    nfe_f = model.nfe(reset=True)
    loss.backward()
    nfe_b = model.nfe(reset=True)
    if reducer is not None:
        reducer.finish()
    opt.step()
    opt.zero_grad()
    return loss, nfe_f, nfe_b


def cpu_baseline(state_dict, cfg, method, min_seconds=12.0, max_seconds=30.0):
    """The reference-equivalent CPU path: same net, the oracle standing in for torchdiffeq (which cannot be
    installed here), on a BOUNDED sample of the workload: whole training iterations at a batch sized so that
    one iteration takes a few seconds, repeated for about `min_seconds`."""
    import torch
    import neural_ode_features_amd as nof
    from oracle import torchdiffeq_restated as tdq
    # threads actually used: the GPU box exposes every host core but gives one GPU slot a 16-core
    # share, and oversubscribing PyTorch-CPU's small convs is slower than not (666 s for two
    # iterations at 256 threads, measured), so the pool is capped at 16.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    torch.manual_seed(23)
    if cfg['blocks'] == 1:
        model = nof.ODENet(3, out=10, n_filters=cfg['filters'], downsample='residual', method=method, tol=cfg['tol'],
                           adjoint=True, t1=1, dropout=0.5)
        model.load_state_dict(state_dict)
        model.odeblock.odeint = tdq.odeint_adjoint       # CPU solver = the checker, timed as the baseline
        bs = cfg['batch']
    else:
        model = nof.StackedODENet(3, out=10, n_filters=cfg['filters'], n_blocks=cfg['blocks'], downsample='residual',
                                  method=method, tol=cfg['tol'], adjoint=True, t1=1, dropout=0.5)
        model.load_state_dict(state_dict)
        for b in model.odeblocks:
            b.odeint = tdq.odeint_adjoint
        bs = 2                                            # cfg 5: ~0.6 TFLOP per image and iteration
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(1234)
    img = cfg['image']
    if cfg['blocks'] == 1:
        xw = torch.randn(8, 3, img, img, generator=gen)
        yw = torch.randint(0, 10, (8,), generator=gen)
        train_step(model, opt, xw, yw)                    # warm-up (thread pool, oneDNN primitives)
    x = torch.randn(bs, 3, img, img, generator=gen)
    y = torch.randint(0, 10, (bs,), generator=gen)
    t0 = time.perf_counter()
    nf = nb = 0
    done = 0
    while done < 1 or (time.perf_counter() - t0 < min_seconds and done < 64):
        _, nf, nb = train_step(model, opt, x, y)
        done += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    dt = time.perf_counter() - t0
    return {'value': done * bs / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d training iterations at bs=%d (fwd + adjoint + SGD), last NFE-F %d NFE-B %d, %.1f s'
                      % (done, bs, nf, nb, dt)}


def pmc_measure(state, conv_path_env, timeout=(300, 180), tol=1e-3):
    """HBM bytes per launch of every library kernel of an adaptive solve (forward + adjoint, the config's tolerance) at this
    workload's state shape, MEASURED in this run: two child processes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no
    trace domain, the program itself behind `--`) over tools/prof_eval.py, started BEFORE this process touches the GPU.
    FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (wide coalesced reads are tallied at half
    their bytes); WRITE_SIZE also counts evictions of the previous kernel's dirty lines.  Returns {kernel name: bytes}
    or None (no rocprofv3, a failed pass): `roofline.traffic` is then null -- never a stale committed number."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if rocprof is None:
        return None
    env = dict(os.environ)
    env.update(conv_path_env)
    env['TMPDIR'] = '/tmp'
    got = {}
    for ci, counter in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):   # (the first child may pay the cold `import torch` of a fresh box)
        d = tempfile.mkdtemp(prefix='node_pmc_', dir='/tmp')
        cmd = [rocprof, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable,
               os.path.join(ROOT, 'tools', 'prof_eval.py'), '--shape', ','.join(str(v) for v in state), '--iters', '2', '--solve', repr(float(tol))]
        try:
            r = subprocess.run(cmd, env=env, cwd='/tmp', capture_output=True, text=True, timeout=timeout[ci])
            if r.returncode != 0:
                return None
            acc = {}
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row['Kernel_Name'].split('(')[0].replace('void ', '')
                    if 'node::' in k and row['Counter_Name'] == counter:
                        a = acc.setdefault(k, [0.0, 0])
                        a[0] += float(row['Counter_Value'])
                        a[1] += 1
            if not acc:
                return None
            for k, (tot, n) in acc.items():
                got.setdefault(k, {})[counter] = tot / n
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k, v in got.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            out[k] = {'read': 2.0 * v['FETCH_SIZE'] * 1024.0, 'written': v['WRITE_SIZE'] * 1024.0,
                      'bytes': (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0}
    return out or None


def held_clock_ghz(N, C, side):
    """The clock the chip holds inside k_w4_gemm64b's K loop (s_memtime / s_memrealtime stamps around the loop).
    The stamps exist in the DIAGNOSTICS build of the library only (libnode_hip_diag.so): a child process measures it
    (tools/held_clock.py), started BEFORE this process touches the GPU.  None where the launch does not take that kernel, the
    diagnostics library is not built, or the child fails."""
    diag = os.path.join(ROOT, 'neural-ode-features_amd', 'csrc', 'libnode_hip_diag.so')
    if not os.path.exists(diag) or C >= 512 or (N * (4 if side == 16 else 1)) % 16 != 0 or C % 64 != 0:
        return None
    env = dict(os.environ)
    env['NODE_HIP_DIAG'] = '1'
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'held_clock.py'), '%d,%d,%d' % (N, C, side)], env=env,
                           capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            return None
        return json.loads(r.stdout.strip().splitlines()[-1]).get('held_clock_ghz')
    except Exception:
        return None


def pmc_lookup(pmc, kernel):
    """bytes per launch of the kernel whose name contains `kernel` (the instance with the most bytes), or None"""
    if not pmc:
        return None
    hits = [v for k, v in pmc.items() if kernel in k]
    return max(hits, key=lambda v: v['bytes']) if hits else None


def other_configs(args):
    """Side blocks of the default line, NOT the metric: the other single-GPU BASELINE configs (3: tol 1e-5; 5: three stacked blocks at
    1024 filters, per-GPU shard) as short runs of this same script in child processes -- so that the record of the default run holds a
    number for them measured on the same box.  A child that fails or runs past its time leaves an `error` entry."""
    out = {}
    for config, extra, limit in ((3, ['--steps', '10', '--warmup', '3'], 300), (5, ['--steps', '4', '--warmup', '2', '--no-dropin'], 420)):
        cmd = [sys.executable, os.path.abspath(__file__), '--config', str(config), '--method', args.method, '--no-cpu-baseline',
               '--no-fresh', '--no-latency', '--no-other-configs'] + extra
        env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not lines:
                out[str(config)] = {'error': 'rc %d: %s' % (r.returncode, r.stderr[-300:])}
                continue
            d = json.loads(lines[-1])
            rf = d.get('roofline') or {}
            out[str(config)] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                                'workload': d['config'].get('workload'), 'retries': d['config'].get('retries'),
                                'dead_steps_per_step': d['config'].get('dead_steps_per_step'),
                                'dropin': (d.get('dropin') or {}).get('value'),
                                'roofline': {k: rf.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_us',
                                                                    'launches', 'bytes_per_launch', 'flops_per_launch')} if rf else None}
        except Exception as e:
            out[str(config)] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def latency_bs1(cfg, method):
    """A side block of the line, NOT the metric: the other regime of the same path -- the reference's NFE census solves every test image
    on its own (evaluate.py:97-142).  Wall time of forward solves of ONE sample of this config's state, host included, per evaluation
    of the dynamics, at the config's tolerance and at 1e-5.  States the chip can hold resident run as one launch per solve
    (csrc/kernels_tiny_solve.hip); the others run the throughput kernels."""
    import torch
    import neural_ode_features_amd as nof
    try:
        C = cfg['filters']
        side = cfg['image'] // 4
        torch.manual_seed(1)
        f = nof.ODEfunc(C).cuda()
        y = torch.randn(1, C, side, side, device='cuda')
        t = torch.tensor([0.0, 1.0], device='cuda')
        rows = []
        with torch.no_grad():
            for tol in sorted({float(cfg['tol']), 1e-5}, reverse=True):
                for _ in range(5):
                    nof.odeint(f, y, t, rtol=tol, atol=tol, method=method)
                torch.cuda.synchronize()
                n = 100
                t0 = time.perf_counter()
                for _ in range(n):
                    nof.odeint(f, y, t, rtol=tol, atol=tol, method=method)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / n
                st = f.last_forward_stats
                rows.append({'tol': tol, 'nfe': st['nfe'], 'solve_us': wall * 1e6, 'us_per_evaluation': wall * 1e6 / max(1, st['nfe'])})
        from neural_ode_features_amd import _lib
        import ctypes
        shape = _lib.NodeShape(1, C, side, side, min(32, C), 1e-5)
        one_launch = bool(_lib.load().node_solve_is_resident(ctypes.byref(shape))) and method == 'dopri5'
        return {'state': [1, C, side, side], 'solves': rows, 'one_launch_per_solve': one_launch,
                'note': 'forward solve of ONE sample, wall time over 100 solves back to back (host included) / evaluations of the dynamics'}
    except Exception as e:      # (a side block never takes the line down)
        return {'error': '%s: %s' % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS), help='BASELINE.json config (1-based)')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch override')
    ap.add_argument('--filters', type=int, default=None)
    ap.add_argument('--tol', type=float, default=None)
    ap.add_argument('--method', default='dopri5')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc child passes (roofline.traffic = null)')
    ap.add_argument('--graphs', action='store_true',
                    help='replay the classifier head from hipGraphs (pays off only with --no-deferred: see graphs.py)')
    ap.add_argument('--settle', type=int, default=24,
                    help='deferred completion: at most this many further UNTIMED steps after the warm-up, until every solve '
                         'runs blind without a spare step (integrate.Deferred.settled), so that the timed region is the '
                         'steady state; reported as config.settle_steps')
    ap.add_argument('--no-dropin', action='store_true', help='skip the second timed region with a read-back per solve')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short side runs of BASELINE configs 3 and 5 that the default (config 2, one GPU) line carries')
    ap.add_argument('--no-latency', action='store_true', help='skip the bs = 1 latency block (evaluate.py:97-142: one image solved on its own)')
    ap.add_argument('--no-fresh', action='store_true', help='skip the timed region on a fresh synthetic batch per step')
    ap.add_argument('--no-deferred', action='store_true',
                    help='every solve ends with a read-back of the device controller (the drop-in default) instead of '
                         'deferred completion with a device-predicated optimizer step (integrate.Deferred)')
    ap.add_argument('--dist-backend', default='nccl', choices=('nccl', 'gloo'),
                    help='collective backend (nccl = RCCL; gloo only for the shared-GPU smoke test)')
    ap.add_argument('--force-dist', action='store_true',
                    help='with --gpus 1: run the N-rank code path anyway -- process group on the chosen backend (nccl = RCCL), '
                         'bucketed all-reduce of the gradients, the deferred-completion flag slot -- as a world of ONE rank; what '
                         'a 1-GPU box can prove about the multi-GPU path')
    ap.add_argument('--global-norm', action='store_true',
                    help='N > 1: GLOBAL-NORM solves (dp.enable_global_norm: the ranks add the sums behind every step decision -- identical '
                         'steps on all ranks, the mixed norm of the unsharded batch; one 32-byte all-reduce per step).  Default: local norms')
    ap.add_argument('--share-gpu', action='store_true',
                    help='TESTING ONLY: ranks share the devices that exist (rank %% device_count); exercises the N-rank code '
                         'path on a 1-GPU box, the number it prints is not a measurement')
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.batch is not None:
        cfg['batch'] = args.batch
    if args.filters is not None:
        cfg['filters'] = args.filters
    if args.tol is not None:
        cfg['tol'] = args.tol

    world, rank, local_rank = resolve_world(args)

    side = cfg['image'] // 4
    state = [cfg['batch'], cfg['filters'], side, side]
    pmc = None
    if world == 1 and not args.no_roofline and not args.no_pmc:
        # HBM traffic of the kernels of this workload, counted in THIS run (child processes under rocprofv3 --pmc), before
        # anything here initialises the GPU.  Single evaluations take the tolerance-gated F(4x4,3x3) pipeline only when
        # forced, so the children run with the conv path the timed solves will take.
        w4_default = os.environ.get('NODE_TUNE_WINO4', '1')
        takes_w4 = args.method == 'dopri5' and cfg['tol'] >= 0.99e-5 and side in (8, 16) and w4_default != '0'
        # (the children run whole solves at the config's tolerance: the adaptive-solve kernels -- fp16-pair GEMMs, weight gradient -- are in)
        pmc = pmc_measure(state, {} if takes_w4 else {'NODE_TUNE_WINO4': '0'}, tol=cfg['tol'])
    held_clk = None
    if world == 1 and not args.no_roofline:
        held_clk = held_clock_ghz(cfg['batch'], cfg['filters'], side)      # (a child process on the diagnostics library)

    others = None
    if world == 1 and args.config == 2 and not args.no_other_configs and args.batch is None and args.filters is None and args.tol is None:
        others = other_configs(args)      # (child processes, before this one touches the GPU)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (there is no CPU fallback for the product path)')
    if args.share_gpu:
        if args.dist_backend != 'gloo':
            raise SystemExit('bench.py: --share-gpu needs --dist-backend gloo (RCCL refuses two ranks on one device)')
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit('bench.py: LOCAL_RANK %d but only %d HIP device(s)' % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # (no library setting to tune: the stem, the ODE block, the head and the optimizer step all run on libnode_hip.so's own
    # kernels -- no MIOpen / hipBLASLt call is left in a training step, profiles/r05_cfg2_steps.txt)

    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    model = build_model(device, cfg, args.method)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(cfg['batch'], 3, cfg['image'], cfg['image'], generator=gen).to(device)   # normalised CIFAR-shaped
    y = torch.randint(0, 10, (cfg['batch'],), generator=gen).to(device)
    model.train()
    if args.graphs:
        # the classifier head (forward and backward) as hipGraphs, captured before RCCL's watchdog thread exists.
        # Same-process A/B at cfg 2 with a read-back per solve (tools/ab_step.py, medians of 60 steps): eager 8.93 ms,
        # head graphed 8.80 ms, stem graphed 9.16 ms, both 8.98 ms.  With deferred completion the host runs ahead of
        # the GPU anyway and the capture's side-stream synchronisation only costs: 8.42 ms eager vs 8.61 ms graphed
        # (profiles/r02_deferred_ab.txt) -- so the default is eager.
        nof.graphs.capture_static_parts(model, x, stem=False, head=True)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    reducer = None
    # SGD lr .1 momentum .9 wd 1e-4 (reproduce.sh:3-6, train.py:136): one fused launch for all parameter tensors
    opt = nof.FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    if dist_on:
        if world > 1:
            nof.dp.broadcast_parameters(model, 0)
        else:
            for p in model.parameters():
                dist.broadcast(p.data, src=0)
        # 2 MB buckets: the stem's gradients leave in several all-reduces while its backward is still running, so only
        # the last, small one (first conv + first block) is exposed behind it; the ODE block's 4.75 MB stay one bucket
        reducer = nof.dp.GradientReducer(model, limits={model.downsample: 2 << 20}, average=False,
                                         collectives_at_world_1=args.force_dist)   # all-reduce SUM in the bucket buffers ...
        opt.grad_scale = 1.0 / world                             # ... its 1/world folded into the optimizer step
        if args.global_norm and world > 1:
            nof.dp.enable_global_norm(model)

    def sync():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(device)

    # Deferred completion: the solves enqueue the step count of the previous iterations and return without a
    # read-back; a miss (too few steps) raises a STICKY device flag on which the optimizer step is predicated, so that
    # update and every later one is skipped until the host -- one iteration late, no stall -- sees the flag and repeats
    # the voided batches in order with a read-back per solve (integrate.DeferredLoop).  A miss costs time, never an update.
    deferred = None if args.no_deferred else integrate.Deferred(device)

    def one_step(xx, yy):
        return train_step(model, opt, xx, yy, reducer)

    loop = integrate.DeferredLoop(deferred, opt, one_step, reducer) if deferred is not None else None

    def run_steps(batches, events=None):
        """K training steps; returns (sum NFE-F, sum NFE-B) over the K committed updates."""
        nf = nb = 0
        for i, (xx, yy) in enumerate(batches):
            done = loop.step(xx, yy) if loop is not None else [one_step(xx, yy)]
            if events is not None:
                events[i + 1].record()
            for _, a, b in done:
                nf += a
                nb += b
        if loop is not None:
            for _, a, b in loop.flush():      # (inside the timed region: the last verdicts, and their repeats if any)
                nf += a
                nb += b
        return nf, nb

    fixed = [(x, y)] * args.steps
    run_steps([(x, y)] * args.warmup)
    # While a solve's step-count history is young ONE spare step is enqueued (launches that return at once); on this
    # fixed batch that takes ~10 iterations.  Those iterations are not the steady state a training run spends its
    # time in, so they stay outside the timed region.
    settle_steps = 0
    if deferred is not None:
        while settle_steps < args.settle:
            deferred.resolve()
            if world > 1:
                ok = torch.tensor([1.0 if deferred.settled() else 0.0], device=device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                calm = bool(ok.item())
            else:
                calm = deferred.settled()
            if calm:
                break
            run_steps([(x, y)])
            settle_steps += 1

    def counters():
        return (loop.retries, loop.miss_events, deferred.dead_steps, deferred.blind_solves) if loop is not None else (0, 0, 0, 0)

    def timed_region(batches, events=None):
        sync()
        if deferred is not None:
            deferred.resolve()
        c0 = counters()
        t0 = time.perf_counter()
        if events is not None:
            events[0].record()
        nf, nb = run_steps(batches, events)
        torch.cuda.synchronize(device)
        own = time.perf_counter() - t0          # this rank's own time for its K steps (before the barrier)
        sync()
        el = time.perf_counter() - t0
        if deferred is not None:
            deferred.resolve()
        c1 = counters()
        return {'elapsed': el, 'own': own, 'nfe_f': nf, 'nfe_b': nb, 'retries': c1[0] - c0[0], 'miss_events': c1[1] - c0[1],
                'dead_steps': c1[2] - c0[2], 'blind_solves': c1[3] - c0[3]}

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    region = timed_region(fixed, ev)
    elapsed, nfe_f, nfe_b = region['elapsed'], region['nfe_f'], region['nfe_b']
    per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]

    def over_ranks(value, op):
        if world == 1:
            return value
        tt = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=op)
        return float(tt.item())

    elapsed = over_ranks(elapsed, dist.ReduceOp.MAX)
    blocks = list(model.odeblocks) if hasattr(model, 'odeblocks') else [model.odeblock]
    # what a multi-GPU line needs to be diagnosable: every rank's own time and step counts (stragglers at the
    # gradient barrier, per-shard step-count divergence, who missed)
    mine = [float(rank), region['own'] / args.steps * 1e3, float(region['retries']), float(region['miss_events']),
            float(region['dead_steps']), region['nfe_f'] / args.steps, region['nfe_b'] / args.steps]
    for b in blocks:
        fs, bs_ = b.odefunc.last_forward_stats, b.odefunc.last_backward_stats
        mine += [float(fs['accepted'] + fs['rejected']), float(bs_['accepted'] + bs_['rejected'])]
    if world > 1:
        gathered = [torch.zeros(len(mine), dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(mine, dtype=torch.float64, device=device))
        rows = [g.tolist() for g in gathered]
    else:
        rows = [mine]
    per_rank = [{'rank': int(r[0]), 'ms_per_step_own': r[1], 'retries': int(r[2]), 'miss_events': int(r[3]),
                 'dead_steps': int(r[4]), 'nfe_forward_per_step': r[5], 'nfe_backward_per_step': r[6],
                 'last_steps_fwd_bwd_per_block': [[int(r[7 + 2 * i]), int(r[8 + 2 * i])] for i in range(len(blocks))]}
                for r in rows]

    # The same loop on a FRESH synthetic batch every step (class-dependent means, as tools/deferred_soak.py): the step
    # counts move with the data, which is what a training run sees; the fixed batch above is the friendliest case for
    # predicting them.
    fresh = None
    if not args.no_fresh:
        gen_d = torch.Generator(device=device).manual_seed(4321 + rank)
        means = torch.randn(10, 3, 1, 1, device=device, generator=gen_d)

        def fresh_batches(k):
            out = []
            for _ in range(k):
                yy = torch.randint(0, 10, (cfg['batch'],), device=device, generator=gen_d)
                out.append((torch.randn(cfg['batch'], 3, cfg['image'], cfg['image'], device=device, generator=gen_d) + means[yy], yy))
            return out

        run_steps(fresh_batches(max(2, args.warmup)))
        fr = timed_region(fresh_batches(args.steps))
        fel = over_ranks(fr['elapsed'], dist.ReduceOp.MAX)
        fresh = {'value': args.steps * cfg['batch'] * world / fel, 'unit': 'images/sec', 'ms_per_step': fel / args.steps * 1e3,
                 'steps': args.steps, 'retries': fr['retries'], 'miss_events': fr['miss_events'],
                 'dead_steps_per_step': fr['dead_steps'] / args.steps,
                 'nfe_forward_per_step': fr['nfe_f'] / args.steps, 'nfe_backward_per_step': fr['nfe_b'] / args.steps,
                 'note': 'a NEW synthetic batch every step (class-dependent means), same model right behind the headline region; '
                         'a missed solve is repeated (retries), never skipped'}
        run_steps([(x, y)] * 3)                 # back on the fixed batch for the regions below
    blocks = list(model.odeblocks) if hasattr(model, 'odeblocks') else [model.odeblock]

    # The same K steps through the DROP-IN behaviour (what the reference's loop gets without opting in to anything):
    # every solve ends with a read-back of the device controller.  Same process, same model, right behind the regions above.
    dropin = None
    if deferred is not None and not args.no_dropin:
        deferred.resolve()
        flag_saved, opt.skip_flag = opt.skip_flag, None
        deferred.armed = False
        for _ in range(2):
            one_step(x, y)                        # untimed: the library's own step-count guesses
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(x, y)
        sync()
        el = over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX)
        dropin = {'value': args.steps * cfg['batch'] * world / el, 'unit': 'images/sec', 'ms_per_step': el / args.steps * 1e3,
                  'steps': args.steps,
                  'note': 'the same steps with a read-back per solve (the drop-in odeint / ODEBlock behaviour; --no-deferred), '
                          'timed in the same run right behind the headline region'}
        opt.skip_flag = flag_saved
        deferred.armed = True

    roofline = None
    if not args.no_roofline:
        # repeat of the timed steps with per-launch HIP events on the launch stream
        # ... with a read-back per solve: under deferred completion a spare step may be enqueued whose kernels return
        # at once, and those empty launches would be averaged into the per-launch duration
        if deferred is not None:
            deferred.resolve()
            opt.skip_flag = None
            deferred.armed = False
        train_step(model, opt, x, y, reducer)     # unprofiled: refreshes the library's own step-count guesses
        integrate.profile_begin()
        for _ in range(min(args.steps, 5)):
            train_step(model, opt, x, y, reducer)
        torch.cuda.synchronize(device)
        prof = integrate.profile_end()
        k = prof['conv3x3_implicit_gemm']
        k4 = prof['w4_component_gemm']
        if k4['launches'] > k['launches']:
            # the convs ran as the F(4x4,3x3) pipeline (dopri5 at rtol, atol >= 1e-5 on 8x8 / 16x16 states): the dominant kernel
            # is the component GEMM; the transforms around it run inside the GroupNorm passes
            avg_ms = k4['total_ms'] / k4['launches']
            algo_per_launch = k4['flops'] / k4['launches']      # direct 3x3 conv: 2*9*C^2*N*H*W
            issued = 36.0 / (16.0 * 9.0)
            ach = algo_per_launch * issued / (avg_ms * 1e-3) / 1e12
            roofline = {'bound': 'mfma', 'kernel': 'k_w4_gemm64 / k_w4_gemm (fp32-MFMA form of the 36 component GEMMs of Winograd F(4x4,3x3), fwd+dgrad: '
                                                    'batches the bf16-triple kernels do not take, NODE_TUNE_W4_BF16X3=0)',
                        'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': ach / MFMA_F32_PEAK_TFLOPS, 'traffic': (pmc_lookup(pmc, 'k_w4_gemm') or {}).get('bytes'),
                        'traffic_detail': pmc_lookup(pmc, 'k_w4_gemm'),
                        'avg_launch_us': avg_ms * 1e3, 'launches': k4['launches'],
                        'flops_per_launch': algo_per_launch * issued,
                        'algorithmic': {'flops_per_launch': algo_per_launch,
                                        'note': 'direct-convolution FLOPs of the conv this launch belongs to (SURVEY.md 8d); its '
                                                'input / output transforms run in the GroupNorm passes around it, so no rate is '
                                                'quoted against this launch alone'},
                        'note': 'achieved/frac = MFMA FLOPs issued (0.25 of the direct-convolution FLOPs) over the fp32 matrix '
                                'peak, i.e. matrix-pipe utilisation of the dominant kernel'}
            quads = 4 if side == 16 else 1     # a 16x16 image runs as four 8x8 quadrants = four "samples" of the GEMMs' layouts
            if os.environ.get('NODE_TUNE_W4_BF16X3', '1') != '0' and (cfg['batch'] * quads) % 16 == 0 and os.environ.get('NODE_TUNE_W4_GEMM64', '1') != '0':
                # k_w4_gemm64b: the same products on the bf16 matrix pipe at fp32 accuracy (every fp32 operand an exact sum
                # of three bf16 parts, six of the nine part products).  Which side bounds it depends on the reduction
                # length C: at cfg 2 14.5 GFLOP issued over 52 MB (V fp32 + the filters' bf16 triples + M fp32) =
                # 279 FLOP/B sits left of the bf16 ridge (2500 TFLOP/s / 8 TB/s = 312 FLOP/B) -> HBM; at cfg 5
                # (C = 1024) 464 GFLOP over 528 MB = 878 FLOP/B -> the matrix pipe.  The other view is reported next to it.
                C, Nn = cfg['filters'], cfg['batch'] * quads
                # round 6: both operands as fp16 PAIRS (x 2^e = h + l, 22 significand bits in the 4 bytes of an fp32; three
                # part products hl, lh, hh per fp32 product, no vector arithmetic in the loop: k_w4_gemm64h / k_w4_gemm128h) --
                # every launch of an adaptive solve but the few of each interval's first augmented evaluation (bf16 triples)
                f16 = (os.environ.get('NODE_TUNE_W4_F16', '1') != '0' and Nn % 16 == 0 and C % 64 == 0 and
                       (C < 512 or (Nn % 32 == 0 and C % 128 == 0 and ((Nn // 32) * (C // 128)) % 2 == 0)))
                nprod = 3.0 if f16 else 6.0
                bytes_algo = 36.0 * 4 * Nn * C * 4 * 2 + 36.0 * C * C * (4 if f16 else 6)
                tbs = bytes_algo / (avg_ms * 1e-3) / 1e12
                flops_bf16 = algo_per_launch * issued * nprod
                issued_bf16 = flops_bf16 / (avg_ms * 1e-3) / 1e12
                lds_tiled = C >= 512 and Nn % 32 == 0 and C % 128 == 0 and os.environ.get('NODE_TUNE_W4_GEMM128', '1') != '0'
                if f16:
                    # long reductions with rows % 256 == 0 and C % 256 == 0: components 0..31 as 256 x 256 workgroup tiles (k_w4_gemm256h),
                    # components 32..35 behind them as k_w4_gemm128h's tail launch -- one ProfScope, so `avg_launch_us` is the PAIR
                    big = (lds_tiled and Nn % 64 == 0 and C % 256 == 0 and os.environ.get('NODE_TUNE_W4_H256', '1') != '0' and
                           (os.environ.get('NODE_TUNE_W4_H256', '1') == '2' or ((Nn // 64) * (C // 256)) % 8 == 0))
                    kname = ('%s (the 36 component GEMMs of Winograd F(4x4,3x3), fwd+dgrad, on fp16 MFMA at fp32 accuracy: both operands '
                             'as scaled fp16 pairs h + l, three products%s)'
                             % (('k_w4_gemm256h + k_w4_gemm128h', '; LDS-tiled: 256x256 workgroup tiles, one wave per SIMD with sixteen accumulators, '
                                 'for components 0..31, 128x128 tiles for 32..35; the duration is that of the pair of launches') if big
                                else ('k_w4_gemm128h', '; LDS-tiled 128x128 workgroup tiles for long reductions') if lds_tiled
                                else ('k_w4_gemm64h', '')))
                    gem_pmc = 'k_w4_gemm256h' if big else 'k_w4_gemm128h' if lds_tiled else 'k_w4_gemm64h'
                else:
                    kname = ('%s (the 36 component GEMMs of Winograd F(4x4,3x3), fwd+dgrad, on bf16 MFMA '
                             'at fp32 accuracy: exact three-way bf16 split of both operands, six products%s)'
                             % (('k_w4_gemm128b', '; LDS-tiled 128x128 workgroup tiles for long reductions') if lds_tiled
                                else ('k_w4_gemm64b', '')))
                    gem_pmc = 'k_w4_gemm128b' if lds_tiled else 'k_w4_gemm64b'
                roofline['traffic'] = (pmc_lookup(pmc, gem_pmc) or {}).get('bytes')
                roofline['traffic_detail'] = pmc_lookup(pmc, gem_pmc)
                mfma_view = {'issued_bf16_tflops': issued_bf16, 'frac_of_bf16_peak': issued_bf16 / MFMA_BF16_PEAK_TFLOPS,
                             'fp32_equivalent_tflops': ach, 'vs_fp32_matrix_peak': ach / MFMA_F32_PEAK_TFLOPS,
                             'part_products_per_fp32_product': nprod,
                             'sustained_on_this_data': 'tools/mfma_power.hip (profiles/r06_mfma_power.txt): the bare v_mfma_f32_32x32x16_f16 loop, sixteen '
                                                       'accumulators per wave, whole chip, on RANDOM fp16 pairs holds 1.46 - 1.60 GHz = 1480 - 1670 TFLOP/s '
                                                       '(power), not the 2500 the fractions here are priced against',
                             'note': 'issued = 16-bit MFMA FLOPs (fp16 and bf16 forms run at the same rate); fp32_equivalent = the '
                                     'component products the fp32 MFMA kernel would issue, over this launch time; against the fp32 '
                                     'matrix peak it may exceed 1 -- the products run at the 16-bit rate'}
                clk = held_clk if not lds_tiled else None
                if clk:
                    mfma_view.update({'held_clock_ghz': clk, 'frac_of_bf16_peak_at_held_clock': issued_bf16 / (MFMA_BF16_PEAK_TFLOPS * clk / 2.4),
                                      'held_clock_note': 'in-kernel clock over the K loop (s_memtime / s_memrealtime, one stamped launch): the '
                                                         'chip lowers its clock under this load; the peaks above are priced at 2.4 GHz'})
                if flops_bf16 / bytes_algo < MFMA_BF16_PEAK_TFLOPS / HBM_PEAK_TBS:
                    roofline.update({
                        'bound': 'hbm', 'kernel': kname,
                        'achieved': tbs, 'peak': HBM_PEAK_TBS, 'unit': 'TB/s', 'frac': tbs / HBM_PEAK_TBS,
                        'bytes_per_launch': bytes_algo, 'mfma': mfma_view,
                        'note': 'achieved = algorithmic bytes per launch (row operand + filters as %s + products fp32) '
                                '/ mean launch duration (HIP events, all component-GEMM launches of the repeated steps); `traffic` = the '
                                'bytes counted by rocprofv3 PMC in this run' % ('fp16 pairs' if f16 else 'fp32 / bf16 triples')})
                else:
                    roofline.update({
                        'bound': 'mfma', 'kernel': kname,
                        'achieved': issued_bf16, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': issued_bf16 / MFMA_BF16_PEAK_TFLOPS,
                        'flops_per_launch': flops_bf16, 'bytes_per_launch': bytes_algo, 'mfma': mfma_view,
                        'hbm_side': {'achieved': tbs, 'unit': 'TB/s', 'frac': tbs / HBM_PEAK_TBS},
                        'note': 'achieved = 16-bit MFMA FLOPs issued per launch (%d part products per fp32 product, 0.25 of the '
                                'direct-convolution FLOPs each) / mean launch duration (HIP events) over the dense bf16 / fp16 matrix '
                                'peak; %.0f FLOP/B sits right of the ridge' % (int(nprod), flops_bf16 / bytes_algo)})
        elif k['launches'] > 0:
            avg_ms = k['total_ms'] / k['launches']
            algo_per_launch = k['flops'] / k['launches']      # direct 3x3 conv: 2*9*C^2*N*H*W
            wino = os.environ.get('NODE_TUNE_CONV_WINO', '2')
            issued = {'2': 16.0 / 36.0, '1': 2.0 / 3.0}.get(wino, 1.0)
            kname = {'2': 'k_conv3x3_w2 (fp32 MFMA, 2-D Winograd F(2x2,3x3), fwd+dgrad)',
                     '1': 'k_conv3x3_w (fp32 MFMA, 1-D Winograd F(2,3), fwd+dgrad)'}.get(wino, 'k_conv3x3 (fp32 MFMA implicit GEMM, fwd+dgrad)')
            algo = algo_per_launch / (avg_ms * 1e-3) / 1e12
            ach = algo * issued
            roofline = {'bound': 'mfma', 'kernel': kname,
                        'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': ach / MFMA_F32_PEAK_TFLOPS, 'traffic': (pmc_lookup(pmc, 'k_conv3x3') or {}).get('bytes'),
                        'traffic_detail': pmc_lookup(pmc, 'k_conv3x3'),
                        'avg_launch_us': avg_ms * 1e3, 'launches': k['launches'],
                        'flops_per_launch': algo_per_launch * issued,
                        'algorithmic': {'flops_per_launch': algo_per_launch, 'achieved': algo,
                                        'speedup_vs_peak': algo / MFMA_F32_PEAK_TFLOPS},
                        'note': 'achieved/frac = MFMA FLOPs issued (%.3f of the direct-convolution FLOPs: Winograd) over the '
                                'fp32 matrix peak, i.e. matrix-pipe utilisation; `algorithmic` counts direct-convolution '
                                'FLOPs (SURVEY.md 8d) and is not a utilisation' % issued}
        w = prof['wgrad_gemm']
        if roofline is not None and w['launches'] > 0:      # (k_wgrad_w2: F(2x2,3x3) domain on either conv path)
            wavg = w['total_ms'] / w['launches']
            walgo = w['flops'] / w['launches'] / (wavg * 1e-3) / 1e12
            wissued = {'2': 16.0 / 36.0, '1': 2.0 / 3.0}.get(os.environ.get('NODE_TUNE_WGRAD_WINO', '2'), 1.0)
            wname = 'k_wgrad_w2 (F(2x2,3x3) domain, split-K slabs)'
            if k4['launches'] > k['launches'] and cfg['filters'] % 128 == 0:
                # behind the F(4x4,3x3) pipeline the weight gradient runs in that domain too (k_w4_wgrad, both layers per launch)
                wissued, wname = 36.0 / 144.0, 'k_w4_wgrad (F(4x4,3x3) domain, both conv layers per launch, no split-K slabs)'
            roofline['wgrad'] = {'kernel': wname, 'achieved': walgo * wissued, 'frac': walgo * wissued / MFMA_F32_PEAK_TFLOPS,
                                 'algorithmic': walgo, 'avg_launch_us': wavg * 1e3, 'launches': w['launches']}
            quads_w = 4 if side == 16 else 1
            if (wname.startswith('k_w4_wgrad') and os.environ.get('NODE_TUNE_W4_F16', '1') != '0' and (cfg['batch'] * quads_w) % 16 == 0 and
                    args.method == 'dopri5'):
                # round 6: V pairs x Z pairs, three fp16 MFMA products per fp32 product (k_w4_wgrad64h), every launch of an adaptive
                # solve but the one of each interval's first evaluation
                roofline['wgrad'].update({
                    'kernel': 'k_w4_wgrad64h (F(4x4,3x3) domain, both conv layers per launch, fp16 pairs, LDS-DMA ring + transposed LDS reads)',
                    'achieved': walgo * wissued * 3.0, 'frac': walgo * wissued * 3.0 / MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s of fp16 MFMA issued',
                    'fp32_equivalent_tflops': walgo * wissued,
                    'traffic': (pmc_lookup(pmc, 'k_w4_wgrad64h') or {}).get('bytes')})
            elif wname.startswith('k_w4_wgrad') and cfg['filters'] >= 512 and os.environ.get('NODE_TUNE_W4_WGRAD128', '1') != '0':
                # long filters: the same sums on bf16 triples, LDS-tiled (k_w4_wgrad128b): six bf16 MFMA products per fp32 product
                roofline['wgrad'].update({
                    'kernel': 'k_w4_wgrad128b (F(4x4,3x3) domain, both conv layers per launch, bf16 MFMA at fp32 accuracy, LDS-tiled)',
                    'achieved': walgo * wissued * 6.0, 'frac': walgo * wissued * 6.0 / MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s of bf16 MFMA issued',
                    'fp32_equivalent_tflops': walgo * wissued})
        # the HBM-bound side ("report both fractions separately"): the GroupNorm / transform passes of the
        # F(4x4,3x3) pipeline, per kernel instance, ALGORITHMIC bytes (every tensor a pass must read or write, once)
        # over the mean launch duration between HIP events
        passes = {n: v for n, v in prof.items() if n.startswith('w4s_pass') and v['launches'] > 0}
        if roofline is not None and passes:
            def hb(v):
                us = v['total_ms'] / v['launches'] * 1e3
                tbs = v['flops'] / v['launches'] / (us * 1e-6) / 1e12
                return {'bytes_per_launch': v['flops'] / v['launches'], 'avg_launch_us': us, 'launches': v['launches'],
                        'achieved': tbs, 'frac': tbs / HBM_PEAK_TBS}
            dom = max(passes, key=lambda n: passes[n]['total_ms'])
            tot = {'launches': sum(v['launches'] for v in passes.values()), 'total_ms': sum(v['total_ms'] for v in passes.values()),
                   'flops': sum(v['flops'] for v in passes.values())}
            # 'w4s_pass<1,2> ...' -> '<1, 2, ' as rocprofv3 prints the instance (third argument: 1 = 8x8 images, 4 = 16x16)
            inst = dom[dom.index('<'):dom.index('>')].replace(',', ', ') + ', '
            roofline['hbm'] = dict(hb(passes[dom]), bound='hbm', kernel='k_' + dom, peak=HBM_PEAK_TBS, unit='TB/s',
                                   traffic=(pmc_lookup(pmc, 'k_w4s_pass' + inst) or {}).get('bytes'),
                                   all_passes=dict(hb(tot), ms_per_step=tot['total_ms'] / min(args.steps, 5)),
                                   per_kernel={n: hb(v) for n, v in passes.items()},
                                   note='algorithmic bytes per launch / mean launch duration (HIP events on the launch stream); '
                                        'peak = HBM3E spec, a streaming copy reaches 6.3 TB/s on this part')

    if rank == 0:
        global_batch = cfg['batch'] * world
        result = {
            'metric': 'images/sec (fwd+adjoint) CIFAR-10 ODE-ResNet bs=128 at 1/2/4/8 GPU',
            'value': args.steps * global_batch / elapsed,
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'step_ms': {'median': statistics.median(per_step), 'min': min(per_step), 'max': max(per_step),
                        'all': [round(v, 3) for v in per_step],
                        'source': 'HIP events on the compute stream of rank 0, one per step'},
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'dtype_note': 'fp32 tensors end to end; the 3x3 convolutions run as Winograd F(4x4,3x3) with every fp32 operand of the component '
                          'GEMMs held as a scaled fp16 pair h + l (22 significand bits, 3 MFMA products per fp32 product): fp32-accuracy '
                          'emulation on the 16-bit matrix pipe, 3e-6 of max|y| per convolution against fp64 (the fp32 chain: 3e-6)',
            'data': 'synthetic',
            'config': {
                'workload': '%s, bs=%d per GPU, SGD step (BASELINE.json configs[%d])' % (cfg['name'], cfg['batch'], args.config - 1),
                'global_batch': global_batch, 'state': state, 'ode_blocks': cfg['blocks'],
                'settle_steps': settle_steps,     # untimed steps beyond `warmup` until no solve enqueued a spare step any more
                'solver_completion': 'read-back per solve' if deferred is None else
                                     'deferred (sticky device flag predicates the optimizer step; a missed solve voids the '
                                     'iteration and it is REPEATED, never skipped): %d blind solves, %d miss events, %d batches '
                                     'repeated in the timed region' % (region['blind_solves'], region['miss_events'], region['retries']),
                'retries': region['retries'], 'miss_events': region['miss_events'],
                # steps enqueued past the end of their interval (every kernel of such a step returns at its first
                # instruction): forward ones are ~28 launches each, augmented ones ~60
                'dead_steps_per_step': region['dead_steps'] / args.steps,
                'collectives': (args.dist_backend + (' (RCCL)' if args.dist_backend == 'nccl' else '')) if dist_on else 'none (one rank)',
                'step_norm': 'global (sums of every step decision all-reduced: identical steps on all ranks)' if (args.global_norm and world > 1) else 'local (each rank adapts its steps on its shard)',
                'parallelism': 'dp%d' % world if not args.share_gpu else 'dp%d (ranks SHARE a GPU: smoke test, not a measurement)' % world, 'head': 'hipGraph' if args.graphs else 'eager',
                'nfe_forward_per_step': nfe_f / args.steps, 'nfe_backward_per_step': nfe_b / args.steps,
                'per_rank': per_rank,
            },
        }
        if fresh is not None:
            result['fresh_batches'] = fresh
        if dropin is not None:
            result['dropin'] = dropin
        if roofline is not None:
            result['roofline'] = roofline
        if others is not None:
            result['other_configs'] = others
        if world == 1 and not args.no_latency:
            result['latency_bs1'] = latency_bs1(cfg, args.method)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(init_state, cfg, args.method)
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
