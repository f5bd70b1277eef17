"""CPU-side checks: the host logic of the boundary, fail-loudly behaviour without a
GPU, and that libnode_hip.so loads and exports every symbol include/node_hip.h declares
(no compute calls here -- there is no GPU in this container)."""
import ctypes as C
import os
import re
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from neural_ode_features_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'node_hip.h')).read()
    declared = set(re.findall(r'\b(node_[a-z_0-9]+)\s*\(', header))
    declared -= {'node_shape', 'node_params', 'node_stats', 'node_solve_opts', 'node_profile'}
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.node_abi_version() == _lib.NODE_ABI_VERSION == 6
    m = re.search(r'#define NODE_ABI_VERSION (\d+)', header)
    assert int(m.group(1)) == _lib.NODE_ABI_VERSION


def test_stem_entry_points_host_side():
    """node_stem_workspace_bytes / argument checks of the stem's C ABI need no GPU: sizes for the BASELINE shapes, refusals
    with a message for what the kernels do not take."""
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    cifar = lib.node_stem_workspace_bytes(C.byref(_lib.NodeStemShape(128, 3, 32, 32, 256, 1e-5)))
    mnist = lib.node_stem_workspace_bytes(C.byref(_lib.NodeStemShape(32, 1, 28, 28, 64, 1e-5)))
    assert 0 < mnist < cifar < (2 << 30)
    for bad, word in ((_lib.NodeStemShape(8, 4, 32, 32, 64, 1e-5), 'in_ch'), (_lib.NodeStemShape(8, 3, 32, 32, 24, 1e-5), 'filters'),
                      (_lib.NodeStemShape(8, 3, 64, 64, 64, 1e-5), 'LDS'), (_lib.NodeStemShape(0, 3, 32, 32, 64, 1e-5), 'shape')):
        assert lib.node_stem_workspace_bytes(C.byref(bad)) == 0
        assert word in lib.node_last_error().decode(), (word, lib.node_last_error())
    shape = _lib.NodeStemShape(8, 3, 32, 32, 64, 1e-5)
    assert lib.node_stem_fwd(C.byref(shape), None, None, None, None, 0, None) == -1          # NODE_ERR_NULL
    geom = _lib.NodeConvGeom(8, 64, 64, 30, 30, 3, 2, 1)
    assert lib.node_stem_conv_workspace_bytes(C.byref(geom)) > 0
    assert lib.node_stem_conv_workspace_bytes(C.byref(_lib.NodeConvGeom(8, 48, 64, 30, 30, 3, 2, 1))) == 0


def test_host_only_entry_points():
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    shape = _lib.NodeShape(128, 256, 8, 8, 32, 1e-5)
    assert lib.node_param_count(C.byref(shape)) == 1186304          # SURVEY.md section 8 (C=256)
    fwd = lib.node_workspace_bytes(C.byref(shape), 0, 0, 2)
    adj = lib.node_workspace_bytes(C.byref(shape), 0, 1, 2)
    assert 0 < fwd < adj < (2 << 30)
    assert lib.node_param_count(C.byref(_lib.NodeShape(1, 64, 7, 7, 32, 1e-5))) == 75392
    # unsupported / inconsistent shapes are refused with a message, never a crash
    for bad in [_lib.NodeShape(4, 30, 8, 8, 30, 1e-5), _lib.NodeShape(4, 64, 8, 8, 7, 1e-5),
                _lib.NodeShape(4, 64, 40, 40, 32, 1e-5), _lib.NodeShape(0, 64, 8, 8, 32, 1e-5)]:
        assert lib.node_workspace_bytes(C.byref(bad), 0, 0, 2) == 0
        assert len(lib.node_last_error()) > 0


def test_new_entry_points_check_their_arguments_without_a_gpu():
    """ABI 4: the head's Linear + loss launches and the generic (flat) solver refuse NULL / inconsistent arguments with a
    message before any HIP call; the diagnostics library is a separate file the product binding never loads by itself."""
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    assert lib.node_head_loss_scratch_bytes(128) > 0 and lib.node_head_loss_scratch_bytes(0) == 0
    assert lib.node_head_loss_fwd(None, None) == -1
    h = _lib.NodeHeadLoss(128, 256, 10, 0, None, None, None, None, None, None, None, None)
    assert lib.node_head_loss_fwd(C.byref(h), None) == -1 and b'logits' in lib.node_last_error()
    h = _lib.NodeHeadLoss(128, 256, 4096, 0, None, None, None, None, 256, None, None, None)
    assert lib.node_head_loss_fwd(C.byref(h), None) == -3                       # more than 1024 classes
    h = _lib.NodeHeadLoss(128, 256, 10, 0, None, None, None, None, 256, None, None, None)
    assert lib.node_head_loss_fwd(C.byref(h), None) == -9                       # neither pooled nor target: nothing to do
    assert lib.node_head_loss_bwd(C.byref(h), None, None) == -1
    assert lib.node_flat_workspace_bytes(1) > 0 and lib.node_flat_workspace_bytes(50) > lib.node_flat_workspace_bytes(1)
    f = _lib.NodeFlatSolve()
    f.nseg, f.n_targets = 0, 1
    assert lib.node_flat_finish_step(C.byref(f), 0, None, None) == -9 and b'nseg' in lib.node_last_error()
    f.nseg = 1
    assert lib.node_flat_finish_step(C.byref(f), 0, None, None) == -9 and b'workspace' in lib.node_last_error()
    assert os.path.basename(_lib.LIB_PATH) == 'libnode_hip.so' and os.path.basename(_lib.LIB_DIAG_PATH) == 'libnode_hip_diag.so'


def test_error_paths_of_the_c_abi_without_a_gpu():
    """NULL / misaligned / undersized arguments are rejected before any HIP call."""
    from neural_ode_features_amd import _lib
    lib = _lib.load()
    shape = _lib.NodeShape(2, 8, 4, 4, 8, 1e-5)
    params = _lib.NodeParams(*([0] * 10))
    stats = _lib.NodeStats()
    tarr = (C.c_float * 2)(0.0, 1.0)
    rc = lib.node_solve_fwd(C.byref(shape), C.byref(params), None, tarr, 2, 1e-3, 1e-3, 0, None, None,
                            C.byref(stats), None, 0, None)
    assert rc == -1 and b'NULL' in lib.node_last_error()
    bad_t = (C.c_float * 3)(0.0, 1.0, 0.5)
    rc = lib.node_solve_fwd(C.byref(shape), C.byref(params), 256, bad_t, 3, 1e-3, 1e-3, 0, None, 256,
                            C.byref(stats), 256, 1 << 20, None)
    assert rc == -9                                                    # non-monotonic time grid
    rc = lib.node_solve_fwd(C.byref(shape), C.byref(params), 256, tarr, 2, 1e-3, 1e-3, 7, None, 256,
                            C.byref(stats), 256, 1 << 20, None)
    assert rc == -9                                                    # unknown method
    rc = lib.node_solve_fwd(C.byref(shape), C.byref(params), 256, tarr, 2, 1e-3, 1e-3, 0, None, 256,
                            C.byref(stats), 256, 1 << 20, None)
    assert rc == -1                                                    # NULL parameter pointers
    good = _lib.NodeParams(*([4096] * 10))
    rc = lib.node_solve_fwd(C.byref(shape), C.byref(good), 256, tarr, 2, 1e-3, 1e-3, 0, None, 256,
                            C.byref(stats), 256, 16, None)
    assert rc == -4 and b'workspace too small' in lib.node_last_error()
    mis = _lib.NodeParams(*([4100] * 10))
    rc = lib.node_solve_fwd(C.byref(shape), C.byref(mis), 256, tarr, 2, 1e-3, 1e-3, 0, None, 256,
                            C.byref(stats), 256, 1 << 20, None)
    assert rc == -9


def test_product_path_has_no_cpu_fallback():
    import neural_ode_features_amd as nof
    blk = nof.ODEBlock(n_filters=8, adjoint=True)
    with pytest.raises(RuntimeError, match='no CPU path'):
        blk(torch.randn(2, 8, 4, 4))
    with pytest.raises(RuntimeError, match='no CPU path'):
        nof.odeint(blk.odefunc, torch.randn(2, 8, 4, 4), torch.tensor([0.0, 1.0]))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from neural_ode_features_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='not built'):
        _lib.load()


def test_argument_validation_mirrors_torchdiffeq():
    from neural_ode_features_amd import integrate
    import neural_ode_features_amd as nof
    f = nof.ODEfunc(8)
    rec = integrate.Recognised(f)
    assert rec.dim == 8 and rec.groups == 8 and len(rec.params) == 10
    assert [n for n, _ in f.named_parameters()] == [
        'norm1.weight', 'norm1.bias', 'conv1._layer.weight', 'conv1._layer.bias',
        'norm2.weight', 'norm2.bias', 'conv2._layer.weight', 'conv2._layer.bias', 'norm3.weight', 'norm3.bias']
    assert integrate._method_id(None) == 0 and integrate._method_id('dopri5') == 0 and integrate._method_id('rk4') == 1
    with pytest.raises(NotImplementedError):
        integrate._method_id('adams')                 # train.py:219 offers it; not implemented
    with pytest.raises(TypeError):
        integrate._host_times(torch.tensor([0, 1]))
    with pytest.raises(ValueError):
        integrate._host_times(torch.tensor([0.0]))
    assert integrate._host_times(torch.tensor([0.0, 0.5, 1.0])) == [0.0, 0.5, 1.0]
    with pytest.raises(NotImplementedError):
        integrate.Recognised(torch.nn.Linear(3, 3))   # not the fused kernels' dynamics (they run the generic solver, generic.py)
    bn = nof.ODEfunc(8, norm='batch')                 # model.py:274: BatchNorm2d without running statistics
    assert isinstance(bn.norm1, torch.nn.BatchNorm2d) and bn.norm1.track_running_stats is False
    with pytest.raises(NotImplementedError):
        integrate.Recognised(bn)
    with pytest.raises(NotImplementedError):
        nof.ODEfunc(8, norm='layer')
    with pytest.raises(ValueError):
        integrate._odeint_impl(lambda t, y: y, _FakeCuda(), torch.tensor([0.0, 1.0]), 1e-3, 1e-3, None, None)


class _FakeCuda(torch.Tensor):
    """A CPU tensor that claims to be a float32 CUDA tensor: lets the host-side checks past
    `_check_state` be exercised without a GPU."""
    @staticmethod
    def __new__(cls):
        return torch.Tensor._make_subclass(cls, torch.zeros(1, 8, 2, 2))

    @property
    def is_cuda(self):
        return True


def test_odeblock_attribute_surface():
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    blk = nof.ODEBlock(n_filters=16, tol=1e-2, method='rk4', adjoint=True, t1=[0.1, 0.2, 1])
    assert blk.odeint is integrate.odeint_adjoint and blk.return_last_only is True
    assert blk.tol == 1e-2 and blk.method == 'rk4'
    assert nof.ODEBlock(n_filters=16).odeint is integrate.odeint
    blk.nfe = 5
    assert blk.odefunc.nfe == 5 and blk.nfe == 5
    net = nof.ODENet(3, n_filters=16, adjoint=True)
    net.odeblock.nfe = 7
    assert net.nfe(reset=True) == 7 and net.nfe() == 0
    net.to_features_extractor()
    assert net.odeblock.return_last_only is False


@pytest.mark.skipif(not os.path.exists('/root/reference/model.py'), reason='reference checkout not present')
def test_reference_model_py_runs_on_top_of_the_package_unchanged():
    """Drop-in: alias the package as `torchdiffeq`, import the reference's model.py as is;
    its ODEBlock then calls OUR odeint with ITS OWN ODEfunc, which is recognised."""
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    saved = sys.modules.get('torchdiffeq')
    sys.modules['torchdiffeq'] = nof
    sys.path.insert(0, '/root/reference')
    try:
        sys.modules.pop('model', None)
        import model as ref_model
        net = ref_model.ODENet(3, n_filters=16, adjoint=True, tol=1e-3)
        assert net.odeblock.odeint is integrate.odeint_adjoint
        rec = integrate.Recognised(net.odeblock.odefunc)       # the reference's own class
        assert rec.dim == 16 and rec.groups == 16
        mine = nof.ODENet(3, n_filters=16, adjoint=True, tol=1e-3)
        assert list(mine.state_dict().keys()) == list(net.state_dict().keys())
        mine.load_state_dict(net.state_dict())
        with pytest.raises(RuntimeError, match='no CPU path'):  # reaches our boundary; no GPU here
            net(torch.randn(2, 3, 32, 32))
    finally:
        sys.path.remove('/root/reference')
        sys.modules.pop('model', None)
        if saved is not None:
            sys.modules['torchdiffeq'] = saved
        else:
            sys.modules.pop('torchdiffeq', None)


def test_header_is_plain_c_and_ctypes_layouts_match(tmp_path):
    """include/node_hip.h must compile as C (it is the drop-in boundary: no C++ / torch types), and every struct the
    ctypes binding mirrors must have the size and field offsets the C compiler gives it."""
    import ctypes as C
    import subprocess
    from neural_ode_features_amd import _lib
    structs = {
        'node_shape': (_lib.NodeShape, ['n', 'c', 'h', 'w', 'groups', 'eps']),
        'node_params': (_lib.NodeParams, ['norm1_w', 'conv1_w', 'norm3_b']),
        'node_stats': (_lib.NodeStats, ['nfe', 'status', 'last_dt', 't_final', 'first_dt']),
        'node_solve_opts': (_lib.NodeSolveOpts, ['max_num_steps', 'n_forced_dt', 'forced_dt', 'record_dt', 'dt_log', 'n_dt_log',
                                                 'blind_steps', 'record', 'miss_flag', 'grad_last_only']),
        'node_step_record': (_lib.NodeStepRecord, ['done', 'status', 'steps', 'accepted', 'rejected', 'miss', 't', 'dt', 'first_dt']),
        'node_sgd_tensor': (_lib.NodeSgdTensor, ['param', 'grad', 'momentum_buf', 'n']),
        'node_profile': (_lib.NodeProfile, ['launches', 'total_ms', 'flops']),
        'node_head_loss': (_lib.NodeHeadLoss, ['n', 'c', 'classes', 'reduction', 'pooled', 'weight', 'bias', 'target', 'logits', 'loss',
                                               'stat', 'scratch']),
        'node_head_loss_grad': (_lib.NodeHeadLossGrad, ['grad_loss', 'grad_logits', 'd_logits', 'd_pooled', 'd_weight', 'd_bias']),
        'node_flat_seg': (_lib.NodeFlatSeg, ['y', 'y1', 'k', 'n']),
        'node_flat_solve': (_lib.NodeFlatSolve, ['nseg', 'has_scalar', 'seg', 'rtol', 'atol', 'tsign', 'n_targets', 'ws', 'ws_bytes']),
        'node_flat_status': (_lib.NodeFlatStatus, ['done', 'status', 'steps', 'accepted', 'rejected', 't', 'dt', 'first_dt', 'scalar']),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "node_hip.h"', 'int main(void) {']
    for name, (_, fields) in structs.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines += ['  return 0;', '}']
    src = tmp_path / 'abi.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'abi'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, (ct, fields) in structs.items():
        assert int(got[name]) == C.sizeof(ct), name
        for f in fields:
            assert int(got['%s.%s' % (name, f)]) == getattr(ct, f).offset, (name, f)


def test_optimizer_and_training_loop_refuse_the_cpu():
    import neural_ode_features_amd as nof
    lin = torch.nn.Linear(3, 2)
    opt = nof.FusedSGD(lin.parameters(), lr=0.1, momentum=0.9)
    lin(torch.ones(1, 3)).sum().backward()
    with pytest.raises(RuntimeError, match='no CPU path'):
        opt.step()
    # torch.optim.SGD's state layout / scheduler surface without touching a device
    assert opt.state_dict()['param_groups'][0]['momentum'] == 0.9
    torch.optim.lr_scheduler.CosineAnnealingLR(opt, 4)
    if not torch.cuda.is_available():
        from neural_ode_features_amd import train as T
        with pytest.raises(SystemExit):
            T.main(['--dataset', 'mnist', '-e', '1', '--run-dir', '/tmp/never_created_run_dir_node'])


def test_asm_load_guard_flags_a_register_touched_in_flight(tmp_path):
    """tools/check_asm_loads.py (run by build.py after a relink): a planted move out of a register whose load has not been waited
    for is reported; the same stream with the wait in front is clean; lanes of the other side of a divergent if are exempt."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_asm_loads', os.path.join(ROOT, 'tools', 'check_asm_loads.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def stream(body):
        text = ''
        for k in mod.KERNELS:
            n = len(k)
            text += '_ZN4node%d%sEv:\n%s\ts_endpgm\n.Lfunc_end_%s:\n' % (n, k, body, k)
        p = tmp_path / 'k.s'
        p.write_text(text)
        return mod.check(str(p))

    load = '\tglobal_load_dwordx4 v[8:11], v[4:5], off\n'
    assert stream(load + '\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v1, v9\n') == []
    bad = stream(load + '\tv_mov_b32 v1, v9\n\ts_waitcnt vmcnt(0)\n')
    nk = len(mod.KERNELS)
    assert len(bad) == nk and 'v9' in bad[0]
    assert len(stream(load + load.replace('v[8:11]', 'v[12:15]') + '\ts_waitcnt vmcnt(1)\n\tv_add_f32 v0, v8, v8\n\tv_add_f32 v0, v12, v0\n')) == nk
    # scratch traffic is a violation by itself except in the kernels listed as spilling outside their loop (SCRATCH_OK) -- and there, too,
    # when it names a register whose request is in flight
    assert len(stream(load + '\tscratch_store_dword off, v1, off\n\ts_waitcnt vmcnt(0)\n')) == nk - len(mod.SCRATCH_OK)
    touched = stream(load + '\tscratch_store_dword off, v9, off\n\ts_waitcnt vmcnt(0)\n')
    assert all(any(k in line and 'touches v9' in line for line in touched) for k in mod.KERNELS)
    other = '\ts_and_saveexec_b64 s[6:7], s[0:1]\n' + load + '\ts_andn2_saveexec_b64 s[6:7], s[6:7]\n' + load + '\ts_or_b64 exec, exec, s[6:7]\n'
    assert stream(other + '\ts_waitcnt vmcnt(0)\n') == []
    assert len(stream(other + '\tv_mov_b32 v1, v8\n')) == nk
    assert len(stream('\tv_mov_b32 v1, v8\n')) == nk         # no load at all: the kernel changed under the guard


def test_loss_helpers_route_cpu_tensors_to_pytorch():
    """`nof.linear` / `nof.cross_entropy` / `nof.linear_cross_entropy` are drop-ins for F.linear / F.cross_entropy: CPU tensors
    (the CPU baseline of bench.py, the gloo tests) take PyTorch's operators, values and gradients unchanged."""
    import torch.nn.functional as F
    import neural_ode_features_amd as nof
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(5, 16, generator=gen, requires_grad=True)
    w = torch.randn(10, 16, generator=gen, requires_grad=True)
    b = torch.randn(10, generator=gen, requires_grad=True)
    y = torch.randint(0, 10, (5,), generator=gen)
    p = nof.linear(x, w, b)
    assert torch.equal(p, F.linear(x, w, b))
    loss = nof.cross_entropy(p, y)
    assert torch.equal(loss, F.cross_entropy(F.linear(x, w, b), y)) and not hasattr(loss, 'node_stat')
    loss2, logits = nof.linear_cross_entropy(x, w, b, y, reduction='sum')
    assert torch.equal(loss2, F.cross_entropy(F.linear(x, w, b), y, reduction='sum')) and torch.equal(logits, p)
    loss.backward()
    assert x.grad is not None and w.grad is not None and b.grad is not None


def test_recognised_is_remembered_per_object_and_follows_replacements():
    """`integrate.recognised(func)` skips the module-tree walk of `Recognised(func)` for a func it has seen -- as long as the five
    sub-modules, the conv layers and the ten parameter OBJECTS are the ones it saw (an optimizer step or `load_state_dict` keeps
    them; assigning a new layer or parameter does not)."""
    import torch
    import neural_ode_features_amd as nof
    from neural_ode_features_amd import integrate
    f = nof.ODEfunc(64)
    r1 = integrate.recognised(f)
    assert integrate.recognised(f) is r1
    f.load_state_dict(f.state_dict())
    with torch.no_grad():
        f.norm2.weight.mul_(2.0)
    assert integrate.recognised(f) is r1 and r1.params[4] is f.norm2.weight
    f.norm1 = torch.nn.GroupNorm(32, 64)
    r2 = integrate.recognised(f)
    assert r2 is not r1 and r2.params[0] is f.norm1.weight
    f.conv1._layer.weight = torch.nn.Parameter(torch.zeros_like(f.conv1._layer.weight))
    r3 = integrate.recognised(f)
    assert r3 is not r2 and r3.params[2] is f.conv1._layer.weight
    f.extra = torch.nn.Parameter(torch.zeros(1))          # an eleventh parameter: no longer the reference's ODEfunc
    with pytest.raises(NotImplementedError):
        integrate.recognised(f)
    with pytest.raises(NotImplementedError):
        integrate.recognised(nof.ODEfunc(64, norm='batch'))
