#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

The reference's ``model.py`` is imported unchanged.  Its single third-party
import (``from torchdiffeq import odeint_adjoint, odeint``, model.py:3) cannot
resolve -- torchdiffeq is an empty submodule -- so a stub module carrying the
oracle restatement (oracle/torchdiffeq_restated.py) is registered under that
name first.  What the fixtures pin:

  odefunc_c{8,16,64}.pt   reference ``ODEfunc`` (model.py:326-348): inputs, parameters,
                          f(t, y) and autograd VJPs (d/dy, d/dt, d/dtheta) for a fixed cotangent
  odefunc_c64_n8.pt,      the same at N = 8, 8x8: the smallest shape the Winograd F(4x4,3x3) pipeline takes (round 3:
  odefunc_c64_n8_kf.pt    ``python tests/golden/make_golden.py w4``); _kf = GroupNorm biases in front of the ReLUs at +8
  odenet_t0.pt            reference ``ODENet(..., t1=0)`` logits (stem + head, no solver; model.py:27-46,363-364)
  odeblock_t1_cases.json  reference ``ODEBlock.t1`` setter semantics (model.py:380-403)
  odenet_rk4.pt /         reference ``ODENet`` driven end-to-end (forward, CE loss, backward) with the
  odenet_dopri5.pt        oracle standing in for torchdiffeq: logits, loss, NFE-F/NFE-B, selected grads

  odenet_ode_features.pt  reference ``ODENet(3, downsample='ode', t1=[.1,.2,.3,1])`` in feature-extractor mode -- the
                          reference's own smoke (model.py:416-421): a second ODE block inside the stem
                          (``ODEDownsample``, model.py:181-196), both trajectories pooled and concatenated
  odenet_ode2_train.pt    reference ``ODENet(3, downsample='ode2')`` (``ODEDownsample2``, model.py:199-223) trained one
                          step end-to-end (two adjoint solves per backward), oracle standing in for torchdiffeq

  stem_residual_c64.pt    reference ``ResDownsample(1, 64)`` (model.py:167-178: Conv2d(1, 64, 3, 1) + two stride-2 ``ResBlock``s,
                          model.py:284-310) on a [2, 1, 28, 28] input: output and the gradient of every parameter for a fixed
                          cotangent (round 4: ``python tests/golden/make_golden.py stem``) -- what the library's own stem
                          kernels (csrc/kernels_stem.hip) are held against

  odenet_rk4_f64.pt       reference ``ODENet(1, n_filters=64, downsample='residual', method='rk4', adjoint=True)`` -- the MODEL of
                          BASELINE.json configs[0] (reproduce.sh:9; MNIST-shaped input, state [n, 64, 7, 7], one rk4 3/8 step) --
                          trained one step end to end at batch 4, dropout off (round 5: ``python tests/golden/make_golden.py cfg1``):
                          at 64 filters the library's own stem kernels take the stem (the 8-filter fixtures run the module
                          sequence), so this is the end-to-end fixture that holds them, the ODE block and the head together

Fixtures are data (tensors / json).  No reference source text is stored.
"""
import json
import os
import sys
import types

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import torchdiffeq_restated as tdq  # noqa: E402

REF = '/root/reference'


def import_reference_model():
    stub = types.ModuleType('torchdiffeq')
    stub.odeint = tdq.odeint
    stub.odeint_adjoint = tdq.odeint_adjoint
    sys.modules['torchdiffeq'] = stub
    sys.path.insert(0, REF)
    import model as ref_model  # noqa
    sys.path.remove(REF)
    return ref_model


def randomize_(module, gen):
    """Make every parameter non-trivial (GroupNorm defaults to w=1, b=0)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.25 * torch.randn(p.shape, generator=gen))
            elif 'norm' in name and name.endswith('bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))


def make_odefunc(ref, C, N, H, W, seed, name=None, kink_free=False):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    func = ref.ODEfunc(C)
    randomize_(func, gen)
    if kink_free:   # GroupNorm biases in front of the two ReLUs at +8: no pre-activation anywhere near a kink
        with torch.no_grad():
            for pname, p in func.named_parameters():
                if pname.startswith(('norm1', 'norm2')) and pname.endswith('bias'):
                    p.add_(8.0)
    y = torch.randn(N, C, H, W, generator=gen)
    cot = torch.randn(N, C, H, W, generator=gen)
    t = torch.tensor(0.37)
    tt = t.clone().requires_grad_(True)
    yy = y.clone().requires_grad_(True)
    params = tuple(func.parameters())
    f = func(tt, yy)
    grads = torch.autograd.grad(f, (tt, yy) + params, cot)
    out = {
        'C': C, 'N': N, 'H': H, 'W': W, 't': t, 'y': y, 'cotangent': cot,
        'state_dict': {k: v.clone() for k, v in func.state_dict().items()},
        'param_names': [n for n, _ in func.named_parameters()],
        'f': f.detach(), 'vjp_t': grads[0], 'vjp_y': grads[1],
        'vjp_params': torch.cat([g.reshape(-1) for g in grads[2:]]),
    }
    torch.save(out, os.path.join(HERE, name or 'odefunc_c%d.pt' % C))
    print('odefunc C=%d: |f|=%.4f |vjp_y|=%.4f vjp_t=%.5f P=%d' % (
        C, f.norm(), grads[1].norm(), grads[0], out['vjp_params'].numel()))


def make_odenet_t0(ref):
    torch.manual_seed(23)  # train.py:224,227
    net = ref.ODENet(1, out=10, n_filters=8, downsample='residual', t1=0)
    net.eval()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 28, 28, generator=gen)
    with torch.no_grad():
        logits = net(x)
    torch.save({'x': x, 'logits': logits, 'state_dict': net.state_dict(),
                'keys': list(net.state_dict().keys())}, os.path.join(HERE, 'odenet_t0.pt'))
    print('odenet_t0 logits', logits.shape, float(logits.abs().mean()))


def make_t1_cases(ref):
    cases = []
    for value in [1, 0, 0.5, [0.1, 0.2, 1], (0, 0.25, 0.5), [0, 1], torch.tensor([0.3, 0.6]), 2.0]:
        blk = ref.ODEBlock(n_filters=8, t1=1)
        import io
        import contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):  # the setter prints value[0] when it prepends 0 (model.py:398)
            blk.t1 = value
        it = blk.integration_time
        cases.append({
            'value': value.tolist() if torch.is_tensor(value) else (list(value) if isinstance(value, tuple) else value),
            'kind': type(value).__name__,
            'integration_time': None if it is None else it.tolist(),
            'dtype': None if it is None else str(it.dtype),
        })
    with open(os.path.join(HERE, 'odeblock_t1_cases.json'), 'w') as fh:
        json.dump(cases, fh, indent=1)
    print('t1 cases', len(cases))


def make_odenet_e2e(ref, name, method, tol, in_ch, filters, hw, bs, t1, seed):
    torch.manual_seed(seed)
    net = ref.ODENet(in_ch, out=10, n_filters=filters, downsample='residual', method=method, tol=tol,
                     adjoint=True, t1=t1, dropout=0)
    gen = torch.Generator().manual_seed(seed + 7)
    randomize_(net.odeblock.odefunc, gen)
    net.train()
    x = torch.rand(bs, in_ch, hw, hw, generator=gen)
    y = torch.randint(0, 10, (bs,), generator=gen)
    p = net(x)
    loss = F.cross_entropy(p, y)
    nfe_f = net.nfe(reset=True)
    loss.backward()
    nfe_b = net.nfe(reset=True)
    grads = {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}
    torch.save({
        'method': method, 'tol': tol, 'in_ch': in_ch, 'filters': filters, 't1': t1,
        'x': x, 'y': y, 'logits': p.detach(), 'loss': loss.detach(), 'nfe_f': nfe_f, 'nfe_b': nfe_b,
        'state_dict': net.state_dict(),
        'grads': grads,
    }, os.path.join(HERE, name))
    print(name, 'loss %.5f nfe_f %d nfe_b %d' % (loss, nfe_f, nfe_b))


def make_ode_stem_features(ref):
    torch.manual_seed(31)
    net = ref.ODENet(3, out=10, n_filters=16, downsample='ode', t1=[.1, .2, .3, 1], tol=1e-3, adjoint=True)
    gen = torch.Generator().manual_seed(32)
    randomize_(net, gen)
    net.eval()
    net.to_features_extractor()
    x = torch.rand(2, 3, 32, 32, generator=gen)
    with torch.no_grad():
        feats = net(x)
    torch.save({'x': x, 'features': feats, 'state_dict': net.state_dict(), 'keys': list(net.state_dict().keys()),
                't1': [.1, .2, .3, 1], 'filters': 16, 'tol': 1e-3, 'nfe_main': net.nfe(),
                'nfe_stem': net.downsample.odeblock.nfe},
               os.path.join(HERE, 'odenet_ode_features.pt'))
    print('odenet_ode_features', tuple(feats.shape), 'nfe stem/main', net.downsample.odeblock.nfe, net.nfe())


def make_ode2_train(ref):
    torch.manual_seed(33)
    net = ref.ODENet(3, out=10, n_filters=16, downsample='ode2', method='dopri5', tol=1e-3, adjoint=True, t1=1, dropout=0)
    gen = torch.Generator().manual_seed(34)
    randomize_(net, gen)
    # dynamics whose ReLUs never switch (GroupNorm biases in front of them at +8): gradients of a whole solve can then be
    # compared to rounding -- with ordinary parameters a pre-activation within fp32 rounding of zero flips its mask
    # between two correct implementations, and at 16 channels one flipped element moves a gradient by several percent
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'odefunc.norm' in name and name.endswith('bias') and 'norm3' not in name:
                p.add_(8.0)
    net.train()
    x = torch.rand(2, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (2,), generator=gen)
    p = net(x)
    loss = F.cross_entropy(p, y)
    nfe_f = (net.downsample.odeblock.nfe, net.nfe())
    loss.backward()
    nfe_b = (net.downsample.odeblock.nfe - nfe_f[0], net.nfe() - nfe_f[1])
    torch.save({'x': x, 'y': y, 'logits': p.detach(), 'loss': loss.detach(), 'nfe_f': nfe_f, 'nfe_b': nfe_b,
                'state_dict': net.state_dict(), 'keys': list(net.state_dict().keys()), 'filters': 16, 'tol': 1e-3,
                'grads': {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}},
               os.path.join(HERE, 'odenet_ode2_train.pt'))
    print('odenet_ode2_train loss %.5f nfe_f %s nfe_b %s' % (loss, nfe_f, nfe_b))


def make_stem(ref):
    torch.manual_seed(47)
    gen = torch.Generator().manual_seed(47)
    stem = ref.ResDownsample(1, out_ch=64)
    randomize_(stem, gen)
    x = torch.rand(2, 1, 28, 28, generator=gen)
    out = stem(x)
    cot = torch.randn(out.shape, generator=gen)
    out.backward(cot)
    torch.save({'x': x, 'cot': cot, 'out': out.detach(), 'state_dict': stem.state_dict(),
                'grads': {k: v.grad.clone() for k, v in stem.named_parameters()}},
               os.path.join(HERE, 'stem_residual_c64.pt'))
    print('stem_residual_c64: out %s, %d parameter tensors' % (tuple(out.shape), len(list(stem.parameters()))))


def main():
    ref = import_reference_model()
    if 'cfg1' in sys.argv[1:]:         # only the fixture added in round 5
        make_odenet_e2e(ref, 'odenet_rk4_f64.pt', 'rk4', 1e-3, 1, 64, 28, 4, 1, seed=61)
        return
    if 'stem' in sys.argv[1:]:         # only the fixture added in round 4
        make_stem(ref)
        return
    if 'w4' in sys.argv[1:]:           # only the fixtures added in round 3
        make_odefunc(ref, 64, 8, 8, 8, seed=41, name='odefunc_c64_n8.pt')
        make_odefunc(ref, 64, 8, 8, 8, seed=42, name='odefunc_c64_n8_kf.pt', kink_free=True)
        return
    if 'stems' in sys.argv[1:]:        # only the fixtures added in round 2
        make_ode_stem_features(ref)
        make_ode2_train(ref)
        return
    make_odefunc(ref, 8, 2, 7, 7, seed=23)
    make_odefunc(ref, 16, 3, 5, 6, seed=24)
    make_odefunc(ref, 64, 2, 8, 8, seed=25)
    make_odefunc(ref, 64, 8, 8, 8, seed=41, name='odefunc_c64_n8.pt')
    make_odefunc(ref, 64, 8, 8, 8, seed=42, name='odefunc_c64_n8_kf.pt', kink_free=True)
    make_odenet_t0(ref)
    make_t1_cases(ref)
    # config-1-like plumbing case (MNIST-shaped, rk4, one 3/8 step) and a small dopri5 case
    make_odenet_e2e(ref, 'odenet_rk4.pt', 'rk4', 1e-3, 1, 8, 28, 4, 1, seed=23)
    make_odenet_e2e(ref, 'odenet_dopri5.pt', 'dopri5', 1e-3, 3, 8, 32, 2, 1, seed=29)
    make_ode_stem_features(ref)
    make_ode2_train(ref)
    make_stem(ref)
    make_odenet_e2e(ref, 'odenet_rk4_f64.pt', 'rk4', 1e-3, 1, 64, 28, 4, 1, seed=61)


if __name__ == '__main__':
    main()
