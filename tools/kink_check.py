import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import make_func, rel_err, robust_grad_err
import neural_ode_features_amd as nof
from oracle.dynamics import odefunc_vjp as oracle_vjp
for kf in (False, True):
    f, twin = make_func(256, seed=2, device='cuda', kink_free=kf)
    gen = torch.Generator().manual_seed(8)
    y = torch.randn(128, 256, 8, 8, generator=gen)
    cot = torch.randn(128, 256, 8, 8, generator=gen)
    fo, vy, vt, vp = nof.odefunc_vjp(f, 0.5, y.cuda(), cot.cuda())
    f_ref, vy_ref, vt_ref, vp_ref = oracle_vjp(0.5, y, dict(twin.named_parameters()), cot)
    e = (vy.cpu() - vy_ref).abs()
    bad = e > 1e-4 * vy_ref.abs().max()
    idx = bad.nonzero()
    print('kink_free', kf, 'rel vy', rel_err(vy, vy_ref), 'vp', rel_err(vp, vp_ref), 'robust', robust_grad_err(vy, vy_ref), 'nbad', int(bad.sum()))
    if len(idx):
        print(' samples', idx[:,0].unique().tolist(), 'rows', idx[:,2].unique().tolist(), 'cols', idx[:,3].unique().tolist(), 'nch', len(idx[:,1].unique()))
