"""Forward-only multi-time-point solve (the feature-extraction use of the path, evaluate.py:56-94): time per call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof
dev = torch.device('cuda:0')
torch.manual_seed(0)
f = nof.ODEfunc(256).to(dev)
y = torch.randn(128, 256, 8, 8, device=dev)
for T in (2, 21, 51):
    t = torch.linspace(0, 1, T, device=dev)
    with torch.no_grad():
        for _ in range(2): out = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3, method='dopri5')
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = nof.odeint(f, y, t, rtol=1e-3, atol=1e-3, method='dopri5')
        torch.cuda.synchronize()
    print('T=%2d: %.3f ms per solve, nfe %d, out %s' % (T, (time.perf_counter() - t0) / 5 * 1e3, f.nfe // 7, tuple(out.shape)))
    f.nfe = 0
