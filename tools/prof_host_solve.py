import cProfile, pstats, sys, os, io, time
import torch
sys.path.insert(0, os.getcwd())
import neural_ode_features_amd as nof
torch.manual_seed(0)
f = nof.ODEfunc(256).cuda()
y = torch.randn(1, 256, 8, 8, device='cuda')
t = torch.tensor([0.0, 1.0], device='cuda')
with torch.no_grad():
    for _ in range(20):
        nof.odeint(f, y, t, rtol=1e-1, atol=1e-1, method='dopri5')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        nof.odeint(f, y, t, rtol=1e-1, atol=1e-1, method='dopri5')
    torch.cuda.synchronize()
    print('wall per solve us', (time.perf_counter() - t0) / 2000 * 1e6, 'nfe', f.last_forward_stats['nfe'])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        nof.odeint(f, y, t, rtol=1e-1, atol=1e-1, method='dopri5')
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:6000])
