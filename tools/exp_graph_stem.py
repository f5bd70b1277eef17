"""How much of the training step is PyTorch stem/head/optimizer time, eager vs hipGraph-captured?"""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof

dev = torch.device('cuda', 0)
torch.manual_seed(23)
model = nof.ODENet(3, out=10, n_filters=256, downsample='residual', method='dopri5', tol=1e-3, adjoint=True, t1=1, dropout=0.5).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, 10, (128,), device=dev)

def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def step_full():
    loss = F.cross_entropy(model(x), y); loss.backward(); opt.step(); opt.zero_grad()
print('full step eager              %.2f ms' % timeit(step_full))

model.odeblock.t1 = 0   # identity ODE block: what remains is stem + head + optimizer
def step_noode():
    loss = F.cross_entropy(model(x), y); loss.backward(); opt.step(); opt.zero_grad()
print('stem+head+opt eager (no ODE) %.2f ms' % timeit(step_noode))
def fwdbwd_noode():
    loss = F.cross_entropy(model(x), y); loss.backward()
print('stem+head fwd+bwd eager      %.2f ms' % timeit(fwdbwd_noode))
opt.zero_grad()
try:
    stem_g = torch.cuda.make_graphed_callables(model.downsample, (x,))
    feat = model.downsample(x).detach().requires_grad_(True)
    head_g = torch.cuda.make_graphed_callables(model.classifier, (feat,))
    def fwdbwd_graph():
        loss = F.cross_entropy(head_g(stem_g(x)), y); loss.backward()
    print('stem+head fwd+bwd graphed    %.2f ms' % timeit(fwdbwd_graph))
    def step_graph():
        loss = F.cross_entropy(head_g(stem_g(x)), y); loss.backward(); opt.step(); opt.zero_grad()
    print('stem+head+opt graphed        %.2f ms' % timeit(step_graph))
except Exception as e:
    print('graph capture failed:', repr(e)[:300])
