"""Head (FCClassifier) forward+backward on [128,256,8,8]: eager vs hipGraph-captured, and kernel count."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_ode_features_amd as nof
dev = torch.device('cuda', 0)
torch.manual_seed(23)
model = nof.ODENet(3, out=10, n_filters=256, downsample='residual', method='dopri5', tol=1e-3, adjoint=True, t1=1, dropout=0.5).to(dev)
feat = torch.randn(128, 256, 8, 8, device=dev, requires_grad=True)
y = torch.randint(0, 10, (128,), device=dev)
def timeit(fn, n=50, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def eager():
    loss = F.cross_entropy(model.classifier(feat), y); loss.backward()
print('head fwd+loss+bwd eager   %.3f ms' % timeit(eager))
head_g = torch.cuda.make_graphed_callables(model.classifier, (feat.detach().clone().requires_grad_(True),))
def graphed():
    loss = F.cross_entropy(head_g(feat), y); loss.backward()
print('head fwd+loss+bwd graphed %.3f ms' % timeit(graphed))
print(model.classifier)
