// Fused optimizer step: SGD with momentum and weight decay exactly as torch.optim.SGD(lr, momentum, weight_decay)
// computes it (the reference's optimizer, train.py:136, stepped and zeroed at train.py:56-58), for EVERY parameter
// tensor of the model in one launch (a table of up to SGD_TABLE tensors travels in the kernel arguments):
//     g   = grad_scale * grad + weight_decay * p
//     buf = momentum * buf + g            (buf starts at zero, so the first step gives buf = g like PyTorch)
//     p   = p - lr * buf
// Gradients are read wherever autograd (or the data-parallel reducer's bucket) left them: no flattening pass.
// HBM-bound: 12 B read + 8 B written per element; ~2.0 M parameters at cfg 2 = 40 MB -> ~10 us.
#include "node_internal.h"

namespace node {

__global__ __launch_bounds__(256) void k_sgd_multi(SgdTable tb, float lr, float momentum, float wd, float gscale, const float* skip) {
  if (skip != nullptr && *skip != 0.f) return;   // a solve of this step reported a miss: nothing is committed
  const SgdEntry e = tb.e[blockIdx.y];
  float* __restrict__ p = e.p;
  const float* __restrict__ g = e.g;
  float* __restrict__ m = e.m;
  const size_t n = e.n;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t start = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) == 0;
  size_t done = 0;
  if (m == nullptr) {   // momentum == 0: torch.optim.SGD keeps no buffer then, p = p - lr (grad_scale grad + wd p)
    for (size_t i = start; i < n; i += stride) p[i] -= lr * (gscale * g[i] + wd * p[i]);
    return;
  }
  if (vec) {
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    for (size_t i = start; i < n4; i += stride) {
      float4 pv = p4[i];
      const float4 gv = g4[i];
      float4 mv = m4[i];
      // same operation order as PyTorch's kernels: (grad + wd * p), (momentum * buf + g), (p - lr * buf)
      mv.x = momentum * mv.x + (gscale * gv.x + wd * pv.x);
      mv.y = momentum * mv.y + (gscale * gv.y + wd * pv.y);
      mv.z = momentum * mv.z + (gscale * gv.z + wd * pv.z);
      mv.w = momentum * mv.w + (gscale * gv.w + wd * pv.w);
      pv.x -= lr * mv.x; pv.y -= lr * mv.y; pv.z -= lr * mv.z; pv.w -= lr * mv.w;
      p4[i] = pv;
      m4[i] = mv;
    }
    done = n4 << 2;
  }
  for (size_t i = done + start; i < n; i += stride) {
    const float b = momentum * m[i] + (gscale * g[i] + wd * p[i]);
    m[i] = b;
    p[i] -= lr * b;
  }
}

void launch_sgd_multi(const SgdTable& tb, int count, size_t max_n, float lr, float momentum, float wd, float gscale,
                      const float* skip, hipStream_t s) {
  size_t bx = (max_n / 4 + 255) / 256;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(k_sgd_multi, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, s, tb, lr, momentum, wd, gscale, skip);
}

}  // namespace node
