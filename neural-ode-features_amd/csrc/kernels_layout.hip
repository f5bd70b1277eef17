// Layout kernels: NCHW <-> NHWC at the solve boundary, weight packing for the
// implicit-GEMM B operand, the time-channel border map, and the theta-segment
// -> PyTorch flat parameter layout conversion.  All HBM-bound and tiny next to
// the convolutions (run once per solve, not per stage).
#include "node_internal.h"
#include <cstring>

namespace node {

// ---------------------------------------------------------------- transposes
// per sample: [C][HW] <-> [HW][C] through a 32x33 LDS tile
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, float* __restrict__ dst,
                                                   int rows, int cols) {
  // src: [n][rows][cols] -> dst: [n][cols][rows]
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* s = src + (size_t)n * rows * cols;
  float* d = dst + (size_t)n * rows * cols;
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) tile[i][tx] = s[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) d[(size_t)c * rows + r] = tile[tx][i];
  }
}

void launch_nchw_to_nhwc(const Dims& d, const float* src, float* dst, hipStream_t s) {
  dim3 grid((d.HW + 31) / 32, (d.C + 31) / 32, d.N);
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, src, dst, d.C, d.HW);
}
void launch_nhwc_to_nchw(const Dims& d, const float* src, float* dst, hipStream_t s) {
  dim3 grid((d.C + 31) / 32, (d.HW + 31) / 32, d.N);
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, s, src, dst, d.HW, d.C);
}

// ------------------------------------------------------------- weight packing
// packed[nt][ch][tap][j][kk]  (64 output columns x 32 K values per (nt, ch, tap) tile, contiguous:
// one ds_read_b128 of a column feeds four MFMA steps)
//   forward : value = W[co = nt*BNE + j][1 + ci = ch*32 + kk][kh][kw],           tap = kh*3 + kw
//   dgrad   : value = W[co = ch*32 + kk][1 + ci = nt*BNE + j][2 - kh][2 - kw]    (flipped, transposed)
// zero outside C / beyond BNE so padded K rows and N columns contribute nothing.
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, float* __restrict__ packed,
                                                      int C, int BNE, int ntile, int nchunk, int dgrad) {
  size_t total = (size_t)ntile * nchunk * 9 * KCH * BN;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = idx % KCH;
    size_t r = idx / KCH;
    const int j = r % BN; r /= BN;
    int tap = r % 9; r /= 9;
    int ch = r % nchunk;
    int nt = r / nchunk;
    int kh = tap / 3, kw = tap % 3;
    int kidx = ch * KCH + kk;        // K index (input channel of this GEMM)
    int nidx = nt * BNE + j;         // N index (output channel of this GEMM)
    float v = 0.f;
    if (j < BNE && kidx < C && nidx < C) {
      if (!dgrad) {
        v = w[(((size_t)nidx * (C + 1) + 1 + kidx) * 3 + kh) * 3 + kw];
      } else {
        v = w[(((size_t)kidx * (C + 1) + 1 + nidx) * 3 + (2 - kh)) * 3 + (2 - kw)];
      }
    }
    packed[idx] = v;
  }
}

void launch_pack_weights(const Dims& d, const float* w, float* packed, int dgrad, hipStream_t s) {
  size_t total = (size_t)d.ntile * d.nchunk * 9 * KCH * BN;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_weights, dim3(blocks), dim3(256), 0, s, w, packed, d.C, d.BNE, d.ntile, d.nchunk, dgrad);
}

// -------------------------------------------------------------- time-channel map
// The constant-t channel (model.py:321-322) is zero-padded like every other input
// channel, so its contribution is t * sum of the taps that fall inside the image:
//   tmap[p][co] = sum_{kh,kw : (h+kh-1, w+kw-1) in bounds} W[co][0][kh][kw]
__global__ __launch_bounds__(256) void k_tmap(const float* __restrict__ w, float* __restrict__ tmap, int C, int H, int W) {
  int HW = H * W;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < HW * C; idx += gridDim.x * blockDim.x) {
    int co = idx % C, p = idx / C;
    int h = p / W, x = p % W;
    float sum = 0.f;
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        int hh = h + kh - 1, ww = x + kw - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) sum += w[(((size_t)co * (C + 1)) * 3 + kh) * 3 + kw];
      }
    tmap[idx] = sum;
  }
}
void launch_tmap(const Dims& d, const float* w, float* tmap, hipStream_t s) {
  int total = d.HW * d.C;
  hipLaunchKernelGGL(k_tmap, dim3((total + 255) / 256), dim3(256), 0, s, w, tmap, d.C, d.H, d.W);
}

// time-channel weights W[co][0][kh][kw] gathered as [tap][co] (d f / d t contracts them with the masked dz sums)
__global__ __launch_bounds__(256) void k_wtime(const float* __restrict__ w, float* __restrict__ wt, int C) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 9 * C; i += gridDim.x * blockDim.x) {
    const int tap = i / C, co = i - tap * C;
    wt[i] = w[((size_t)co * (C + 1)) * 9 + tap];
  }
}
void launch_wtime(const Dims& d, const float* w, float* wtime, hipStream_t s) {
  hipLaunchKernelGGL(k_wtime, dim3((9 * d.C + 255) / 256), dim3(256), 0, s, w, wtime, d.C);
}

// Both layers' border maps (and, for an augmented solve, their gathered time-channel taps) in ONE launch per solve:
// blockIdx.y = job.  Four to six launches of ~4.5 us each (their floor on this box) per training step otherwise.
struct TimePrepArgs { const float* w[2]; float* tmap[2]; float* wtime[2]; float* zero[12]; size_t zero_n[12]; int nzero; int njobs; };
__global__ __launch_bounds__(256) void k_time_prep(TimePrepArgs a, int C, int H, int W) {
  const int job = blockIdx.y, layer = job & 1;
  if (job >= a.njobs) {   // the solve's zero fills (zero rows behind the conv inputs, the never-written stage derivative,
                          // the finalize kernel's arrival counter): they used to be one hipMemsetAsync each
    const int z = job - a.njobs;
    float* p = a.zero[z];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.zero_n[z]; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
    return;
  }
  const float* w = a.w[layer];
  if (job < 2) {
    float* tmap = a.tmap[layer];
    const int HW = H * W;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < HW * C; idx += gridDim.x * blockDim.x) {
      const int co = idx % C, p = idx / C;
      const int h = p / W, x = p % W;
      float sum = 0.f;
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
          const int hh = h + kh - 1, ww = x + kw - 1;
          if (hh >= 0 && hh < H && ww >= 0 && ww < W) sum += w[(((size_t)co * (C + 1)) * 3 + kh) * 3 + kw];
        }
      tmap[idx] = sum;
    }
  } else {
    float* wt = a.wtime[layer];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 9 * C; i += gridDim.x * blockDim.x) {
      const int tap = i / C, co = i - tap * C;
      wt[i] = w[((size_t)co * (C + 1)) * 9 + tap];
    }
  }
}
void launch_time_prep(const Dims& d, const float* w1, const float* w2, float* tmap1, float* tmap2, float* wtime1, float* wtime2,
                      float* const* zero, const size_t* zero_n, int nzero, hipStream_t s) {
  TimePrepArgs a;
  memset(&a, 0, sizeof(a));
  a.w[0] = w1; a.w[1] = w2; a.tmap[0] = tmap1; a.tmap[1] = tmap2; a.wtime[0] = wtime1; a.wtime[1] = wtime2;
  a.njobs = wtime1 != nullptr ? 4 : 2;
  a.nzero = nzero > 12 ? 12 : nzero;
  for (int i = 0; i < a.nzero; ++i) { a.zero[i] = zero[i]; a.zero_n[i] = zero_n[i]; }
  const int total = d.HW * d.C;
  int bx = (total + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_time_prep, dim3(bx, a.njobs + a.nzero), dim3(256), 0, s, a, d.C, d.H, d.W);
}

// ---------------------------------------------- theta internal -> PyTorch flat
// flat (parameters() order): norm1.w, norm1.b, conv1.w [C][C+1][3][3], conv1.b, norm2.w, ...
__global__ __launch_bounds__(256) void k_theta_to_torch(const float* __restrict__ th, float* __restrict__ flat, int C) {
  const ThetaLayout L = theta_layout(C);
  const size_t cw = (size_t)C * (C + 1) * 9;
  const size_t P = 6 * (size_t)C + 2 * (cw + C);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    float v;
    int layer = 0;
    for (;;) {
      if (r < (size_t)C) { v = th[L.g[layer] + r]; break; }
      r -= C;
      if (r < (size_t)C) { v = th[L.b[layer] + r]; break; }
      r -= C;
      if (layer == 2) { v = 0.f; break; }  // unreachable
      if (r < cw) {
        int kw = r % 3; size_t q = r / 3;
        int kh = q % 3; q /= 3;
        int cin = q % (C + 1);
        int co = q / (C + 1);
        int tap = kh * 3 + kw;
        if (cin == 0) v = th[L.wt[layer] + (size_t)tap * C + co];
        else v = th[L.wc[layer] + ((size_t)tap * C + (cin - 1)) * C + co];
        break;
      }
      r -= cw;
      if (r < (size_t)C) { v = th[L.cb[layer] + r]; break; }
      r -= C;
      ++layer;
    }
    flat[idx] = v;
  }
}
// The same conversion through LDS tiles for C % 32 == 0: the conv weights are [tap][ci][co] inside and
// [co][1 + ci][tap] in PyTorch's layout, so the element-wise kernel above reads with a stride of C*C floats (26 us at
// C = 256).  A workgroup moves a (32 co x 16 ci x 9 taps) tile: reads 128-byte runs along co, writes 576-byte runs
// along (ci, tap).  The last workgroups copy the small vectors and scatter the time-channel taps.
__global__ __launch_bounds__(256) void k_theta_to_torch_tiled(const float* __restrict__ th, float* __restrict__ flat, int C) {
  __shared__ float tile[9 * 16 * 33];
  const ThetaLayout L = theta_layout(C);
  const size_t cw = (size_t)C * (C + 1) * 9;
  const size_t per_layer = 3 * (size_t)C + cw;          // g, b, conv.w, conv.b
  const int nco = C / 32, nci = C / 16;
  const int nblk = 2 * nco * nci;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b < nblk) {
    const int layer = b / (nco * nci), rem = b - layer * nco * nci;
    const int co0 = (rem / nci) * 32, ci0 = (rem % nci) * 16;
    for (int e = tid; e < 9 * 16 * 32; e += 256) {
      const int co = e & 31, ci = (e >> 5) & 15, tap = e >> 9;
      tile[(tap * 16 + ci) * 33 + co] = th[L.wc[layer] + ((size_t)tap * C + ci0 + ci) * C + co0 + co];
    }
    __syncthreads();
    const size_t wbase = layer * per_layer + 2 * (size_t)C;
    for (int e = tid; e < 32 * 144; e += 256) {
      const int co = e / 144, r = e - co * 144;
      const int ci = r / 9, tap = r - ci * 9;
      flat[wbase + ((size_t)(co0 + co) * (C + 1) + 1 + ci0 + ci) * 9 + tap] = tile[(tap * 16 + ci) * 33 + co];
    }
    return;
  }
  // small parts: 3 x (g, b), 2 x conv bias, 2 x time-channel taps
  const int nsmall = 26 * C;
  for (int q = (b - nblk) * 256 + tid; q < nsmall; q += (gridDim.x - nblk) * 256) {
    if (q < 6 * C) {
      const int layer = q / (2 * C), r = q - layer * 2 * C;
      const int which = r / C, c = r - which * C;
      flat[layer * per_layer + (size_t)which * C + c] = th[(which ? L.b[layer] : L.g[layer]) + c];
    } else if (q < 8 * C) {
      const int r = q - 6 * C, layer = r / C, c = r - layer * C;
      flat[layer * per_layer + 2 * (size_t)C + cw + c] = th[L.cb[layer] + c];
    } else {
      const int r = q - 8 * C, layer = r / (9 * C), rr = r - layer * 9 * C;
      const int tap = rr / C, co = rr - tap * C;
      flat[layer * per_layer + 2 * (size_t)C + ((size_t)co * (C + 1)) * 9 + tap] = th[L.wt[layer] + (size_t)tap * C + co];
    }
  }
}
void launch_theta_to_torch(const Dims& d, const float* theta_int, float* flat, hipStream_t s) {
  if (d.C % 32 == 0) {
    const int nblk = 2 * (d.C / 32) * (d.C / 16);
    int extra = (26 * d.C + 255) / 256;
    if (extra > 64) extra = 64;
    hipLaunchKernelGGL(k_theta_to_torch_tiled, dim3(nblk + extra), dim3(256), 0, s, theta_int, flat, d.C);
    return;
  }
  int blocks = (int)((d.P + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_theta_to_torch, dim3(blocks), dim3(256), 0, s, theta_int, flat, d.C);
}

}  // namespace node
