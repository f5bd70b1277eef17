// Classifier head of the ODE-Net, the caller right behind the hot path:
//   GroupNorm -> ReLU -> global average pool -> [Dropout]     (reference: model.py:231-250 FCClassifier,
//   model.py:268-271 normalization('group') = nn.GroupNorm(min(32, C), C)); the Linear layer and the loss stay
// with the caller.  On PyTorch-ROCm this is ~15 launch-bound kernels (0.45 ms per training step at cfg 2, of
// which 0.27 ms survive hipGraph capture); here it is one launch forward and one backward, one workgroup per
// sample, fp32 throughout, NCHW in and out (the ODE block's external layout).
//
// forward:  mean_g, rstd_g over (C/G channels x HW);  v = relu((z - mean) * rstd * gamma + beta);
//           pooled[n][c] = scale[n][c] * mean_px v          (scale = dropout mask / (1 - p), or absent)
// backward: gv = g[n][c] * scale / HW on every pixel; gu = gv * (u > 0); dgamma, dbeta partials per sample;
//           dz = rstd * (gu*gamma - mean_g(gu*gamma) - xhat * mean_g(gu*gamma*xhat))
//
// k_head_*_v<LPC>: the sample ([C][HW], 64 KB at cfg 2) is read ONCE with coalesced 16-B loads -- LPC lanes
// per channel, 256 / LPC channels per pass -- and kept in registers for all phases; channel sums are LPC-lane
// DPP reductions, group sums go through LDS.  k_head_*_g: one thread per channel, any geometry (fallback).
#include "node_internal.h"
#include "../../include/node_hip.h"
#include <cstdio>

namespace node {

namespace {
constexpr int HEAD_MAXIT = 16;   // float4 per thread kept in registers by the vector kernels

template <int CTRL>
__device__ inline float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over aligned groups of LPC consecutive lanes (LPC = 1, 4, 16, 64), result in every lane of the group
template <int LPC>
__device__ inline float lanes_sum(float v) {
  if (LPC >= 4) { v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); }        // quad_perm [1,0,3,2], [2,3,0,1]
  if (LPC >= 16) { v = dpp_add<0x141>(v); v = dpp_add<0x140>(v); }     // row_half_mirror, row_mirror
  if (LPC >= 64) {
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast15
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast31
    v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  }
  return v;
}

// per-channel values chv[C] (LDS) -> per-group sums grp[G] (LDS); one thread per group
__device__ inline void groups_from_channels(const float* chv, float* grp, int G, int cpg) {
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < cpg; ++k) s += chv[g * cpg + k];
    grp[g] = s;
  }
  __syncthreads();
}

// 256-thread block sum, result in every thread (red: 4 floats of LDS; safe to call back to back)
__device__ inline float block_sum_256h(float v, float* red) {
  v = lanes_sum<64>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// sum of `v` over the channels of this thread's group (thread = channel; fallback kernels)
__device__ inline float group_sum(float v, float* ch, int c, int c_lo, int cpg, bool on) {
  __syncthreads();               // previous use of `ch` finished
  if (on) ch[c] = v;
  __syncthreads();
  float s = 0.f;
  if (on)
    for (int k = 0; k < cpg; ++k) s += ch[c_lo + k];
  return s;
}
}  // namespace

template <int LPC>
__global__ __launch_bounds__(256) void k_head_fwd_v(const float* __restrict__ z, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ scale,
                                                    float* __restrict__ pooled, float* __restrict__ stats, int C, int HW,
                                                    int G, float eps) {
  extern __shared__ float sm[];   // chv[C], grp[G], grp2[G]
  float* chv = sm;
  float* grp = sm + C;
  float* grp2 = grp + G;
  constexpr int CPP = 256 / LPC;             // channels per pass
  const int n = blockIdx.x, cpg = C / G;
  const int lc = threadIdx.x / LPC, lq = threadIdx.x % LPC;   // channel within the pass, float4 within the channel
  const int q4 = HW >> 2;                    // float4 per channel == LPC
  const int iters = (C + CPP - 1) / CPP;
  const float4* zs = reinterpret_cast<const float4*>(z + (size_t)n * C * HW);
  float4 v[HEAD_MAXIT];
#pragma unroll
  for (int i = 0; i < HEAD_MAXIT; ++i) {
    const int c = i * CPP + lc;
    v[i] = (i < iters && c < C) ? zs[(size_t)c * q4 + lq] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_m = 1.0f / (float)(cpg * HW);
#pragma unroll
  for (int i = 0; i < HEAD_MAXIT; ++i) {
    const int c = i * CPP + lc;
    const float s = lanes_sum<LPC>((v[i].x + v[i].y) + (v[i].z + v[i].w));
    if (i < iters && c < C && lq == 0) chv[c] = s;
  }
  groups_from_channels(chv, grp, G, cpg);    // grp = group sums -> means below
#pragma unroll
  for (int i = 0; i < HEAD_MAXIT; ++i) {
    const int c = i * CPP + lc;
    const float mean = (i < iters && c < C) ? grp[c / cpg] * inv_m : 0.f;
    const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
    const float s = lanes_sum<LPC>((a * a + b * b) + (cc * cc + dd * dd));
    if (i < iters && c < C && lq == 0) chv[c] = s;
  }
  groups_from_channels(chv, grp2, G, cpg);   // grp2 = centred sums of squares
#pragma unroll
  for (int i = 0; i < HEAD_MAXIT; ++i) {
    const int c = i * CPP + lc;
    if (i < iters && c < C) {
      const int g = c / cpg;
      const float mean = grp[g] * inv_m, rstd = 1.0f / sqrtf(grp2[g] * inv_m + eps);
      const float gm = gamma[c], bt = beta[c];
      float acc = fmaxf(((v[i].x - mean) * rstd) * gm + bt, 0.f) + fmaxf(((v[i].y - mean) * rstd) * gm + bt, 0.f) +
                  fmaxf(((v[i].z - mean) * rstd) * gm + bt, 0.f) + fmaxf(((v[i].w - mean) * rstd) * gm + bt, 0.f);
      acc = lanes_sum<LPC>(acc) * (1.0f / (float)HW);
      if (lq == 0) {
        if (scale) acc *= scale[(size_t)n * C + c];
        pooled[(size_t)n * C + c] = acc;
        if (c == g * cpg) { stats[((size_t)n * G + g) * 2] = mean; stats[((size_t)n * G + g) * 2 + 1] = rstd; }
      }
    } else {
      (void)lanes_sum<LPC>(0.f);   // keep the DPP sequence uniform across the wave
    }
  }
}

template <int LPC>
__global__ __launch_bounds__(256) void k_head_bwd_v(const float* __restrict__ z, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ scale,
                                                    const float* __restrict__ stats, const float* __restrict__ gpool,
                                                    float* __restrict__ dz, float* __restrict__ gpart, int C, int HW, int G) {
  extern __shared__ float sm[];   // chv[C], chw[C], grp[G], grp2[G]
  float* chv = sm;
  float* chw = sm + C;
  float* grp = chw + C;
  float* grp2 = grp + G;
  constexpr int CPP = 256 / LPC;
  const int n = blockIdx.x, cpg = C / G;
  const int lc = threadIdx.x / LPC, lq = threadIdx.x % LPC;
  const int q4 = HW >> 2;
  const int iters = (C + CPP - 1) / CPP;
  const float4* zs = reinterpret_cast<const float4*>(z + (size_t)n * C * HW);
  float4* dzs = reinterpret_cast<float4*>(dz + (size_t)n * C * HW);
  const float inv_m = 1.0f / (float)(cpg * HW);
  float4 xh[HEAD_MAXIT];   // xhat, then reused
  float gvv[HEAD_MAXIT];
#pragma unroll
  for (int i = 0; i < HEAD_MAXIT; ++i) {
    const int c = i * CPP + lc;
    const bool on = i < iters && c < C;
    float4 zv = on ? zs[(size_t)c * q4 + lq] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int g = on ? c / cpg : 0;
    const float mean = on ? stats[((size_t)n * G + g) * 2] : 0.f, rstd = on ? stats[((size_t)n * G + g) * 2 + 1] : 0.f;
    xh[i] = make_float4((zv.x - mean) * rstd, (zv.y - mean) * rstd, (zv.z - mean) * rstd, (zv.w - mean) * rstd);
    float gv = 0.f;
    if (on) {
      gv = gpool[(size_t)n * C + c] * (1.0f / (float)HW);
      if (scale) gv *= scale[(size_t)n * C + c];
    }
    gvv[i] = gv;
    const float gm = on ? gamma[c] : 0.f, bt = on ? beta[c] : 0.f;
    // gu per pixel (gv where the ReLU passed), kept as a 4-bit mask folded into the sign of nothing: recomputed below
    const float g0 = (xh[i].x * gm + bt > 0.f) ? gv : 0.f, g1 = (xh[i].y * gm + bt > 0.f) ? gv : 0.f;
    const float g2 = (xh[i].z * gm + bt > 0.f) ? gv : 0.f, g3 = (xh[i].w * gm + bt > 0.f) ? gv : 0.f;
    const float dg = lanes_sum<LPC>((g0 * xh[i].x + g1 * xh[i].y) + (g2 * xh[i].z + g3 * xh[i].w));
    const float db = lanes_sum<LPC>((g0 + g1) + (g2 + g3));
    if (on && lq == 0) {
      gpart[((size_t)n * 2 + 0) * C + c] = dg;
      gpart[((size_t)n * 2 + 1) * C + c] = db;
      chv[c] = db * gm;   // sum_px dxhat
      chw[c] = dg * gm;   // sum_px dxhat * xhat
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < cpg; ++k) { s1 += chv[g * cpg + k]; s2 += chw[g * cpg + k]; }
    grp[g] = s1 * inv_m;
    grp2[g] = s2 * inv_m;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < HEAD_MAXIT; ++i) {
    const int c = i * CPP + lc;
    if (i < iters && c < C) {
      const int g = c / cpg;
      const float rstd = stats[((size_t)n * G + g) * 2 + 1], m1 = grp[g], m2 = grp2[g];
      const float gm = gamma[c], bt = beta[c], gv = gvv[i];
      float4 o;
      o.x = rstd * (((xh[i].x * gm + bt > 0.f) ? gv * gm : 0.f) - m1 - xh[i].x * m2);
      o.y = rstd * (((xh[i].y * gm + bt > 0.f) ? gv * gm : 0.f) - m1 - xh[i].y * m2);
      o.z = rstd * (((xh[i].z * gm + bt > 0.f) ? gv * gm : 0.f) - m1 - xh[i].z * m2);
      o.w = rstd * (((xh[i].w * gm + bt > 0.f) ? gv * gm : 0.f) - m1 - xh[i].w * m2);
      dzs[(size_t)c * q4 + lq] = o;
    }
  }
}

// ---- any geometry: one thread per channel, three passes over its pixels ----
__global__ __launch_bounds__(256) void k_head_fwd_g(const float* __restrict__ z, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ scale,
                                                    float* __restrict__ pooled, float* __restrict__ stats, int C, int HW,
                                                    int G, float eps) {
  extern __shared__ float ch[];   // [C]
  const int n = blockIdx.x, cpg = C / G;
  const float inv_m = 1.0f / (float)(cpg * HW);
  for (int cb = 0; cb < C; cb += 256) {   // cpg divides 256 or C <= 256 (checked on the host): groups never straddle passes
    const int c = cb + threadIdx.x;
    const bool on = c < C;
    const int g = on ? c / cpg : 0, c_lo = g * cpg;
    const float* zc = z + ((size_t)n * C + (on ? c : 0)) * HW;
    float s = 0.f;
    if (on)
      for (int p = 0; p < HW; ++p) s += zc[p];
    const float mean = group_sum(s, ch, c, c_lo, cpg, on) * inv_m;
    float s2 = 0.f;
    if (on)
      for (int p = 0; p < HW; ++p) { const float dv = zc[p] - mean; s2 += dv * dv; }
    const float var = group_sum(s2, ch, c, c_lo, cpg, on) * inv_m;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (on) {
      const float gm = gamma[c], bt = beta[c];
      float acc = 0.f;
      for (int p = 0; p < HW; ++p) acc += fmaxf(((zc[p] - mean) * rstd) * gm + bt, 0.f);   // (subtract first: groups of one element have rstd = eps^-1/2)
      acc *= 1.0f / (float)HW;
      if (scale) acc *= scale[(size_t)n * C + c];
      pooled[(size_t)n * C + c] = acc;
      if (c == c_lo) { stats[((size_t)n * G + g) * 2] = mean; stats[((size_t)n * G + g) * 2 + 1] = rstd; }
    }
  }
}

__global__ __launch_bounds__(256) void k_head_bwd_g(const float* __restrict__ z, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ scale,
                                                    const float* __restrict__ stats, const float* __restrict__ gpool,
                                                    float* __restrict__ dz, float* __restrict__ gpart, int C, int HW, int G) {
  extern __shared__ float ch[];   // [C]
  const int n = blockIdx.x, cpg = C / G;
  const float inv_m = 1.0f / (float)(cpg * HW);
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + threadIdx.x;
    const bool on = c < C;
    const int g = on ? c / cpg : 0, c_lo = g * cpg;
    const size_t base = ((size_t)n * C + (on ? c : 0)) * HW;
    const float mean = on ? stats[((size_t)n * G + g) * 2] : 0.f, rstd = on ? stats[((size_t)n * G + g) * 2 + 1] : 0.f;
    const float gm = on ? gamma[c] : 0.f, bt = on ? beta[c] : 0.f;
    float gv = 0.f;
    if (on) {
      gv = gpool[(size_t)n * C + c] * (1.0f / (float)HW);
      if (scale) gv *= scale[(size_t)n * C + c];
    }
    float dg = 0.f, db = 0.f;   // sums over the pixels of gu * xhat and gu  (gu = gv where the ReLU passed)
    if (on)
      for (int p = 0; p < HW; ++p) {
        const float xh = (z[base + p] - mean) * rstd;
        const float gu = (xh * gm + bt > 0.f) ? gv : 0.f;
        dg += gu * xh;
        db += gu;
      }
    if (on) {
      gpart[((size_t)n * 2 + 0) * C + c] = dg;
      gpart[((size_t)n * 2 + 1) * C + c] = db;
    }
    const float m1 = group_sum(db * gm, ch, c, c_lo, cpg, on) * inv_m;   // mean_g(dxhat),  dxhat = gu * gamma
    const float m2 = group_sum(dg * gm, ch, c, c_lo, cpg, on) * inv_m;   // mean_g(dxhat * xhat)
    if (on)
      for (int p = 0; p < HW; ++p) {
        const float xh = (z[base + p] - mean) * rstd;
        const float gu = (xh * gm + bt > 0.f) ? gv : 0.f;
        dz[base + p] = rstd * (gu * gm - m1 - xh * m2);
      }
  }
}

// ----------------------------------------------------------------------------
// GroupNorm (+ReLU) of the stem's residual blocks (model.py:284-310: `relu(norm(x))` in front of every conv),
// forward and backward, NCHW.  One workgroup per (sample, group): in NCHW a group is ONE contiguous run of
// cpg * HW floats.  Three passes over that run (it stays in L1/L2); per-channel (dgamma, dbeta) partials per
// sample for the caller to sum.  Replaces PyTorch's 3 + 5 kernels per GroupNorm-ReLU pair, among them a
// 56-us GammaBetaBackward.
//   forward:  out = [relu]((z - mean) * rstd * gamma + beta)
//   backward: gu = g_out * [(u > 0)];  dz = rstd * (gu*gamma - mean_g(gu*gamma) - xhat * mean_g(gu*gamma*xhat))
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gn_relu_fwd(const float* __restrict__ z, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ out,
                                                     float* __restrict__ stats, int C, int HW, int G, float eps, int relu) {
  __shared__ float red[4];
  const int n = blockIdx.x / G, g = blockIdx.x - n * G, cpg = C / G;
  const int M = cpg * HW;
  const size_t base = ((size_t)n * C + (size_t)g * cpg) * HW;
  const float* zg = z + base;
  float s = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) s += zg[i];
  const float mean = block_sum_256h(s, red) / (float)M;
  float s2 = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) { const float dv = zg[i] - mean; s2 += dv * dv; }
  const float rstd = 1.0f / sqrtf(block_sum_256h(s2, red) / (float)M + eps);
  if (threadIdx.x == 0) { stats[(size_t)blockIdx.x * 2] = mean; stats[(size_t)blockIdx.x * 2 + 1] = rstd; }
  for (int cc = 0; cc < cpg; ++cc) {
    const float gm = gamma[g * cpg + cc], bt = beta[g * cpg + cc];
    for (int p = threadIdx.x; p < HW; p += 256) {
      float v = ((zg[cc * HW + p] - mean) * rstd) * gm + bt;
      if (relu) v = fmaxf(v, 0.f);
      out[base + cc * HW + p] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_gn_relu_bwd(const float* __restrict__ z, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ stats,
                                                     const float* __restrict__ gout, float* __restrict__ dz,
                                                     float* __restrict__ gpart, int C, int HW, int G, int relu) {
  __shared__ float red[4];
  const int n = blockIdx.x / G, g = blockIdx.x - n * G, cpg = C / G;
  const int M = cpg * HW;
  const size_t base = ((size_t)n * C + (size_t)g * cpg) * HW;
  const float mean = stats[(size_t)blockIdx.x * 2], rstd = stats[(size_t)blockIdx.x * 2 + 1];
  float s1 = 0.f, s2 = 0.f;   // sums over the group of dxhat and dxhat * xhat
  for (int cc = 0; cc < cpg; ++cc) {
    const int c = g * cpg + cc;
    const float gm = gamma[c], bt = beta[c];
    float dg = 0.f, db = 0.f;
    for (int p = threadIdx.x; p < HW; p += 256) {
      const float xh = (z[base + cc * HW + p] - mean) * rstd;
      float gu = gout[base + cc * HW + p];
      if (relu && !(xh * gm + bt > 0.f)) gu = 0.f;
      dg += gu * xh;
      db += gu;
    }
    dg = block_sum_256h(dg, red);
    db = block_sum_256h(db, red);
    if (threadIdx.x == 0) {
      gpart[((size_t)n * 2 + 0) * C + c] = dg;
      gpart[((size_t)n * 2 + 1) * C + c] = db;
    }
    s1 += db * gm;
    s2 += dg * gm;
  }
  const float m1 = s1 / (float)M, m2 = s2 / (float)M;
  for (int cc = 0; cc < cpg; ++cc) {
    const float gm = gamma[g * cpg + cc], bt = beta[g * cpg + cc];
    for (int p = threadIdx.x; p < HW; p += 256) {
      const float xh = (z[base + cc * HW + p] - mean) * rstd;
      float gu = gout[base + cc * HW + p];
      if (relu && !(xh * gm + bt > 0.f)) gu = 0.f;
      dz[base + cc * HW + p] = rstd * (gu * gm - m1 - xh * m2);
    }
  }
}

void launch_gn_relu_fwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, int relu, float* out,
                        float* stats, hipStream_t s) {
  hipLaunchKernelGGL(k_gn_relu_fwd, dim3(sh.n * sh.groups), dim3(256), 0, s, z, gamma, beta, out, stats, sh.c, sh.h * sh.w,
                     sh.groups, sh.eps, relu);
}
void launch_gn_relu_bwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, const float* stats,
                        int relu, const float* gout, float* dz, float* gpart, hipStream_t s) {
  hipLaunchKernelGGL(k_gn_relu_bwd, dim3(sh.n * sh.groups), dim3(256), 0, s, z, gamma, beta, stats, gout, dz, gpart, sh.c,
                     sh.h * sh.w, sh.groups, relu);
}

int head_check(const node_shape* sh, char* why, size_t why_len) {
  if (!sh) { snprintf(why, why_len, "shape is NULL"); return NODE_ERR_NULL; }
  if (sh->n <= 0 || sh->c <= 0 || sh->h <= 0 || sh->w <= 0 || sh->groups <= 0) { snprintf(why, why_len, "non-positive dimension"); return NODE_ERR_SHAPE; }
  if (sh->c % sh->groups != 0) { snprintf(why, why_len, "groups (%d) must divide channels (%d)", sh->groups, sh->c); return NODE_ERR_SHAPE; }
  const int cpg = sh->c / sh->groups;
  if (sh->c > 256 && 256 % cpg != 0) { snprintf(why, why_len, "C = %d with %d channels per group: groups would straddle the 256-channel passes", sh->c, cpg); return NODE_ERR_UNSUPPORTED; }
  if ((size_t)sh->c * sizeof(float) > 24 * 1024) { snprintf(why, why_len, "C = %d does not fit the LDS scratch", sh->c); return NODE_ERR_UNSUPPORTED; }
  return NODE_OK;
}

// lanes per channel of the vector kernels, or 0: the sample must fit HEAD_MAXIT float4 per thread
static int head_lpc(const node_shape& sh) {
  const int HW = sh.h * sh.w;
  if (HW % 4 != 0) return 0;
  const int q4 = HW / 4;
  if (q4 != 1 && q4 != 4 && q4 != 16 && q4 != 64) return 0;
  const int cpp = 256 / q4;
  if ((sh.c + cpp - 1) / cpp > HEAD_MAXIT) return 0;
  return q4;
}

void launch_head_fwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, const float* scale,
                     float* pooled, float* stats, hipStream_t s) {
  const int HW = sh.h * sh.w;
  const size_t lds = (size_t)(sh.c + 2 * sh.groups) * sizeof(float);
#define HEAD_FWD(LPC) hipLaunchKernelGGL((k_head_fwd_v<LPC>), dim3(sh.n), dim3(256), lds, s, z, gamma, beta, scale, pooled, stats, sh.c, HW, sh.groups, sh.eps)
  switch (head_lpc(sh)) {
    case 1: HEAD_FWD(1); return;
    case 4: HEAD_FWD(4); return;
    case 16: HEAD_FWD(16); return;
    case 64: HEAD_FWD(64); return;
    default: break;
  }
#undef HEAD_FWD
  hipLaunchKernelGGL(k_head_fwd_g, dim3(sh.n), dim3(256), (size_t)sh.c * sizeof(float), s, z, gamma, beta, scale, pooled, stats,
                     sh.c, HW, sh.groups, sh.eps);
}
// gsum[j] = sum_n gpart[n][j], j < 2 C: the per-sample (dgamma, dbeta) partials of a backward pass summed over the batch in
// a fixed order (the caller's `gpart.sum(0)`: an ATen reduction launch + three indexing operators on the host)
__global__ __launch_bounds__(256) void k_head_gsum(const float* __restrict__ gpart, float* __restrict__ gsum, int N, int W) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float s = 0.f;
  if (col < W)
    for (int n = part; n < N; n += 4) s += gpart[(size_t)n * W + col];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && col < W) gsum[col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
void launch_head_gsum(const float* gpart, float* gsum, int N, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_head_gsum, dim3((2 * C + 63) / 64), dim3(256), 0, s, gpart, gsum, N, 2 * C);
}

void launch_head_bwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, const float* scale,
                     const float* stats, const float* gpool, float* dz, float* gpart, hipStream_t s) {
  const int HW = sh.h * sh.w;
  const size_t lds = (size_t)(2 * sh.c + 2 * sh.groups) * sizeof(float);
#define HEAD_BWD(LPC) hipLaunchKernelGGL((k_head_bwd_v<LPC>), dim3(sh.n), dim3(256), lds, s, z, gamma, beta, scale, stats, gpool, dz, gpart, sh.c, HW, sh.groups)
  switch (head_lpc(sh)) {
    case 1: HEAD_BWD(1); return;
    case 4: HEAD_BWD(4); return;
    case 16: HEAD_BWD(16); return;
    case 64: HEAD_BWD(64); return;
    default: break;
  }
#undef HEAD_BWD
  hipLaunchKernelGGL(k_head_bwd_g, dim3(sh.n), dim3(256), (size_t)sh.c * sizeof(float), s, z, gamma, beta, scale, stats, gpool, dz,
                     gpart, sh.c, HW, sh.groups);
}

}  // namespace node
