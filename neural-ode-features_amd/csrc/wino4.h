// Winograd F(4x4, 3x3) as a three-stage pipeline (gfx950 only): the 3x3 convolutions of an 8x8 state whose
// solver tolerance leaves room for the transform's rounding (see Solver::choose_w4, node_api.hip).
//
//   V = B^T d B   6x6 input transform of every 4x4 output tile's 6x6 patch -- written by the PRODUCER of the conv
//                 input (k_combine_gn, k_gn_bwd: w4_emit_v below), not by the conv
//   M_c = V_c U_c 36 independent [rows x C] x [C x C] products, rows = samples x 4 tiles: k_w4_gemm64 (fp32 MFMA,
//                 operands straight from L2 into registers in MFMA-ready blocks, no LDS, no transform in the loop)
//   Y = A^T M A   4x4 output transform -- done by the CONSUMER (the GroupNorm pass behind the conv: w4_load_tile)
//
// 36 multiplies per 16 outputs = 0.25 of the direct convolution's (F(2x2,3x3): 0.444).  Interpolation points
// (0, 1, -1, 1/2, -2, inf): max error 3.2e-6 of max|y| at C = 256 against an fp64 direct convolution (the textbook
// points (0, +-1, +-2, inf): 8.4e-6; F(2x2,3x3): 4.9e-7; tools/wino_error.py).
//
// Layouts (H = W = 8: T = 4 tiles per sample, R = 4 N rows, RB = N / 8 row blocks of 32; G8 = C / 8):
//   V  [comp 36][rb][g G8][s 8][hi 2][t 4][e 4]   channel = 8 g + 4 hi + e, sample = 8 rb + s: one contiguous 1 KB
//      block per (comp, rb, g) = ONE 16-B load per lane for four MFMA k-steps; a producer workgroup (one sample,
//      32 channels) writes whole 128-B lines
//   U  [comp 36][cb C/32][g G8][hi 2][col 32][e 4]   the same for the filter operand (k_w4_pack, once per solve)
//   M  [comp 36][row R][C]
#pragma once
#include "node_internal.h"

namespace node {

constexpr int W4_COMPS = 36;
constexpr int W4_SCRATCH = 36 * 4 * 36;   // floats of LDS scratch the two helpers below need (they may share it)

// B^T (6x6), A^T (4x6), G (6x3) for the points (0, 1, -1, 1/2, -2, inf)
__device__ constexpr float W4_BT[6][6] = {{1.f, -1.5f, -2.f, 1.5f, 1.f, 0.f}, {0.f, -1.f, 0.5f, 2.5f, 1.f, 0.f},
                                          {0.f, 1.f, -2.5f, 0.5f, 1.f, 0.f},  {0.f, -2.f, -1.f, 2.f, 1.f, 0.f},
                                          {0.f, 0.5f, -1.f, -0.5f, 1.f, 0.f}, {0.f, 1.f, -1.5f, -2.f, 1.5f, 1.f}};
__device__ constexpr float W4_AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                                          {0.f, 1.f, -1.f, 0.5f, -2.f, 0.f},
                                          {0.f, 1.f, 1.f, 0.25f, 4.f, 0.f},
                                          {0.f, 1.f, -1.f, 0.125f, -8.f, 1.f}};
__device__ constexpr double W4_G[6][3] = {{1.0, 0.0, 0.0},
                                          {1.0 / 3, 1.0 / 3, 1.0 / 3},
                                          {-1.0 / 3, 1.0 / 3, -1.0 / 3},
                                          {-16.0 / 15, -8.0 / 15, -4.0 / 15},
                                          {1.0 / 15, -2.0 / 15, 4.0 / 15},
                                          {0.0, 0.0, 1.0}};

struct W4Geom {
  int N, C, RB, G8, R;   // R = 4 N rows, RB = N / 8, G8 = C / 8
};
__host__ __device__ inline W4Geom w4_geom(int N, int C) {
  W4Geom g;
  g.N = N; g.C = C; g.RB = N / 8; g.G8 = C / 8; g.R = 4 * N;
  return g;
}
constexpr int W4_SLACK = 16 * 256;         // floats behind V and U that k_w4_gemm's operand ring may read (never uses)
__host__ __device__ inline size_t w4_v_elems(int N, int C) { return (size_t)W4_COMPS * 4 * N * C + W4_SLACK; }
__host__ __device__ inline size_t w4_u_elems(int C) { return (size_t)W4_COMPS * C * C + W4_SLACK; }

// Input transform of one sample's [64 px][32 ch] slab held in LDS (`tile`, row stride `ld` floats) into the blocked
// V layout.  256 threads; `scratch` = W4_SCRATCH floats of LDS.  The caller has synchronised after writing `tile`;
// the function ends without a barrier (it only reads `tile`).
__device__ inline void w4_emit_v(const float* tile, int ld, int n, int c0, float* __restrict__ V, const W4Geom gm,
                                 float* scratch, int tid) {
  const int c = tid & 31, t = (tid >> 5) & 3, h = tid >> 7;
  const int y0 = 4 * (t >> 1) - 1, x0 = 4 * (t & 1) - 1;
  float w[6][6];   // W = d B:  W[j][l] = sum_k d[j][k] B^T[l][k]
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float dr[6];
    const int y = y0 + j;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int x = x0 + k;
      const bool in = y >= 0 && y < 8 && x >= 0 && x < 8;
      dr[k] = in ? tile[(y * 8 + x) * ld + c] : 0.f;
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (W4_BT[l][k] != 0.f) s += W4_BT[l][k] * dr[k];
      w[j][l] = s;
    }
  }
  const int gq = c >> 3, hi = (c >> 2) & 1, e = c & 3;
#pragma unroll
  for (int ii = 0; ii < 3; ++ii) {
    // rows 3h .. 3h+2 of V = B^T W (h is uniform per half of the workgroup; both variants are unrolled)
#pragma unroll
    for (int l = 0; l < 6; ++l) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (W4_BT[ii][j] != 0.f) s0 += W4_BT[ii][j] * w[j][l];
        if (W4_BT[ii + 3][j] != 0.f) s1 += W4_BT[ii + 3][j] * w[j][l];
      }
      const int i = ii + 3 * h;
      scratch[((i * 6 + l) * 4 + gq) * 36 + hi * 16 + t * 4 + e] = h ? s1 : s0;
    }
  }
  __syncthreads();
  const int rb = n >> 3, s = n & 7;
  for (int u = tid; u < W4_COMPS * 32; u += 256) {
    const int comp = u >> 5, g4 = (u >> 3) & 3, q = u & 7;   // q = hi * 4 + t
    const float4 v = *reinterpret_cast<const float4*>(scratch + (comp * 4 + g4) * 36 + q * 4);
    const size_t dst = (((((size_t)comp * gm.RB + rb) * gm.G8 + (c0 >> 3) + g4) * 8 + s) * 8 + q) * 4;
    *reinterpret_cast<float4*>(V + dst) = v;
  }
}

// Output transform: the 36 component rows of one sample's four tiles, channels [c0, c0 + 32), -> `tile`
// [64 px][ld] (+ bias[c] + tval * tmap[p][c] when bias != nullptr).  256 threads; `scratch` = W4_SCRATCH floats.
// Ends WITHOUT a barrier: the caller synchronises before reading `tile`.
__device__ inline void w4_load_tile(const float* __restrict__ M, int n, int c0, const W4Geom gm, float* tile, int ld,
                                    float* scratch, const float* __restrict__ bias, const float* __restrict__ tmap,
                                    float tval, int tid) {
  // all of a thread's requests first (4.5 x 16 B), then the LDS writes: written as one loop the compiler waits for
  // every request before it issues the next -- five dependent round trips at the head of every pass
  constexpr int NU = W4_COMPS * 32;
  float4 mv[5];
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const int u = min(tid + it * 256, NU - 1);   // clamped, not predicated: a masked request makes the compiler wait
    const int comp = u >> 5, t = (u >> 3) & 3, q = u & 7;
    mv[it] = *reinterpret_cast<const float4*>(M + ((size_t)comp * gm.R + 4 * n + t) * gm.C + c0 + 4 * q);
  }
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const int u = min(tid + it * 256, NU - 1);   // (the clamped duplicates rewrite the last unit with its own value)
    const int comp = u >> 5, t = (u >> 3) & 3, q = u & 7;
    *reinterpret_cast<float4*>(scratch + (comp * 4 + t) * 32 + 4 * q) = mv[it];
  }
  __syncthreads();
  const int c = tid & 31, t = (tid >> 5) & 3, h = tid >> 7;
  float z[2][6];   // rows 2h, 2h+1 of A^T M
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float m[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = scratch[((j * 6 + k) * 4 + t) * 32 + c];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (W4_AT[ii][j] != 0.f) s0 += W4_AT[ii][j] * m[j];
        if (W4_AT[ii + 2][j] != 0.f) s1 += W4_AT[ii + 2][j] * m[j];
      }
      z[ii][k] = h ? s1 : s0;
    }
  }
  const float bv = bias ? bias[c0 + c] : 0.f;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int y = 4 * (t >> 1) + ii + 2 * h;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (W4_AT[l][k] != 0.f) s += z[ii][k] * W4_AT[l][k];
      const int p = y * 8 + 4 * (t & 1) + l;
      if (bias) s += bv + tval * tmap[(size_t)p * gm.C + c0 + c];
      tile[p * ld + c] = s;
    }
  }
}

// launchers (kernels_w4.hip)
struct W4PackJobs { const float* w[4]; float* u[4]; int dgrad[4]; };
void launch_w4_pack(const W4PackJobs& jobs, int count, int C, hipStream_t s);
void launch_w4_gemm(const float* V, const float* U, float* M, const Ctrl* ctrl, int N, int C, hipStream_t s);
// stand-alone transforms (diagnostics: node_conv3x3_w4)
void launch_w4_input(const float* x_nhwc, float* V, int N, int C, hipStream_t s);
void launch_w4_output(const float* M, float* y_nhwc, int N, int C, hipStream_t s);

}  // namespace node
