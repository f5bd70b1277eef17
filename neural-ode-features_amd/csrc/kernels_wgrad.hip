// Weight-gradient kernels of the adjoint dynamics (gfx950).
//
//   dW[tap][ci][co] = sum_{n,p} act[n, p + tap, ci] * dz[n, p, co]        K = N*H*W pixels
//
// 64 ci x 64 co x 9 taps per workgroup; 4 waves (one per SIMD), each holding the
// 32x32 tile of all nine taps = 144 accumulator registers; K split over `nsplit`
// ranges of (sample, row band) units; deterministic partial slabs, reduced by
// k_theta_finalize.  fp32 v_mfma_f32_32x32x2_f32 at 64 cycles per MFMA is the bound,
// and one wave per SIMD has nobody to hide behind, so the inner loop is built to have
// NOTHING but MFMAs and LDS reads in it:
//  * nine independent accumulator chains: consecutive MFMAs never depend on each other;
//  * operands of pixel pair k+1 are read while the nine MFMAs of pair k execute, in
//    the order the next step consumes them (LDS returns in order: counted lgkmcnt);
//  * k_wgrad_t<W, RB>: the image geometry is a template parameter, so every LDS
//    address in the loop is `per-lane base + immediate` -- zero VALU / SALU
//    instructions between the MFMAs.  Measured (tools/mfma_rate2): address
//    arithmetic interleaved with 32x32x2 MFMAs costs 10+ cycles of matrix-pipe time
//    per VALU instruction (85.9 vs 64.2 cycles per MFMA for the table-driven step).
//    Pixel pairs never straddle image rows (odd W gets one zero pad pixel per row);
//  * units are double-buffered in LDS: the global loads of unit u+1 are issued before
//    the MFMAs of unit u and written to the other buffer after them.
// k_wgrad_p is the geometry-generic fallback (slot table + five VALU per step).
//
// The masked column sums of dz (conv-bias, time-channel-weight and d/dt terms) are a
// separate HBM-bound kernel (k_colsum): inside the GEMM they unbalanced a quarter of
// the workgroups by 50 %.
#include "node_internal.h"
#include <cstdlib>

namespace node {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_wgrad_variant = -1;
int g_wgrad_wino = -1;

#ifdef NODE_STAMPS
#define WSTAMP(buf, slot, INS)                                                                   \
  do {                                                                                           \
    if ((buf) != nullptr && (threadIdx.x & 63) == 0) {                                           \
      unsigned long long _t;                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      asm volatile(INS " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                       \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      (buf)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = _t; \
    }                                                                                            \
  } while (0)
#else
#define WSTAMP(buf, slot, INS) do { } while (0)
#endif

constexpr int WG_MAXA = 6;   // float4 staging units per thread: activations ((RB + 2) * W * 16 <= 6 * 256)
constexpr int WG_MAXZ = 4;   // dz (RB * W * 16 <= 4 * 256)

// ----------------------------------------------------------------------------
// shared pieces: staging of one unit, slab store
// ----------------------------------------------------------------------------
struct WgGeom {
  int ci0, co0, sp, wi, wj, l31, hi, tid, q16;
  bool ci_ok, co_ok;
};

template <int NTAP>
__device__ inline void wg_store_slab(const WgradArgs& a, const Dims& d, const WgGeom& g, const f32x16 (&acc)[NTAP]) {
  const size_t CC = (size_t)d.C * d.C;
  float* wp = a.wpart + (size_t)g.sp * NTAP * CC;
  const int co = g.co0 + g.wj * 32 + g.l31;
  if (co < d.C) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = g.ci0 + g.wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g.hi;
        if (ci < d.C) wp[(size_t)t * CC + (size_t)ci * d.C + co] = acc[t][r];
      }
    }
  }
}

// ============================================================================
// k_wgrad_t<W, RB>: geometry-templated, immediate-offset inner loop
//   requires H % RB == 0, (RB + 2) * W * 16 <= 6 * 256, RB * WZ * 16 <= 4 * 256
// ============================================================================
template <int W, int RB>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_t(WgradArgs a, Dims d) {
  if (a.ctrl != nullptr && a.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)

  WSTAMP(a.stamps, 0, "s_memrealtime");
  WSTAMP(a.stamps, 1, "s_memtime");
  constexpr int WP = W + 2;
  constexpr int MARGIN = WP + 1;
  constexpr int NJ = (W + 1) / 2;          // pixel pairs per image row
  constexpr int WZ = 2 * NJ;               // dz row length in LDS (odd W: one zero pad pixel)
  constexpr int AROWS = (RB + 2) * WP + 2 * MARGIN;
  constexpr int ASZ = AROWS * 64, ZSZ = RB * WZ * 64;
  constexpr int NUA = (RB + 2) * W * 16, NUZ = RB * W * 16;
  static_assert(NUA <= WG_MAXA * WG_THREADS && NUZ <= WG_MAXZ * WG_THREADS, "staging registers");
  static_assert(((RB + 2) * WP + WZ + 2) * 256 < 65536, "LDS immediates");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WgGeom g;
  g.tid = tid; g.l31 = lane & 31; g.hi = lane >> 5; g.wi = wave >> 1; g.wj = wave & 1;
  const int ntc = (d.C + 63) / 64;
  const int ci_t = blockIdx.x / ntc, co_t = blockIdx.x - ci_t * ntc;
  g.ci0 = ci_t * 64; g.co0 = co_t * 64; g.sp = blockIdx.y; g.q16 = tid & 15;
  g.ci_ok = g.ci0 + g.q16 * 4 < d.C; g.co_ok = g.co0 + g.q16 * 4 < d.C;

  float* As0 = smem;            // 2 x [AROWS][64]
  float* Zs0 = smem + 2 * ASZ;  // 2 x [RB * WZ][64]
  for (int i = tid * 4; i < 2 * ASZ + 2 * ZSZ; i += WG_THREADS * 4)
    *reinterpret_cast<float4*>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nbands = d.H / RB;
  const int U = d.N * nbands;
  const int u_begin = (int)(((long long)g.sp * U) / d.nsplit);
  const int u_end = (int)(((long long)(g.sp + 1) * U) / d.nsplit);

  // staging descriptors: unit v -> (pixel, float4 column); all geometry compile-time
  float4 ra[WG_MAXA], rz[WG_MAXZ];
  auto stage_load = [&](int u) {
    const int n = u / nbands, band = u - n * nbands;
    const int row0 = band * RB;
#pragma unroll
    for (int i = 0; i < WG_MAXA; ++i) {
      const int v = tid + i * WG_THREADS;
      const int px = v >> 4, hr = px / W, x = px - hr * W;
      const int ih = row0 + hr - 1;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < NUA && ih >= 0 && ih < d.H && g.ci_ok)
        ra[i] = *reinterpret_cast<const float4*>(a.act + ((size_t)n * d.HW + ih * W + x) * d.C + g.ci0 + g.q16 * 4);
    }
#pragma unroll
    for (int i = 0; i < WG_MAXZ; ++i) {
      const int v = tid + i * WG_THREADS;
      rz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < NUZ && g.co_ok)
        rz[i] = *reinterpret_cast<const float4*>(a.dz + ((size_t)n * d.HW + row0 * W + (v >> 4)) * d.C + g.co0 + g.q16 * 4);
    }
  };
  auto stage_write = [&](int buf) {
    float* As = As0 + buf * ASZ;
    float* Zs = Zs0 + buf * ZSZ;
#pragma unroll
    for (int i = 0; i < WG_MAXA; ++i) {
      const int v = tid + i * WG_THREADS;
      const int px = v >> 4, hr = px / W, x = px - hr * W;
      if (v < NUA) *reinterpret_cast<float4*>(As + (MARGIN + hr * WP + x + 1) * 64 + g.q16 * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < WG_MAXZ; ++i) {
      const int v = tid + i * WG_THREADS;
      const int px = v >> 4, r = px / W, x = px - r * W;
      if (v < NUZ) *reinterpret_cast<float4*>(Zs + (r * WZ + x) * 64 + g.q16 * 4) = rz[i];
    }
  };

  __syncthreads();  // zero fill visible
  if (u_begin < u_end) {
    stage_load(u_begin);
    stage_write(0);
  }
  __syncthreads();
  WSTAMP(a.stamps, 2, "s_memtime");

  // per-lane bases; slot of tap (kh, kw) of pixel (r, x) = MARGIN + (r + kh) * WP + x + kw
  const int abase = (MARGIN + g.hi) * 64 + g.wi * 32 + g.l31;
  const int zbase = g.hi * 64 + g.wj * 32 + g.l31;
  int buf = 0;
  for (int u = u_begin; u < u_end; ++u) {
    const float* As = As0 + buf * ASZ + abase;
    const float* Zs = Zs0 + buf * ZSZ + zbase;
    const bool more = (u + 1) < u_end;
    if (more) stage_load(u + 1);  // in flight during this unit's MFMAs

    float av[2][9], bv[2];
    bv[0] = Zs[0];
#pragma unroll
    for (int t = 0; t < 9; ++t) av[0][t] = As[((t / 3) * WP + (t % 3)) * 64];
#pragma unroll
    for (int s = 0; s < RB * NJ; ++s) {
      constexpr int NS = RB * NJ;
      const int cur = s & 1, nxt = cur ^ 1;
      const int sn = s + 1 < NS ? s + 1 : s;            // (last step re-reads itself: no branch)
      const int rn = sn / NJ, jn = sn - rn * NJ;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][t], bv[cur], acc[t], 0, 0, 0);
        if (t == 0) bv[nxt] = Zs[(rn * WZ + 2 * jn) * 64];
        av[nxt][t] = As[((rn + t / 3) * WP + 2 * jn + (t % 3)) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) stage_write(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  WSTAMP(a.stamps, 3, "s_memtime");
  wg_store_slab<9>(a, d, g, acc);
  WSTAMP(a.stamps, 4, "s_memtime");
  WSTAMP(a.stamps, 5, "s_memrealtime");
}

// ============================================================================
// k_wgrad_w<W, RB>: the weight gradient in the 1-D Winograd F(2,3) domain of k_conv3x3_w
//   dU_j[kh][ci][co] = sum_{n,h,t} V_j[n, h + kh - 1, t, ci] * Z_j[n, h, t, co]        j = 0..3, kh = 0..2
//   V = input transform of the activations (as in the forward kernel), Z = transposed output transform
//   of dz:  Z0 = dy0, Z1 = dy0 + dy1, Z2 = dy0 - dy1, Z3 = -dy1   (dy0, dy1 = the pixel pair of tile t)
//   and, before the slab is stored,  dW[kh][0] = dU0 + (dU1+dU2)/2, dW[kh][1] = (dU1-dU2)/2, dW[kh][2] = dU3 + (dU1+dU2)/2.
// K = tile-rows (half the pixels), 12 "taps" instead of 9: 1.5 x fewer MFMAs.  Same structure as
// k_wgrad_t: 4 waves x 32x32 x 12 accumulators (192 AGPRs), immediate-offset inner loop, operands one
// step ahead in consumption order, double-buffered units; both transforms happen at staging time.
// Requires H % RB == 0 and an even number of column pairs (W % 4 == 0).
// ============================================================================
template <int W, int RB>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_w(WgradArgs a, Dims d) {
  if (a.ctrl != nullptr && a.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)

  WSTAMP(a.stamps, 0, "s_memrealtime");
  WSTAMP(a.stamps, 1, "s_memtime");
  constexpr int NT = W / 2;                  // column pairs per image row
  constexpr int VROWS = RB + 2;              // band rows with one halo row above and below
  constexpr int VSZ = VROWS * NT * 4 * 64;   // V image: [row][t][j][64 ci]
  constexpr int ZSZ = RB * NT * 4 * 64;      // Z image: [row][t][j][64 co]
  constexpr int NUV = VROWS * NT * 16;       // staging units (slot, float4 column) of the V image
  constexpr int NUZ = RB * NT * 16;
  constexpr int MAXV = (NUV + WG_THREADS - 1) / WG_THREADS;
  constexpr int MAXZ = (NUZ + WG_THREADS - 1) / WG_THREADS;
  static_assert(NT % 2 == 0, "pairs of tile-rows must not straddle image rows");
  static_assert((VROWS * NT * 4 * 64) * 4 < 65536, "LDS immediates");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WgGeom g;
  g.tid = tid; g.l31 = lane & 31; g.hi = lane >> 5; g.wi = wave >> 1; g.wj = wave & 1;
  const int ntc = (d.C + 63) / 64;
  const int ci_t = blockIdx.x / ntc, co_t = blockIdx.x - ci_t * ntc;
  g.ci0 = ci_t * 64; g.co0 = co_t * 64; g.sp = blockIdx.y; g.q16 = tid & 15;
  g.ci_ok = g.ci0 + g.q16 * 4 < d.C; g.co_ok = g.co0 + g.q16 * 4 < d.C;

  float* Vs0 = smem;            // 2 x VSZ
  float* Zs0 = smem + 2 * VSZ;  // 2 x ZSZ

  f32x16 acc[12];
#pragma unroll
  for (int t = 0; t < 12; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nbands = d.H / RB;
  const int U = d.N * nbands;
  const int u_begin = (int)(((long long)g.sp * U) / d.nsplit);
  const int u_end = (int)(((long long)(g.sp + 1) * U) / d.nsplit);

  float4 rv[MAXV][4], rz[MAXZ][2];
  auto stage_load = [&](int u) {
    const int n = u / nbands, band = u - n * nbands;
    const int row0 = band * RB;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = tid + i * WG_THREADS;
      const int slot = v >> 4, vr = slot / NT, t = slot - vr * NT;
      const int ih = row0 + vr - 1;
      const bool ok = v < NUV && ih >= 0 && ih < d.H && g.ci_ok;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = 2 * t - 1 + e;
        rv[i][e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && x >= 0 && x < W)
          rv[i][e] = *reinterpret_cast<const float4*>(a.act + ((size_t)n * d.HW + ih * W + x) * d.C + g.ci0 + g.q16 * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXZ; ++i) {
      const int v = tid + i * WG_THREADS;
      const int slot = v >> 4, zr = slot / NT, t = slot - zr * NT;
      const bool ok = v < NUZ && g.co_ok;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        rz[i][e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) rz[i][e] = *reinterpret_cast<const float4*>(a.dz + ((size_t)n * d.HW + (row0 + zr) * W + 2 * t + e) * d.C + g.co0 + g.q16 * 4);
      }
    }
  };
  auto f4 = [](float x, float y, float z, float w) { return make_float4(x, y, z, w); };
  auto stage_write = [&](int buf) {
    float* Vs = Vs0 + buf * VSZ;
    float* Zs = Zs0 + buf * ZSZ;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = tid + i * WG_THREADS;
      if (v < NUV) {
        const float4 d0 = rv[i][0], d1 = rv[i][1], d2 = rv[i][2], d3 = rv[i][3];
        float* dst = Vs + (v >> 4) * 256 + g.q16 * 4;
        *reinterpret_cast<float4*>(dst) = f4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        *reinterpret_cast<float4*>(dst + 64) = f4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        *reinterpret_cast<float4*>(dst + 128) = f4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        *reinterpret_cast<float4*>(dst + 192) = f4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXZ; ++i) {
      const int v = tid + i * WG_THREADS;
      if (v < NUZ) {
        const float4 y0 = rz[i][0], y1 = rz[i][1];
        float* dst = Zs + (v >> 4) * 256 + g.q16 * 4;
        *reinterpret_cast<float4*>(dst) = y0;
        *reinterpret_cast<float4*>(dst + 64) = f4(y0.x + y1.x, y0.y + y1.y, y0.z + y1.z, y0.w + y1.w);
        *reinterpret_cast<float4*>(dst + 128) = f4(y0.x - y1.x, y0.y - y1.y, y0.z - y1.z, y0.w - y1.w);
        *reinterpret_cast<float4*>(dst + 192) = f4(-y1.x, -y1.y, -y1.z, -y1.w);
      }
    }
  };

  if (u_begin < u_end) {
    stage_load(u_begin);
    stage_write(0);
  }
  __syncthreads();
  WSTAMP(a.stamps, 2, "s_memtime");

  // per-lane bases: tile-row (h, t = 2 tau + hi); V slot of row tap kh = ((h + kh) * NT + t), Z slot = h * NT + t
  const int vbase = g.hi * 256 + g.wi * 32 + g.l31;
  const int zbase = g.hi * 256 + g.wj * 32 + g.l31;
  int buf = 0;
  for (int u = u_begin; u < u_end; ++u) {
    const float* Vs = Vs0 + buf * VSZ + vbase;
    const float* Zs = Zs0 + buf * ZSZ + zbase;
    const bool more = (u + 1) < u_end;
    if (more) stage_load(u + 1);  // in flight during this unit's MFMAs

    float av[2][12], bv[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bv[0][j] = Zs[j * 64];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) av[0][kh * 4 + j] = Vs[(kh * NT) * 256 + j * 64];
    }
    constexpr int NS = RB * NT / 2;   // steps: two tile-rows each
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      const int sn = s + 1 < NS ? s + 1 : s;           // (last step re-reads itself: no branch)
      const int hn = (2 * sn) / NT, tn = 2 * sn - hn * NT;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          acc[kh * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][kh * 4 + j], bv[cur][j], acc[kh * 4 + j], 0, 0, 0);
          // reads of the next step in the order it consumes them: Z_j first, then the three row taps of V_j
          if (kh == 0) bv[nxt][j] = Zs[((hn * NT + tn) * 4 + j) * 64];
          av[nxt][kh * 4 + j] = Vs[(((hn + kh) * NT + tn) * 4 + j) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (more) stage_write(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  WSTAMP(a.stamps, 3, "s_memtime");
  // back to the nine filter taps before the slab leaves the registers (G^T applied per row tap kh):
  //   dW[kh][0] = dU0 + (dU1 + dU2)/2,  dW[kh][1] = (dU1 - dU2)/2,  dW[kh][2] = dU3 + (dU1 + dU2)/2
  f32x16 wacc[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float u0 = acc[kh * 4 + 0][r], u1 = acc[kh * 4 + 1][r], u2 = acc[kh * 4 + 2][r], u3 = acc[kh * 4 + 3][r];
      const float hs = 0.5f * (u1 + u2);
      wacc[kh * 3 + 0][r] = u0 + hs;
      wacc[kh * 3 + 1][r] = 0.5f * (u1 - u2);
      wacc[kh * 3 + 2][r] = u3 + hs;
    }
  wg_store_slab<9>(a, d, g, wacc);
  WSTAMP(a.stamps, 4, "s_memtime");
  WSTAMP(a.stamps, 5, "s_memrealtime");
}

// ============================================================================
// k_wgrad_w2<UT>: the weight gradient in the 2-D Winograd F(2x2, 3x3) domain of k_conv3x3_w2
//   M[a][b][ci][co] = sum over 2x2 tiles of  V[a][b][tile][ci] * Z[a][b][tile][co]        a, b = 0..3
//   V = B^T d B   of the 4x4 activation patch (the forward kernel's input transform),
//   Z = A dy A^T  of the 2x2 dz tile  (per dimension: z0 = y0, z1 = y0 + y1, z2 = y0 - y1, z3 = -y1),
//   dW = G^T M G  before the slab leaves the registers (per dimension: w0 = m0 + (m1+m2)/2, w1 = (m1-m2)/2,
//   w2 = m3 + (m1+m2)/2).  K = tiles (a quarter of the pixels) x 16 components: 16/36 of the direct MFMA work
//   (the 1-D kernel above: 24/36).
// Eight waves, two per SIMD: wave (h, wi, wj) holds the 32x32 block (wi, wj) of the eight components
// a in {2h, 2h+1} (128 AGPRs).  With one wave per SIMD every staging instruction -- request, transform,
// LDS write -- was paid in matrix-pipe time (measured: 7050 cycles per 4096-cycle unit); here each wave's
// staging overlaps its partner's MFMAs, and all of it is cut into single items placed between the MFMAs:
// the first half of a unit transforms and writes the next unit (its raw pixels were requested during the
// previous unit), the second half requests the unit after that into the registers just freed.  A unit is UT
// consecutive tiles of one sample (whole tile rows); images [tile][component][64 channels] with a component
// stride of 68 floats (conflict-free 16-B staging writes); operands by ds_read_b32 at immediate offsets,
// one step (tile pair) ahead; units double-buffered, one barrier per unit.  The two h halves meet through
// LDS once, after the loop: each wave finishes (and stores) half of the 32x32 block.
// Requires even H, W; UT % (W/2) == 0; a tail of C zeros behind act and dz.
// ============================================================================
constexpr int WG2_THREADS = 512;
template <int UT>
__global__ __launch_bounds__(WG2_THREADS) void k_wgrad_w2(WgradArgs a, Dims d) {
  if (a.ctrl != nullptr && a.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  if (blockIdx.z != 0) { a.act = a.act2; a.dz = a.dz2; a.wpart = a.wpart2; }   // second layer of a paired launch

  WSTAMP(a.stamps, 0, "s_memrealtime");
  WSTAMP(a.stamps, 1, "s_memtime");
  constexpr int CS = 68;              // component stride
  constexpr int TS = 16 * CS;         // tile stride
  constexpr int IMG = UT * TS;        // one image (V or Z) of one unit
  static_assert(UT == 4 || UT == 8, "unit sizes with an instance");
  static_assert((UT * TS) * 4 < 65536, "LDS immediates");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int ntc = (d.C + 63) / 64;
  // XCD-aware order (speed only; blocks b and b + 8 share an XCD and its L2): all output tiles of one K split
  // read the same pixels, so a split's tiles go to one XCD instead of being dealt over all eight
  // (HBM reads per launch at cfg 2: 50.7 -> 17.1 MB, the algorithmic 16.8 MB)
  int tile = blockIdx.x, sp = blockIdx.y;
  {
    const int nt2 = gridDim.x, L = blockIdx.x + nt2 * blockIdx.y;
    if ((gridDim.y & 7) == 0) {
      const int j = L >> 3;
      sp = (L & 7) + 8 * (j / nt2);
      tile = j - (j / nt2) * nt2;
    }
  }
  const int ci_t = tile / ntc, co_t = tile - ci_t * ntc;
  const int ci0 = ci_t * 64, co0 = co_t * 64;

  float* Vs0 = smem;             // 2 x IMG
  float* Zs0 = smem + 2 * IMG;   // 2 x IMG

  f32x16 acc[8];   // component (2 wh + (c >> 2)) * 4 + (c & 3) = 8 wh + c
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int TW = d.W >> 1, TPS = (d.H >> 1) * TW;
  const int groups = TPS / UT, rows_per_unit = UT / TW;
  const int U = d.N * groups;
  // wave-uniform by construction, but the 64-bit division leaves them in vector registers and the whole unit
  // loop then runs on the VALU under an exec-mask loop: readfirstlane puts it on the scalar unit
  const int u_begin = __builtin_amdgcn_readfirstlane((int)(((long long)sp * U) / d.nsplit));
  const int u_end = __builtin_amdgcn_readfirstlane((int)(((long long)(sp + 1) * U) / d.nsplit));

  // ---- staging descriptors: the tile of a thread is its wave index (UT = 4: waves 4-7 stage nothing) ----
  // Requests are "scalar base of the unit + fixed 32-bit lane offset"; pixels outside the image read the row of
  // C zeros the host keeps behind both tensors.
  // V: thread = (tile, channel quad vq, patch row vr): lanes of a quad hold the four rows of one patch
  const int st = wave;   // staged tile
  const bool st_on = st < UT;
  const int vr = tid & 3, vq = (tid >> 2) & 15;
  const int s_thl = st / TW, s_tw = st - s_thl * TW;
  unsigned v_off[4];   // byte offset from (unit base = pixel (2 th0 - 1, -1), channel ci0)
  bool v_xok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int x = 2 * s_tw - 1 + e;
    v_xok[e] = st_on && x >= 0 && x < d.W && ci0 + vq * 4 < d.C;
    v_off[e] = (unsigned)((((size_t)(2 * s_thl + vr) * d.W + (2 * s_tw + e)) * d.C + vq * 4) * sizeof(float));
  }
  const bool v_top = vr == 0 && s_thl == 0, v_bot = vr == 3 && s_thl == rows_per_unit - 1;
  // Z: thread = (tile, row-transform index za, channel quad zq); always inside the image
  const int zq = tid & 15, za = (tid >> 4) & 3;
  const bool z_on = st_on && co0 + zq * 4 < d.C;
  unsigned z_off[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    z_off[e] = (unsigned)((((size_t)(2 * s_thl + (e >> 1)) * d.W + 2 * s_tw + (e & 1)) * d.C + zq * 4) * sizeof(float));
  // R_a = zc0 * y0* + zc1 * y1*  (rows 1, -1 apart; row a = 3 is +y1*, the NEGATIVE of A's textbook row -y1*, to
  // match the negated row 3 of the V transform below: the product V[3][b] * Z[3][b] is unchanged)
  const float zc0 = za == 3 ? 0.f : 1.f, zc1 = za == 0 ? 0.f : za == 2 ? -1.f : 1.f;

  // next unit to request: (sample, group), advanced incrementally on the scalar unit; requests past the last
  // unit repeat it (harmless; every request and every write stays unconditional)
  const int TH = d.H >> 1;
  (void)TH;
  int ld_u = u_begin;
  int ld_n = __builtin_amdgcn_readfirstlane(u_begin / groups);
  int ld_g = __builtin_amdgcn_readfirstlane(u_begin - ld_n * groups);
  const char* vb = nullptr;
  const char* zb = nullptr;
  unsigned vzero = 0, zzero = 0;
  bool rbad = false;
  auto stage_prepare = [&]() {
    const int n = ld_n, th0 = ld_g * rows_per_unit;
    const bool first = ld_g == 0, last = ld_g == groups - 1;
    if (ld_u + 1 < u_end) {
      ++ld_u;
      if (++ld_g == groups) { ld_g = 0; ++ld_n; }
    }
    const ptrdiff_t vrel = (((ptrdiff_t)n * d.HW + (ptrdiff_t)(2 * th0 - 1) * d.W - 1) * d.C + ci0) * (ptrdiff_t)sizeof(float);
    const ptrdiff_t zrel = (((ptrdiff_t)n * d.HW + (ptrdiff_t)(2 * th0) * d.W) * d.C + co0) * (ptrdiff_t)sizeof(float);
    vb = reinterpret_cast<const char*>(a.act) + vrel;
    zb = reinterpret_cast<const char*>(a.dz) + zrel;
    vzero = (unsigned)((ptrdiff_t)(d.numel * sizeof(float)) - vrel) + vq * 16;
    zzero = (unsigned)((ptrdiff_t)(d.numel * sizeof(float)) - zrel) + zq * 16;
    rbad = (first && v_top) || (last && v_bot);
  };
  float4 rv[4], rz[4];
  auto load_piece = [&](int k) {   // k is a compile-time constant at every call site
    if (k < 4) rv[k] = *reinterpret_cast<const float4*>(vb + ((!rbad && v_xok[k]) ? v_off[k] : vzero));
    else rz[k - 4] = *reinterpret_cast<const float4*>(zb + (z_on ? z_off[k - 4] : zzero));
  };
  // (as in k_conv3x3_w2: row 3 negated -> every lane is "own + (+/-) other row": one DPP-fused XOR + one add per element)
  const int spmask = vr == 1 ? 0 : (int)0x80000000;
#define QP2(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), 0x5A /* quad_perm [2,2,1,1] */, 0xf, 0xf, true) ^ spmask)
  auto f4 = [](float x, float y, float z, float w) { return make_float4(x, y, z, w); };
  float4 zr0, zr1;   // the row-transformed dz pair of this thread (columns 0 / 1)
  // one staging item: k = 0..3 column nu = k of V = B^T d B (x transform of this lane's patch row, then the y
  // transform with ONE other row of the quad:  xi = 0: e(0) - e(2)  xi = 1: e(1) + e(2)  xi = 2: e(2) - e(1)
  // xi = 3: e(3) - e(1), lane r = xi owns e(r));  k = 4..7 column b = k - 4 of Z = A dy A^T
  auto write_piece = [&](int k, float* Vs, float* Zs) {
    if (!st_on) return;   // wave-uniform
    if (k < 4) {
      const int nu = k;
      const float4 pl = nu == 0 ? rv[0] : nu == 2 ? rv[2] : rv[1];
      const float4 pr = nu == 0 ? rv[2] : nu == 2 ? rv[1] : nu == 1 ? rv[2] : rv[3];
      const float sg = nu == 1 ? 1.f : -1.f;
      const float e0 = pl.x + sg * pr.x, e1 = pl.y + sg * pr.y, e2 = pl.z + sg * pr.z, e3 = pl.w + sg * pr.w;
      // (channel quad of patch rows 2, 3 stored at vq ^ 2: ds_write_b128 groups are 8 contiguous lanes on 32 banks and the
      //  rows of a quad sit 16 banks apart -- rows 0/2 and 1/3 collided; the readers of components a >= 2 flip channel bit 3)
      *reinterpret_cast<float4*>(Vs + st * TS + (vr * 4 + nu) * CS + (vq ^ (vr & 2)) * 4) =
          f4(e0 + QP2(e0), e1 + QP2(e1), e2 + QP2(e2), e3 + QP2(e3));
    } else {
      const int b = k - 4;
      if (b == 0) {
        zr0 = f4(zc0 * rz[0].x + zc1 * rz[2].x, zc0 * rz[0].y + zc1 * rz[2].y, zc0 * rz[0].z + zc1 * rz[2].z, zc0 * rz[0].w + zc1 * rz[2].w);
        zr1 = f4(zc0 * rz[1].x + zc1 * rz[3].x, zc0 * rz[1].y + zc1 * rz[3].y, zc0 * rz[1].z + zc1 * rz[3].z, zc0 * rz[1].w + zc1 * rz[3].w);
      }
      const float4 o = b == 0 ? zr0
                     : b == 1 ? f4(zr0.x + zr1.x, zr0.y + zr1.y, zr0.z + zr1.z, zr0.w + zr1.w)
                     : b == 2 ? f4(zr0.x - zr1.x, zr0.y - zr1.y, zr0.z - zr1.z, zr0.w - zr1.w)
                              : f4(-zr1.x, -zr1.y, -zr1.z, -zr1.w);
      *reinterpret_cast<float4*>(Zs + st * TS + (za * 4 + b) * CS + zq * 4) = o;
    }
  };
#define SBW __builtin_amdgcn_sched_barrier(0)

  // prologue: unit u_begin -> buffer 0, unit u_begin + 1 -> registers
  stage_prepare();
#pragma unroll
  for (int k = 0; k < 8; ++k) load_piece(k);
  SBW;
#pragma unroll
  for (int k = 0; k < 8; ++k) write_piece(k, Vs0, Zs0);
  SBW;
  stage_prepare();
#pragma unroll
  for (int k = 0; k < 8; ++k) load_piece(k);
  SBW;
  __syncthreads();
  WSTAMP(a.stamps, 2, "s_memtime");

  // per-lane bases: step s covers tiles 2 s + hi; component 8 wh + c at + c * CS
  const int vbase = hi * TS + (8 * wh) * CS + ((wi * 32 + l31) ^ (8 * wh));   // (wave half wh holds a = 2 wh, 2 wh + 1: see the staging write)
  const int zbase = hi * TS + (8 * wh) * CS + wj * 32 + l31;
  int buf = 0;
#ifdef NODE_STAMPS
  unsigned long long tk_prev, tk_acc[4] = {0, 0, 0, 0};
#define WTICK0 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_prev)::"memory")
#define WTICK(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tk_acc[k] += t_ - tk_prev; tk_prev = t_; } while (0)
#else
#define WTICK0 do { } while (0)
#define WTICK(k) do { } while (0)
#endif
  WTICK0;
  for (int u = u_begin; u < u_end; ++u) {
    const float* Vs = Vs0 + buf * IMG + vbase;
    const float* Zs = Zs0 + buf * IMG + zbase;
    float* Vw = Vs0 + (buf ^ 1) * IMG;
    float* Zw = Zs0 + (buf ^ 1) * IMG;

    float av[2][8], bv[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      av[0][c] = Vs[c * CS];
      bv[0][c] = Zs[c * CS];
    }
    constexpr int NS = UT / 2;
    constexpr int M = NS * 8;          // MFMAs per unit and wave
    constexpr int WGAP = (M / 2) / 8;  // MFMAs between two staging items (2 for UT = 8, 1 for UT = 4)
    WTICK(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      const int sn = s + 1 < NS ? s + 1 : s;           // (last step re-reads itself: no branch)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][c], bv[cur][c], acc[c], 0, 0, 0);
        av[nxt][c] = Vs[2 * sn * TS + c * CS];
        bv[nxt][c] = Zs[2 * sn * TS + c * CS];
        const int m = s * 8 + c;
        if (m < M / 2) {            // first half: unit u + 1 goes from the registers to the other buffer
          if (m % WGAP == 0) write_piece(m / WGAP, Vw, Zw);
        } else {                    // second half: unit u + 2 is requested into the registers just freed
          if (m == M / 2) stage_prepare();
          if ((m - M / 2) % WGAP == 0) load_piece((m - M / 2) / WGAP);
        }
        SBW;
      }
    }
    WTICK(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SBW;
    WTICK(2);
    buf ^= 1;
  }
#ifdef NODE_STAMPS
  if (a.stamps != nullptr && (threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k)
      a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + 12 + k] = tk_acc[k];
#endif
#undef WTICK0
#undef WTICK
#undef QP2
#undef SBW
  WSTAMP(a.stamps, 3, "s_memtime");

  // ---- dW = G^T M G.  Column transform (b -> kw) and this wave's share of the row transform (a -> kh) in
  //      registers; the two halves of a block meet in LDS: wave wh finishes and stores rows r in [8 wh, 8 wh + 8).
  //      G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1] ----
  float* X = smem + ((wi * 2 + wj) * 2) * (72 * 64);   // [destination half][9 taps x 8 rows][64 lanes]
  float own[72];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float t[2][3];
#pragma unroll
    for (int al = 0; al < 2; ++al) {
      const float m0 = acc[al * 4 + 0][r], m1 = acc[al * 4 + 1][r], m2 = acc[al * 4 + 2][r], m3 = acc[al * 4 + 3][r];
      const float hs = 0.5f * (m1 + m2);
      t[al][0] = m0 + hs;
      t[al][1] = 0.5f * (m1 - m2);
      t[al][2] = m3 + hs;
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float p0, p1, p2;
      if (wh == 0) {   // a = 0, 1
        p0 = t[0][kw] + 0.5f * t[1][kw]; p1 = 0.5f * t[1][kw]; p2 = 0.5f * t[1][kw];
      } else {         // a = 2, 3
        p0 = 0.5f * t[0][kw]; p1 = -0.5f * t[0][kw]; p2 = 0.5f * t[0][kw] + t[1][kw];
      }
      const bool mine = (r >> 3) == wh;   // wave-uniform per unrolled r
      const int rr = r & 7;
      if (mine) {
        own[(0 * 3 + kw) * 8 + rr] = p0; own[(1 * 3 + kw) * 8 + rr] = p1; own[(2 * 3 + kw) * 8 + rr] = p2;
      } else {
        float* dst = X + (1 - wh) * (72 * 64) + lane;
        dst[((0 * 3 + kw) * 8 + rr) * 64] = p0; dst[((1 * 3 + kw) * 8 + rr) * 64] = p1; dst[((2 * 3 + kw) * 8 + rr) * 64] = p2;
      }
    }
  }
  __syncthreads();
  {
    const size_t CC = (size_t)d.C * d.C;
    float* wp = a.wpart + (size_t)sp * 9 * CC;
    const int co = co0 + wj * 32 + l31;
    const float* src = X + wh * (72 * 64) + lane;
    if (co < d.C) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int r = 8 * wh + rr;
          const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (ci < d.C) wp[(size_t)t * CC + (size_t)ci * d.C + co] = own[t * 8 + rr] + src[(t * 8 + rr) * 64];
        }
    }
  }
  WSTAMP(a.stamps, 4, "s_memtime");
  WSTAMP(a.stamps, 5, "s_memrealtime");
}

// ============================================================================
// k_wgrad_p: geometry-generic fallback (slot table; five VALU instructions per step)
// ============================================================================
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_p(WgradArgs a, Dims d) {
  if (a.ctrl != nullptr && a.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)

  WSTAMP(a.stamps, 0, "s_memrealtime");
  WSTAMP(a.stamps, 1, "s_memtime");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WgGeom g;
  g.tid = tid; g.l31 = lane & 31; g.hi = lane >> 5; g.wi = wave >> 1; g.wj = wave & 1;
  const int l31 = g.l31, hi = g.hi, wi = g.wi, wj = g.wj;
  const int ntc = (d.C + 63) / 64;
  const int ci_t = blockIdx.x / ntc, co_t = blockIdx.x - ci_t * ntc;
  const int ci0 = ci_t * 64, co0 = co_t * 64;
  g.ci0 = ci0; g.co0 = co0; g.sp = blockIdx.y; g.q16 = tid & 15;
  const int sp = g.sp;

  const int band_slots = (d.RB + 2) * d.Wp;
  const int AROWS = band_slots + 2 * d.MARGIN;
  const int band_px = d.RB * d.W;
  const int ASZ = AROWS * 64, ZSZ = (band_px + 1) * 64;
  float* As0 = smem;                 // 2 x [AROWS][64]
  float* Zs0 = smem + 2 * ASZ;       // 2 x [band_px + 1][64]   (last row = zeros: odd pixel counts)
  int* atab = reinterpret_cast<int*>(Zs0 + 2 * ZSZ);   // [band_px + 4]: float offset of pixel p's slot in the A image

  for (int i = tid * 4; i < 2 * ASZ + 2 * ZSZ; i += WG_THREADS * 4)
    *reinterpret_cast<float4*>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * d.Wp + (t % 3 - 1)) * 64;

  const int U = d.N * d.nbands;
  const int u_begin = (int)(((long long)sp * U) / d.nsplit);
  const int u_end = (int)(((long long)(sp + 1) * U) / d.nsplit);

  // ---- staging descriptors (unit-independent part) ----
  const int q16 = tid & 15;          // float4 column of the 64-channel tile
  const int nunitsA = (d.RB + 2) * d.W * 16, nunitsZ = band_px * 16;
  int a_hr[WG_MAXA], a_x[WG_MAXA], a_lds[WG_MAXA];
#pragma unroll
  for (int i = 0; i < WG_MAXA; ++i) {
    const int v = tid + i * WG_THREADS;
    const int px = v >> 4;
    a_hr[i] = px / d.W;
    a_x[i] = px - a_hr[i] * d.W;
    a_lds[i] = (d.MARGIN + a_hr[i] * d.Wp + a_x[i] + 1) * 64 + q16 * 4;
  }
  const bool ci_ok = ci0 + q16 * 4 < d.C, co_ok = co0 + q16 * 4 < d.C;
  float4 ra[WG_MAXA], rz[WG_MAXZ];

  auto stage_load = [&](int u) {
    const int n = u / d.nbands, band = u - n * d.nbands;
    const int row0 = band * d.RB;
    const int rbe = min(d.RB, d.H - row0);
    const int npx = rbe * d.W;
#pragma unroll
    for (int i = 0; i < WG_MAXA; ++i) {
      const int v = tid + i * WG_THREADS;
      const int ih = row0 + a_hr[i] - 1;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < nunitsA && ih >= 0 && ih < d.H && a_hr[i] <= rbe + 1 && ci_ok)
        ra[i] = *reinterpret_cast<const float4*>(a.act + ((size_t)n * d.HW + ih * d.W + a_x[i]) * d.C + ci0 + q16 * 4);
    }
#pragma unroll
    for (int i = 0; i < WG_MAXZ; ++i) {
      const int v = tid + i * WG_THREADS;
      const int r = v >> 4;
      rz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < nunitsZ && r < npx && co_ok)
        rz[i] = *reinterpret_cast<const float4*>(a.dz + ((size_t)n * d.HW + row0 * d.W + r) * d.C + co0 + q16 * 4);
    }
  };
  auto stage_write = [&](int buf) {
    float* As = As0 + buf * ASZ;
    float* Zs = Zs0 + buf * ZSZ;
#pragma unroll
    for (int i = 0; i < WG_MAXA; ++i)
      if (tid + i * WG_THREADS < nunitsA) *reinterpret_cast<float4*>(As + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < WG_MAXZ; ++i) {
      const int v = tid + i * WG_THREADS;
      if (v < nunitsZ) *reinterpret_cast<float4*>(Zs + (v >> 4) * 64 + q16 * 4) = rz[i];
    }
  };

  for (int p = tid; p < band_px + 4; p += WG_THREADS) {
    const int pp = min(p, band_px - 1);
    const int pr = pp / d.W;
    atab[p] = (d.MARGIN + (pr + 1) * d.Wp + (pp - pr * d.W) + 1) * 64;
  }
  __syncthreads();   // zero fill + table visible
  if (u_begin < u_end) {
    stage_load(u_begin);
    stage_write(0);
  }
  __syncthreads();
  WSTAMP(a.stamps, 2, "s_memtime");

  int buf = 0;
  for (int u = u_begin; u < u_end; ++u) {
    const int band = u % d.nbands;
    const int row0 = band * d.RB;
    const int rbe = min(d.RB, d.H - row0);
    const int npx = rbe * d.W;
    const float* As = As0 + buf * ASZ;
    const float* Zs = Zs0 + buf * ZSZ;
    const bool more = (u + 1) < u_end;
    if (more) stage_load(u + 1);   // in flight during this unit's MFMAs

    // ---- MFMA: K = pixel pairs of the band, operands one pair ahead ----
    const int npairs = (npx + 1) >> 1;
    const int abase = wi * 32 + l31;
    const int bbase = wj * 32 + l31;
    const int azero = d.MARGIN * 64;   // a slot with finite data for the masked half of an odd last pair (its dz is 0)
    float av0[9], bv0, av1[9], bv1;
    int slot_nxt;
    {
      const bool ok = hi < npx;
      const int aoff = abase + (ok ? atab[hi] : azero);
      bv0 = Zs[(ok ? hi : band_px) * 64 + bbase];
#pragma unroll
      for (int t = 0; t < 9; ++t) av0[t] = As[aoff + toff[t]];
      slot_nxt = atab[2 + hi];
    }
#define WSTEP(CUR, BCUR, NXT, BNXT, KP)                                                             \
  {                                                                                                 \
    const int p1 = 2 * ((KP) + 1) + hi;                                                             \
    const bool ok = p1 < npx;                                                                       \
    const int aoff = abase + (ok ? slot_nxt : azero);                                               \
    const int zoff = (ok ? p1 : band_px) * 64 + bbase;                                              \
    /* reads in the order the next step consumes them; the slot of pair k+2 goes first */           \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                 \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(CUR[t], BCUR, acc[t], 0, 0, 0);                 \
      if (t == 0) { slot_nxt = atab[min(p1 + 2, band_px + 3)]; BNXT = Zs[zoff]; }                   \
      NXT[t] = As[aoff + toff[t]];                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
  }
    int kp = 0;
    for (; kp + 1 < npairs; kp += 2) {
      WSTEP(av0, bv0, av1, bv1, kp)
      WSTEP(av1, bv1, av0, bv0, kp + 1)
    }
    if (kp < npairs) {
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv0, acc[t], 0, 0, 0);
    }
#undef WSTEP
    if (more) stage_write(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  WSTAMP(a.stamps, 3, "s_memtime");
  wg_store_slab<9>(a, d, g, acc);
  WSTAMP(a.stamps, 4, "s_memtime");
  WSTAMP(a.stamps, 5, "s_memrealtime");
}

// ============================================================================
// k_colsum: masked column sums of dz per sample,
//   spart[n][tap][c] = sum_{pixels p of sample n whose tap neighbour p + tap is inside the image} dz[n, p, c]
// (the conv-bias gradient is tap 4; t * these are the time-channel weight gradients; their
// contraction with the time-channel weights is d f / d t).  One workgroup per sample; nine
// inclusion-exclusion terms from: total, first/last row, first/last column, four corners.
// ============================================================================
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dz, float* __restrict__ spart, Dims d) {
  __shared__ float4 red[9 * 256];      // [pixel group][tap][quad]
  __shared__ unsigned char flg[256];   // per pixel: bit0 first row, bit1 last row, bit2 first column, bit3 last column
  const int n = blockIdx.x, tid = threadIdx.x;
  const int c4n = d.C >> 2;                         // float4 columns
  const int QB = min(c4n, 256);                     // quads per pass
  const int ngrp = max(1, min(8, 256 / QB));        // pixel groups working in parallel on one quad
  for (int p = tid; p < d.HW; p += 256) {
    const int h = p / d.W, x = p - h * d.W;
    flg[p] = (unsigned char)((h == 0 ? 1 : 0) | (h == d.H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == d.W - 1 ? 8 : 0));
  }
  __syncthreads();
  const int ql = tid % QB, pg = tid / QB;
  const bool active = pg < ngrp;
  auto add = [](float4& s, const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
  auto sub = [](float4& s, const float4& v) { s.x -= v.x; s.y -= v.y; s.z -= v.z; s.w -= v.w; };
  for (int q0 = 0; q0 < c4n; q0 += QB) {
    const int q = q0 + ql;
    const bool on = active && q < c4n;
    float4 T = make_float4(0.f, 0.f, 0.f, 0.f), rf = T, rl = T, cf = T, cl = T, k00 = T, k01 = T, k10 = T, k11 = T;
    if (on) {
      const float* base = dz + (size_t)n * d.HW * d.C + q * 4;
      for (int p0 = pg; p0 < d.HW; p0 += 4 * ngrp) {
        float4 v[4];
        int f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // four independent loads in flight
          const int p = p0 + i * ngrp;
          const bool ok = p < d.HW;
          v[i] = ok ? *reinterpret_cast<const float4*>(base + (size_t)p * d.C) : make_float4(0.f, 0.f, 0.f, 0.f);
          f[i] = ok ? flg[p] : 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          add(T, v[i]);
          if (f[i] & 1) add(rf, v[i]);
          if (f[i] & 2) add(rl, v[i]);
          if (f[i] & 4) add(cf, v[i]);
          if (f[i] & 8) add(cl, v[i]);
          if ((f[i] & 5) == 5) add(k00, v[i]);
          if ((f[i] & 9) == 9) add(k01, v[i]);
          if ((f[i] & 6) == 6) add(k10, v[i]);
          if ((f[i] & 10) == 10) add(k11, v[i]);
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {   // tap (kh, kw) excludes the first (k == 0) / last (k == 2) row and column
        const int kh = t / 3, kw = t % 3;
        float4 o = T;
        if (kh == 0) sub(o, rf);
        if (kh == 2) sub(o, rl);
        if (kw == 0) sub(o, cf);
        if (kw == 2) sub(o, cl);
        if (kh == 0 && kw == 0) add(o, k00);
        if (kh == 0 && kw == 2) add(o, k01);
        if (kh == 2 && kw == 0) add(o, k10);
        if (kh == 2 && kw == 2) add(o, k11);
        red[(pg * 9 + t) * QB + ql] = o;
      }
    }
    __syncthreads();
    if (on && pg == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float4 o = red[t * QB + ql];
        for (int r = 1; r < ngrp; ++r) add(o, red[(r * 9 + t) * QB + ql]);
        *reinterpret_cast<float4*>(spart + ((size_t)n * 9 + t) * d.C + q * 4) = o;
      }
    }
    __syncthreads();
  }
}

void launch_colsum(const Dims& d, const float* dz, float* spart, hipStream_t s) {
  hipLaunchKernelGGL(k_colsum, dim3(d.N), dim3(256), 0, s, dz, spart, d);
}

// ----------------------------------------------------------------------------
// launch
// ----------------------------------------------------------------------------
size_t wgrad_lds_bytes(const Dims& d) {
  const int band_slots = (d.RB + 2) * d.Wp;
  const int AROWS = band_slots + 2 * d.MARGIN;
  const int band_px = d.RB * (d.W + 1);   // covers the padded dz rows of the templated kernel
  return (2 * (size_t)AROWS * 64 + 2 * (size_t)(band_px + 1) * 64 + (size_t)band_px + 4) * sizeof(float);
}

int wgrad_variant() {
  if (g_wgrad_variant >= 0) return g_wgrad_variant;
  static int v = -1;
  if (v < 0) { const char* e = getenv("NODE_TUNE_WGRAD_VARIANT"); v = e ? atoi(e) : 1; }
  return v;
}

template <int W, int RB>
static void launch_wgrad_w(const Dims& d, const WgradArgs& a, dim3 grid, hipStream_t s) {
  static bool attr[MAX_DEVICES];
  allow_full_lds((const void*)k_wgrad_w<W, RB>, attr);
  const size_t lds = (size_t)(2 * (RB + 2) * (W / 2) * 256 + 2 * RB * (W / 2) * 256) * sizeof(float);
  hipLaunchKernelGGL((k_wgrad_w<W, RB>), grid, dim3(WG_THREADS), lds, s, a, d);
}

template <int UT>
static void launch_wgrad_w2(const Dims& d, const WgradArgs& a, dim3 grid, hipStream_t s) {
  static bool attr[MAX_DEVICES];
  allow_full_lds((const void*)k_wgrad_w2<UT>, attr);
  const size_t lds_loop = (size_t)(4 * UT * 16 * 68) * sizeof(float), lds_x = (size_t)8 * 72 * 64 * sizeof(float);
  hipLaunchKernelGGL((k_wgrad_w2<UT>), grid, dim3(WG2_THREADS), lds_loop > lds_x ? lds_loop : lds_x, s, a, d);
}

template <int W, int RB>
static void launch_wgrad_t(const Dims& d, const WgradArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static bool attr[MAX_DEVICES];
  allow_full_lds((const void*)k_wgrad_t<W, RB>, attr);
  hipLaunchKernelGGL((k_wgrad_t<W, RB>), grid, dim3(WG_THREADS), lds, s, a, d);
}

// variant 1 (production): templated kernel where the geometry has an instance, generic otherwise;
// variant 0: always the generic kernel.
void launch_wgrad(const Dims& d, const WgradArgs& a, hipStream_t s) {
  const int ntc = (d.C + 63) / 64;
  const dim3 grid(ntc * ntc, d.nsplit);
  const size_t lds = wgrad_lds_bytes(d);
  if (d.wgrad_wino == 2) {   // 2-D Winograd domain (make_dims sets wut)
    const dim3 grid2(ntc * ntc, d.nsplit, a.act2 != nullptr ? 2 : 1);
    if (d.wut == 8) { launch_wgrad_w2<8>(d, a, grid2, s); return; }
    if (d.wut == 4) { launch_wgrad_w2<4>(d, a, grid2, s); return; }
  }
  if (d.wgrad_wino == 1) {   // 1-D Winograd-domain accumulation, ordinary nine-tap slabs (make_dims sets RB)
    if (d.W == 8 && d.RB == 8) { launch_wgrad_w<8, 8>(d, a, grid, s); return; }
    if (d.W == 16 && d.RB == 2) { launch_wgrad_w<16, 2>(d, a, grid, s); return; }
    if (d.W == 4 && d.RB == 4) { launch_wgrad_w<4, 4>(d, a, grid, s); return; }
  }
  if (wgrad_variant() >= 1 && d.H % d.RB == 0) {
    if (d.W == 8 && d.RB == 8) { launch_wgrad_t<8, 8>(d, a, grid, lds, s); return; }
    if (d.W == 16 && d.RB == 4) { launch_wgrad_t<16, 4>(d, a, grid, lds, s); return; }
    if (d.W == 7 && d.RB == 7) { launch_wgrad_t<7, 7>(d, a, grid, lds, s); return; }
    if (d.W == 4 && d.RB == 4) { launch_wgrad_t<4, 4>(d, a, grid, lds, s); return; }
  }
  static bool attr[MAX_DEVICES];
  allow_full_lds((const void*)k_wgrad_p, attr);
  hipLaunchKernelGGL(k_wgrad_p, grid, dim3(WG_THREADS), lds, s, a, d);
}

}  // namespace node
