// fp32-MFMA implicit-GEMM kernels of the ODE dynamics (gfx950).
//
//   k_conv3x3<MT>  out[m, co] = sum_{tap, ci} A[pix(m) + tap, ci] * Wp[tap, ci, co]
//                  M = N*H*W pixels, K = 9*C, fp32 in / fp32 accumulate on
//                  v_mfma_f32_32x32x2_f32 (exact fp32: the embedded-error estimator
//                  of dopri5 cannot tolerate bf16 noise).
//                  Forward conv and data-gradient share the kernel (dgrad = conv
//                  with flipped/transposed packed weights).
//   k_wgrad        dW[tap, ci, co] = sum_pix A[pix + tap, ci] * dZ[pix, co]   (split-K over samples)
//
// MI355X-first design points
//  * M tiles are aligned to WHOLE SAMPLES (S samples of H*W pixels per tile), and N
//    tiles to whole GroupNorm groups, so the GroupNorm that follows every conv in
//    ODEfunc (model.py:343-347) -- and, in the backward, the ReLU mask + GroupNorm
//    backward that follows every dgrad -- is computed entirely in the epilogue from
//    the accumulator tile staged once through LDS.  One ODEfunc eval is three
//    kernels (combine+GN1, conv1+GN2+ReLU, conv2+GN3) instead of ~12 ATen ops.
//  * The activation chunk (S samples x 32 channels) is staged ONCE per K chunk into
//    a zero-haloed LDS image; the nine 3x3 taps are nine constant LDS offsets into
//    that image -- no im2col, no per-tap reload, no border predication.
//  * The constant-time channel of ConcatConv2d (model.py:321-322) is not carried
//    through K: its contribution is t * tmap[p, co] (border-aware tap sums), added
//    with the bias in the epilogue.
//  * 8 waves (4 M x 2 N) per workgroup, two waves per SIMD so one wave's LDS
//    operand reads hide behind its partner's 64-cycle MFMAs; global->LDS staging
//    is register-prefetched one piece ahead (issue early / write late).
#include "node_internal.h"
#include <cstdlib>
#include <cstdio>

namespace node {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_conv_variant = -1;

// Diagnostic builds only (tools/kbench.hip, -DNODE_STAMPS): per-wave s_memtime / s_memrealtime stamps
// written to a buffer nothing else reads.  Production builds compile this to nothing.
#ifdef NODE_STAMPS
#define STAMP(buf, slot)                                                                         \
  do {                                                                                           \
    if ((buf) != nullptr && (threadIdx.x & 63) == 0) {                                           \
      unsigned long long _t;                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                 \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      (buf)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (slot)] = _t; \
    }                                                                                            \
  } while (0)
#define STAMP_REAL(buf, slot)                                                                    \
  do {                                                                                           \
    if ((buf) != nullptr && (threadIdx.x & 63) == 0) {                                           \
      unsigned long long _t;                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");             \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      (buf)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (slot)] = _t; \
    }                                                                                            \
  } while (0)
#define WABL(bit) (a.ablate & (bit))
#else
#define STAMP(buf, slot) do { } while (0)
#define STAMP_REAL(buf, slot) do { } while (0)
#define WABL(bit) 0
#endif

__device__ inline float wave_sum_c(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline int slot_of(int p, int W, int Wp) {
  const int h = p / W;
  return (h + 1) * Wp + (p - h * W) + 1;
}

// ============================================================================
// conv3x3 implicit GEMM + fused epilogue
// ============================================================================
template <int MT>
__global__ __launch_bounds__(CONV_THREADS) void k_conv3x3(ConvArgs a, Dims d) {
  STAMP_REAL(a.stamps, 0);
  STAMP(a.stamps, 1);
  constexpr int BM = 128 * MT;
  constexpr int NA = 2 * MT;          // float4 staging units per thread for one A chunk
  constexpr int CT = BN + 1;          // epilogue tile stride
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int mtile = blockIdx.x, nt = blockIdx.y;
  const int n0 = mtile * d.S;
  const int c0 = nt * d.BNE;
  const int nsamp = min(d.S, d.N - n0);
  const int rows_valid = nsamp * d.HW;

  const int AROWS = d.S * d.SLOTS + 2 * d.MARGIN;
  const int ABUF = (AROWS * AST + 3) & ~3;
  float* Abuf0 = smem;
  float* Abuf1 = smem + ABUF;
  float* Bbuf0 = smem + 2 * ABUF;
  float* Bbuf1 = Bbuf0 + KCH * BN;

  // ---- zero both A images (halo, margins, channel padding) ----
  for (int i = tid; i < 2 * ABUF; i += CONV_THREADS) smem[i] = 0.f;

  // ---- per-thread staging descriptors for the A chunk ----
  size_t gofs[NA];
  int lofs[NA];
  bool aval[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tid + i * CONV_THREADS;
    const int row = u >> 3, q = u & 7;
    aval[i] = row < rows_valid;
    const int rr = aval[i] ? row : 0;
    const int s = rr / d.HW, p = rr - s * d.HW;
    gofs[i] = ((size_t)(n0 + s) * d.HW + p) * d.C + q * 4;
    lofs[i] = (d.MARGIN + s * d.SLOTS + slot_of(p, d.W, d.Wp)) * AST + q * 4;
  }
  // ---- per-lane MFMA A-row offsets ----
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = wm * (32 * MT) + mt * 32 + l31;
    int slot = 0;
    if (row < d.S * d.HW) {
      const int s = row / d.HW, p = row - s * d.HW;
      slot = s * d.SLOTS + slot_of(p, d.W, d.Wp);
    }
    arow[mt] = (d.MARGIN + slot) * AST + hi;
  }
  const int boff = hi * BN + wn * 32 + l31;

  const float* wbase = a.wpacked + (size_t)nt * d.nchunk * 9 * (KCH * BN);
  const int Q = d.nchunk * 9;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  float4 areg[NA];
  float4 breg;
  __syncthreads();  // zero fill visible

  // prologue: chunk 0 + piece 0
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int qq = (tid + i * CONV_THREADS) & 7;
    areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aval[i] && qq * 4 < d.C) areg[i] = *reinterpret_cast<const float4*>(a.in + gofs[i]);
  }
  breg = *reinterpret_cast<const float4*>(wbase + tid * 4);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int qq = (tid + i * CONV_THREADS) & 7;
    if (aval[i] && qq * 4 < d.C) {
      float* dst = Abuf0 + lofs[i];
      dst[0] = areg[i].x; dst[1] = areg[i].y; dst[2] = areg[i].z; dst[3] = areg[i].w;
    }
  }
  *reinterpret_cast<float4*>(Bbuf0 + tid * 4) = breg;
  __syncthreads();
  STAMP(a.stamps, 2);

  for (int q = 0; q < Q; ++q) {
    const int chunk = q / 9, tap = q - chunk * 9;
    const bool has_next = (q + 1) < Q;
    const bool next_new_chunk = has_next && (tap == 8);
    // ---- issue global loads for piece q+1 (consumed after the MFMA block) ----
    if (has_next) breg = *reinterpret_cast<const float4*>(wbase + (size_t)(q + 1) * (KCH * BN) + tid * 4);
    if (next_new_chunk) {
      const int cbase = (chunk + 1) * KCH;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int qq = (tid + i * CONV_THREADS) & 7;
        areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aval[i] && cbase + qq * 4 < d.C)
          areg[i] = *reinterpret_cast<const float4*>(a.in + gofs[i] + cbase);
      }
    }
    // ---- MFMA block on the current piece ----
    const float* Ab = (chunk & 1) ? Abuf1 : Abuf0;
    const float* Bb = (q & 1) ? Bbuf1 : Bbuf0;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int toff = ((kh - 1) * d.Wp + (kw - 1)) * AST;
#pragma unroll
    for (int kk = 0; kk < KCH; kk += 2) {
      const float b = Bb[boff + kk * BN];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float av = Ab[arow[mt] + toff + kk];
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[mt], 0, 0, 0);
      }
    }
    // ---- write the prefetched piece into the other buffers ----
    if (has_next) *reinterpret_cast<float4*>(((q & 1) ? Bbuf0 : Bbuf1) + tid * 4) = breg;
    if (next_new_chunk) {
      float* An = (chunk & 1) ? Abuf0 : Abuf1;
      const int cbase = (chunk + 1) * KCH;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int qq = (tid + i * CONV_THREADS) & 7;
        if (aval[i] && cbase + qq * 4 < d.C) {
          float* dst = An + lofs[i];
          dst[0] = areg[i].x; dst[1] = areg[i].y; dst[2] = areg[i].z; dst[3] = areg[i].w;
        }
      }
    }
    __syncthreads();
  }

  STAMP(a.stamps, 3);
  // ==========================================================================
  // epilogue: accumulators -> LDS tile -> GroupNorm (fwd or bwd) -> HBM
  // ==========================================================================
  float* Ct = smem;                 // [BM][CT]
  float* Xt = smem + BM * CT;       // [BM][CT]   (bwd only)
  float* st0 = smem + 2 * BM * CT;  // [S*BN] mean / m1
  float* st1 = st0 + d.S * BN;      // [S*BN] rstd / m2
  float* cred = st1 + d.S * BN;     // [512][2]

  const bool fwd = a.mode != CM_BWD_RELU_GN;
  const int ncols = min(d.BNE, d.C - c0);
  const float tval = fwd ? eval_time(a.et) : 0.f;
  {
    const int col = wn * 32 + l31;
    const int c = c0 + col;
    const bool cok = col < ncols;
    const float bias = (fwd && cok) ? a.bias[c] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (32 * MT) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = acc[mt][r];
        if (fwd && cok && row < rows_valid) {
          const int p = row % d.HW;
          v += bias + tval * a.tmap[(size_t)p * d.C + c];
        }
        Ct[row * CT + col] = v;
      }
    }
  }
  __syncthreads();

  const int col = tid & 63, rg = tid >> 6;  // 64 columns x 8 row groups
  const int c = c0 + col;
  const bool cok = col < ncols;
  const int GT = ncols / d.cpg;             // whole groups in this tile
  const int npairs = nsamp * GT;
  const int m = d.HW * d.cpg;
  const float inv_m = 1.0f / (float)m;

  if (fwd) {
    for (int pair = wave; pair < npairs; pair += 8) {
      const int s = pair / GT, gl = pair - s * GT;
      float sum = 0.f;
      for (int e = lane; e < m; e += 64) {
        const int p = e / d.cpg, cc = e - p * d.cpg;
        sum += Ct[(s * d.HW + p) * CT + gl * d.cpg + cc];
      }
      const float mean = wave_sum_c(sum) * inv_m;
      float s2 = 0.f;
      for (int e = lane; e < m; e += 64) {
        const int p = e / d.cpg, cc = e - p * d.cpg;
        const float dv = Ct[(s * d.HW + p) * CT + gl * d.cpg + cc] - mean;
        s2 += dv * dv;
      }
      const float var = wave_sum_c(s2) * inv_m;
      const float rstd = 1.0f / sqrtf(var + d.eps);
      if (lane == 0) {
        st0[pair] = mean;
        st1[pair] = rstd;
        if (a.rstd_out) a.rstd_out[(size_t)(n0 + s) * d.G + c0 / d.cpg + gl] = rstd;
      }
    }
    __syncthreads();
    if (cok) {
      const float gm = a.gamma[c], bt = a.beta[c];
      const int gl = col / d.cpg;
      const bool relu = a.mode == CM_FWD_GN_RELU;
      for (int row = rg; row < rows_valid; row += 8) {
        const int s = row / d.HW, p = row - s * d.HW;
        const float xh = (Ct[row * CT + col] - st0[s * GT + gl]) * st1[s * GT + gl];
        float o = xh * gm + bt;
        if (relu) o = fmaxf(o, 0.f);
        const size_t off = ((size_t)(n0 + s) * d.HW + p) * d.C + c;
        a.out[off] = a.osign * o;
        if (a.xhat_out) a.xhat_out[off] = xh;
      }
    }
  } else {
    // ReLU mask, dxhat = du * gamma, channel partials of (dgamma, dbeta)
    float dg = 0.f, db = 0.f;
    if (cok) {
      const float gm = a.gamma[c];
      for (int row = rg; row < rows_valid; row += 8) {
        const int s = row / d.HW, p = row - s * d.HW;
        const size_t off = ((size_t)(n0 + s) * d.HW + p) * d.C + c;
        const float x = a.xhat[off];
        const float du = a.act[off] > 0.f ? Ct[row * CT + col] : 0.f;
        dg += du * x;
        db += du;
        Ct[row * CT + col] = du * gm;
        Xt[row * CT + col] = x;
      }
    }
    cred[tid * 2] = dg;
    cred[tid * 2 + 1] = db;
    __syncthreads();
    if (rg == 0 && cok) {
#pragma unroll
      for (int r = 1; r < 8; ++r) { dg += cred[(r * 64 + col) * 2]; db += cred[(r * 64 + col) * 2 + 1]; }
      a.gpart[((size_t)mtile * 2 + 0) * d.C + c] = dg;
      a.gpart[((size_t)mtile * 2 + 1) * d.C + c] = db;
    }
    for (int pair = wave; pair < npairs; pair += 8) {
      const int s = pair / GT, gl = pair - s * GT;
      float s1 = 0.f, s2 = 0.f;
      for (int e = lane; e < m; e += 64) {
        const int p = e / d.cpg, cc = e - p * d.cpg;
        const int idx = (s * d.HW + p) * CT + gl * d.cpg + cc;
        const float dxh = Ct[idx];
        s1 += dxh;
        s2 += dxh * Xt[idx];
      }
      s1 = wave_sum_c(s1) * inv_m;
      s2 = wave_sum_c(s2) * inv_m;
      if (lane == 0) { st0[pair] = s1; st1[pair] = s2; }
    }
    __syncthreads();
    if (cok) {
      const int gl = col / d.cpg;
      for (int row = rg; row < rows_valid; row += 8) {
        const int s = row / d.HW, p = row - s * d.HW;
        const float r = a.rstd[(size_t)(n0 + s) * d.G + c0 / d.cpg + gl];
        const float dx = r * (Ct[row * CT + col] - st0[s * GT + gl] - Xt[row * CT + col] * st1[s * GT + gl]);
        a.out[((size_t)(n0 + s) * d.HW + p) * d.C + c] = a.osign * dx;
      }
    }
  }
  STAMP(a.stamps, 4);
  STAMP_REAL(a.stamps, 5);
}

static size_t conv_v0_lds_bytes(const Dims& d) {
  const int AROWS = d.S * d.SLOTS + 2 * d.MARGIN;
  const size_t abuf = ((size_t)AROWS * AST + 3) & ~(size_t)3;
  const size_t main_loop = 2 * abuf + 2 * KCH * BN;
  const size_t epi = 2 * (size_t)d.BM * (BN + 1) + 2 * (size_t)d.S * BN + 2 * CONV_THREADS;
  return (main_loop > epi ? main_loop : epi) * sizeof(float);
}
// geometry check (make_dims): the tile must fit LDS under either kernel variant
size_t conv_lds_bytes(const Dims& d, int /*mode*/) {
  const size_t a = conv_v0_lds_bytes(d), b = conv_p_lds_bytes(d);
  return a > b ? a : b;
}

int conv_variant() {
  if (g_conv_variant >= 0) return g_conv_variant;
  static int v = -1;
  if (v < 0) { const char* e = getenv("NODE_TUNE_CONV_VARIANT"); v = e ? atoi(e) : 1; }
  return v;
}

static size_t tune_min_lds() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("NODE_TUNE_CONV_MIN_LDS"); v = e ? atol(e) : 0; }
  return (size_t)v;
}

// ---- debugging aid (NODE_DEBUG_CONV_XCHECK=1): every launch runs BOTH kernels and reports the largest
// difference of the primary output; development only (allocates, synchronises, prints) ----
static const float* g_xc_v1[8];
static float* g_xc_v0[8];
static int g_xc_n = 0;
static bool xcheck_on() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("NODE_DEBUG_CONV_XCHECK"); v = e ? atoi(e) : 0; }
  return v != 0;
}
void xcheck_register(const Dims& d, const float* wraw, const float* packed_v1, int dgrad, hipStream_t s) {
  if (!xcheck_on()) return;
  const size_t wsz = (size_t)d.ntile * d.nchunk * 9 * KCH * BN;
  int slot = -1;
  for (int i = 0; i < g_xc_n; ++i) if (g_xc_v1[i] == packed_v1) slot = i;
  if (slot < 0) { slot = g_xc_n++ % 8; g_xc_v1[slot] = packed_v1; (void)hipMalloc((void**)&g_xc_v0[slot], wsz * sizeof(float)); }
  launch_pack_weights(d, wraw, g_xc_v0[slot], dgrad, 0, s);
}
__global__ void k_xc_diff(const float* a, const float* b, size_t n, float* out) {
  float m = 0.f, r = 0.f;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) { m = fmaxf(m, fabsf(a[i] - b[i])); r = fmaxf(r, fabsf(b[i])); }
  for (int off = 32; off > 0; off >>= 1) { m = fmaxf(m, __shfl_xor(m, off, 64)); r = fmaxf(r, __shfl_xor(r, off, 64)); }
  __shared__ float sm[32];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; sm[16 + (threadIdx.x >> 6)] = r; }
  __syncthreads();
  if (threadIdx.x == 0) { for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { m = fmaxf(m, sm[i]); r = fmaxf(r, sm[16 + i]); } out[0] = m; out[1] = r; }
}
static void launch_conv_v0(const Dims& d, const ConvArgs& a, hipStream_t s);

void launch_conv(const Dims& d, const ConvArgs& a, hipStream_t s) {
  if (conv_variant() >= 1 && xcheck_on()) {
    static float *tmp = nullptr, *dres = nullptr;
    static size_t tmpn = 0;
    if (tmpn < d.numel) { if (tmp) (void)hipFree(tmp); (void)hipMalloc((void**)&tmp, d.numel * sizeof(float)); tmpn = d.numel; }
    if (!dres) (void)hipMalloc((void**)&dres, 2 * sizeof(float));
    launch_conv_p(d, a, s);
    ConvArgs b = a;
    for (int i = 0; i < 8; ++i) if (g_xc_v1[i] == a.wpacked) b.wpacked = g_xc_v0[i];
    static float *tmpx = nullptr, *tmpr = nullptr;
    if (!tmpx) { (void)hipMalloc((void**)&tmpx, (size_t)64 << 20); (void)hipMalloc((void**)&tmpr, (size_t)1 << 20); }
    b.out = tmp; b.xhat_out = a.xhat_out ? tmpx : nullptr; b.rstd_out = a.rstd_out ? tmpr : nullptr;
    float* gp2 = nullptr;
    if (a.mode == CM_BWD_RELU_GN) { (void)hipMalloc((void**)&gp2, (size_t)d.mtiles * 2 * d.C * sizeof(float)); b.gpart = gp2; }
    launch_conv_v0(d, b, s);
    hipLaunchKernelGGL(k_xc_diff, dim3(1), dim3(1024), 0, s, a.out, tmp, d.numel, dres);
    float h[2];
    (void)hipMemcpyAsync(h, dres, sizeof(h), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    static int call = 0;
    fprintf(stderr, "[xcheck] conv call %3d mode %d osign %+.0f  max|v1 - v0| = %.3e (ref max %.3e)%s\n", call++, a.mode, a.osign, h[0], h[1],
            h[0] > 1e-4f * h[1] + 1e-12f ? "   <<<<<" : "");
    if (a.xhat_out) {
      hipLaunchKernelGGL(k_xc_diff, dim3(1), dim3(1024), 0, s, a.xhat_out, tmpx, d.numel, dres);
      (void)hipMemcpyAsync(h, dres, sizeof(h), hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      fprintf(stderr, "[xcheck]      xhat_out max diff %.3e (ref %.3e)%s\n", h[0], h[1], h[0] > 1e-4f * h[1] ? "   <<<<<" : "");
      hipLaunchKernelGGL(k_xc_diff, dim3(1), dim3(1024), 0, s, a.rstd_out, tmpr, (size_t)d.N * d.G, dres);
      (void)hipMemcpyAsync(h, dres, sizeof(h), hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      fprintf(stderr, "[xcheck]      rstd_out max diff %.3e (ref %.3e)%s\n", h[0], h[1], h[0] > 1e-4f * h[1] ? "   <<<<<" : "");
    }
    if (gp2) {
      hipLaunchKernelGGL(k_xc_diff, dim3(1), dim3(1024), 0, s, a.gpart, gp2, (size_t)d.mtiles * 2 * d.C, dres);
      (void)hipMemcpyAsync(h, dres, sizeof(h), hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      fprintf(stderr, "[xcheck]      gpart max diff %.3e (ref %.3e)%s\n", h[0], h[1], h[0] > 1e-3f * h[1] ? "   <<<<<" : "");
      (void)hipFree(gp2);
    }
    return;
  }
  if (conv_variant() >= 1) { launch_conv_p(d, a, s); return; }
  launch_conv_v0(d, a, s);
}

static void launch_conv_v0(const Dims& d, const ConvArgs& a, hipStream_t s) {
  size_t lds = conv_v0_lds_bytes(d);
  if (lds < tune_min_lds()) lds = tune_min_lds();
  dim3 grid(d.mtiles, d.ntile);
  if (d.BM == 128) {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)k_conv3x3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL(k_conv3x3<1>, grid, dim3(CONV_THREADS), lds, s, a, d);
  } else {
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute((const void*)k_conv3x3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr2 = true; }
    hipLaunchKernelGGL(k_conv3x3<2>, grid, dim3(CONV_THREADS), lds, s, a, d);
  }
}

}  // namespace node
