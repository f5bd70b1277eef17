// Winograd F(4x4,3x3) pipeline, stage 2: the 36 component GEMMs, and the once-per-solve filter transform.
// gfx950 (MI355X / CDNA4) only.  See wino4.h for the data layouts.
#include "wino4.h"
#include <cstdlib>
#include <cstring>

namespace node {

typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* w4_lds_ptr_t;

// Big once-written, once-read results (M, dU): stored WRITE-THROUGH (agent scope = sc1 on gfx950) so that they leave the
// XCD's L2 while the kernel still runs instead of as one write-back burst at its end (MI355X_MICROARCH.md, `boundary`:
// + B / 6 TB/s behind B dirty bytes).  NODE_WT_STORES=0 at build time: plain stores (A/B measurements).
#ifndef NODE_WT_STORES
#define NODE_WT_STORES 1
#endif
__device__ __forceinline__ void st_wt(float* p, float v) {
#if NODE_WT_STORES
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

// ----------------------------------------------------------------------------
// U = G g G^T for every (co, ci) pair, written in MFMA-ready blocks (wino4.h).  dgrad: the data-gradient filter
// g'[ci][co][kh][kw] = g[co][ci][2-kh][2-kw].  Weights are [C][C+1][3][3] (input channel 0 = time, model.py:320-323).
// ----------------------------------------------------------------------------
// With jobs.ub: additionally the split-precision form k_w4_gemm64b reads -- every fp32 value u as three bf16 parts
// h = bf16(u), m = bf16(u - h), l = bf16(u - h - m) (u = h + m + l exactly), laid out [comp][cb][g/2][part 3][hi 2][col 32]
// [8 values: g even e 0..3, g odd e 0..3], i.e. one 16-B load per lane and part feeds one K = 16 bf16 MFMA.
__device__ __forceinline__ unsigned short w4_bf16_rne(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);   // round to nearest even (finite inputs)
}
__device__ __forceinline__ float w4_bf16_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

__global__ __launch_bounds__(256) void k_w4_pack(W4PackJobs jobs, int C) {
  const float* __restrict__ w = jobs.w[blockIdx.y];
  float* __restrict__ U = jobs.u[blockIdx.y];
  unsigned short* __restrict__ Ub = jobs.ub[blockIdx.y];
  _Float16* __restrict__ Uh = reinterpret_cast<_Float16*>(jobs.uh[blockIdx.y]);
  const float uscale = Uh != nullptr ? ldexpf(1.f, *jobs.uh_exp[blockIdx.y]) : 1.f;
  const int dgrad = jobs.dgrad[blockIdx.y];
  const int CI = jobs.plain[blockIdx.y] ? C : C + 1, c_off = jobs.plain[blockIdx.y] ? 0 : 1;   // input-channel stride / first data channel
  const int G8 = C >> 3;
  const size_t total = (size_t)C * C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    // consecutive threads write consecutive elements of the layout this launch produces: fp32 [cb][g][hi][col][e], or
    // the bf16 triples [cb][g/2][part][hi][col][g & 1][e] (eight values = one 16-B operand of a lane)
    int e, col, hi, g, cb;
    if (Ub == nullptr && Uh == nullptr) {
      e = idx & 3; col = (idx >> 2) & 31; hi = (idx >> 7) & 1;
      g = (int)((idx >> 8) % G8); cb = (int)((idx >> 8) / G8);
    } else {
      e = idx & 3; col = (idx >> 3) & 31; hi = (idx >> 8) & 1;
      const int g2 = (int)((idx >> 9) % (G8 >> 1));
      g = 2 * g2 + (int)((idx >> 2) & 1); cb = (int)((idx >> 9) / (G8 >> 1));
    }
    const int nidx = cb * 32 + col, kidx = 8 * g + 4 * hi + e;   // output column / reduction index of the GEMM
    double gg[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        gg[kh][kw] = dgrad ? (double)w[(((size_t)kidx * CI + c_off + nidx) * 3 + (2 - kh)) * 3 + (2 - kw)]
                           : (double)w[(((size_t)nidx * CI + c_off + kidx) * 3 + kh) * 3 + kw];
    double gt[6][3];   // G g
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) gt[i][kw] = W4_G[i][0] * gg[0][kw] + W4_G[i][1] * gg[1][kw] + W4_G[i][2] * gg[2][kw];
    const size_t fidx = ((((size_t)cb * G8 + g) * 2 + hi) * 32 + col) * 4 + e;
    const size_t bidx = ((((size_t)cb * (G8 >> 1) + (g >> 1)) * 3) * 64 + (size_t)(hi * 32 + col)) * 8 + (g & 1) * 4 + e;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        const double v = gt[i][0] * W4_G[l][0] + gt[i][1] * W4_G[l][1] + gt[i][2] * W4_G[l][2];
        const float uf = (float)v;
        if (Uh != nullptr) {          // "U pairs" (wino4.h): [cb][g2][part][hi][col][gp][e], the parts 512 halves apart
          const float us = uf * uscale;
          const _Float16 hh = (_Float16)us;
          const _Float16 lh = (_Float16)(us - (float)hh);
          _Float16* o = Uh + (size_t)(i * 6 + l) * total * 2 + ((((size_t)cb * (G8 >> 1) + (g >> 1)) * 2) * 64 + (size_t)(hi * 32 + col)) * 8 + (g & 1) * 4 + e;
          o[0] = hh;
          o[512] = lh;
        }
        if (Ub != nullptr) {          // (an augmented solve prepares both: its first evaluations run the triples, wino4.h)
          const unsigned short hb = w4_bf16_rne(uf);
          const float r1 = uf - w4_bf16_f32(hb);
          const unsigned short mb = w4_bf16_rne(r1);
          const unsigned short lb = w4_bf16_rne(r1 - w4_bf16_f32(mb));
          unsigned short* o = Ub + (size_t)(i * 6 + l) * total * 3 + bidx;
          o[0] = hb;
          o[512] = mb;
          o[1024] = lb;
        }
        if (Ub == nullptr && Uh == nullptr) U[(size_t)(i * 6 + l) * total + fidx] = uf;
      }
  }
}
void launch_w4_pack(const W4PackJobs& jobs, int count, int C, hipStream_t s) {
  int blocks = (int)(((size_t)C * C + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_w4_pack, dim3(blocks, count), dim3(256), 0, s, jobs, C);
}

// ----------------------------------------------------------------------------
// k_w4_gemm: M_c[rows, C] = V_c[rows, C] x U_c[C, C] for the 36 components -- the first version, kept for batches that
// are a multiple of 8 but not of 16 (k_w4_gemm64 below is 2 - 4 us faster where it applies).
// Workgroup = one 32-row block x one pair of 32-column blocks x NINE components (grid = N/8 x C/64 x 4): eight
// waves take one component each over the whole K range (two accumulators sharing the row operand), the ninth
// component is cut into eight K slices, one per wave, and summed through LDS -- 288 MFMAs per wave, 576 per SIMD,
// no wave idles.  No operand is shared between waves, so nothing is staged through LDS: every (component, block,
// eight channels) operand is a contiguous 1 KB block and one 16-B load per lane feeds four MFMA k-steps; four
// loads deep in registers.  Per eight MFMAs a wave issues three vector loads and nothing else.
// ----------------------------------------------------------------------------
constexpr int W4_DEPTH = 4;    // operand sets in flight in the main loop (C % 64 == 0: C / 8 is a multiple)
constexpr int W4_SDEPTH = 4;   // ... in a K slice of the shared component

__device__ __forceinline__ void w4_mac(float16_t& acc0, float16_t& acc1, const float4& a, const float4& b0, const float4& b1) {
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
}

// acc += V[comp, rb, g0 .. g0+ng) x U[comp, cb0 / cb0+1, g0 .. g0+ng).  A ring of W4_DEPTH operand sets: each set is
// refilled, right behind the MFMAs that consumed it, with the block W4_DEPTH ahead -- unconditionally, so the last
// sets read up to W4_DEPTH blocks past the range (the buffers carry that much slack, see w4_v_elems / w4_u_elems).
template <int D>
struct W4Ring {
  float4 a[D], b0[D], b1[D];
  const float4 *qa, *q0, *q1;
};
// requests in the steady state's order, pinned: the compiler merges the wait state of the loop entry into every
// iteration, so any other order here would make each iteration wait for its youngest request
template <int D>
__device__ __forceinline__ void w4_ring_fill(W4Ring<D>& r, const float* pa, const float* pb0, const float* pb1) {
  r.qa = reinterpret_cast<const float4*>(pa);
  r.q0 = reinterpret_cast<const float4*>(pb0);
  r.q1 = reinterpret_cast<const float4*>(pb1);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    r.a[i] = r.qa[i * 64];
    r.b0[i] = r.q0[i * 64];
    r.b1[i] = r.q1[i * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  r.qa += D * 64; r.q0 += D * 64; r.q1 += D * 64;
}
// GUARD: ng need not be a multiple of D and nothing is refilled (the K slices of the shared component: ng <= D)
template <int D, bool GUARD>
__device__ __forceinline__ void w4_ring_run(W4Ring<D>& r, float16_t& acc0, float16_t& acc1, int ng) {
  for (int g = 0; g < ng; g += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (!GUARD || g + i < ng) w4_mac(acc0, acc1, r.a[i], r.b0[i], r.b1[i]);
      __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the MFMAs that read the old contents
      if (!GUARD) {
        r.a[i] = r.qa[i * 64];
        r.b0[i] = r.q0[i * 64];
        r.b1[i] = r.q1[i * 64];
      }
    }
    r.qa += D * 64; r.q0 += D * 64; r.q1 += D * 64;
  }
}

// AB: timing-only ablations (tools/w4_time.py, NODE_TUNE_W4_ABLATE): 1 no shared component, 2 no main loop, 4 no stores
template <int AB>
__global__ __launch_bounds__(512) void k_w4_gemm(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                 const Ctrl* ctrl, W4Geom gm, int xmap) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [8 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCP = gm.C >> 6, nRB = gm.RB;
  // Workgroups b and b + 8 share an XCD (and its L2).  An XCD takes one component group and half of the column
  // pairs for every row block: its filter operands (9 components x nCP/2 pairs) stay resident in its L2 while the
  // row operands stream through, each fetched by two XCDs.
  int cg, rb, cp;
  {
    const int L = blockIdx.x;
    if (xmap == 1 && (nRB & 1) == 0) {   // (A/B) an XCD = one component group x HALF of the row blocks x every column pair
      const int xcd = L & 7, slot = L >> 3;
      cg = xcd >> 1;
      rb = (xcd & 1) * (nRB >> 1) + slot / nCP;
      cp = slot % nCP;
    } else if ((nCP & 1) == 0 && ((nRB * nCP * 4) & 7) == 0) {
      const int xcd = L & 7, slot = L >> 3, half = nCP >> 1;
      cg = xcd >> 1;
      rb = slot / half;
      cp = (xcd & 1) * half + slot % half;
    } else {
      cg = L & 3;
      const int r = L >> 2;
      cp = r % nCP;
      rb = r / nCP;
    }
  }
  const int G8 = gm.G8, CB = gm.C >> 5;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  const int b_off = lane * 4;
  auto vblk = [&](int comp) { return V + (((size_t)comp * nRB + rb) * G8) * 256 + a_off; };
  auto ublk = [&](int comp, int cb) { return U + (((size_t)comp * CB + cb) * G8) * 256 + b_off; };

  // --- this wave's own component, whole K range
  const int comp = cg * 9 + wave;
  {
    W4Ring<W4_DEPTH> ring;
    w4_ring_fill(ring, vblk(comp), ublk(comp, 2 * cp), ublk(comp, 2 * cp + 1));
    float16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (!(AB & 2)) w4_ring_run<W4_DEPTH, false>(ring, acc0, acc1, G8);
    // M is [n][C/32][36][4 t][32 c] (wino4.h): accumulator register r of a lane holds row (r & 3) + 8 (r >> 2) + 4 hi
    // = tile r & 3 of sample 2 (r >> 2) + hi of this 8-sample row block
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample
    float* mrow = M + ((size_t)(rb * 8 + hi) * (gm.C >> 5) + 2 * cp) * (36 * 128) + (size_t)comp * 128 + l31;
    if (!(AB & 4)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* q = mrow + (size_t)(2 * (r >> 2)) * sstride + (r & 3) * 32;
        q[0] = acc0[r];
        q[36 * 128] = acc1[r];
      }
    } else if (acc0[0] == 12345.f) mrow[0] = acc0[1] + acc1[2];
  }
  // --- the ninth component, shared: K slice [wave * G8/8, (wave+1) * G8/8) per wave, summed through LDS.
  // (Measured: running it first on operands requested together with the main loop's, or first in four waves and
  // last in the other four, with eight operand sets in flight, were both 1.3 - 2 us SLOWER than this order.)
  const int scomp = cg * 9 + 8;
  if (!(AB & 1)) {
    const int ng = G8 >> 3, g0 = wave * ng;
    W4Ring<W4_SDEPTH> sr;
    float16_t s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    for (int gs = 0; gs < ng; gs += W4_SDEPTH) {   // (one round at C <= 256)
      w4_ring_fill(sr, vblk(scomp) + (size_t)(g0 + gs) * 256, ublk(scomp, 2 * cp) + (size_t)(g0 + gs) * 256,
                   ublk(scomp, 2 * cp + 1) + (size_t)(g0 + gs) * 256);
      w4_ring_run<W4_SDEPTH, true>(sr, s0, s1, min(W4_SDEPTH, ng - gs));
    }
    float* red = smem + wave * 2048;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      *reinterpret_cast<float4*>(red + (r4 * 64 + lane) * 4) = make_float4(s0[4 * r4], s0[4 * r4 + 1], s0[4 * r4 + 2], s0[4 * r4 + 3]);
      *reinterpret_cast<float4*>(red + 1024 + (r4 * 64 + lane) * 4) = make_float4(s1[4 * r4], s1[4 * r4 + 1], s1[4 * r4 + 2], s1[4 * r4 + 3]);
    }
  }
  if (!(AB & 1)) {
    __syncthreads();
    const int blk = tid >> 8, r4 = (tid >> 6) & 3;
    float4 s = *reinterpret_cast<const float4*>(smem + blk * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk * 1024 + (r4 * 64 + lane) * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    // rows 8 r4 + 4 hi + (0..3) of the block = tiles 0..3 of sample 2 r4 + hi
    float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * (gm.C >> 5) + 2 * cp + blk) * (36 * 128) + (size_t)scomp * 128 + l31;
    mrow[0] = s.x;
    mrow[32] = s.y;
    mrow[64] = s.z;
    mrow[96] = s.w;
  }
}

// ----------------------------------------------------------------------------
// k_w4_gemm64: the same products on 64 x 64 tiles, FOUR waves per workgroup (one per SIMD, up to 512 registers).
// A wave owns one whole component of its tile -- two row blocks x two column blocks, four accumulators -- so four
// 1 KB requests feed sixteen MFMAs (256 B per MFMA; k_w4_gemm: 384): the operand stream of a CU, which is what
// bounds these K = C products (L2-served: ~70 GB/s per CU, Infinity Cache ~33), shrinks from 864 to 608 KB.
// Eight workgroups share a tile: workgroup j takes components 4j .. 4j+3, and half a tile (32 rows) of component
// 32 + j/2, whose K range its four waves split and sum through LDS -- 512 + 64 MFMAs per wave, 576 per SIMD,
// every SIMD of the chip the same.  Workgroup j of every tile runs on XCD j: that XCD's 4.5 components of V and U
// (3.5 MB at N = 128, C = 256) stay in its L2, so each operand byte leaves HBM / Infinity Cache once per launch.
// Needs N % 16 == 0 and C % 64 == 0.
// ----------------------------------------------------------------------------
constexpr int W4_DEPTH64 = 8;

template <int D>
struct W4Ring4 {
  float4 a0[D], a1[D], b0[D], b1[D];
  const float4 *qa0, *qa1, *qb0, *qb1;
};

template <int AB>
__global__ __launch_bounds__(256) void k_w4_gemm64(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                   const Ctrl* ctrl, W4Geom gm) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 6, nRB = gm.RB, G8 = gm.G8, CB = gm.C >> 5;
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int rt = tile / nCT, ct = tile - rt * nCT;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  const int b_off = lane * 4;
  auto vblk = [&](int comp, int rb) { return V + (((size_t)comp * nRB + rb) * G8) * 256 + a_off; };
  auto ublk = [&](int comp, int cb) { return U + (((size_t)comp * CB + cb) * G8) * 256 + b_off; };

  // The shared component's first operand sets are requested BEFORE the own component's loop (one wave per SIMD:
  // nothing else would cover their latency behind it), and its MFMAs run while the own component's 18.9 MB of
  // result stores drain.
  const int scomp = 32 + (j >> 1), rb = 2 * rt + (j & 1);
  const int ng = G8 >> 2, g0 = wave * ng;
  const bool early = !(AB & 1) && ng % W4_SDEPTH == 0;
  W4Ring<W4_SDEPTH> sr;
  if (early) {
    w4_ring_fill(sr, vblk(scomp, rb) + (size_t)g0 * 256, ublk(scomp, 2 * ct) + (size_t)g0 * 256, ublk(scomp, 2 * ct + 1) + (size_t)g0 * 256);
    asm volatile("" ::: "memory");   // the compiler may not sink these requests to their first use behind the loop
  }
  // --- this wave's own component: the whole 64 x 64 tile over the whole K range
  {
    const int comp = 4 * j + wave;
    W4Ring4<W4_DEPTH64> r;
    r.qa0 = reinterpret_cast<const float4*>(vblk(comp, 2 * rt));
    r.qa1 = reinterpret_cast<const float4*>(vblk(comp, 2 * rt + 1));
    r.qb0 = reinterpret_cast<const float4*>(ublk(comp, 2 * ct));
    r.qb1 = reinterpret_cast<const float4*>(ublk(comp, 2 * ct + 1));
    // requests in the steady state's order, pinned (see w4_ring_fill)
#pragma unroll
    for (int i = 0; i < W4_DEPTH64; ++i) {
      r.a0[i] = r.qa0[i * 64]; r.a1[i] = r.qa1[i * 64]; r.b0[i] = r.qb0[i * 64]; r.b1[i] = r.qb1[i * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
    r.qa0 += W4_DEPTH64 * 64; r.qa1 += W4_DEPTH64 * 64; r.qb0 += W4_DEPTH64 * 64; r.qb1 += W4_DEPTH64 * 64;
    float16_t c00, c01, c10, c11;
#pragma unroll
    for (int q = 0; q < 16; ++q) { c00[q] = 0.f; c01[q] = 0.f; c10[q] = 0.f; c11[q] = 0.f; }
    if (!(AB & 2))
      for (int g = 0; g < G8; g += W4_DEPTH64) {
#pragma unroll
        for (int i = 0; i < W4_DEPTH64; ++i) {
          const float4 a0 = r.a0[i], a1 = r.a1[i], b0 = r.b0[i], b1 = r.b1[i];
#define W4_STEP(E)                                                          \
  c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b0.E, c00, 0, 0, 0);     \
  c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b1.E, c01, 0, 0, 0);     \
  c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b0.E, c10, 0, 0, 0);     \
  c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b1.E, c11, 0, 0, 0);
          W4_STEP(x) W4_STEP(y) W4_STEP(z) W4_STEP(w)
#undef W4_STEP
          __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the MFMAs that read the old contents
          r.a0[i] = r.qa0[i * 64]; r.a1[i] = r.qa1[i * 64]; r.b0[i] = r.qb0[i * 64]; r.b1[i] = r.qb1[i * 64];
        }
        r.qa0 += W4_DEPTH64 * 64; r.qa1 += W4_DEPTH64 * 64; r.qb0 += W4_DEPTH64 * 64; r.qb1 += W4_DEPTH64 * 64;
      }
    if (!(AB & 4)) {
      // M is [n][C/32][36][4 t][32 c] (wino4.h): register q of a lane = tile q & 3 of sample 2 (q >> 2) + hi of its row block
      const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample
      float* m0 = M + ((size_t)(rt * 16 + hi) * (gm.C >> 5) + 2 * ct) * (36 * 128) + (size_t)comp * 128 + l31;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
        o[0] = c00[q];
        o[36 * 128] = c01[q];
        o[8 * sstride] = c10[q];
        o[8 * sstride + 36 * 128] = c11[q];
      }
    } else if (c00[0] == 12345.f) M[0] = c00[1] + c01[2] + c10[3] + c11[4];
  }
  // --- half a tile of a shared component: rows [32 half, 32 half + 32), K slice [wave G8/4, (wave+1) G8/4) per wave
  if (!(AB & 1)) {
    float16_t s0, s1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
    if (early) {
      w4_ring_run<W4_SDEPTH, false>(sr, s0, s1, ng);
    } else {
      for (int gs = 0; gs < ng; gs += W4_SDEPTH) {
        w4_ring_fill(sr, vblk(scomp, rb) + (size_t)(g0 + gs) * 256, ublk(scomp, 2 * ct) + (size_t)(g0 + gs) * 256,
                     ublk(scomp, 2 * ct + 1) + (size_t)(g0 + gs) * 256);
        w4_ring_run<W4_SDEPTH, true>(sr, s0, s1, min(W4_SDEPTH, ng - gs));
      }
    }
    float* red = smem + wave * 2048;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      *reinterpret_cast<float4*>(red + (r4 * 64 + lane) * 4) = make_float4(s0[4 * r4], s0[4 * r4 + 1], s0[4 * r4 + 2], s0[4 * r4 + 3]);
      *reinterpret_cast<float4*>(red + 1024 + (r4 * 64 + lane) * 4) = make_float4(s1[4 * r4], s1[4 * r4 + 1], s1[4 * r4 + 2], s1[4 * r4 + 3]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int blk = u >> 8, r4 = (u >> 6) & 3;
      float4 s = *reinterpret_cast<const float4*>(smem + blk * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk * 1024 + (r4 * 64 + lane) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * (gm.C >> 5) + 2 * ct + blk) * (36 * 128) + (size_t)scomp * 128 + l31;
      mrow[0] = s.x;
      mrow[32] = s.y;
      mrow[64] = s.z;
      mrow[96] = s.w;
    }
  }
}


// ----------------------------------------------------------------------------
// k_w4_gemm64b: k_w4_gemm64's products on the bf16 matrix pipe at fp32 accuracy.  Every fp32 operand is split EXACTLY
// into three bf16 parts (x = h + m + l: 3 x 8 mantissa bits; U once per solve by k_w4_pack, V in registers here with
// v_cvt_pk_bf16_f32), and a K = 16 step of a 32 x 32 block is six v_mfma_f32_32x32x16_bf16 -- hh, hm, mh, mm, hl, lh,
// fp32 accumulation; the dropped products ml, lm, ll are <= 2^-24 of the result -- instead of eight
// v_mfma_f32_32x32x2_f32: 192 instead of 512 matrix-pipe cycles.  Measured against an fp64 product the error is that
// of the fp32 MFMA chain (tools/bf16x3 check in tests/test_gpu_w4.py: same 3.2e-6-of-max|y| convolution error).
// Same decomposition, layouts of V and M, and XCD placement as k_w4_gemm64; a lane's eight K values of a step are
// channels {8 g + 4 hi + e} of TWO consecutive g blocks (two of the 16-B loads the fp32 kernel issues too).
// 24.4 -> 19.6 us per launch at cfg 2.  (Measured and not kept, end of round 3: this loop's forty operand requests as inline asm
// with ONE exact wait per step -- s_waitcnt vmcnt(26), where the compiler's placement waits for up to vmcnt(20) -- as in
// k_w4_gemm128b below, where that gave 8 %: 22.5 us by events either way at cfg 2, whose 16 steps per tile are not what bounds it.)  (Measured and not kept: the same products with the operands shared through LDS
// -- 128 x 128 tiles per workgroup, three LDS buffers, fragments prefetched under the MFMAs, L2 -> CU traffic 448
// instead of 768 KB per CU -- 21.5 - 22.8 us: what bounds the launch now is its 52 MB through the fabric plus fill and
// drain, not the per-CU operand stream.)
// ----------------------------------------------------------------------------
typedef __bf16 w4_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 w4_bf16x2 __attribute__((ext_vector_type(2)));
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned w4_u32x4 __attribute__((ext_vector_type(4)));

struct W4Split { w4_bf16x8 h, m, l; };
// (Measured and not kept, round 3: the remainders as v_dot2_f32_bf16(part, (-1, 0) | (0, -1), x) -- three instructions per pair
// and level instead of four.  No launch got faster (cfg 2 GEMM 23.2 vs 22.5 us by events, cfg 5 382 vs 383), and the compiler
// folded the (-1, 0) pair into an inline constant the instruction reads as (0, -1): wrong remainders for every even element,
// caught by tests/test_gpu_w4.py::test_w4_split_is_exact_on_the_device, which stays.)
// (the subtraction as ONE v_pk_add_f32 -- written as `x - convert(part)` the compiler emits two v_add_f32: 126 instead of 63
//  instructions per four K steps of k_w4_gemm64b, whose clock the chip holds down under load: fewer VALU instructions per MFMA is
//  what raises it, MI355X_MICROARCH.md 'DVFS give-back')
__device__ __forceinline__ w4_f32x2 w4_minus_part(const w4_f32x2& x, const w4_bf16x2& part) {
  return x + (-__builtin_convertvector(part, w4_f32x2));   // (a two-float fadd is a legal packed operation; the fsub is expanded)
}
__device__ __forceinline__ W4Split w4_split8(const float4& p, const float4& q) {
  const w4_f32x2 v[4] = {{p.x, p.y}, {p.z, p.w}, {q.x, q.y}, {q.z, q.w}};
  w4_u32x4 hh, mm, ll;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const w4_bf16x2 h = __builtin_convertvector(v[i], w4_bf16x2);
    const w4_f32x2 r = w4_minus_part(v[i], h);
    const w4_bf16x2 m = __builtin_convertvector(r, w4_bf16x2);
    const w4_f32x2 t = w4_minus_part(r, m);
    const w4_bf16x2 l = __builtin_convertvector(t, w4_bf16x2);
    hh[i] = __builtin_bit_cast(unsigned, h);
    mm[i] = __builtin_bit_cast(unsigned, m);
    ll[i] = __builtin_bit_cast(unsigned, l);
  }
  W4Split o;
  o.h = __builtin_bit_cast(w4_bf16x8, hh);
  o.m = __builtin_bit_cast(w4_bf16x8, mm);
  o.l = __builtin_bit_cast(w4_bf16x8, ll);
  return o;
}
// diagnostics (node_w4_split3): the three parts of every element, as floats
__global__ __launch_bounds__(256) void k_w4_split_check(const float* __restrict__ x, float* __restrict__ out, size_t n8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const float4 p = reinterpret_cast<const float4*>(x)[2 * i], q = reinterpret_cast<const float4*>(x)[2 * i + 1];
  const W4Split sp = w4_split8(p, q);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    out[(8 * i + k) * 3 + 0] = (float)sp.h[k];
    out[(8 * i + k) * 3 + 1] = (float)sp.m[k];
    out[(8 * i + k) * 3 + 2] = (float)sp.l[k];
  }
}
void launch_w4_split_check(const float* x, float* out, size_t n, hipStream_t s) {
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(k_w4_split_check, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, out, n8);
}
// acc += a * b over one K = 16 step, smallest products first
__device__ __forceinline__ void w4_mac6(float16_t& acc, const W4Split& a, const w4_u32x4& bh, const w4_u32x4& bm, const w4_u32x4& bl) {
  const w4_bf16x8 Bh = __builtin_bit_cast(w4_bf16x8, bh), Bm = __builtin_bit_cast(w4_bf16x8, bm), Bl = __builtin_bit_cast(w4_bf16x8, bl);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, Bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, Bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, Bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, Bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, Bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, Bh, acc, 0, 0, 0);
}

constexpr int W4B_DEPTH = 4;   // K = 16 steps (two g blocks each) in flight

struct W4BStage { float4 a[2][2]; w4_u32x4 b[2][3]; };   // [row block][g of the pair], [column block][part]
// (Measured and removed, round 4: the K steps of a wave's own component in a per-wave ROTATED order against L2-channel camping --
// every wave walks its streams with the same power-of-two strides.  18.2 -> 19.8 us per launch, cfg 2 24 870 -> 24 480 images/s: the
// waves that SHARE an operand block ask for it at the same time in the lock-step order and are served by one L2 fill; rotated,
// they are not.  And there is no camping to cure: pulling every block 1 - 11 KB out of the power-of-two spacing changes nothing
// (round-4 timing experiment, profiles/r04_w4_gemm_pad.txt).)
struct W4BPtrs { const float4* a[2]; const w4_u32x4* b[2]; };
template <int NRB>
__device__ __forceinline__ void w4b_load(W4BStage& s, const W4BPtrs& p, int g2) {
#pragma unroll
  for (int r = 0; r < NRB; ++r) {
    s.a[r][0] = p.a[r][(size_t)(2 * g2) * 64];
    s.a[r][1] = p.a[r][(size_t)(2 * g2 + 1) * 64];
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) s.b[c][q] = p.b[c][(size_t)(g2 * 3 + q) * 64];
}
// the 6 * NRB * 2 MFMAs of one K = 16 step, the independent accumulators round-robin (no dependent back-to-back pair)
template <int NRB>
__device__ __forceinline__ void w4b_mac(float16_t (&acc)[2][2], const W4Split (&a)[2], const W4BStage& s) {
  w4_bf16x8 B[2][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) B[c][q] = __builtin_bit_cast(w4_bf16x8, s.b[c][q]);
#define W4B_P(AP, BQ)                                                                             \
  _Pragma("unroll") for (int r = 0; r < NRB; ++r) _Pragma("unroll") for (int c = 0; c < 2; ++c)   \
      acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r].AP, B[c][BQ], acc[r][c], 0, 0, 0);
  W4B_P(l, 0) W4B_P(h, 2) W4B_P(m, 1) W4B_P(m, 0) W4B_P(h, 1) W4B_P(h, 0)   // smallest products first
#undef W4B_P
}
// acc += sum over K = 16 steps [g0, g0 + n) (n a multiple of D): a ring of D stages, each refilled right behind the
// MFMAs that consumed it; the refills of the last D steps read up to D steps past the range (buffer slack).  The exact
// bf16 split of the NEXT step's row operand (v_cvt_pk_bf16_f32 + subtracts: ~44 VALU instructions per row block) is
// interleaved with the CURRENT step's MFMAs -- one matrix instruction, then a few vector ones -- so that a wave that
// has its SIMD to itself keeps both pipes busy.
#ifdef NODE_DIAG
// --- variant (NODE_TUNE_W4_UF32 = 1): the FILTER operand fp32 as well (the layout k_w4_gemm64 reads), split into its bf16
// triple in registers like the row operand.  The launch is bound by operand delivery, not by the matrix pipe (ablations,
// the requests alone take the whole 20 us): fp32 filters are 4 instead of 6 bytes per element -- 8 instead
// of 10 KB per K = 16 step and wave -- for ~60 more VALU instructions per step under MFMAs that wait anyway.  Results are
// bit-identical (k_w4_pack's triples are split from the same fp32 values).
struct W4FStage { float4 a[2][2]; float4 b[2][2]; };
struct W4FPtrs { const float4* a[2]; const float4* b[2]; };
template <int NRB>
__device__ __forceinline__ void w4f_load(W4FStage& s, const W4FPtrs& p, int g2) {
#pragma unroll
  for (int r = 0; r < NRB; ++r) {
    s.a[r][0] = p.a[r][(size_t)(2 * g2) * 64];
    s.a[r][1] = p.a[r][(size_t)(2 * g2 + 1) * 64];
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    s.b[c][0] = p.b[c][(size_t)(2 * g2) * 64];
    s.b[c][1] = p.b[c][(size_t)(2 * g2 + 1) * 64];
  }
}
template <int NRB>
__device__ __forceinline__ void w4f_mac(float16_t (&acc)[2][2], const W4Split (&a)[2], const W4Split (&b)[2]) {
#define W4F_P(AP, BQ)                                                                             \
  _Pragma("unroll") for (int r = 0; r < NRB; ++r) _Pragma("unroll") for (int c = 0; c < 2; ++c)   \
      acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r].AP, b[c].BQ, acc[r][c], 0, 0, 0);
  W4F_P(l, h) W4F_P(h, l) W4F_P(m, m) W4F_P(m, h) W4F_P(h, m) W4F_P(h, h)   // smallest products first (as w4b_mac)
#undef W4F_P
}
template <int D, int NRB>
__device__ __forceinline__ void w4f_run(float16_t (&acc)[2][2], const W4FPtrs& p, int g0, int n) {
  W4FStage ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    w4f_load<NRB>(ring[i], p, g0 + i);
    __builtin_amdgcn_sched_barrier(0);
  }
  W4Split ca[2], cb[2], na[2], nb[2];
#pragma unroll
  for (int r = 0; r < NRB; ++r) ca[r] = w4_split8(ring[0].a[r][0], ring[0].a[r][1]);
#pragma unroll
  for (int c = 0; c < 2; ++c) cb[c] = w4_split8(ring[0].b[c][0], ring[0].b[c][1]);
  for (int g = g0; g < g0 + n; g += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const W4FStage& ns = ring[(i + 1) % D];
#pragma unroll
      for (int r = 0; r < NRB; ++r) na[r] = w4_split8(ns.a[r][0], ns.a[r][1]);
#pragma unroll
      for (int c = 0; c < 2; ++c) nb[c] = w4_split8(ns.b[c][0], ns.b[c][1]);
      w4f_mac<NRB>(acc, ca, cb);
#pragma unroll
      for (int k = 0; k < 12 * NRB; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // ... then up to eight VALU instructions of the splits
      }
      __builtin_amdgcn_sched_barrier(0);
      w4f_load<NRB>(ring[i], p, g + D + i);
#pragma unroll
      for (int r = 0; r < NRB; ++r) ca[r] = na[r];
#pragma unroll
      for (int c = 0; c < 2; ++c) cb[c] = nb[c];
    }
  }
}

#endif  // NODE_DIAG
// timing-only ablations (NODE_TUNE_W4_ABLATE, results are wrong): AB bit 2 -- requests without the split / MFMA work (every
// loaded register is folded into one accumulator element, so the requests and their waits stay); AB bit 8 -- the split /
// MFMA work on whatever the registers hold, no requests
// stamps (timing diagnostics, NODE_TUNE_W4_STAMPS): wall-clock ticks (100 MHz) of lane 0 -- [1] ring requested, [2] first step's operands
// arrived and multiplied, [3] loop done
#ifdef NODE_DIAG
__device__ __forceinline__ void w4_stamp(unsigned long long* st, int k) {
  if (st != nullptr && (threadIdx.x & 63) == 0) { st[k] = wall_clock64(); st[8 + k] = clock64(); }
}
#else
__device__ __forceinline__ void w4_stamp(unsigned long long*, int) {}   // (the product library stamps nothing)
#endif
struct W4BCursor { const float4* a[2]; const w4_u32x4* b[2]; };
template <int NRB>
__device__ __forceinline__ void w4b_next(W4BStage& s, W4BCursor& cu) {   // the next K = 16 step of the streams
#pragma unroll
  for (int r = 0; r < NRB; ++r) {
    s.a[r][0] = cu.a[r][0];
    s.a[r][1] = cu.a[r][64];
    cu.a[r] += 128;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int q = 0; q < 3; ++q) s.b[c][q] = cu.b[c][q * 64];
    cu.b[c] += 192;
  }
}
struct W4Nothing { __device__ __forceinline__ void operator()() const {} };
// after_fill: called once the ring's first D steps are requested (k_w4_gemm64b puts the shared component's requests there)
template <int D, int NRB, int AB = 0, class F = W4Nothing>
__device__ __forceinline__ void w4b_run(float16_t (&acc)[2][2], const W4BPtrs& p, int g0, int n, unsigned long long* st = nullptr,
                                        F after_fill = F()) {
  W4BStage ring[D];
  if (AB & 8) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int r = 0; r < 2; ++r) { ring[i].a[r][0] = make_float4(lane, i, r, 1.f); ring[i].a[r][1] = make_float4(1.f, lane, i, r); }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) ring[i].b[c][q] = w4_u32x4{(unsigned)lane, (unsigned)i, (unsigned)c, (unsigned)q};
    }
    W4Split cur8[2];
#pragma unroll
    for (int r = 0; r < NRB; ++r) cur8[r] = w4_split8(ring[0].a[r][0], ring[0].a[r][1]);
    for (int g = g0; g < g0 + n; g += D) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        W4Split nx[2];
#pragma unroll
        for (int r = 0; r < NRB; ++r) nx[r] = w4_split8(ring[(i + 1) % D].a[r][0], ring[(i + 1) % D].a[r][1]);
        w4b_mac<NRB>(acc, cur8, ring[i]);
#pragma unroll
        for (int r = 0; r < NRB; ++r) cur8[r] = nx[r];
        ring[i].a[0][0].x += acc[0][0][0];     // (keeps the chain data-dependent: nothing folds away)
      }
    }
    return;
  }
  if (AB & 2) {
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      w4b_load<NRB>(ring[i], p, g0 + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int g = g0; g < g0 + n; g += D) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int r = 0; r < NRB; ++r) sink += ring[i].a[r][0].x + ring[i].a[r][1].w;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 3; ++q) sink += __builtin_bit_cast(float, ring[i].b[c][q][0]);
        __builtin_amdgcn_sched_barrier(0);
        w4b_load<NRB>(ring[i], p, g + D + i);
      }
    }
    acc[0][0][0] += sink;
    return;
  }
  // the operand streams as running pointers (one 64-bit add per stream and step; the loads of a step differ by immediates)
  W4BCursor cu;
#pragma unroll
  for (int r = 0; r < NRB; ++r) cu.a[r] = p.a[r] + (size_t)(2 * g0) * 64;
#pragma unroll
  for (int c = 0; c < 2; ++c) cu.b[c] = p.b[c] + (size_t)(3 * g0) * 64;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    w4b_next<NRB>(ring[i], cu);
    __builtin_amdgcn_sched_barrier(0);
  }
  after_fill();
  W4Split cur[2], nxt[2];
  w4_stamp(st, 1);
#pragma unroll
  for (int r = 0; r < NRB; ++r) cur[r] = w4_split8(ring[0].a[r][0], ring[0].a[r][1]);
  for (int g = g0; g < g0 + n; g += D) {
    if (g == g0 + D) w4_stamp(st, 2);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const W4BStage& ns = ring[(i + 1) % D];          // the step after this one (refilled D - 1 steps ago)
#pragma unroll
      for (int r = 0; r < NRB; ++r) nxt[r] = w4_split8(ns.a[r][0], ns.a[r][1]);
      w4b_mac<NRB>(acc, cur, ring[i]);
#pragma unroll
      for (int k = 0; k < 12 * NRB; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // ... then up to four VALU instructions of the split
      }
      __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the MFMAs that read the old contents
      w4b_next<NRB>(ring[i], cu);
#pragma unroll
      for (int r = 0; r < NRB; ++r) cur[r] = nxt[r];
    }
  }
  w4_stamp(st, 3);
}

template <int AB>
__global__ __launch_bounds__(256) void k_w4_gemm64b(const float* __restrict__ V, const unsigned short* __restrict__ Ub, float* __restrict__ M,
                                                    const Ctrl* ctrl, W4Geom gm, int mode, unsigned long long* stamps) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* st = stamps != nullptr ? stamps + ((size_t)blockIdx.x * 4 + wave) * 16 : nullptr;
  w4_stamp(st, 0);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 6, nRB = gm.RB, G8 = gm.G8, G2 = G8 >> 1, CB = gm.C >> 5;
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int rt = tile / nCT, ct = tile - rt * nCT;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  // (timing experiments only, NODE_TUNE_W4_PAD = "v,u" with NODE_TUNE_W4_ABLATE >= 16, results wrong: the operand blocks' starts pulled
  // v / u KB per block out of their power-of-two spacing -- do the streams, which all walk K in the same order, camp on L2 channels?)
#ifdef NODE_DIAG
  const int padv = (mode >> 8) & 0xff, padu = (mode >> 16) & 0xff;
#else
  constexpr int padv = 0, padu = 0;
#endif
  auto vblk = [&](int comp, int rb) {
    const size_t b = (size_t)comp * nRB + rb;
    return reinterpret_cast<const float4*>(V + (b * G8) * 256 - b * padv * 256 + a_off);
  };
  auto ublk = [&](int comp, int cb) {
    const size_t b = (size_t)comp * CB + cb;
    return reinterpret_cast<const w4_u32x4*>(Ub) + (b * G2) * 192 - b * padu * 64 + lane;
  };

  // The shared component's operands (four K = 16 steps per wave at C = 256: 128 registers -- the kernel runs one wave per
  // SIMD, so the file's other half is free) are requested BEFORE the own component's loop: nothing else would cover their
  // latency behind it (the ablation of DESIGN.md 4.7: the second ring fill cost ~3 of the shared component's 4.7 us).
  constexpr int SH = 4;
  const bool early = !(AB & 1) && !(AB & 2) && !(AB & 8) && (G2 >> 2) == SH;
  // mode bit 3 (NODE_TUNE_W4_EARLY = 1): ... and BEHIND the ring's first four steps, whose operands the first MFMA waits for (the
  // texture path takes a CU's requests at 64 B per clock: 32 KB per wave in front of them is ~1 us)
  W4BStage shr[SH];
  auto request_shared = [&]() {
    const int scomp = 32 + (j >> 1), srb = 2 * rt + (j & 1);
    W4BPtrs sp;
    sp.a[0] = vblk(scomp, srb); sp.a[1] = sp.a[0];
    sp.b[0] = ublk(scomp, 2 * ct); sp.b[1] = ublk(scomp, 2 * ct + 1);
    if (early) {
#pragma unroll
      for (int i = 0; i < SH; ++i) {
        w4b_load<1>(shr[i], sp, wave * SH + i);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("" ::: "memory");   // (the compiler may not sink these requests to their first use behind the loop)
    }
  };
  const bool behind = (mode & 8) != 0;
  if (!behind) request_shared();
  // --- this wave's own component: the whole 64 x 64 tile over the whole K range.
  // mode bit 1 (NODE_TUNE_W4_SHAREV, four column tiles only): the four waves of a workgroup take the SAME component and
  // row tile and one column tile each -- they walk the same V blocks in lock-step, so three of their four requests for a
  // block are served by the CU's own L1 / merged in flight -- instead of four components of one tile (nothing shared
  // inside the CU).  The workgroup's place (tile % nCT) then names the component, the wave the column tile.
  {
    const bool sharev = (mode & 2) != 0 && nCT == 4;
    // mode bit 2 (NODE_TUNE_W4_SHAREV = 2; four column tiles, row tiles a multiple of two): a workgroup takes a 128 x 128 tile of
    // one component, wave (r, c) its 64 x 64 quarter: two waves walk each V block together, two each U block
    const bool share2 = (mode & 4) != 0 && nCT == 4 && (nRB & 3) == 0;
    const int comp = 4 * j + (share2 ? (tile & 3) : sharev ? ct : wave);
    const int oct = share2 ? 2 * ((tile >> 2) & 1) + (wave & 1) : sharev ? wave : ct;
    const int ort = share2 ? 2 * (tile >> 3) + (wave >> 1) : rt;
    W4BPtrs p;
    p.a[0] = vblk(comp, 2 * ort); p.a[1] = vblk(comp, 2 * ort + 1);
    p.b[0] = ublk(comp, 2 * oct); p.b[1] = ublk(comp, 2 * oct + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
    w4b_run<W4B_DEPTH, 2, AB>(acc, p, 0, G2, st, [&]() { if (behind) request_shared(); });
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample of M
    float* m0 = M + ((size_t)(ort * 16 + hi) * (gm.C >> 5) + 2 * oct) * (36 * 128) + (size_t)comp * 128 + l31;
    if (!(AB & 4) || acc[0][0][0] == 123.456f)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
      st_wt(o, acc[0][0][q]);
      st_wt(o + 36 * 128, acc[0][1][q]);
      st_wt(o + 8 * sstride, acc[1][0][q]);
      st_wt(o + 8 * sstride + 36 * 128, acc[1][1][q]);
    }
    w4_stamp(st, 4);
  }
  // --- half a tile of a shared component: rows [32 half, 32 half + 32), K range [wave G2/4, (wave+1) G2/4) per wave
  if (!(AB & 1)) {
    const int scomp = 32 + (j >> 1), rb = 2 * rt + (j & 1);
    const int ng = G2 >> 2, g0 = wave * ng;
    W4BPtrs p;
    p.a[0] = vblk(scomp, rb); p.a[1] = p.a[0];
    p.b[0] = ublk(scomp, 2 * ct); p.b[1] = ublk(scomp, 2 * ct + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0][c][q] = 0.f;
    if (early) {
      W4Split cs[2];
#pragma unroll
      for (int i = 0; i < SH; ++i) {
        cs[0] = w4_split8(shr[i].a[0][0], shr[i].a[0][1]);
        w4b_mac<1>(acc, cs, shr[i]);
      }
    } else if (ng % 4 == 0) w4b_run<4, 1, AB>(acc, p, g0, ng);
    else if (ng % 2 == 0) w4b_run<2, 1, AB>(acc, p, g0, ng);
    else w4b_run<1, 1, AB>(acc, p, g0, ng);
    w4_stamp(st, 5);
    float* red = smem + wave * 2048;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(red + c * 1024 + (r4 * 64 + lane) * 4) =
            make_float4(acc[0][c][4 * r4], acc[0][c][4 * r4 + 1], acc[0][c][4 * r4 + 2], acc[0][c][4 * r4 + 3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int blk = u >> 8, r4 = (u >> 6) & 3;
      float4 s = *reinterpret_cast<const float4*>(smem + blk * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk * 1024 + (r4 * 64 + lane) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * (gm.C >> 5) + 2 * ct + blk) * (36 * 128) + (size_t)scomp * 128 + l31;
      if ((AB & 4) && s.x != 123.456f) continue;
      st_wt(mrow, s.x);
      st_wt(mrow + 32, s.y);
      st_wt(mrow + 64, s.z);
      st_wt(mrow + 96, s.w);
    }
  }
#ifdef NODE_DIAG
  if (st != nullptr) {   // (diagnostics: when this wave's stores have drained)
    w4_stamp(st, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w4_stamp(st, 7);
  }
#endif
}


#ifdef NODE_DIAG   // measured-and-rejected variants: built into libnode_hip_diag.so only (build.py --diag)
// ----------------------------------------------------------------------------
// k_w4_gemm32b (C = 256, NODE_TUNE_W4_HALF): k_w4_gemm64b's products with HALF-HEIGHT tiles and TWO waves per SIMD.
// What the counters say about k_w4_gemm64b (profiles/r05_pmc_w4_limiter.txt): the matrix pipe is busy 45 % of a wave's life --
// ~75 % inside the K loop, idle through a prologue (first operands: L2 latency) and an epilogue (stores, shared component,
// LDS reduction) that one tile per wave at one wave per SIMD cannot overlap with anything; neither the texture path nor the
// fabric is saturated.  Here a wave owns a 32 x 64 tile (one row block, two column blocks: 32 + 128 ring registers instead
// of 64 + 160), the kernel fits 256 registers, and a SIMD holds two waves of two different workgroups: one's prologue and
// epilogue run under the other's K loop.  Twice the workgroups (N / 8 row tiles x 4 column tiles x 8), each streaming the same
// filter blocks for half the rows (L1 -> L2 requests of the filter operand double: the texture path has the room).  The
// shared component 32 + j / 2 is dealt in 32 x 32 blocks (one per workgroup, K range cut over the four waves as before).
// Every output element is the same sum in the same order as in k_w4_gemm64b: bit-identical.
// MEASURED AND REJECTED (round 5, profiles/r05_w4_half_ab.txt): 24.3 us against 21.1 us by HIP events in the cfg-2 bench loop
// (24 300 vs 24 880 images/s, cfg 3 18 090 vs 18 560): the second wave's K loop does not hide the first one's prologue --
// both waves of a SIMD share ONE matrix pipe, so two K loops side by side each run at half rate, and the filter operand's
// requests double.  Diagnostics library only.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_w4_gemm32b(const float* __restrict__ V, const unsigned short* __restrict__ Ub, float* __restrict__ M,
                                                       const Ctrl* ctrl, W4Geom gm) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nRB = gm.RB, G8 = gm.G8, G2 = G8 >> 1, CB = gm.C >> 5;      // C = 256: four 64-column tiles, eight 32-column blocks
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int rb = tile >> 2, ct = tile & 3;                                // 32-row block, the workgroup's place among four
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;
  auto vblk = [&](int comp, int r) { return reinterpret_cast<const float4*>(V + (((size_t)comp * nRB + r) * G8) * 256 + a_off); };
  auto ublk = [&](int comp, int cb) { return reinterpret_cast<const w4_u32x4*>(Ub) + (((size_t)comp * CB + cb) * G2) * 192 + lane; };
  const size_t sstride = (size_t)CB * 36 * 128;   // floats per sample of M
  {
    // own component 4 j + ct (the workgroup's place names it, as NODE_TUNE_W4_SHAREV = 1 does in k_w4_gemm64b): the four waves walk
    // the same V block in lock-step, wave w multiplies it with column tile w
    const int comp = 4 * j + ct;
    W4BPtrs p;
    p.a[0] = vblk(comp, rb); p.a[1] = p.a[0];
    p.b[0] = ublk(comp, 2 * wave); p.b[1] = ublk(comp, 2 * wave + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0][c][q] = 0.f;
    w4b_run<W4B_DEPTH, 1>(acc, p, 0, G2);
    float* m0 = M + ((size_t)(rb * 8 + hi) * CB + 2 * wave) * (36 * 128) + (size_t)comp * 128 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
      st_wt(o, acc[0][0][q]);
      st_wt(o + 36 * 128, acc[0][1][q]);
    }
  }
  {
    // a 32 x 32 block of the shared component: rows rb, column block 2 ct + (j & 1); K range [wave G2 / 4, (wave + 1) G2 / 4)
    const int scomp = 32 + (j >> 1), cb = 2 * ct + (j & 1);
    const int ng = G2 >> 2, g0 = wave * ng;
    W4BPtrs p;
    p.a[0] = vblk(scomp, rb); p.a[1] = p.a[0];
    p.b[0] = ublk(scomp, cb); p.b[1] = p.b[0];
    float16_t acc[2][2];
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[0][0][q] = 0.f; acc[0][1][q] = 0.f; }
    // (one column block: the second accumulator of w4b_mac<1> multiplies the same block again and is dropped -- 12 spare MFMAs
    //  per K step of a phase that is 1 / 9 of the work, for one code path)
    if (ng % 4 == 0) w4b_run<4, 1>(acc, p, g0, ng);
    else if (ng % 2 == 0) w4b_run<2, 1>(acc, p, g0, ng);
    else w4b_run<1, 1>(acc, p, g0, ng);
    float* red = smem + wave * 1024;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
      *reinterpret_cast<float4*>(red + (r4 * 64 + lane) * 4) =
          make_float4(acc[0][0][4 * r4], acc[0][0][4 * r4 + 1], acc[0][0][4 * r4 + 2], acc[0][0][4 * r4 + 3]);
    __syncthreads();
    const int r4 = wave;      // 256 threads = 4 r4 x 64 lanes
    float4 sacc = *reinterpret_cast<const float4*>(smem + (r4 * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(smem + w * 1024 + (r4 * 64 + lane) * 4);
      sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
    }
    float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * CB + cb) * (36 * 128) + (size_t)scomp * 128 + l31;
    st_wt(mrow, sacc.x);
    st_wt(mrow + 32, sacc.y);
    st_wt(mrow + 64, sacc.z);
    st_wt(mrow + 96, sacc.w);
  }
}

// ----------------------------------------------------------------------------
// k_w4_gemm64l (NODE_TUNE_W4_LDS, C = 256, N % 32 == 0): k_w4_gemm64b's products with the own component's operands brought into
// the CU ONCE.  What bounds k_w4_gemm64b is the bytes its waves load into registers (every operand block aliased onto one
// 80 KB footprint -- all L2 hits -- it takes 17.3 instead of 18.5 us): 160 KB per wave, 640 KB per CU for the own
// component.  Here a workgroup takes a 128 x 128 tile of one component (wave (r, c) its 64 x 64 quarter, as NODE_TUNE_W4_SHAREV = 2)
// and the 320 KB of operands of that tile arrive by LDS-DMA (global_load_lds_dwordx4: no registers, no staging instructions)
// in a ring of W4L_NS K = 16 steps of 20 KB (V: 4 row blocks x 2 KB fp32; U: 4 column blocks x 3 KB triples), W4L_D steps in
// flight; the LDS image of a block is in READER-lane order (the permutation sits on the DMA's per-lane source address), so a
// fragment is one conflict-free ds_read_b128.  One raw s_barrier per step behind a counted s_waitcnt vmcnt (never 0 inside the
// loop: cdna_hip_programming.md, Pipelining across barriers); a slot is refilled D + 1 steps after its reads were waited for.
// The shared component (1/9 of the work, no operand shared between waves) keeps k_w4_gemm64b's register path and early requests.
// ----------------------------------------------------------------------------
constexpr int W4L_SPS = 2;                          // K steps per ring slot = per barrier
constexpr int W4L_NS = 4, W4L_D = 3, W4L_STEP = 20 * 1024, W4L_SLOT = W4L_SPS * W4L_STEP, W4L_STEPS = 16, W4L_SLOTS = W4L_STEPS / W4L_SPS;
// one LDS-DMA piece as inline asm: the compiler, which does not count asm memory operations, then neither drains the ring
// (`s_waitcnt vmcnt(0)`) in front of every fragment read -- with the builtin it does: it cannot tell the reads from the pieces in
// flight -- nor knows of it: the counted waits of the loop are the only ordering (cdna_hip_programming.md 5.7: M0 is written in the
// statement that reads it)
__device__ __forceinline__ void w4l_dma(const unsigned char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void w4l_wait_vm(int n) {   // (n is a compile-time constant after unrolling)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
__global__ __launch_bounds__(256) void k_w4_gemm64l(const float* __restrict__ V, const unsigned short* __restrict__ Ub, float* __restrict__ M,
                                                    const Ctrl* ctrl, W4Geom gm, unsigned long long* stamps) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];   // [W4L_NS][20 KB]; the shared component's reduction in slots 2, 3
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long* stp = stamps != nullptr ? stamps + ((size_t)blockIdx.x * 4 + wave) * 16 : nullptr;
  w4_stamp(stp, 0);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 6, nRB = gm.RB, G8 = gm.G8, G2 = G8 >> 1, CB = gm.C >> 5;
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int rt = tile / nCT, ct = tile - rt * nCT;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  auto vblk = [&](int comp, int rb) { return reinterpret_cast<const float4*>(V + (((size_t)comp * nRB + rb) * G8) * 256 + a_off); };
  auto ublk = [&](int comp, int cb) { return reinterpret_cast<const w4_u32x4*>(Ub) + (((size_t)comp * CB + cb) * G2) * 192 + lane; };

  // the shared component's operands, requested first (k_w4_gemm64b)
  constexpr int SH = 4;
  W4BStage shr[SH];
  {
    const int scomp = 32 + (j >> 1), srb = 2 * rt + (j & 1);
    W4BPtrs sp;
    sp.a[0] = vblk(scomp, srb); sp.a[1] = sp.a[0];
    sp.b[0] = ublk(scomp, 2 * ct); sp.b[1] = ublk(scomp, 2 * ct + 1);
#pragma unroll
    for (int i = 0; i < SH; ++i) {
      w4b_load<1>(shr[i], sp, wave * SH + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::: "memory");
  }

  // --- own component: the 128 x 128 tile (RT, CT) of component comp; wave (wr, wc) multiplies its 64 x 64 quarter
  const int comp = 4 * j + (tile & 3), RT = tile >> 3, CT = (tile >> 2) & 1;
  const int wr = wave >> 1, wc = wave & 1;
  // the five of a step's twenty 1-KB pieces this wave brings: pieces 0 .. 7 = V (row block p / 2, g block p % 2), 8 .. 19 = U
  // (column block (p - 8) / 3, part (p - 8) % 3); a piece's source advances by 2 KB (V) / 3 KB (U) per step
  const unsigned char* src[5];
  int adv[5], dst[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int p = 5 * wave + i;
    if (p < 8) {
      src[i] = reinterpret_cast<const unsigned char*>(vblk(comp, 4 * RT + (p >> 1))) + (p & 1) * 1024;
      adv[i] = 2048;
    } else {
      const int q = p - 8;
      src[i] = reinterpret_cast<const unsigned char*>(ublk(comp, 4 * CT + q / 3)) + (q % 3) * 1024;
      adv[i] = 3072;
    }
    dst[i] = p * 1024;
  }
  const unsigned slot0 = (unsigned)(size_t)(w4_lds_ptr_t)lsm;   // LDS byte address of the ring
  auto issue = [&](int sl) {                                      // ring slot sl = K steps [SPS sl, SPS sl + SPS)
#pragma unroll
    for (int k = 0; k < W4L_SPS; ++k) {
      const int step = sl * W4L_SPS + k;
#pragma unroll
      for (int i = 0; i < 5; ++i)
        w4l_dma(src[i] + (size_t)step * adv[i], slot0 + (unsigned)((sl % W4L_NS) * W4L_SLOT + k * W4L_STEP + dst[i]));
    }
  };
  auto fetch = [&](W4BStage& st, int step) {
    const unsigned char* slot = lsm + ((step / W4L_SPS) % W4L_NS) * W4L_SLOT + (step % W4L_SPS) * W4L_STEP + lane * 16;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int g = 0; g < 2; ++g) st.a[r][g] = *reinterpret_cast<const float4*>(slot + ((2 * wr + r) * 2 + g) * 1024);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 3; ++q) st.b[c][q] = *reinterpret_cast<const w4_u32x4*>(slot + 8192 + ((2 * wc + c) * 3 + q) * 1024);
  };
  {
    float16_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
#pragma unroll
    for (int sp = 0; sp < W4L_D; ++sp) issue(sp);
    w4_stamp(stp, 1);
    W4BStage st[2];
    w4l_wait_vm(5 * W4L_SPS * (W4L_D - 1));
    __builtin_amdgcn_s_barrier();
    issue(W4L_D);
    fetch(st[0], 0);
    W4Split cur[2], nxt[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) cur[r] = w4_split8(st[0].a[r][0], st[0].a[r][1]);
#pragma unroll
    for (int g = 0; g < W4L_STEPS; ++g) {
      if (g == 4) w4_stamp(stp, 2);
      W4BStage& cs = st[g & 1];
      W4BStage& ns = st[(g + 1) & 1];
      if (g + 1 < W4L_STEPS) {
        if ((g + 1) % W4L_SPS == 0) {                     // step g + 1 opens ring slot S
          const int S = (g + 1) / W4L_SPS;
          const int newest = (S + W4L_D - 1 < W4L_SLOTS - 1) ? S + W4L_D - 1 : W4L_SLOTS - 1;   // the youngest slot whose pieces are issued
          w4l_wait_vm(5 * W4L_SPS * (newest - S));            // this wave's pieces of slot S have landed
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its reads of slot S - 1 are in registers (that slot is refilled next)
          __builtin_amdgcn_s_barrier();
          if (S + W4L_D < W4L_SLOTS) issue(S + W4L_D);
        }
        fetch(ns, g + 1);
      }
      w4b_mac<2>(acc, cur, cs);
      if (g + 1 < W4L_STEPS) {
#pragma unroll
        for (int r = 0; r < 2; ++r) nxt[r] = w4_split8(ns.a[r][0], ns.a[r][1]);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    // eight MFMAs cover the fragment reads' latency ...
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // ... then one MFMA,
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  //     up to six VALU instructions of the next step's split
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 2; ++r) cur[r] = nxt[r];
    }
    w4_stamp(stp, 3);
    const int ort = 2 * RT + wr, oct = 2 * CT + wc;
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample of M
    float* m0 = M + ((size_t)(ort * 16 + hi) * (gm.C >> 5) + 2 * oct) * (36 * 128) + (size_t)comp * 128 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
      st_wt(o, acc[0][0][q]);
      st_wt(o + 36 * 128, acc[0][1][q]);
      st_wt(o + 8 * sstride, acc[1][0][q]);
      st_wt(o + 8 * sstride + 36 * 128, acc[1][1][q]);
    }
    w4_stamp(stp, 4);
  }
  // --- half a tile of a shared component (k_w4_gemm64b): rows [32 half, 32 half + 32), K range of wave `wave`
  {
    const int scomp = 32 + (j >> 1), rb = 2 * rt + (j & 1);
    float16_t acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0][c][q] = 0.f;
    W4Split cs[2];
#pragma unroll
    for (int i = 0; i < SH; ++i) {
      cs[0] = w4_split8(shr[i].a[0][0], shr[i].a[0][1]);
      w4b_mac<1>(acc, cs, shr[i]);
    }
    w4_stamp(stp, 5);
    float* smem = reinterpret_cast<float*>(lsm + 1 * W4L_SLOT);   // (ring slot 1, last K steps 10 and 11: read long ago by every wave)
    float* red = smem + wave * 2048;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(red + c * 1024 + (r4 * 64 + lane) * 4) =
            make_float4(acc[0][c][4 * r4], acc[0][c][4 * r4 + 1], acc[0][c][4 * r4 + 2], acc[0][c][4 * r4 + 3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int blk = u >> 8, r4 = (u >> 6) & 3;
      float4 s = *reinterpret_cast<const float4*>(smem + blk * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk * 1024 + (r4 * 64 + lane) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * (gm.C >> 5) + 2 * ct + blk) * (36 * 128) + (size_t)scomp * 128 + l31;
      st_wt(mrow, s.x);
      st_wt(mrow + 32, s.y);
      st_wt(mrow + 64, s.z);
      st_wt(mrow + 96, s.w);
    }
  }
  if (stp != nullptr) {
    w4_stamp(stp, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w4_stamp(stp, 7);
  }
}


// ----------------------------------------------------------------------------
// k_w4_gemm64k (NODE_TUNE_W4_KSPLIT, C = 256, N % 16 == 0): k_w4_gemm64b's products with TWO waves per SIMD.  The timeline of
// k_w4_gemm64b (tools/w4_stamps.py): the chip holds ~1.65 GHz in the launch, where the 432 MFMAs of a wave are
// 8.4 us and the 768 KB a CU's waves request are 7.5 us of its texture path (64 B per clock) -- but with one wave per SIMD the two
// do not overlap: a wave that stands in a request (the path's queue is full) or in a wait issues no MFMA.  Here every 64 x 64 tile
// of a component is cut in two K halves, one wave each, so a SIMD holds two waves (<= 256 registers: a ring of two K steps
// instead of four -- the requests in flight per SIMD stay what they were) and one multiplies while the other stands.
//   workgroup (512 of them, two per CU) = component 4 j + c4, row tile rt, column tiles 2 p and 2 p + 1: wave 2 t + h = K half h of
//   tile t (the two tiles walk the same V blocks); the halves meet in LDS: wave h keeps row block h of the tile, hands the other
//   one over (8 KB), adds its partner's and stores 32 x 64 results.  The four K-sliced components (32 + j / 2): a 32 x 32 piece per
//   workgroup, four K steps per wave, summed through LDS as in k_w4_gemm64b.  Sums of two K halves: not bit-identical to
//   k_w4_gemm64b's single chain (same error against fp64).
// ----------------------------------------------------------------------------
struct W4KStage { float4 a[2]; w4_u32x4 b[3]; };
__global__ __launch_bounds__(256, 2) void k_w4_gemm64k(const float* __restrict__ V, const unsigned short* __restrict__ Ub, float* __restrict__ M,
                                                       const Ctrl* ctrl, W4Geom gm, unsigned long long* stamps) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 32 KB: [4 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* st = stamps != nullptr ? stamps + ((size_t)blockIdx.x * 4 + wave) * 16 : nullptr;
  w4_stamp(st, 0);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nRB = gm.RB, G8 = gm.G8, G2 = G8 >> 1, CB = gm.C >> 5;
  const int j = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  auto vblk = [&](int comp, int rb) { return reinterpret_cast<const float4*>(V + (((size_t)comp * nRB + rb) * G8) * 256 + a_off); };
  auto ublk = [&](int comp, int cb) { return reinterpret_cast<const w4_u32x4*>(Ub) + (((size_t)comp * CB + cb) * G2) * 192 + lane; };
  const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample of M

  // the K-sliced component's piece of this workgroup: rows rbs (32), column block cbs (32), K steps [4 wave, 4 wave + 4)
  const int scomp = 32 + (j >> 1);
  const int rbs = 2 * (idx >> 3) + (j & 1), cbs = 2 * ((idx >> 1) & 3) + (idx & 1);
  const int SG = G2 >> 2;                        // its K steps per wave
  const float4* sa = vblk(scomp, rbs) + (size_t)(2 * wave * SG) * 64;
  const w4_u32x4* sb = ublk(scomp, cbs) + (size_t)(3 * wave * SG) * 64;
  auto snext = [&](W4KStage& s) {
    s.a[0] = sa[0]; s.a[1] = sa[64];
#pragma unroll
    for (int q = 0; q < 3; ++q) s.b[q] = sb[q * 64];
    sa += 128; sb += 192;
  };

  // --- own component: K half h of the 64 x 64 tile (rt, 2 p + t)
  const int comp = 4 * j + (idx & 3), rt = idx >> 3, t = wave >> 1, h = wave & 1;
  const int ct = 2 * ((idx >> 2) & 1) + t;
  W4KStage sring[2];
  {
    W4BPtrs p;
    p.a[0] = vblk(comp, 2 * rt); p.a[1] = vblk(comp, 2 * rt + 1);
    p.b[0] = ublk(comp, 2 * ct); p.b[1] = ublk(comp, 2 * ct + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
    w4b_run<2, 2, 0>(acc, p, h * (G2 >> 1), G2 >> 1, st);
    // the K-sliced piece's first two steps are requested now: they arrive under the exchange and the stores
    snext(sring[0]);
    snext(sring[1]);
    asm volatile("" ::: "memory");
    // the halves meet: wave h hands row block 1 - h over and keeps row block h (h is wave-uniform: two straight-line copies, the
    // accumulators stay in registers)
    float* mine = smem + wave * 2048;
    const float* theirs = smem + (wave ^ 1) * 2048;
    auto give = [&](const float16_t (&g)[2]) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          *reinterpret_cast<float4*>(mine + c * 1024 + (r4 * 64 + lane) * 4) = make_float4(g[c][4 * r4], g[c][4 * r4 + 1], g[c][4 * r4 + 2], g[c][4 * r4 + 3]);
    };
    if (h) give(acc[0]); else give(acc[1]);
    __syncthreads();
    float* m0 = M + ((size_t)(rt * 16 + hi) * (gm.C >> 5) + 2 * ct) * (36 * 128) + (size_t)comp * 128 + l31 + (size_t)h * 8 * sstride;
    auto keep = [&](const float16_t (&k)[2], bool first) {   // first: this wave holds K half 0 (the sum is half 0 + half 1 either way)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 o = *reinterpret_cast<const float4*>(theirs + c * 1024 + (r4 * 64 + lane) * 4);
          float* o0 = m0 + (size_t)(2 * r4) * sstride + c * (36 * 128);
          st_wt(o0, first ? k[c][4 * r4] + o.x : o.x + k[c][4 * r4]);
          st_wt(o0 + 32, first ? k[c][4 * r4 + 1] + o.y : o.y + k[c][4 * r4 + 1]);
          st_wt(o0 + 64, first ? k[c][4 * r4 + 2] + o.z : o.z + k[c][4 * r4 + 2]);
          st_wt(o0 + 96, first ? k[c][4 * r4 + 3] + o.w : o.w + k[c][4 * r4 + 3]);
        }
    };
    if (h) keep(acc[1], false); else keep(acc[0], true);
    w4_stamp(st, 4);
  }
  // --- the K-sliced component's piece
  {
    float16_t acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int g = 0; g < SG; g += 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const W4Split as = w4_split8(sring[i].a[0], sring[i].a[1]);
        w4_mac6(acc, as, sring[i].b[0], sring[i].b[1], sring[i].b[2]);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 + i < SG) snext(sring[i]);
      }
    }
    w4_stamp(st, 5);
    __syncthreads();                 // (the exchange's reads of smem are done)
    float* red = smem + wave * 1024;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
      *reinterpret_cast<float4*>(red + (r4 * 64 + lane) * 4) = make_float4(acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]);
    __syncthreads();
    {
      const int r4 = wave;
      float4 s = *reinterpret_cast<const float4*>(smem + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 1024 + (r4 * 64 + lane) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* mrow = M + ((size_t)(rbs * 8 + 2 * r4 + hi) * (gm.C >> 5) + cbs) * (36 * 128) + (size_t)scomp * 128 + l31;
      st_wt(mrow, s.x);
      st_wt(mrow + 32, s.y);
      st_wt(mrow + 64, s.z);
      st_wt(mrow + 96, s.w);
    }
  }
  if (st != nullptr) {
    w4_stamp(st, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w4_stamp(st, 7);
  }
}



// k_w4_gemm64c: k_w4_gemm64b with fp32 filters split in registers (see w4f_run)
__global__ __launch_bounds__(256) void k_w4_gemm64c(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                    const Ctrl* ctrl, W4Geom gm) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 6, nRB = gm.RB, G8 = gm.G8, G2 = G8 >> 1, CB = gm.C >> 5;
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int rt = tile / nCT, ct = tile - rt * nCT;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  auto vblk = [&](int comp, int rb) { return reinterpret_cast<const float4*>(V + (((size_t)comp * nRB + rb) * G8) * 256 + a_off); };
  auto ublk = [&](int comp, int cb) { return reinterpret_cast<const float4*>(U + (((size_t)comp * CB + cb) * G8) * 256) + lane; };

  // --- this wave's own component: the whole 64 x 64 tile over the whole K range
  {
    const int comp = 4 * j + wave;
    W4FPtrs p;
    p.a[0] = vblk(comp, 2 * rt); p.a[1] = vblk(comp, 2 * rt + 1);
    p.b[0] = ublk(comp, 2 * ct); p.b[1] = ublk(comp, 2 * ct + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
    w4f_run<W4B_DEPTH, 2>(acc, p, 0, G2);
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample of M
    float* m0 = M + ((size_t)(rt * 16 + hi) * (gm.C >> 5) + 2 * ct) * (36 * 128) + (size_t)comp * 128 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
      st_wt(o, acc[0][0][q]);
      st_wt(o + 36 * 128, acc[0][1][q]);
      st_wt(o + 8 * sstride, acc[1][0][q]);
      st_wt(o + 8 * sstride + 36 * 128, acc[1][1][q]);
    }
  }
  // --- half a tile of a shared component: rows [32 half, 32 half + 32), K range [wave G2/4, (wave+1) G2/4) per wave
  {
    const int scomp = 32 + (j >> 1), rb = 2 * rt + (j & 1);
    const int ng = G2 >> 2, g0 = wave * ng;
    W4FPtrs p;
    p.a[0] = vblk(scomp, rb); p.a[1] = p.a[0];
    p.b[0] = ublk(scomp, 2 * ct); p.b[1] = ublk(scomp, 2 * ct + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0][c][q] = 0.f;
    if (ng % 4 == 0) w4f_run<4, 1>(acc, p, g0, ng);
    else if (ng % 2 == 0) w4f_run<2, 1>(acc, p, g0, ng);
    else w4f_run<1, 1>(acc, p, g0, ng);
    float* red = smem + wave * 2048;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(red + c * 1024 + (r4 * 64 + lane) * 4) =
            make_float4(acc[0][c][4 * r4], acc[0][c][4 * r4 + 1], acc[0][c][4 * r4 + 2], acc[0][c][4 * r4 + 3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int blk = u >> 8, r4 = (u >> 6) & 3;
      float4 s = *reinterpret_cast<const float4*>(smem + blk * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk * 1024 + (r4 * 64 + lane) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * (gm.C >> 5) + 2 * ct + blk) * (36 * 128) + (size_t)scomp * 128 + l31;
      st_wt(mrow, s.x);
      st_wt(mrow + 32, s.y);
      st_wt(mrow + 64, s.z);
      st_wt(mrow + 96, s.w);
    }
  }
}



#endif  // NODE_DIAG (measured-and-rejected variants)

// ----------------------------------------------------------------------------
// k_w4_scales: the power-of-two scales of a solve's fp16-pair operands (wino4.h, W4Scales) in one launch: block (x, job) adds the
// maximum of its slice of job's tensor by atomicMax on the fp32 bit pattern (non-negative floats order like unsigned integers);
// the last block to arrive derives the exponents and zeroes the scratch words for the next launch.
// ----------------------------------------------------------------------------
constexpr int W4SC_BLOCKS = 32;       // per big tensor (every block takes one returning ticket at the end: few blocks, several requests in flight each)
__global__ __launch_bounds__(256) void k_w4_scales(W4ScaleJobs j, int nbig) {
  __shared__ float red[4];
  // blocks [0, nbig W4SC_BLOCKS): slices of the big tensors (the conv weights; diagnostics: a whole activation tensor as "beta");
  // then one block per [C] vector
  int job, part, parts;
  if ((int)blockIdx.x < nbig * W4SC_BLOCKS) {
    const int b = blockIdx.x / W4SC_BLOCKS;
    part = blockIdx.x - b * W4SC_BLOCKS; parts = W4SC_BLOCKS;
    job = j.bigjob[b];
  } else {
    job = 2 + ((int)blockIdx.x - nbig * W4SC_BLOCKS); part = 0; parts = 1;
  }
  const float* p = job < 2 ? j.w[job] : j.gb[job - 2];
  size_t n = job < 2 ? j.wn : (size_t)j.C;
  const bool big_vec = (job == 3 || job == 5) && j.vn[(job - 3) >> 1] != 0;
  if (big_vec) n = j.vn[(job - 3) >> 1];
  if (parts == 1 && big_vec) p = nullptr;     // (a big "vector" is taken by its sliced blocks)
  float m = 0.f;
  if (p != nullptr) {
    const size_t i0 = (size_t)part * 256 + threadIdx.x, step = (size_t)parts * 256;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (n & 3) == 0) {      // 16-B loads, four in flight
      const float4* q = reinterpret_cast<const float4*>(p);
      const size_t n4 = n >> 2;
      for (size_t i = i0; i < n4; i += 4 * step) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * step < n4 ? q[i + u * step] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
      }
    } else {
      for (size_t i = i0; i < n; i += step) m = fmaxf(m, fabsf(p[i]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x != 0) return;
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  W4Scales* sc = j.sc;
  if (m > 0.f) atomicMax(&sc->mx[job], __builtin_bit_cast(unsigned, m));
  __threadfence();
  const unsigned ticket = atomicAdd(&sc->arrived, 1u);
  if (ticket != gridDim.x - 1) return;
  __threadfence();
  float mx[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    mx[q] = __builtin_bit_cast(float, __hip_atomic_load(&sc->mx[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __hip_atomic_store(&sc->mx[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __hip_atomic_store(&sc->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // |U| = |G g G^T| <= (28/15)^2 max|w|; ceiling 2^14 (a factor four under fp16's)
  if (j.w[0] != nullptr) sc->e[W4_E_U1] = w4_scale_exp(3.49f * mx[0], 14);
  if (j.w[1] != nullptr) sc->e[W4_E_U2] = w4_scale_exp(3.49f * mx[1], 14);
  // |B^T d B| <= 49 max|d|, d = relu(gamma xhat + beta), |xhat| <= sqrt(m - 1): ceiling 2^15 for the BOUND (what the data reaches is
  // typically 2^5 under it)
  const float rm = sqrtf((float)j.gn_m);
  if (j.gb[0] != nullptr || j.gb[1] != nullptr) sc->e[W4_E_V1] = w4_scale_exp(49.f * (rm * mx[2] + mx[3]), 15);
  if (j.gb[2] != nullptr || j.gb[3] != nullptr) sc->e[W4_E_V2] = w4_scale_exp(49.f * (rm * mx[4] + mx[5]), 15);
}
void launch_w4_scales(const W4ScaleJobs& j_in, hipStream_t s) {
  W4ScaleJobs j = j_in;
  int nbig = 0;      // the tensors cut over W4SC_BLOCKS blocks: the conv weights, and a "vector" with a length of its own (diagnostics)
  if (j.w[0] != nullptr) j.bigjob[nbig++] = 0;
  if (j.w[1] != nullptr) j.bigjob[nbig++] = 1;
  if (j.vn[0] != 0) j.bigjob[nbig++] = 3;
  if (j.vn[1] != 0) j.bigjob[nbig++] = 5;
  hipLaunchKernelGGL(k_w4_scales, dim3(nbig * W4SC_BLOCKS + 4), dim3(256), 0, s, j, nbig);
}

// ----------------------------------------------------------------------------
// k_w4_gemm64h: k_w4_gemm64b's products on fp16 PAIRS (wino4.h): both operands arrive split -- V pairs written by the GroupNorm
// pass in front, U pairs by k_w4_pack -- so a K = 16 step of a 32 x 32 block is three v_mfma_f32_32x32x16_f16 (l h, h l, h h) on
// registers the loads delivered: no conversion, no subtraction, no vector instruction at all between the MFMAs (k_w4_gemm64b: six
// MFMAs and ~44 VALU instructions per row block and step, the matrix pipe busy 45 % of a wave's life).  Same decomposition, XCD
// placement and M layout as k_w4_gemm64b: a wave owns a 64 x 64 tile of one component over the whole reduction; eight workgroups
// share a tile (workgroup j: components 4 j .. 4 j + 3 and half a tile of component 32 + j / 2, K range cut over its waves).
// The result leaves unscaled: M = acc * 2^-(v_exp + u_exp).
// ----------------------------------------------------------------------------
typedef _Float16 w4_f16x8 __attribute__((ext_vector_type(8)));
struct W4HStage { w4_u32x4 a[2][2], b[2][2]; };    // [row block][part h, l], [column block][part]
struct W4HCursor { const w4_u32x4* a[2]; const w4_u32x4* b[2]; };
template <int NRB>
__device__ __forceinline__ void w4h_next(W4HStage& s, W4HCursor& cu) {   // the next K = 16 step of the streams (2 KB per stream and step)
#pragma unroll
  for (int r = 0; r < NRB; ++r) {
    s.a[r][0] = cu.a[r][0];
    s.a[r][1] = cu.a[r][64];
    cu.a[r] += 128;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    s.b[c][0] = cu.b[c][0];
    s.b[c][1] = cu.b[c][64];
    cu.b[c] += 128;
  }
}
template <int NRB>
__device__ __forceinline__ void w4h_mac(float16_t (&acc)[2][2], const W4HStage& s) {
  w4_f16x8 A[2][2], B[2][2];
#pragma unroll
  for (int r = 0; r < NRB; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) A[r][q] = __builtin_bit_cast(w4_f16x8, s.a[r][q]);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 2; ++q) B[c][q] = __builtin_bit_cast(w4_f16x8, s.b[c][q]);
#define W4H_P(AP, BQ)                                                                             \
  _Pragma("unroll") for (int r = 0; r < NRB; ++r) _Pragma("unroll") for (int c = 0; c < 2; ++c)   \
      acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[r][AP], B[c][BQ], acc[r][c], 0, 0, 0);
  W4H_P(1, 0) W4H_P(0, 1) W4H_P(0, 0)   // smallest products first
#undef W4H_P
}
// acc += sum over K = 16 steps [g0, g0 + n) (n a multiple of D): a ring of D stages, each refilled right behind the MFMAs that
// consumed it (the refills of the last D steps read up to D steps past the range: buffer slack)
template <int D, int NRB, class F = W4Nothing>
__device__ __forceinline__ void w4h_run(float16_t (&acc)[2][2], const W4HCursor& start, int n, F after_fill = F(), unsigned long long* st = nullptr) {
  W4HStage ring[D];
  W4HCursor cu = start;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    w4h_next<NRB>(ring[i], cu);
    __builtin_amdgcn_sched_barrier(0);
  }
  after_fill();
  w4_stamp(st, 1);
  for (int g = 0; g < n; g += D) {
    if (g == D) w4_stamp(st, 2);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      w4h_mac<NRB>(acc, ring[i]);
      __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the MFMAs that read the old contents
      w4h_next<NRB>(ring[i], cu);
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_w4_gemm64h(const unsigned* __restrict__ Vh, const unsigned* __restrict__ Uh, float* __restrict__ M,
                                                    const Ctrl* ctrl, W4Geom gm, int mode, const int* v_exp, const int* u_exp, unsigned long long* stamps) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* st = stamps != nullptr ? stamps + ((size_t)blockIdx.x * 4 + wave) * 16 : nullptr;   // (diagnostics build only)
  w4_stamp(st, 0);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 6, nRB = gm.RB, G2 = gm.G8 >> 1, CB = gm.C >> 5;
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int rt = tile / nCT, ct = tile - rt * nCT;
  const float inv = ldexpf(1.f, -(*v_exp + *u_exp));
  const int a_off = ((l31 >> 2) * 8) + hi * 4 + (l31 & 3);   // lane (row = 4 s + t, k-half hi): its 16 B inside a 1 KB part
  auto vblk = [&](int comp, int rb) { return reinterpret_cast<const w4_u32x4*>(Vh) + (((size_t)comp * nRB + rb) * G2) * 128 + a_off; };
  auto ublk = [&](int comp, int cb) { return reinterpret_cast<const w4_u32x4*>(Uh) + (((size_t)comp * CB + cb) * G2) * 128 + lane; };

  // the shared component's operands are requested behind the own component's first ring (one wave per SIMD: nothing else would
  // cover their latency at the end), and multiplied while the own component's stores drain
  constexpr int SH = 4;
  const int sng = G2 >> 2;                 // its K steps per wave
  const bool early = sng == SH;
  W4HStage shr[SH];
  const int scomp = 32 + (j >> 1), srb = 2 * rt + (j & 1);
  W4HCursor scu;
  scu.a[0] = vblk(scomp, srb) + (size_t)(wave * sng) * 128; scu.a[1] = scu.a[0];
  scu.b[0] = ublk(scomp, 2 * ct) + (size_t)(wave * sng) * 128; scu.b[1] = ublk(scomp, 2 * ct + 1) + (size_t)(wave * sng) * 128;
  {
    // mode bit 1 (NODE_TUNE_W4_SHAREV = 1, four column tiles): the four waves of a workgroup take the SAME component and row tile
    // and one column tile each (they walk the same V blocks in lock-step); bit 2: a 128 x 128 tile of one component (k_w4_gemm64b)
    const bool sharev = (mode & 2) != 0 && nCT == 4;
    const bool share2 = (mode & 4) != 0 && nCT == 4 && (nRB & 3) == 0;
    const int comp = 4 * j + (share2 ? (tile & 3) : sharev ? ct : wave);
    const int oct = share2 ? 2 * ((tile >> 2) & 1) + (wave & 1) : sharev ? wave : ct;
    const int ort = share2 ? 2 * (tile >> 3) + (wave >> 1) : rt;
    W4HCursor cu;
    cu.a[0] = vblk(comp, 2 * ort); cu.a[1] = vblk(comp, 2 * ort + 1);
    cu.b[0] = ublk(comp, 2 * oct); cu.b[1] = ublk(comp, 2 * oct + 1);
    float16_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
    auto request_shared = [&]() {
      if (early) {
        W4HCursor c2 = scu;
#pragma unroll
        for (int i = 0; i < SH; ++i) {
          w4h_next<1>(shr[i], c2);
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");   // (the compiler may not sink these requests to their first use behind the loop)
      }
    };
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample of M
    float* m0 = M + ((size_t)(ort * 16 + hi) * (gm.C >> 5) + 2 * oct) * (36 * 128) + (size_t)comp * 128 + l31;
    if ((mode & 16) != 0 && G2 % D == 0 && G2 >= 2 * D) {
      // mode bit 4 (NODE_TUNE_W4_HSPLIT, default): the tile as TWO 32-row halves one after the other on ONE operand ring -- the first
      // half's 16 KB of results drain while the second half's operands stream in (as one 64 x 64 tile every wave of the chip loads, then
      // every wave stores: 3 us of a 12 us launch in which nothing is read); the column operand is fetched twice (from L2).
      W4HStage ring[D];
      W4HCursor cc;
      cc.a[0] = cu.a[0]; cc.a[1] = cu.a[0]; cc.b[0] = cu.b[0]; cc.b[1] = cu.b[1];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        w4h_next<1>(ring[i], cc);
        __builtin_amdgcn_sched_barrier(0);
      }
      request_shared();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        for (int g = 0; g < G2; g += D) {
          if (h == 0 && g + D == G2) { cc.a[0] = cu.a[1]; cc.b[0] = cu.b[0]; cc.b[1] = cu.b[1]; }   // this round's refills open the second half
#pragma unroll
          for (int i = 0; i < D; ++i) {
            w4h_mac<1>(acc, ring[i]);
            __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the MFMAs that read the old contents
            w4h_next<1>(ring[i], cc);
          }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32 + (h ? 8 * sstride : 0);
          st_wt(o, acc[0][0][q] * inv);
          st_wt(o + 36 * 128, acc[0][1][q] * inv);
          acc[0][0][q] = 0.f; acc[0][1][q] = 0.f;
        }
      }
    } else {
      w4h_run<D, 2>(acc, cu, G2, request_shared, st);
      w4_stamp(st, 3);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
        st_wt(o, acc[0][0][q] * inv);
        st_wt(o + 36 * 128, acc[0][1][q] * inv);
        st_wt(o + 8 * sstride, acc[1][0][q] * inv);
        st_wt(o + 8 * sstride + 36 * 128, acc[1][1][q] * inv);
      }
      w4_stamp(st, 4);
    }
  }
  // --- half a tile of a shared component: rows [32 half, 32 half + 32), K range [wave G2/4, (wave+1) G2/4) per wave
  {
    float16_t acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0][c][q] = 0.f;
    if (early) {
#pragma unroll
      for (int i = 0; i < SH; ++i) w4h_mac<1>(acc, shr[i]);
    } else if (sng % 4 == 0) w4h_run<4, 1>(acc, scu, sng);
    else if (sng % 2 == 0) w4h_run<2, 1>(acc, scu, sng);
    else w4h_run<1, 1>(acc, scu, sng);
    w4_stamp(st, 5);
    float* red = smem + wave * 2048;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(red + c * 1024 + (r4 * 64 + lane) * 4) =
            make_float4(acc[0][c][4 * r4], acc[0][c][4 * r4 + 1], acc[0][c][4 * r4 + 2], acc[0][c][4 * r4 + 3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int blk = u >> 8, r4 = (u >> 6) & 3;
      float4 sm = *reinterpret_cast<const float4*>(smem + blk * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk * 1024 + (r4 * 64 + lane) * 4);
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
      float* mrow = M + ((size_t)(srb * 8 + 2 * r4 + hi) * (gm.C >> 5) + 2 * ct + blk) * (36 * 128) + (size_t)scomp * 128 + l31;
      st_wt(mrow, sm.x * inv);
      st_wt(mrow + 32, sm.y * inv);
      st_wt(mrow + 64, sm.z * inv);
      st_wt(mrow + 96, sm.w * inv);
    }
  }
#ifdef NODE_DIAG
  if (st != nullptr) {   // (diagnostics: when this wave's stores have drained)
    w4_stamp(st, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    w4_stamp(st, 7);
  }
#endif
}

// ----------------------------------------------------------------------------
// k_w4_gemm128b: k_w4_gemm64b's products for LONG reductions (C >= 512: cfg 5's 16x16 states at 1024 filters), as a
// classic LDS-tiled GEMM.  k_w4_gemm64b gives every wave a component of its own, so the four waves of a workgroup share
// nothing and an XCD works on 4.5 components at once: at C = 1024 that is 47 MB of operands against a 4 MB L2, every
// 64-row tile streams all of the filter triples in from the Infinity Cache again (3.6 GB per launch, measured 472 us =
// 0.39 of the bf16 pipe).  Here a workgroup owns a 128 x 128 tile of ONE component, its waves 64 x 64 quarters; per
// K = 16 step each wave fetches one quarter of the tile's operands (2 + 3 KB instead of 4 + 6), splits its row block
// into bf16 triples ONCE for the workgroup, and the MFMA-ready 1 KB blocks go through a two-stage LDS ring (one barrier
// per step; two workgroups per CU cover each other's barriers).  An XCD walks through its components one at a time --
// 64 concurrent workgroups = every tile of a component at cfg 5 -- so what is live in its L2 is one K slice of V and U.
// Same V / Ub / M layouts.  Needs 4 N % 128 == 0 and C % 128 == 0.
// (The first version of this kernel, round 3, was measured at cfg 2 -- C = 256, 16 steps per tile -- and lost to
// k_w4_gemm64b there, 21.5 - 22.8 against 19.6 us: fill and drain dominate so short a loop.)
// ----------------------------------------------------------------------------
__device__ __forceinline__ void w4c_mac(float16_t (&acc)[2][2], const w4_u32x4 (&a)[2][3], const w4_u32x4 (&b)[2][3]) {
  w4_bf16x8 A[2][3], B[2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) { A[r][q] = __builtin_bit_cast(w4_bf16x8, a[r][q]); B[r][q] = __builtin_bit_cast(w4_bf16x8, b[r][q]); }
#define W4C_P(AP, BQ)                                                                           \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int c = 0; c < 2; ++c)   \
      acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[r][AP], B[c][BQ], acc[r][c], 0, 0, 0);
  W4C_P(2, 0) W4C_P(0, 2) W4C_P(1, 1) W4C_P(1, 0) W4C_P(0, 1) W4C_P(0, 0)   // parts 0 = h, 1 = m, 2 = l: smallest products first
#undef W4C_P
}

// A wave's share of one K = 16 step: its row block (two g blocks of V), its column block (three parts of Ub).  The five
// requests are inline asm with HAND-PLACED waits: left to the compiler, the wait state of the loop entry merged into the
// steady state made every other step wait for all but one of the ten requests in flight -- the step's own prefetch
// (s_waitcnt vmcnt(1) where vmcnt(5) is exact).  W4C_WAIT ties the wait to the registers, so no use can move above it, and
// the registers stay allocated to the request while it is in flight.
typedef float w4_f32x4 __attribute__((ext_vector_type(4)));
struct W4CLoad { w4_f32x4 a0, a1; w4_u32x4 b0, b1, b2; };
#define W4C_FETCH(L, PA, PB)                                                                        \
  {                                                                                                 \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).a0) : "v"(PA) : "memory");            \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).a1) : "v"(PA) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).b0) : "v"(PB) : "memory");            \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).b1) : "v"(PB) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"((L).b2) : "v"(PB) : "memory"); \
  }
#define W4C_WAIT(N, L) \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"((L).a0), "+v"((L).a1), "+v"((L).b0), "+v"((L).b1), "+v"((L).b2) : : "memory")

__global__ __launch_bounds__(256, 2) void k_w4_gemm128b(const float* __restrict__ V, const unsigned short* __restrict__ Ub,
                                                        float* __restrict__ M, const Ctrl* ctrl, W4Geom gm) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) w4_u32x4 tile_lds[];   // [2 stages][A 4 row blocks x 3 parts | B 4 column blocks x 3 parts][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 7, nRT = gm.R >> 7, nT = nRT * nCT, G8 = gm.G8, G2 = G8 >> 1, CB = gm.C >> 5, nRB = gm.RB;
  // workgroup -> (component, tile): XCD j (= blockIdx % 8) takes components 4 j .. 4 j + 3 one after the other, then half of the
  // tiles of component 32 + j / 2
  const int j = blockIdx.x & 7, i = blockIdx.x >> 3;
  int comp, tile;
  if (i < 4 * nT) { comp = 4 * j + i / nT; tile = i % nT; }
  else { comp = 32 + (j >> 1); tile = (j & 1) * (nT >> 1) + (i - 4 * nT); }
  const int RT = tile / nCT, CT = tile - RT * nCT;
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;   // lane (row = 4 s + t, k-half hi) inside a V block
  // this lane's requests of step g2: V at pa + g2 * 2 KB (+ 1 KB: the second g block), Ub at pb + g2 * 3 KB (+ 1, 2 KB: the parts)
  const char* pa = reinterpret_cast<const char*>(V + (((size_t)comp * nRB + 4 * RT + wave) * G8) * 256 + a_off);
  const char* pb = reinterpret_cast<const char*>(reinterpret_cast<const w4_u32x4*>(Ub) + (((size_t)comp * CB + 4 * CT + wave) * G2) * 192 + lane);
  // LDS block (stage, kind 0 = A / 1 = B, block 0..3, part): 64 lanes x 16 B, every access lane * 16 B -- conflict-free
  auto blk = [&](int stage, int kind, int b, int part) { return tile_lds + ((((stage * 2 + kind) * 4 + b) * 3 + part) * 64 + lane); };
#define W4C_STASH(L, STAGE)                                                                                  \
  {                                                                                                          \
    const W4Split sp_ = w4_split8(make_float4((L).a0.x, (L).a0.y, (L).a0.z, (L).a0.w),                       \
                                  make_float4((L).a1.x, (L).a1.y, (L).a1.z, (L).a1.w));                      \
    *blk(STAGE, 0, wave, 0) = __builtin_bit_cast(w4_u32x4, sp_.h);                                           \
    *blk(STAGE, 0, wave, 1) = __builtin_bit_cast(w4_u32x4, sp_.m);                                           \
    *blk(STAGE, 0, wave, 2) = __builtin_bit_cast(w4_u32x4, sp_.l);                                           \
    *blk(STAGE, 1, wave, 0) = (L).b0;                                                                        \
    *blk(STAGE, 1, wave, 1) = (L).b1;                                                                        \
    *blk(STAGE, 1, wave, 2) = (L).b2;                                                                        \
  }
  const int wr = wave >> 1, wc = wave & 1;
  float16_t acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;

  // Two register sets, never copied: at the top of an even step k, ldx holds step k + 1 and ldy step k + 2 (five requests
  // each, ldx's the older) -- two steps of cover.  (Reads past the reduction's end land in the buffers' slack.)
  W4CLoad ldx, ldy;
  W4C_FETCH(ldx, pa, pb)
  W4C_WAIT(0, ldx);
  W4C_STASH(ldx, 0)
  W4C_FETCH(ldx, pa + 2048, pb + 3072)
  W4C_FETCH(ldy, pa + 4096, pb + 6144)
  pa += 3 * 2048; pb += 3 * 3072;          // -> step 3
  __syncthreads();
#define W4C_STEP(ST, LD)                                                                                          \
  {                                                                                                               \
    w4_u32x4 fa[2][3], fb[2][3];                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int q = 0; q < 3; ++q) {                 \
      fa[r][q] = *blk(ST, 0, 2 * wr + r, q);                                                                      \
      fb[r][q] = *blk(ST, 1, 2 * wc + r, q);                                                                      \
    }                                                                                                             \
    w4c_mac(acc, fa, fb);                                                                                         \
    W4C_WAIT(5, LD); /* the older five of the ten in flight */                                                    \
    W4C_STASH(LD, (ST) ^ 1) /* the next step -> the other stage (everybody left it at the last barrier) */        \
    W4C_FETCH(LD, pa, pb)                                                                                         \
    pa += 2048; pb += 3072;                                                                                       \
    __syncthreads();                                                                                              \
  }
  for (int k = 0; k < G2; k += 2) {   // (G2 = C / 16 is even: C % 128 == 0)
    W4C_STEP(0, ldx)
    W4C_STEP(1, ldy)
  }
  W4C_WAIT(0, ldx);                   // nothing may still be landing in registers the epilogue reuses
  W4C_WAIT(0, ldy);
#undef W4C_STEP
#undef W4C_STASH
  // M is [n][C/32][36][4 t][32 c] (wino4.h): this wave's 64 x 64 quarter = 64-row tile 2 RT + wr, 64-column tile 2 CT + wc
  const int rt = 2 * RT + wr, ct = 2 * CT + wc;
  const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;   // floats per sample of M
  float* m0 = M + ((size_t)(rt * 16 + hi) * (gm.C >> 5) + 2 * ct) * (36 * 128) + (size_t)comp * 128 + l31;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
    st_wt(o, acc[0][0][q]);
    st_wt(o + 36 * 128, acc[0][1][q]);
    st_wt(o + 8 * sstride, acc[1][0][q]);
    st_wt(o + 8 * sstride + 36 * 128, acc[1][1][q]);
  }
}


// ----------------------------------------------------------------------------
// k_w4_gemm_small: the component GEMMs of a batch of at most 16 samples (the bs = 1 census, evaluate.py:97-142).  The
// throughput kernels give every wave a whole K range however few rows there are (23.9 us per launch at ONE sample);
// here a workgroup owns ONE 32 x 32 block of one component, its four waves split K (all operand requests of a wave in
// flight at once, 32 MFMAs at C = 256) and meet through LDS: 36 x C/32 x N8/8 workgroups, one memory round trip deep.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_w4_gemm_small(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                       const Ctrl* ctrl, W4Geom gm) {
  if (ctrl != nullptr && ctrl->done) return;
  __shared__ __attribute__((aligned(16))) float red[4 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int CB = gm.C >> 5, G8 = gm.G8;
  const int cb = blockIdx.x % CB, comp = (blockIdx.x / CB) % W4_COMPS, rb = blockIdx.x / (CB * W4_COMPS);
  const int a_off = (((l31 >> 2) * 8) + hi * 4 + (l31 & 3)) * 4;
  const int ng = G8 >> 2, g0 = wave * ng;          // this wave's K quarter (C % 64 == 0: ng is a multiple of 2)
  const float4* pa = reinterpret_cast<const float4*>(V + (((size_t)comp * gm.RB + rb) * G8 + g0) * 256 + a_off);
  const float4* pb = reinterpret_cast<const float4*>(U + (((size_t)comp * CB + cb) * G8 + g0) * 256 + lane * 4);
  float16_t acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int g = 0; g < ng; g += 8) {
    float4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // (clamped, not predicated: a masked request makes the compiler wait)
      const int gi = g + i < ng ? g + i : ng - 1;
      a[i] = pa[(size_t)gi * 64];
      b[i] = pb[(size_t)gi * 64];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (g + i < ng) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[i].w, acc, 0, 0, 0);
      }
  }
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4)
    *reinterpret_cast<float4*>(red + wave * 1024 + (r4 * 64 + lane) * 4) = make_float4(acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]);
  __syncthreads();
  {   // thread (r4 = wave, lane): registers 4 r4 .. 4 r4 + 3 of the block = tiles 0..3 of sample 2 r4 + hi
    const int r4 = wave;
    float4 s = *reinterpret_cast<const float4*>(red + (r4 * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(red + w * 1024 + (r4 * 64 + lane) * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* mrow = M + ((size_t)(rb * 8 + 2 * r4 + hi) * CB + cb) * (36 * 128) + (size_t)comp * 128 + l31;
    mrow[0] = s.x;
    mrow[32] = s.y;
    mrow[64] = s.z;
    mrow[96] = s.w;
  }
}


// The A/B switches that select the component-GEMM kernel.  ONE reader for the packer (which filter forms a solve
// prepares) and the launcher (which kernel reads them), all of them read on every call: a process that changes a
// switch between solves (the tests do) can never pack for one kernel and launch another.
// The PRODUCT library (no -DNODE_DIAG) reads only switches under which every result stays correct: which kernel family
// multiplies (fp32 MFMA / bf16 triples / small batches) and how a component's tiles are dealt to waves (bit-identical).
// The timing ablations (results wrong by design), the stamps, the padded operand spacing and the measured-and-rejected
// kernels exist in libnode_hip_diag.so only (build.py --diag; loaded by tools/ with NODE_HIP_DIAG=1).
struct W4Switches { int g64, b16, ablate, small, uf32, sharev, lds, early, ksplit, gemm128, wgrad128, half, f16, hdepth, hsplit, h128, h256; };
static W4Switches w4_read_switches() {
  auto rd = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
  // NODE_TUNE_W4_GEMM128 / _WGRAD128 = 0 never / 1 wherever it fits / unset (-1): long reductions (C >= 512)
#ifdef NODE_DIAG
  return {rd("NODE_TUNE_W4_GEMM64", 1), rd("NODE_TUNE_W4_BF16X3", 1), rd("NODE_TUNE_W4_ABLATE", 0), rd("NODE_TUNE_W4_SMALL", 1),
          rd("NODE_TUNE_W4_UF32", 0), rd("NODE_TUNE_W4_SHAREV", 1), rd("NODE_TUNE_W4_LDS", 0), rd("NODE_TUNE_W4_EARLY", 0), rd("NODE_TUNE_W4_KSPLIT", 0),
          rd("NODE_TUNE_W4_GEMM128", -1), rd("NODE_TUNE_W4_WGRAD128", -1), rd("NODE_TUNE_W4_HALF", 0), rd("NODE_TUNE_W4_F16", 1), rd("NODE_TUNE_W4_HDEPTH", 4), rd("NODE_TUNE_W4_HSPLIT", 0), rd("NODE_TUNE_W4_H128", -1), rd("NODE_TUNE_W4_H256", 1)};
#else
  return {rd("NODE_TUNE_W4_GEMM64", 1), rd("NODE_TUNE_W4_BF16X3", 1), 0, rd("NODE_TUNE_W4_SMALL", 1), 0, rd("NODE_TUNE_W4_SHAREV", 1), 0, 0, 0,
          rd("NODE_TUNE_W4_GEMM128", -1), rd("NODE_TUNE_W4_WGRAD128", -1), 0, rd("NODE_TUNE_W4_F16", 1), rd("NODE_TUNE_W4_HDEPTH", 4), rd("NODE_TUNE_W4_HSPLIT", 0), rd("NODE_TUNE_W4_H128", -1), rd("NODE_TUNE_W4_H256", 1)};
#endif
}
// The switches are read from the environment ONCE PER C-ABI CALL (w4_refresh_tuning at the top of every entry point that
// launches these kernels), not once per launch: a training step launches ~100 component GEMMs, and eleven getenv scans in
// front of each were a quarter of a millisecond of host time per step -- on the drop-in path, where the host is what
// bounds the step, that is throughput.  Tests that flip a switch between two calls still see it.
static thread_local W4Switches g_w4_sw;
static thread_local bool g_w4_sw_valid = false;
void w4_refresh_tuning() { g_w4_sw = w4_read_switches(); g_w4_sw_valid = true; }
static const W4Switches& w4_switches() {
  if (!g_w4_sw_valid) w4_refresh_tuning();
  return g_w4_sw;
}
static bool w4_takes_small(const W4Switches& sw, int N) { return sw.small != 0 && N <= 16 && sw.ablate == 0; }   // (ablations time the throughput kernels)
// fp32 filters, bf16-triple products (k_w4_gemm64c): 8x8 / 16x16 batches of C < 512 (the LDS-tiled kernel of long reductions
// shares its split filter blocks through LDS and keeps the packed triples)
static bool w4_takes_uf32(const W4Switches& sw, int N, int C) {
  return sw.uf32 != 0 && !w4_takes_small(sw, N) && N % 16 == 0 && sw.g64 != 0 && sw.b16 != 0 && sw.ablate == 0 && C < 512;
}
bool w4_uses_bf16(int N, int C) {
  const W4Switches sw = w4_switches();
  if (w4_takes_small(sw, N)) return false;   // (k_w4_gemm_small reads the fp32 filters)
  if (w4_takes_uf32(sw, N, C)) return false; // (k_w4_gemm64c splits them in registers)
  return N % 16 == 0 && sw.g64 != 0 && sw.b16 != 0 && (sw.ablate == 0 || sw.ablate >= 16);
}

// one 1 KB LDS-DMA piece (lane-linear destination) as inline asm: the compiler neither counts it nor drains it in front of LDS reads
__device__ __forceinline__ void w4wh_dma(const unsigned char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ----------------------------------------------------------------------------
// k_w4_gemm128h: k_w4_gemm64h's products as an LDS-tiled GEMM (the skeleton of k_w4_gemm128b): a workgroup owns a 128 x 128 tile of
// ONE component, its waves 64 x 64 quarters; per K = 16 step each wave fetches ONE quarter of the tile's operands -- its 32-row block
// of V pairs and its 32-column block of U pairs, parts h and l: four 16-B requests per lane, already the MFMA fragments -- and
// the 16 KB of the step go through a four-stage LDS ring (one barrier per step), so a CU takes in 16 KB per step where
// k_w4_gemm64h's four independent 64 x 64 tiles take 32 KB.  Why: at long reductions k_w4_gemm64h, like k_w4_gemm64b, re-reads its
// operands from the Infinity Cache (371 against 262 us at cfg 5).  (At SHORT reductions k_w4_gemm64h stays: its launch runs at the
// memory system's pace for its bytes -- DESIGN.md 4.3; the texture path looked like its limit and is not, profiles/r06_gemm64v_ab.txt.)
// Requests as inline asm with hand-placed waits, NSET register sets in flight that are never copied (k_w4_gemm128b).  (Measured and
// removed, round 6: the same tile with the operands brought by LDS-DMA -- no staging instructions -- 14.1 - 17.9 us against
// k_w4_gemm64h's 12.8 at cfg 2: a CU's four DMA streams deliver less than its register loads do.)
// Work: components 0..31 give 32 nT tiles (nT = rows / 128 x C / 128); XCD j takes 4 j .. 4 j + 3 one after the other, and every
// workgroup adds one EIGHTH of a tile of component 32 + j / 2 (a 32 x 64 block, K range cut over its four waves and summed through
// LDS, operands straight into registers: k_w4_gemm64h's shared component), so every workgroup does the same work.
// Needs rows % 128 == 0 (N % 32 == 0), C % 128 == 0, nT even.
// ----------------------------------------------------------------------------
struct W4GLoad { w4_u32x4 a0, a1, b0, b1; };
#define W4G_FETCH(L, PA, PB)                                                                        \
  {                                                                                                 \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).a0) : "v"(PA) : "memory");            \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).a1) : "v"(PA) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).b0) : "v"(PB) : "memory");            \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).b1) : "v"(PB) : "memory"); \
  }
#define W4G_WAIT(N, L) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"((L).a0), "+v"((L).a1), "+v"((L).b0), "+v"((L).b1) : : "memory")
template <int WPC>
__global__ __launch_bounds__(256, WPC) void k_w4_gemm128h(const unsigned* __restrict__ Vh, const unsigned* __restrict__ Uh, float* __restrict__ M,
                                                           const Ctrl* ctrl, W4Geom gm, const int* v_exp, const int* u_exp, int tail) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) w4_u32x4 gtile[];   // [4 stages][A 4 row blocks x 2 parts | B 4 column blocks x 2 parts][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 7, nRT = gm.R >> 7, nT = nRT * nCT, G2 = gm.G8 >> 1, CB = gm.C >> 5, nRB = gm.RB;
  const int j = blockIdx.x & 7, i = blockIdx.x >> 3;            // XCD, slot: 4 nT slots per XCD
  // tail != 0 (behind k_w4_gemm256h, which multiplies components 0..31): the grid is the 4 nT tiles of components 32..35 themselves, a pair
  // of XCDs per component, and there is no shared piece
  const int comp = tail ? 32 + (j >> 1) : 4 * j + i / nT, tile = tail ? (j & 1) * (nT >> 1) + i : i % nT;
  const int RT = tile / nCT, CT = tile - RT * nCT;
  const float inv = ldexpf(1.f, -(*v_exp + *u_exp));
  const int a_off = ((l31 >> 2) * 8) + hi * 4 + (l31 & 3);      // lane (row = 4 s + t, k-half hi): its 16 B inside a 1 KB part of V
  // --- the shared component's piece: component 32 + j / 2; the pair of XCDs holds 8 nT workgroups = nT tiles x 8 pieces (4 row blocks x 2 column halves)
  const int sidx = (j & 1) * 4 * nT + i;
  const int stile = sidx >> 3, spiece = sidx & 7;
  const int scomp = 32 + (j >> 1);
  const int srb = 4 * (stile / nCT) + (spiece >> 1);             // its 32-row block
  const int scb = 4 * (stile % nCT) + 2 * (spiece & 1);          // the first of its two 32-column blocks
  const int sng = G2 >> 2;                                       // K steps per wave
  W4HCursor scu;
  scu.a[0] = reinterpret_cast<const w4_u32x4*>(Vh) + (((size_t)scomp * nRB + srb) * G2 + (size_t)wave * sng) * 128 + a_off; scu.a[1] = scu.a[0];
  scu.b[0] = reinterpret_cast<const w4_u32x4*>(Uh) + (((size_t)scomp * CB + scb) * G2 + (size_t)wave * sng) * 128 + lane;
  scu.b[1] = reinterpret_cast<const w4_u32x4*>(Uh) + (((size_t)scomp * CB + scb + 1) * G2 + (size_t)wave * sng) * 128 + lane;

  // this lane's requests of step g2: V at pa + g2 * 2 KB (+ 1 KB: part l), U at pb + g2 * 2 KB (+ 1 KB)
  const char* pa = reinterpret_cast<const char*>(reinterpret_cast<const w4_u32x4*>(Vh) + (((size_t)comp * nRB + 4 * RT + wave) * G2) * 128 + a_off);
  const char* pb = reinterpret_cast<const char*>(reinterpret_cast<const w4_u32x4*>(Uh) + (((size_t)comp * CB + 4 * CT + wave) * G2) * 128 + lane);
  auto blk = [&](int stage, int kind, int b, int part) { return gtile + (((stage * 2 + kind) * 4 + b) * 2 + part) * 64 + lane; };
#define W4G_STASH(L, STAGE)                   \
  {                                           \
    *blk(STAGE, 0, wave, 0) = (L).a0;         \
    *blk(STAGE, 0, wave, 1) = (L).a1;         \
    *blk(STAGE, 1, wave, 0) = (L).b0;         \
    *blk(STAGE, 1, wave, 1) = (L).b1;         \
  }
  const int wr = wave >> 1, wc = wave & 1;
  float16_t acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;

  // FOUR register sets, never copied (step s travels in set s % 4), FOUR LDS stages (step s sits in stage s % 4), TWO fragment sets:
  // step k multiplies fragments read from LDS a step earlier while step k + 1's travel LDS -> registers, stashes step k + 2 (requested
  // four steps ago) and requests step k + 6 -- one barrier per step, nothing a wave waits for was issued less than a step ago.
  // (Reads past the reduction's end land in the buffers' slack: W4_SLACK.)
  W4GLoad l0, l1, l2, l3;
  W4HStage fA, fB;
  W4G_FETCH(l0, pa, pb)
  W4G_FETCH(l1, pa + 2048, pb + 2048)
  W4G_FETCH(l2, pa + 4096, pb + 4096)
  W4G_FETCH(l3, pa + 6144, pb + 6144)
  W4G_WAIT(12, l0);
  W4G_STASH(l0, 0)
  W4G_FETCH(l0, pa + 8192, pb + 8192)
  W4G_WAIT(12, l1);
  W4G_STASH(l1, 1)
  W4G_FETCH(l1, pa + 10240, pb + 10240)
  pa += 6 * 2048; pb += 6 * 2048;            // -> step 6
  __syncthreads();
#define W4G_FRAGS(F, ST)                                                                          \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int q = 0; q < 2; ++q) {   \
    (F).a[r][q] = *blk(ST, 0, 2 * wr + r, q);                                                     \
    (F).b[r][q] = *blk(ST, 1, 2 * wc + r, q);                                                     \
  }
  W4G_FRAGS(fA, 0)
  // step k: FCUR = fragments of step k, FNEXT <- stage (k + 1) % 4, LSET = set (k + 2) % 4 -> stage (k + 2) % 4, then refilled with step k + 6
#define W4G_STEP(FCUR, FNEXT, STN, LSET, STS)                                                     \
  {                                                                                               \
    W4G_FRAGS(FNEXT, STN)                                                                         \
    w4h_mac<2>(acc, FCUR);                                                                        \
    W4G_WAIT(12, LSET);                                                                           \
    W4G_STASH(LSET, STS)                                                                          \
    W4G_FETCH(LSET, pa, pb)                                                                       \
    pa += 2048; pb += 2048;                                                                       \
    __syncthreads();                                                                              \
  }
  for (int k = 0; k < G2; k += 4) {   // (G2 = C / 16 is a multiple of 4: C % 128 == 0 gives 8)
    W4G_STEP(fA, fB, 1, l2, 2)
    W4G_STEP(fB, fA, 2, l3, 3)
    W4G_STEP(fA, fB, 3, l0, 0)
    W4G_STEP(fB, fA, 0, l1, 1)
  }
  W4G_WAIT(0, l0);                    // nothing may still be landing in registers the epilogue reuses
  W4G_WAIT(0, l1);
  W4G_WAIT(0, l2);
  W4G_WAIT(0, l3);
#undef W4G_FRAGS
#undef W4G_STEP
#undef W4G_STASH
  {
    const int rt = 2 * RT + wr, ct = 2 * CT + wc;                // this wave's 64 x 64 quarter
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;       // floats per sample of M
    float* m0 = M + ((size_t)(rt * 16 + hi) * (gm.C >> 5) + 2 * ct) * (36 * 128) + (size_t)comp * 128 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* o = m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32;
      st_wt(o, acc[0][0][q] * inv);
      st_wt(o + 36 * 128, acc[0][1][q] * inv);
      st_wt(o + 8 * sstride, acc[1][0][q] * inv);
      st_wt(o + 8 * sstride + 36 * 128, acc[1][1][q] * inv);
    }
  }
  // --- the shared piece: rows srb (32), column blocks scb, scb + 1; K range [wave sng, (wave + 1) sng) per wave
  if (!tail) {
    float16_t sa[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) sa[0][c][q] = 0.f;
    if (sng % 4 == 0) w4h_run<4, 1>(sa, scu, sng);
    else if (sng % 2 == 0) w4h_run<2, 1>(sa, scu, sng);
    else w4h_run<1, 1>(sa, scu, sng);
    float* smem = reinterpret_cast<float*>(gtile);               // (everybody left the ring at the loop's last barrier)
    float* red = smem + wave * 2048;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(red + c * 1024 + (r4 * 64 + lane) * 4) =
            make_float4(sa[0][c][4 * r4], sa[0][c][4 * r4 + 1], sa[0][c][4 * r4 + 2], sa[0][c][4 * r4 + 3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int blk2 = u >> 8, r4 = (u >> 6) & 3;
      float4 sm = *reinterpret_cast<const float4*>(smem + blk2 * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + blk2 * 1024 + (r4 * 64 + lane) * 4);
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
      float* mrow = M + ((size_t)(srb * 8 + 2 * r4 + hi) * (gm.C >> 5) + scb + blk2) * (36 * 128) + (size_t)scomp * 128 + l31;
      st_wt(mrow, sm.x * inv);
      st_wt(mrow + 32, sm.y * inv);
      st_wt(mrow + 64, sm.z * inv);
      st_wt(mrow + 96, sm.w * inv);
    }
  }
}
#undef W4G_FETCH
#undef W4G_WAIT

// ----------------------------------------------------------------------------
// k_w4_gemm256h (long reductions, rows % 256 == 0, C % 256 == 0): k_w4_gemm128h with FOUR times the tile.  What holds k_w4_gemm128h at
// 43 % matrix duty (cfg 5) is not the LDS (half its fragment reads removed: -2 %; three quarters of its writes: -5 %; profiles/
// r06_gemm256h.txt) but operand DELIVERY: 16 KB per 192 matrix cycles and workgroup, 18 % of it cold in L2, requested 1.1 us ahead.
// Here a workgroup owns a 256 x 256 tile of one component, ONE wave per SIMD, each wave a 128 x 128 quarter: sixteen 32 x 32 accumulators
// (256 registers), 48 MFMAs per K = 16 step and wave against 16 fragment reads -- half the LDS and half the global bytes per MFMA, and
// a step lasts 768 matrix cycles, so the same four steps of look-ahead are 1.5 us.  Per step a wave requests TWO row blocks and TWO
// column blocks (both parts: eight 16-B requests per lane, inline asm, FOUR register sets in flight, counted waits), four 32 KB LDS
// stages; the step is cut in four quadrants so that fragment halves travel under MFMAs (details at the loop).
// Components 0..31 only (32 nT tiles: XCD j takes components 4 j .. 4 j + 3); components 32..35 follow as k_w4_gemm128h<..>(tail = 1).
// ----------------------------------------------------------------------------
struct W4KLoad { w4_u32x4 a00, a01, a10, a11, b00, b01, b10, b11; };      // [block 0 | 1][part h | l] of V, of U
#define W4K_FETCH(L)                                                                                     \
  {                                                                                                      \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).a00) : "v"(pa0) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).a01) : "v"(pa0) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).a10) : "v"(pa1) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).a11) : "v"(pa1) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).b00) : "v"(pb0) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).b01) : "v"(pb0) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).b10) : "v"(pb1) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).b11) : "v"(pb1) : "memory");   \
    pa0 += 2048; pa1 += 2048; pb0 += 2048; pb1 += 2048;                                                  \
  }
#define W4K_WAIT(N, L)                                                                                                         \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"((L).a00), "+v"((L).a01), "+v"((L).a10), "+v"((L).a11), "+v"((L).b00), "+v"((L).b01), \
               "+v"((L).b10), "+v"((L).b11) : : "memory")
struct W4KB { w4_u32x4 v[2][2]; };      // a fragment half: two row (or column) blocks, parts h | l
__global__ __launch_bounds__(256, 1) void k_w4_gemm256h(const unsigned* __restrict__ Vh, const unsigned* __restrict__ Uh, float* __restrict__ M,
                                                        const Ctrl* ctrl, W4Geom gm, const int* v_exp, const int* u_exp) {
  if (ctrl != nullptr && ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  extern __shared__ __attribute__((aligned(16))) w4_u32x4 ktile[];   // [4 stages][A | B][8 blocks][2 parts][64 lanes]: 128 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nCT = gm.C >> 8, nRT = gm.R >> 8, nT = nRT * nCT, G2 = gm.G8 >> 1, CB = gm.C >> 5, nRB = gm.RB;
  const int j = blockIdx.x & 7, i = blockIdx.x >> 3;            // XCD, slot: 4 nT slots per XCD
  const int comp = 4 * j + i / nT, tile = i % nT;
  const int RT = tile / nCT, CT = tile - RT * nCT;
  const float inv = ldexpf(1.f, -(*v_exp + *u_exp));
  const int a_off = ((l31 >> 2) * 8) + hi * 4 + (l31 & 3);      // lane (row = 4 s + t, k-half hi): its 16 B inside a 1 KB part of V
  const size_t blk_bytes = (size_t)G2 * 2048;                   // one 32-row / 32-column block over the whole reduction
  const char* pa0 = reinterpret_cast<const char*>(reinterpret_cast<const w4_u32x4*>(Vh) + (((size_t)comp * nRB + 8 * RT + 2 * wave) * G2) * 128 + a_off);
  const char* pa1 = pa0 + blk_bytes;
  const char* pb0 = reinterpret_cast<const char*>(reinterpret_cast<const w4_u32x4*>(Uh) + (((size_t)comp * CB + 8 * CT + 2 * wave) * G2) * 128 + lane);
  const char* pb1 = pb0 + blk_bytes;
  auto blk = [&](int stage, int kind, int b, int part) { return ktile + (((stage * 2 + kind) * 8 + b) * 2 + part) * 64 + lane; };
#define W4K_STASH(L, STAGE)                       \
  {                                               \
    *blk(STAGE, 0, 2 * wave, 0) = (L).a00;        \
    *blk(STAGE, 0, 2 * wave, 1) = (L).a01;        \
    *blk(STAGE, 0, 2 * wave + 1, 0) = (L).a10;    \
    *blk(STAGE, 0, 2 * wave + 1, 1) = (L).a11;    \
    *blk(STAGE, 1, 2 * wave, 0) = (L).b00;        \
    *blk(STAGE, 1, 2 * wave, 1) = (L).b01;        \
    *blk(STAGE, 1, 2 * wave + 1, 0) = (L).b10;    \
    *blk(STAGE, 1, 2 * wave + 1, 1) = (L).b11;    \
  }
  const int wr = wave >> 1, wc = wave & 1;
  float16_t acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
  // fragments travel as HALVES of the wave's quarter: two row blocks (W4K_AH: half H of its four) or two column blocks (W4K_BH), both parts
#define W4K_AH(F, ST, H) \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int q = 0; q < 2; ++q) (F).v[b][q] = *blk(ST, 0, 4 * wr + 2 * (H) + b, q);
#define W4K_BH(F, ST, H) \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int q = 0; q < 2; ++q) (F).v[b][q] = *blk(ST, 1, 4 * wc + 2 * (H) + b, q);
  // one quadrant: acc[2 RH + r][2 CH + c] += A x B; per 32 x 32 tile the products (l,h), (h,l), (h,h) in this order (k_w4_gemm64h), four
  // independent MFMAs between two on the same accumulator
#define W4K_MAC1(AF, BF, RH, CH, PA, PB)                                                            \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int c = 0; c < 2; ++c)       \
      acc[2 * (RH) + r][2 * (CH) + c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w4_f16x8, (AF).v[r][PA]), __builtin_bit_cast(w4_f16x8, (BF).v[c][PB]), acc[2 * (RH) + r][2 * (CH) + c], 0, 0, 0);
#define W4K_MAC(AF, BF, RH, CH) { W4K_MAC1(AF, BF, RH, CH, 1, 0) W4K_MAC1(AF, BF, RH, CH, 0, 1) W4K_MAC1(AF, BF, RH, CH, 0, 0) }
  // FOUR register sets, never copied (step s travels in set s % 4: requested four steps before it is waited for -- with two sets the loop
  // ran 3100 cycles per step against 1536 of matrix work: 18 % of the requests are cold in L2 and come back after ~3 us), FOUR LDS stages
  // (step s sits in stage s % 4).  A step is four quadrants in snake order -- (A0,B0) (A0,B1) (A1,B1) (A1,B0) -- so that each transition
  // brings ONE new fragment half under the MFMAs of the quadrant before (64 fragment registers live, not 112): step k reads B1, A1, B0
  // again, then -- under its last quadrant -- A0 and B0 of step k + 1 (stashed in step k - 1, behind that step's barrier); between the
  // quadrants set (k + 2) % 4 goes to stage (k + 2) % 4 and is refilled with step k + 6.  One barrier per step.
  // (Reads past the reduction's end land in the buffers' slack: W4_SLACK.)
  W4KLoad l0, l1, l2, l3;
  W4KB xa, ya, xb, yb;      // xa: A0;  ya: A1;  xb / yb: B0 and B1, their roles swapping every step
  W4K_FETCH(l0)
  W4K_FETCH(l1)
  W4K_FETCH(l2)
  W4K_FETCH(l3)
  W4K_WAIT(24, l0);
  W4K_STASH(l0, 0)
  W4K_FETCH(l0)
  W4K_WAIT(24, l1);
  W4K_STASH(l1, 1)
  W4K_FETCH(l1)
  __syncthreads();
  W4K_AH(xa, 0, 0)
  W4K_BH(xb, 0, 0)
  // step k in stage ST, next stage STN; B0 in B0R on entry, B1 goes to B1R; on exit B0 of step k + 1 sits in B1R
#define W4K_FENCE __builtin_amdgcn_sched_barrier(0);   /* the order below IS the schedule: left alone the compiler sinks every fragment read
                                                          to a few MFMAs before its use, and with one wave per SIMD nothing covers the LDS latency */
// Everything that is not an MFMA goes in pieces of FOUR instructions behind groups of four MFMAs (one part product of a quadrant): the matrix
// pipe works 128 cycles on a group while the vector / memory issue it blocks for 32 of them is free for the rest.  (As ONE block between two
// quadrants -- wait, eight staging writes, eight requests, their pointer sums, four fragment reads -- the same instructions cost ~330 cycles
// of a drained pipe per step: 14 of a round's 108 us, measured by leaving them out.)
#define W4K_STASH_A(L, STAGE) { *blk(STAGE, 0, 2 * wave, 0) = (L).a00; *blk(STAGE, 0, 2 * wave, 1) = (L).a01; *blk(STAGE, 0, 2 * wave + 1, 0) = (L).a10; *blk(STAGE, 0, 2 * wave + 1, 1) = (L).a11; }
#define W4K_STASH_B(L, STAGE) { *blk(STAGE, 1, 2 * wave, 0) = (L).b00; *blk(STAGE, 1, 2 * wave, 1) = (L).b01; *blk(STAGE, 1, 2 * wave + 1, 0) = (L).b10; *blk(STAGE, 1, 2 * wave + 1, 1) = (L).b11; }
#define W4K_FETCH_A(L)                                                                                   \
  {                                                                                                      \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).a00) : "v"(pa0) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).a01) : "v"(pa0) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).a10) : "v"(pa1) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).a11) : "v"(pa1) : "memory");   \
    pa0 += 2048; pa1 += 2048;                                                                            \
  }
#define W4K_FETCH_B(L)                                                                                   \
  {                                                                                                      \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).b00) : "v"(pb0) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).b01) : "v"(pb0) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"((L).b10) : "v"(pb1) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"((L).b11) : "v"(pb1) : "memory");   \
    pb0 += 2048; pb1 += 2048;                                                                            \
  }
#define W4K_STEP(ST, STN, B0R, B1R, LSET, STS)                                                      \
  {                                                                                                 \
    W4K_BH(B1R, ST, 1)                                                                              \
    W4K_FENCE                                                                                       \
    W4K_MAC1(xa, B0R, 0, 0, 1, 0) W4K_FENCE                                                         \
    W4K_AH(ya, ST, 1)             W4K_FENCE                                                         \
    W4K_MAC1(xa, B0R, 0, 0, 0, 1) W4K_FENCE                                                         \
    W4K_MAC1(xa, B0R, 0, 0, 0, 0) W4K_FENCE                                                         \
    W4K_MAC1(xa, B1R, 0, 1, 1, 0) W4K_FENCE                                                         \
    W4K_WAIT(24, LSET);                                                                             \
    W4K_STASH_A(LSET, STS)        W4K_FENCE                                                         \
    W4K_MAC1(xa, B1R, 0, 1, 0, 1) W4K_FENCE                                                         \
    W4K_STASH_B(LSET, STS)        W4K_FENCE                                                         \
    W4K_MAC1(xa, B1R, 0, 1, 0, 0) W4K_FENCE                                                         \
    W4K_FETCH_A(LSET)             W4K_FENCE                                                         \
    W4K_MAC1(ya, B1R, 1, 1, 1, 0) W4K_FENCE                                                         \
    W4K_FETCH_B(LSET)             W4K_FENCE                                                         \
    W4K_MAC1(ya, B1R, 1, 1, 0, 1) W4K_FENCE                                                         \
    W4K_BH(B0R, ST, 0)            W4K_FENCE                                                         \
    W4K_MAC1(ya, B1R, 1, 1, 0, 0) W4K_FENCE                                                         \
    W4K_AH(xa, STN, 0)            W4K_FENCE                                                         \
    W4K_MAC1(ya, B0R, 1, 0, 1, 0) W4K_FENCE                                                         \
    W4K_BH(B1R, STN, 0)           W4K_FENCE                                                         \
    W4K_MAC1(ya, B0R, 1, 0, 0, 1) W4K_FENCE                                                         \
    W4K_MAC1(ya, B0R, 1, 0, 0, 0) W4K_FENCE                                                         \
    __syncthreads();                                                                                \
  }
  for (int k = 0; k < G2; k += 4) {   // (G2 = C / 16 is a multiple of 4)
    W4K_STEP(0, 1, xb, yb, l2, 2)
    W4K_STEP(1, 2, yb, xb, l3, 3)
    W4K_STEP(2, 3, xb, yb, l0, 0)
    W4K_STEP(3, 0, yb, xb, l1, 1)
  }
  W4K_WAIT(0, l0);                    // nothing may still be landing in registers the epilogue reuses
  W4K_WAIT(0, l1);
  W4K_WAIT(0, l2);
  W4K_WAIT(0, l3);
#undef W4K_STEP
#undef W4K_STASH_A
#undef W4K_STASH_B
#undef W4K_FETCH_A
#undef W4K_FETCH_B
#undef W4K_FENCE
#undef W4K_MAC
#undef W4K_MAC1
#undef W4K_AH
#undef W4K_BH
#undef W4K_STASH
  {
    // (the lane's output coordinates are formed HERE, from a lane id the compiler cannot trace back: kept alive across the loop they spill)
    int lane2 = (int)threadIdx.x;
    asm volatile("" : "+v"(lane2));
    const int l31e = lane2 & 31, hie = (lane2 >> 5) & 1;
    const size_t sstride = (size_t)(gm.C >> 5) * 36 * 128;       // floats per sample of M
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rb = 8 * RT + 4 * wr + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cb = 8 * CT + 4 * wc + c;
        float* m0 = M + ((size_t)(rb * 8 + hie) * (gm.C >> 5) + cb) * (36 * 128) + (size_t)comp * 128 + l31e;
#pragma unroll
        for (int q = 0; q < 16; ++q) st_wt(m0 + (size_t)(2 * (q >> 2)) * sstride + (q & 3) * 32, acc[r][c][q] * inv);
      }
    }
  }
}
#undef W4K_FETCH
#undef W4K_WAIT

// NODE_TUNE_W4_F16 = 0: never the fp16-pair operands (the bf16-triple kernels everywhere: A/B measurements, tests); read per call
bool w4_f16_fits(int N, int C) {
  const W4Switches sw = w4_switches();
  if (!(sw.f16 != 0 && sw.ablate == 0 && w4_uses_bf16(N, C) && N % 16 == 0 && C % 64 == 0)) return false;
  if (C < 512 || getenv("NODE_TUNE_W4_F16_ANYC") != nullptr) return true;      // (the second: experiments -- k_w4_gemm64h at long reductions)
  return sw.h128 != 0 && N % 32 == 0 && C % 128 == 0 && (((N / 32) * (C >> 7)) & 1) == 0;    // long reductions: the LDS-tiled kernel only
}
void launch_w4_gemm_f16(const unsigned* Vh, const unsigned* Uh, float* M, const Ctrl* ctrl, int N, int C, const int* v_exp, const int* u_exp,
                        hipStream_t s) {
  const W4Switches sw = w4_switches();
  const int mode = (sw.sharev == 1 ? 2 : 0) | (sw.sharev == 2 ? 4 : 0) | (sw.hsplit ? 16 : 0);
  const W4Geom gm = w4_geom(N, C);
  {
    // NODE_TUNE_W4_H128 = 0 never / 1 wherever it fits / unset (-1): long reductions (C >= 512) -- the LDS-tiled kernel (at cfg 2 it takes
    // 14.4 us against k_w4_gemm64h's 12.7: profiles/r06_w4h_kernels.txt)
    const int nT = (N / 32) * (C >> 7);
    if ((sw.h128 == 1 || (sw.h128 < 0 && C >= 512)) && N % 32 == 0 && C % 128 == 0 && (nT & 1) == 0) {
      const size_t lds = 4 * 16 * 64 * 16;      // four stages of sixteen 1 KB blocks
      static bool attr1[MAX_DEVICES] = {}, attr2[MAX_DEVICES] = {};
      allow_full_lds(reinterpret_cast<const void*>(k_w4_gemm128h<1>), attr1);
      allow_full_lds(reinterpret_cast<const void*>(k_w4_gemm128h<2>), attr2);
      // NODE_TUNE_W4_H256 = 1 (default): 256 x 256 tiles for components 0..31, components 32..35 behind, where the geometry has them AND
      // they fill whole rounds of the chip (one workgroup per CU: 32 nT2 workgroups, a multiple of 256 -- a quarter-full round is slower
      // than k_w4_gemm128h's many small tiles); 2: wherever the geometry has them (tests); 0: never
      const int nT2 = (N / 64) * (C >> 8);
      if (sw.h256 != 0 && C >= 512 && N % 64 == 0 && C % 256 == 0 && (sw.h256 == 2 || nT2 % 8 == 0)) {
        static bool attr3[MAX_DEVICES] = {};
        allow_full_lds(reinterpret_cast<const void*>(k_w4_gemm256h), attr3);
        hipLaunchKernelGGL(k_w4_gemm256h, dim3(32 * nT2), dim3(256), 4 * 32 * 64 * 16, s, Vh, Uh, M, ctrl, gm, v_exp, u_exp);
        hipLaunchKernelGGL(k_w4_gemm128h<2>, dim3(4 * nT), dim3(256), lds, s, Vh, Uh, M, ctrl, gm, v_exp, u_exp, 1);
        return;
      }
      if (C >= 512) hipLaunchKernelGGL(k_w4_gemm128h<2>, dim3(32 * nT), dim3(256), lds, s, Vh, Uh, M, ctrl, gm, v_exp, u_exp, 0);
      else hipLaunchKernelGGL(k_w4_gemm128h<1>, dim3(32 * nT), dim3(256), lds, s, Vh, Uh, M, ctrl, gm, v_exp, u_exp, 0);
      return;
    }
  }
  const int grid64 = (N / 16) * (C >> 6) * 8;
  const size_t lds64 = 4 * 2048 * sizeof(float);
  unsigned long long* stamps = nullptr;
#ifdef NODE_DIAG
  { const char* e = getenv("NODE_TUNE_W4_STAMPS"); if (e != nullptr) stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)); }
#endif
  if (sw.hdepth == 8 && (C >> 4) % 8 == 0) hipLaunchKernelGGL(k_w4_gemm64h<8>, dim3(grid64), dim3(256), lds64, s, Vh, Uh, M, ctrl, gm, mode, v_exp, u_exp, stamps);
  else if ((C >> 4) % 4 == 0) hipLaunchKernelGGL(k_w4_gemm64h<4>, dim3(grid64), dim3(256), lds64, s, Vh, Uh, M, ctrl, gm, mode, v_exp, u_exp, stamps);
  else hipLaunchKernelGGL(k_w4_gemm64h<2>, dim3(grid64), dim3(256), lds64, s, Vh, Uh, M, ctrl, gm, mode, v_exp, u_exp, stamps);
}

void launch_w4_gemm(const float* V, const float* U, float* M, const Ctrl* ctrl, int N, int C, hipStream_t s, const unsigned short* Ub) {
  static bool attr[4][MAX_DEVICES] = {};
  const W4Switches sw = w4_switches();
  int mode = (sw.sharev == 1 ? 2 : 0) | (sw.sharev == 2 ? 4 : 0) | (sw.early ? 8 : 0);
  unsigned long long* stamps = nullptr;
#ifdef NODE_DIAG
  const int ab = sw.ablate;
  // NODE_TUNE_W4_STAMPS = device address of [grid * 4][16] u64 (tools/w4_stamps.py)
  { const char* e = getenv("NODE_TUNE_W4_STAMPS"); if (e != nullptr) stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)); }
  if (ab >= 16) {
    const char* e = getenv("NODE_TUNE_W4_PAD");
    int pv = 0, pu = 0;
    if (e != nullptr && sscanf(e, "%d,%d", &pv, &pu) == 2) mode |= ((pv & 0xff) << 8) | ((pu & 0xff) << 16);
  }
#endif
  static int xm = -1;   // NODE_TUNE_W4_XCD: workgroup -> XCD assignment of k_w4_gemm (see the kernel; results unchanged)
  if (xm < 0) { const char* e = getenv("NODE_TUNE_W4_XCD"); xm = e ? atoi(e) : 0; }
  const W4Geom gm = w4_geom(N, C);
  if (w4_takes_small(sw, N)) {   // NODE_TUNE_W4_SMALL = 0: never the small-batch kernel (A/B measurements)
    hipLaunchKernelGGL(k_w4_gemm_small, dim3(gm.RB * W4_COMPS * (C >> 5)), dim3(256), 0, s, V, U, M, ctrl, gm);
    return;
  }
  const int grid = gm.RB * (C >> 6) * 4;
  const size_t lds = 8 * 2048 * sizeof(float);
#define W4_LAUNCH(AB, SLOT)                                                               \
  {                                                                                        \
    allow_full_lds(reinterpret_cast<const void*>(k_w4_gemm<AB>), attr[SLOT]);               \
    hipLaunchKernelGGL(k_w4_gemm<AB>, dim3(grid), dim3(512), lds, s, V, U, M, ctrl, gm, xm); \
  }
  const int g64 = sw.g64;   // NODE_TUNE_W4_GEMM64 = 0: k_w4_gemm (eight waves, 32 x 64 tiles) everywhere
  // NODE_TUNE_W4_BF16X3 = 0: the fp32 MFMA kernel (A/B measurements, tests; read on every call like NODE_TUNE_WINO4)
  const bool b16 = w4_uses_bf16(N, C);
  if (g64 && N % 16 == 0) {
    const int grid64 = (N / 16) * (C >> 6) * 8;
    const size_t lds64 = 4 * 2048 * sizeof(float);
#ifdef NODE_DIAG
    if (w4_takes_uf32(sw, N, C) && Ub == nullptr) {
      hipLaunchKernelGGL(k_w4_gemm64c, dim3(grid64), dim3(256), lds64, s, V, U, M, ctrl, gm);
      return;
    }
    if (b16 && Ub != nullptr && ab >= 16) {      // NODE_TUNE_W4_ABLATE = 16 + bits: timing-only ablations of k_w4_gemm64b (results are wrong)
      switch (ab - 16) {
#define W4B_AB(X) case X: hipLaunchKernelGGL(k_w4_gemm64b<X>, dim3(grid64), dim3(256), lds64, s, V, Ub, M, ctrl, gm, mode, stamps); return;
        W4B_AB(1) W4B_AB(2) W4B_AB(4) W4B_AB(5) W4B_AB(6) W4B_AB(7) W4B_AB(8) W4B_AB(12) W4B_AB(13) W4B_AB(14) W4B_AB(15) W4B_AB(3) W4B_AB(9) W4B_AB(10) W4B_AB(11)
#undef W4B_AB
        default: break;
      }
      hipLaunchKernelGGL(k_w4_gemm64b<0>, dim3(grid64), dim3(256), lds64, s, V, Ub, M, ctrl, gm, mode, stamps);
      return;
    }
#endif
    if (b16 && Ub != nullptr && sw.ablate == 0) {
      const int g128 = sw.gemm128;   // 0 never / 1 wherever it fits / -1: long reductions (C >= 512)
      const bool fits = N % 32 == 0 && C % 128 == 0 && (((N / 32) * (C >> 7)) & 1) == 0;
      if (fits && (g128 == 1 || (g128 < 0 && C >= 512))) {
        static bool attr128[MAX_DEVICES] = {};
        const int nT = (N / 32) * (C >> 7);
        const size_t lds128 = 2 * 24 * 64 * 16;
        allow_full_lds(reinterpret_cast<const void*>(k_w4_gemm128b), attr128);
        hipLaunchKernelGGL(k_w4_gemm128b, dim3(8 * (4 * nT + nT / 2)), dim3(256), lds128, s, V, Ub, M, ctrl, gm);
        return;
      }
#ifdef NODE_DIAG
      if (sw.ksplit != 0 && C == 256) {   // NODE_TUNE_W4_KSPLIT: two waves per SIMD, a tile's K range in two halves
        hipLaunchKernelGGL(k_w4_gemm64k, dim3(64 * (N / 16)), dim3(256), 8 * 1024 * sizeof(float), s, V, Ub, M, ctrl, gm, stamps);
        return;
      }
      if (sw.lds != 0 && C == 256 && N % 32 == 0) {      // NODE_TUNE_W4_LDS: the own component's operands through an LDS-DMA ring
        static bool attrl[MAX_DEVICES] = {};
        allow_full_lds(reinterpret_cast<const void*>(k_w4_gemm64l), attrl);
        hipLaunchKernelGGL(k_w4_gemm64l, dim3(grid64), dim3(256), (size_t)W4L_NS * W4L_SLOT, s, V, Ub, M, ctrl, gm, stamps);
        return;
      }
#endif
#ifdef NODE_DIAG
      if (sw.half != 0 && C == 256 && sw.sharev == 1) {    // NODE_TUNE_W4_HALF: half-height tiles, two waves per SIMD (bit-identical)
        hipLaunchKernelGGL(k_w4_gemm32b, dim3((N / 8) * 4 * 8), dim3(256), 4 * 1024 * sizeof(float), s, V, Ub, M, ctrl, gm);
        return;
      }
#endif
      hipLaunchKernelGGL(k_w4_gemm64b<0>, dim3(grid64), dim3(256), lds64, s, V, Ub, M, ctrl, gm, mode, stamps);
      return;
    }
#ifdef NODE_DIAG
    if (ab == 1) { hipLaunchKernelGGL(k_w4_gemm64<1>, dim3(grid64), dim3(256), lds64, s, V, U, M, ctrl, gm); return; }
    if (ab == 2) { hipLaunchKernelGGL(k_w4_gemm64<2>, dim3(grid64), dim3(256), lds64, s, V, U, M, ctrl, gm); return; }
    if (ab == 4) { hipLaunchKernelGGL(k_w4_gemm64<4>, dim3(grid64), dim3(256), lds64, s, V, U, M, ctrl, gm); return; }
#endif
    hipLaunchKernelGGL(k_w4_gemm64<0>, dim3(grid64), dim3(256), lds64, s, V, U, M, ctrl, gm);
    return;
  }
#ifdef NODE_DIAG
  if (ab == 1) { W4_LAUNCH(1, 1) return; }
  if (ab == 2) { W4_LAUNCH(2, 2) return; }
  if (ab == 4) { W4_LAUNCH(4, 3) return; }
#endif
  W4_LAUNCH(0, 0)
#undef W4_LAUNCH
}


// ----------------------------------------------------------------------------
// k_w4_wgrad: the weight gradients of BOTH conv layers of an augmented evaluation in the F(4x4,3x3) domain,
//   dU_c[ci][co] = sum_rows V_c[row][ci] Z_c[row][co]        c = 0..35, rows = samples x 4 tiles
// with V = B^T d B the forward conv's own row operand (as its GroupNorm pass left it for k_w4_gemm64 -- no second copy)
// and Z = A dz A^T of the conv output's cotangent (written by the pass that produces dz, wino4.h).  dW = G^T dU G
// happens in k_theta_finalize.  2.4 GFLOP per layer instead of the F(2x2,3x3) domain's 4.3, and NO split-K slabs:
// the decomposition mirrors k_w4_gemm64 -- a wave owns one whole component of a (128 ci x 32 co) tile over the WHOLE
// reduction (four 32x32 accumulators), eight workgroups share a tile, workgroup j takes components 4j .. 4j+3 and two of
// the four accumulator blocks of component 32 + j/2, whose reduction range its four waves split and sum through LDS:
// 1024 + 128 MFMAs per wave, every SIMD of the chip the same; every result element is written once.
// Operands: a lane's 16 B of V hold FOUR ci of one row -- they feed four MFMAs with four different accumulator blocks
// (block e = channels 8 g + 4 hi + e: any assignment of channels to MFMA rows is as good as another), so V is read in
// the layout the conv wants; a lane's 16 B of Z hold four ROWS of one co ([comp][co/32][sample][co%32][tile]: one
// contiguous 1 KB per wave request).  Workgroup j of every tile runs on XCD j: its 4.5 components of V and Z stream
// through that XCD's L2 once.  Needs N % 8 == 0, C % 128 == 0.
// (Round 3, measured: the same kernel on the bf16 pipe -- both operands split into exact bf16 triples in registers, as
// k_w4_gemm64b does -- takes the same 42 us: at 94 MB of operands and results per launch the memory side, not the
// matrix pipe, bounds it.  So it stays on the fp32 instructions.)
// ----------------------------------------------------------------------------
struct W4WgOps { float4 a0, a1, a2, a3, z; };
template <int NSUB>
__device__ __forceinline__ void w4_wg_mac(float16_t (&acc)[4], const W4WgOps& o, int sub0) {
#define W4WG_STEP(A, ZC)                                                                    \
  if (NSUB == 4) {                                                                           \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, ZC, acc[0], 0, 0, 0);                  \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, ZC, acc[1], 0, 0, 0);                  \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, ZC, acc[2], 0, 0, 0);                  \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, ZC, acc[3], 0, 0, 0);                  \
  } else {                                                                                   \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sub0 ? A.z : A.x, ZC, acc[0], 0, 0, 0);     \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sub0 ? A.w : A.y, ZC, acc[1], 0, 0, 0);     \
  }
  W4WG_STEP(o.a0, o.z.x) W4WG_STEP(o.a1, o.z.y) W4WG_STEP(o.a2, o.z.z) W4WG_STEP(o.a3, o.z.w)
#undef W4WG_STEP
}
// operands of reduction unit q (eight rows = samples 2q, 2q+1): pa / pz point at unit 0 of this lane
__device__ __forceinline__ void w4_wg_load(W4WgOps& o, const float* pa, const float4* pz, int q, size_t rbs) {
  const float4* a = reinterpret_cast<const float4*>(pa + (size_t)(q >> 2) * rbs + (q & 3) * 64);
  o.a0 = a[0]; o.a1 = a[1]; o.a2 = a[2]; o.a3 = a[3];
  o.z = pz[(size_t)q * 64];
}
// acc += sum over units [q0, q0 + nq): a ring of R units in registers, each refilled right behind the MFMAs that
// consumed it.  nq must be a multiple of R.
template <int R, int NSUB>
__device__ __forceinline__ void w4_wg_run(float16_t (&acc)[4], const float* pa, const float4* pz, int q0, int nq, size_t rbs, int sub0) {
  W4WgOps ring[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    w4_wg_load(ring[i], pa, pz, q0 + i, rbs);
    __builtin_amdgcn_sched_barrier(0);
  }
  int q = q0;
  for (; q + R < q0 + nq; q += R) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      w4_wg_mac<NSUB>(acc, ring[i], sub0);
      __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the MFMAs that read the old contents
      w4_wg_load(ring[i], pa, pz, q + R + i, rbs);
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    w4_wg_mac<NSUB>(acc, ring[i], sub0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int R>
__global__ __launch_bounds__(256) void k_w4_wgrad(W4WgradArgs a) {
  if (a.ctrl != nullptr && a.ctrl->done) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2 blocks][4 r4][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int C = a.C, N = a.N;
  const int nCO = C >> 5, per_layer = (C >> 7) * nCO;
  const int j = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int layer = tile / per_layer, tl = tile - layer * per_layer;
  const int cit = tl / nCO, cot = tl - cit * nCO;
  const float* __restrict__ V = layer ? a.V2 : a.V1;
  const float4* __restrict__ Z = reinterpret_cast<const float4*>(layer ? a.Z2 : a.Z1);
  float* __restrict__ dU = a.dU + (size_t)layer * 36 * C * C;
  const int g = l31 >> 1, hic = l31 & 1;                       // accumulator row j <-> channels 8 g + 4 hic + e (block e)
  const size_t cs = (size_t)4 * N * C, rbs = (size_t)(C >> 3) * 256;   // floats per component / per 8-sample row block of V
  const int Q = N >> 1;                                        // reduction units of eight rows
  const float* pa0 = V + (size_t)(cit * 16 + g) * 256 + h * 32 + hic * 16;
  const float4* pz0 = Z + (size_t)cot * N * 32 + h * 32 + l31;
  const size_t zcs = (size_t)nCO * N * 32;                     // float4s per component of Z

  // --- this wave's own component over the whole reduction.  a.sharev (NODE_TUNE_W4_SHAREV, nCO % 4 == 0): the four waves of a
  // workgroup take ONE component and four neighbouring co tiles -- they walk the same V blocks (the 128-ci operand, 80 % of the
  // launch's operand bytes) in lock-step and share them inside the CU -- instead of four components of one tile.
  {
    const bool sharev = a.sharev != 0 && (nCO & 3) == 0;
    const int comp = 4 * j + (sharev ? (cot & 3) : wave);
    const int ocot = sharev ? (cot & ~3) + wave : cot;
    float16_t acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    w4_wg_run<R, 4>(acc, pa0 + (size_t)comp * cs, Z + (size_t)ocot * N * 32 + h * 32 + l31 + (size_t)comp * zcs, 0, Q, rbs, 0);
    float* o = dU + ((size_t)comp * C + cit * 128) * C + ocot * 32 + l31;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;          // accumulator row -> ci = 8 (m >> 1) + 4 (m & 1) + e
        st_wt(o + (size_t)(8 * (m >> 1) + 4 * (m & 1) + e) * C, acc[e][r]);
      }
  }
  // --- two accumulator blocks of a shared component: reduction range [wave Q/4, (wave+1) Q/4) per wave
  {
    const int scomp = 32 + (j >> 1), sub0 = j & 1;             // blocks e = 2 sub0, 2 sub0 + 1
    const int nq = Q >> 2, q0 = wave * nq;
    float16_t acc[4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    if (nq % 4 == 0) w4_wg_run<4, 2>(acc, pa0 + (size_t)scomp * cs, pz0 + (size_t)scomp * zcs, q0, nq, rbs, sub0);
    else for (int q = q0; q < q0 + nq; ++q) w4_wg_run<1, 2>(acc, pa0 + (size_t)scomp * cs, pz0 + (size_t)scomp * zcs, q, 1, rbs, sub0);
    float* red = smem + wave * 2048;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<float4*>(red + e * 1024 + (r4 * 64 + lane) * 4) =
            make_float4(acc[e][4 * r4], acc[e][4 * r4 + 1], acc[e][4 * r4 + 2], acc[e][4 * r4 + 3]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = tid + it * 256;
      const int e = u >> 8, r4 = (u >> 6) & 3;
      float4 s = *reinterpret_cast<const float4*>(smem + e * 1024 + (r4 * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(smem + w * 2048 + e * 1024 + (r4 * 64 + lane) * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int ee = 2 * sub0 + e;
      float* o = dU + ((size_t)scomp * C + cit * 128) * C + cot * 32 + l31;
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i + 8 * r4 + 4 * h;                       // register 4 r4 + i of the block
        st_wt(o + (size_t)(8 * (m >> 1) + 4 * (m & 1) + ee) * C, sv[i]);
      }
    }
  }
}


// ----------------------------------------------------------------------------
// k_w4_wgrad128b: k_w4_wgrad's sums on the bf16 matrix pipe at fp32 accuracy, as an LDS-tiled GEMM (the skeleton of
// k_w4_gemm128b): a workgroup owns a (128 ci x 128 co) tile of ONE component of one layer, its waves 64 x 64 quarters,
// and walks the reduction (rows = samples x 4 tiles) in K = 16 units.  The exact bf16 split costs more here than in the
// forward GEMM -- BOTH operands are fp32 activations -- and in k_w4_wgrad's decomposition (a wave = a 128 x 32 tile of its
// own component) every wave would split the same 128 ci x 16 rows again for each of the C / 32 column tiles (measured
// earlier in round 3: no faster than fp32).  Here an operand element is split once per 128-wide tile: per unit a wave
// splits 16 values per lane (88 VALU instructions) under its 24 MFMAs.
// Operands without a transposed copy: a loader lane's eight 16-B loads of V ([s 2][t 4] x four channels e) hold, for
// each e, the eight reduction rows of one channel -- the K half of an MFMA row operand -- so accumulator block e =
// channels 8 g + 4 hic + e as in k_w4_wgrad; its two 16-B loads of Z hold the eight rows of one output channel.
// Waves 0 / 1 load and split the V patch of the first / second sample of the K half (half of every block's LDS entries each),
// waves 2 / 3 two 32-column blocks of Z each.  Needs N % 4 == 0 (whole units; the launcher asks for N % 8) and C % 128 == 0.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_w4_wgrad128b(W4WgradArgs a) {
  if (a.ctrl != nullptr && a.ctrl->done) return;
  extern __shared__ __attribute__((aligned(16))) w4_u32x4 tile_lds[];   // [2 stages][A 4 e blocks x 3 parts | B 4 column blocks x 3 parts][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int C = a.C, N = a.N;
  const int nCT = C >> 7, nT = nCT * nCT;            // tiles per (layer, component)
  // XCD j (= blockIdx % 8): components 4 j .. 4 j + 3 of layer 0, then of layer 1, then half of the tiles of component 32 + j / 2
  // of either layer
  const int j = blockIdx.x & 7, i = blockIdx.x >> 3;
  int layer, comp, tile;
  if (i < 8 * nT) { layer = i / (4 * nT); const int r = i - layer * 4 * nT; comp = 4 * j + r / nT; tile = r % nT; }
  else { const int r = i - 8 * nT; layer = r / (nT >> 1); comp = 32 + (j >> 1); tile = (j & 1) * (nT >> 1) + r % (nT >> 1); }
  const int cit = tile / nCT, cot = tile - cit * nCT;
  const float* __restrict__ V = layer ? a.V2 : a.V1;
  const float4* __restrict__ Z = reinterpret_cast<const float4*>(layer ? a.Z2 : a.Z1);
  float* __restrict__ dU = a.dU + (size_t)layer * 36 * C * C;
  const size_t cs = (size_t)4 * N * C, rbs = (size_t)(C >> 3) * 256;   // floats per component / per 8-sample row block of V
  const int g = l31 >> 1, hic = l31 & 1;
  // unit u = samples 4 u .. 4 u + 3: this lane's K half h = samples 4 u + 2 h, + 1 (row block u / 2, s = 4 (u & 1) + 2 h + s')
  const float* pa = V + (size_t)comp * cs + (size_t)(cit * 16 + g) * 256 + (2 * h) * 32 + hic * 16;
  const float4* pz = Z + (size_t)comp * (size_t)(C >> 5) * N * 32 + (size_t)(4 * cot) * N * 32 + (2 * h) * 32 + l31;
  const int U = N >> 2;
  auto blk = [&](int stage, int kind, int b, int part) { return tile_lds + ((((stage * 2 + kind) * 4 + b) * 3 + part) * 64 + lane); };
  // What this wave loads for unit u, four 16-B requests each: wave 0 / 1 the V patch of sample s' = 0 / 1 of the lane's K half
  // ([t 4] x four channels e: the first / second four of the eight reduction rows of the four blocks e -- it writes the
  // first / second 8 bytes of their lanes' LDS entries), waves 2 / 3 two column blocks of Z ([block 2][s' 2]).  Requests as
  // inline asm with hand-placed waits, two register sets that are never copied, as in k_w4_gemm128b (left to the compiler
  // this loop waited for vmcnt(0) at the top of every unit: no prefetch at all).
  const bool ldv = wave < 2;
  const char* pvb = reinterpret_cast<const char*>(pa + (wave & 1) * 32);
  const char* pz0 = reinterpret_cast<const char*>(pz + (size_t)(2 * (wave & 1)) * N * 32);
  const char* pz1 = pz0 + (size_t)N * 32 * 16;
#define W4WG_FETCH(L, UU)                                                                                              \
  {                                                                                                                    \
    const int u_ = (UU) < U ? (UU) : U - 1; /* clamped: the last units prefetch one nobody consumes */                 \
    if (ldv) {                                                                                                         \
      const char* p_ = pvb + ((size_t)(u_ >> 1) * rbs + (u_ & 1) * 128) * 4;                                           \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L##0) : "v"(p_) : "memory");                               \
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(L##1) : "v"(p_) : "memory");                     \
      asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(L##2) : "v"(p_) : "memory");                     \
      asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(L##3) : "v"(p_) : "memory");                     \
    } else {                                                                                                           \
      const char* q0_ = pz0 + (size_t)u_ * 2048;                                                                       \
      const char* q1_ = pz1 + (size_t)u_ * 2048;                                                                       \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L##0) : "v"(q0_) : "memory");                              \
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(L##1) : "v"(q0_) : "memory");                   \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L##2) : "v"(q1_) : "memory");                              \
      asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(L##3) : "v"(q1_) : "memory");                   \
    }                                                                                                                  \
  }
  // wait until only the younger set's four requests are in flight; NALL: wait for everything
#define W4WG_WAIT(L, NALL)                                                                                              \
  {                                                                                                                     \
    if (NALL) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L##0), "+v"(L##1), "+v"(L##2), "+v"(L##3) : : "memory");         \
    else asm volatile("s_waitcnt vmcnt(4)" : "+v"(L##0), "+v"(L##1), "+v"(L##2), "+v"(L##3) : : "memory");              \
  }
#define W4WG_PUT(STAGE, KIND, B, SP)                                        \
  {                                                                         \
    *blk(STAGE, KIND, B, 0) = __builtin_bit_cast(w4_u32x4, (SP).h);         \
    *blk(STAGE, KIND, B, 1) = __builtin_bit_cast(w4_u32x4, (SP).m);         \
    *blk(STAGE, KIND, B, 2) = __builtin_bit_cast(w4_u32x4, (SP).l);         \
  }
  // V waves: the split of (block e, block e + 1) x four rows each; the first 8 bytes of a part belong to block e, the last to e + 1
#define W4WG_PUT_HALVES(STAGE, E, SP)                                                                                     \
  {                                                                                                                       \
    const w4_u32x4 h_ = __builtin_bit_cast(w4_u32x4, (SP).h), m_ = __builtin_bit_cast(w4_u32x4, (SP).m),                  \
                   l_ = __builtin_bit_cast(w4_u32x4, (SP).l);                                                             \
    typedef unsigned w4_u32x2_ __attribute__((ext_vector_type(2)));                                                       \
    w4_u32x2_* d0_ = reinterpret_cast<w4_u32x2_*>(blk(STAGE, 0, E, 0)) + (wave & 1);                                      \
    w4_u32x2_* d1_ = reinterpret_cast<w4_u32x2_*>(blk(STAGE, 0, (E) + 1, 0)) + (wave & 1);                                \
    d0_[0] = w4_u32x2_{h_[0], h_[1]};   d1_[0] = w4_u32x2_{h_[2], h_[3]};                                                 \
    d0_[128] = w4_u32x2_{m_[0], m_[1]}; d1_[128] = w4_u32x2_{m_[2], m_[3]};   /* parts are 64 lanes x 16 B apart */       \
    d0_[256] = w4_u32x2_{l_[0], l_[1]}; d1_[256] = w4_u32x2_{l_[2], l_[3]};                                               \
  }
#define W4WG_F4(A, B, C, D) make_float4(A, B, C, D)
#define W4WG_STASH(L, STAGE)                                                                                                  \
  {                                                                                                                           \
    if (ldv) {                                                                                                                \
      const W4Split s0_ = w4_split8(W4WG_F4((L##0).x, (L##1).x, (L##2).x, (L##3).x), W4WG_F4((L##0).y, (L##1).y, (L##2).y, (L##3).y)); \
      W4WG_PUT_HALVES(STAGE, 0, s0_)                                                                                          \
      const W4Split s1_ = w4_split8(W4WG_F4((L##0).z, (L##1).z, (L##2).z, (L##3).z), W4WG_F4((L##0).w, (L##1).w, (L##2).w, (L##3).w)); \
      W4WG_PUT_HALVES(STAGE, 2, s1_)                                                                                          \
    } else {                                                                                                                  \
      const W4Split s0_ = w4_split8(W4WG_F4((L##0).x, (L##0).y, (L##0).z, (L##0).w), W4WG_F4((L##1).x, (L##1).y, (L##1).z, (L##1).w)); \
      W4WG_PUT(STAGE, 1, 2 * (wave & 1), s0_)                                                                                 \
      const W4Split s1_ = w4_split8(W4WG_F4((L##2).x, (L##2).y, (L##2).z, (L##2).w), W4WG_F4((L##3).x, (L##3).y, (L##3).z, (L##3).w)); \
      W4WG_PUT(STAGE, 1, 2 * (wave & 1) + 1, s1_)                                                                             \
    }                                                                                                                         \
  }
  const int wr = wave >> 1, wc = wave & 1;
  float16_t acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;

  // units in flight: at the top of an even unit u, la holds unit u + 1 and lb unit u + 2 (la's requests the older)
  w4_f32x4 la0, la1, la2, la3, lb0, lb1, lb2, lb3;
  W4WG_FETCH(la, 0)
  W4WG_WAIT(la, 1)
  W4WG_STASH(la, 0)
  W4WG_FETCH(la, 1)
  W4WG_FETCH(lb, 2)
  __syncthreads();
#define W4WG_STEP(ST, L, UNEXT)                                                                                   \
  {                                                                                                               \
    w4_u32x4 fa[2][3], fb[2][3];                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int q = 0; q < 3; ++q) {                 \
      fa[r][q] = *blk(ST, 0, 2 * wr + r, q);                                                                      \
      fb[r][q] = *blk(ST, 1, 2 * wc + r, q);                                                                      \
    }                                                                                                             \
    w4c_mac(acc, fa, fb);                                                                                         \
    W4WG_WAIT(L, 0)                                                                                               \
    W4WG_STASH(L, (ST) ^ 1) /* the next unit -> the other stage (everybody left it at the last barrier) */        \
    W4WG_FETCH(L, UNEXT)                                                                                          \
    __syncthreads();                                                                                              \
  }
  for (int u = 0; u < U; u += 2) {   // (U = N / 4 is even: N % 8 == 0)
    W4WG_STEP(0, la, u + 3)
    W4WG_STEP(1, lb, u + 4)
  }
  W4WG_WAIT(la, 1)                    // nothing may still be landing in registers the epilogue reuses
  W4WG_WAIT(lb, 1)
#undef W4WG_STEP
#undef W4WG_FETCH
#undef W4WG_WAIT
#undef W4WG_PUT
#undef W4WG_PUT_HALVES
#undef W4WG_F4
#undef W4WG_STASH
  // block (e = 2 wr + r, column block 2 wc + c): accumulator row m = (q & 3) + 8 (q >> 2) + 4 h <-> ci = 8 (m >> 1) + 4 (m & 1) + e
  float* o = dU + ((size_t)comp * C + cit * 128) * C + cot * 128 + l31;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
        st_wt(o + (size_t)(8 * (m >> 1) + 4 * (m & 1) + 2 * wr + r) * C + 32 * (2 * wc + c), acc[r][c][q]);
      }
}

// ----------------------------------------------------------------------------
// k_w4_wgrad64h: k_w4_wgrad's sums on fp16 PAIRS (wino4.h): dU_c[ci][co] = 2^-(ev + ez) sum_rows V_c[row][ci] Z_c[row][co] with both
// operands as the GroupNorm passes left them -- V pairs (the forward GEMM's own row operand) and Z pairs in the SAME layout
// ([comp][rb][g2][part][s][hi][t][8 halves]: rows x 16 channels per 1 KB) -- three v_mfma_f32_32x32x16_f16 per 32 x 32 block and
// 16 rows: 14.5 GFLOP on the 2.5 PFLOP/s pipe instead of 4.8 GFLOP on the 157 TFLOP/s one.  The reduction runs over ROWS, so an
// MFMA operand is eight rows of one channel: the [row][channel] images go through LDS as they are (LDS-DMA, 1 KB per wave
// instruction, no registers) and come out transposed by ds_read_b64_tr_b16 -- a group of 16 lanes reads 4 rows (the four tiles
// of one sample) x 16 channels and each lane receives its channel's four rows.
//   workgroup = one 128 ci x 128 co tile of one (layer, component), waves = 64 x 64 quarters; K step = 16 rows = half a row block
//   (four samples): 8 KB of V + 8 KB of Z per step through a ring of W4WH_NST stages, W4WH_D steps in flight, ONE s_barrier per step
//   behind a counted s_waitcnt (the DMA pieces are inline asm: the compiler neither drains them in front of every LDS read nor
//   knows of them -- the counted waits are the only ordering, as in k_w4_gemm64l).  Odd 16-channel blocks sit in the image with
//   their samples' 128-B slots swapped pairwise (s ^ 1, applied on the DMA's per-lane SOURCE address): the two blocks a 32-lane
//   half reads then fall on different banks.  (layer, component) pairs are dealt to XCDs nine each, the tiles of a pair together:
//   its 1 MB of operands stays in that XCD's L2 for the second tile that reads it.  Two workgroups per CU.
// Needs N % 8 == 0 (whole row blocks), C % 128 == 0.
// ----------------------------------------------------------------------------
constexpr int W4WH_NST = 4, W4WH_D = 3, W4WH_STAGE = 16384;
struct W4WgradHArgs {
  const unsigned* V[2]; const unsigned* Z[2];   // per layer (Z[1] / V[1] nullable: one layer)
  float* dU; const Ctrl* ctrl; int N, C, layers;
  const int* v_exp[2]; const int* z_exp;
};
typedef short w4_s16x4 __attribute__((ext_vector_type(4)));
typedef short w4_s16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256, 2) void k_w4_wgrad64h(W4WgradHArgs a) {
  if (a.ctrl != nullptr && a.ctrl->done) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];   // [W4WH_NST][V image 8 KB | Z image 8 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C, nRB = a.N >> 3, G16 = C >> 4, nCT = C >> 7, T = nCT * nCT;
  const int per_xcd = (36 * a.layers) >> 3;                       // (layer, component) pairs per XCD: 9 or 4.5 -> see the launcher
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const bool by_xcd = ((36 * a.layers) & 7) == 0;                 // (one layer -- the stem's: 36 pairs do not split over 8 XCDs: in order)
  const int pair = by_xcd ? xcd * per_xcd + slot / T : (int)blockIdx.x / T, tile = by_xcd ? slot % T : (int)blockIdx.x % T;
  const int layer = pair / 36, comp = pair - layer * 36;
  const int cit = tile / nCT, cot = tile - cit * nCT;
  const size_t cbytes = (size_t)nRB * G16 * 2048;                 // bytes per component
  const unsigned char* Vb = reinterpret_cast<const unsigned char*>(a.V[layer]) + comp * cbytes;
  const unsigned char* Zb = reinterpret_cast<const unsigned char*>(a.Z[layer]) + comp * cbytes;
  const float inv = ldexpf(1.f, -(*a.v_exp[layer] + *a.z_exp));

  // --- DMA roles: wave w brings instructions i = 4 w .. 4 w + 3 of a stage: i < 8 the V block g2l = i (lanes 0-31 part h, 32-63
  // part l), else the Z block g2l = i - 8.  Lane -> its 16-B chunk of the 512-B half part: LDS slot (s', hi, t) <- source (s' ^ odd, hi, t)
  const unsigned char* src[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * wave + k, isz = i >> 3, g2l = i & 7;
    const int part = lane >> 5, sg = lane & 31, ss = (sg >> 3) ^ (g2l & 1), hi = (sg >> 2) & 1, t = sg & 3;
    const int g2 = (isz ? cot : cit) * 8 + g2l;
    src[k] = (isz ? Zb : Vb) + ((size_t)g2 * 2 + part) * 1024 + ((ss * 2 + hi) * 4 + t) * 16;
  }
  const size_t rb_bytes = (size_t)G16 * 2048;
  const unsigned lds0 = (unsigned)(size_t)(w4_lds_ptr_t)wsm + (unsigned)(4 * wave) * 1024u;
  auto issue = [&](int q) {                                       // K step q = (row block q / 2, half q % 2)
    const size_t off = (size_t)(q >> 1) * rb_bytes + (size_t)(q & 1) * 512;
    const unsigned dst = lds0 + (unsigned)(q % W4WH_NST) * W4WH_STAGE;
#pragma unroll
    for (int k = 0; k < 4; ++k) w4wh_dma(src[k] + off, dst + (unsigned)k * 1024u);
  };
  // --- fragment reads (ds_read_b64_tr_b16): lane = (kh, g2a, m = 4 q4 + p): rows = tiles q4 of samples 2 kh, 2 kh + 1; 8-byte column quad p
  // = channels 4 p .. 4 p + 3 of its 16-channel block (hi = p & 1, gp = p >> 1); the lane receives channel m of the block
  const int wr = wave >> 1, wc = wave & 1;
  const int kh = lane >> 5, g2a = (lane >> 4) & 1, m = lane & 15, q4 = m >> 2, pq = m & 3;
  const int lane_off = (pq & 1) * 64 + q4 * 16 + (pq >> 1) * 8;
  const int s_lo = (2 * kh) ^ g2a, s_hi = (2 * kh + 1) ^ g2a;   // LDS slots of the two samples (odd blocks are stored swapped)
  auto frag = [&](const unsigned char* img, int g2l0, int part) -> w4_f16x8 {
    const unsigned char* pb = img + ((g2l0 + g2a) * 2 + part) * 512 + lane_off;
    const w4_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w4_s16x4*)(pb + s_lo * 128));
    const w4_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w4_s16x4*)(pb + s_hi * 128));
    const w4_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(w4_f16x8, v);
  };
  float16_t acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;

  const int nK = 2 * nRB;
#pragma unroll
  for (int q = 0; q < W4WH_D; ++q)
    if (q < nK) issue(q);
  for (int q = 0; q < nK; ++q) {
    const int ahead = min(q + W4WH_D - 1, nK - 1) - q;          // stages issued behind stage q
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of stage q - 1 are in registers (its slot is refilled next)
    __builtin_amdgcn_s_barrier();
    if (q + W4WH_D < nK) issue(q + W4WH_D);
    const unsigned char* st = wsm + (q % W4WH_NST) * W4WH_STAGE;
    w4_f16x8 A[2][2], B[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        A[r][part] = frag(st, 4 * wr + 2 * r, part);
        B[r][part] = frag(st + 8192, 4 * wc + 2 * r, part);
      }
#define W4WH_P(AP, BQ)                                                                       \
  _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int c = 0; c < 2; ++c)  \
      acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[r][AP], B[c][BQ], acc[r][c], 0, 0, 0);
    W4WH_P(1, 0) W4WH_P(0, 1) W4WH_P(0, 0)   // smallest products first
#undef W4WH_P
  }
  // accumulator register q of a lane: row (q & 3) + 8 (q >> 2) + 4 kh = channel ci of the block, column lane & 31 = co
  float* o = a.dU + ((size_t)layer * 36 + comp) * C * C + (size_t)(cit * 128 + 64 * wr) * C + cot * 128 + 64 * wc + (lane & 31);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
        st_wt(o + (size_t)(32 * r + row) * C + 32 * c, acc[r][c][q] * inv);
      }
}
bool w4_wgrad_f16_fits(int N, int C) {
  const W4Switches sw = w4_switches();
  return sw.f16 != 0 && N % 8 == 0 && C % 128 == 0;
}
void launch_w4_wgrad_f16(const unsigned* V1, const unsigned* Z1, const unsigned* V2, const unsigned* Z2, float* dU, const Ctrl* ctrl, int N, int C,
                         const int* v1_exp, const int* v2_exp, const int* z_exp, hipStream_t s) {
  static bool attr[MAX_DEVICES] = {};
  W4WgradHArgs a;
  memset(&a, 0, sizeof(a));
  a.V[0] = V1; a.Z[0] = Z1; a.V[1] = V2; a.Z[1] = Z2; a.dU = dU; a.ctrl = ctrl; a.N = N; a.C = C;
  a.layers = V2 != nullptr ? 2 : 1;
  a.v_exp[0] = v1_exp; a.v_exp[1] = v2_exp; a.z_exp = z_exp;
  const int T = (C >> 7) * (C >> 7);
  allow_full_lds(reinterpret_cast<const void*>(k_w4_wgrad64h), attr);
  hipLaunchKernelGGL(k_w4_wgrad64h, dim3(36 * a.layers * T), dim3(256), (size_t)W4WH_NST * W4WH_STAGE, s, a);
}

void launch_w4_wgrad(const W4WgradArgs& a_in, hipStream_t s) {
  W4WgradArgs a = a_in;
  a.sharev = w4_switches().sharev;
  // NODE_TUNE_W4_WGRAD128 = 0 never / 1 wherever it fits / unset: long filters (C >= 512), where the fp32 kernel is bound by
  // the matrix pipe (read on every call: tests run both)
  {
    const int w128 = w4_switches().wgrad128;
    const int nT = (a.C >> 7) * (a.C >> 7);
    if (a.V2 != nullptr && a.N % 8 == 0 && a.C % 128 == 0 && (nT & 1) == 0 && (w128 == 1 || (w128 < 0 && a.C >= 512))) {
      hipLaunchKernelGGL(k_w4_wgrad128b, dim3(8 * (8 * nT + nT)), dim3(256), 2 * 24 * 64 * 16, s, a);
      return;
    }
  }
  const int grid = (a.V2 != nullptr ? 2 : 1) * (a.C >> 7) * (a.C >> 5) * 8;     // (layer = tile / tiles per layer)
  const size_t lds = 4 * 2048 * sizeof(float);
  if (a.N % 16 == 0) hipLaunchKernelGGL(k_w4_wgrad<8>, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(k_w4_wgrad<4>, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace node
