// k_conv3x3 -- software-pipelined fp32-MFMA implicit GEMM for the 3x3 convolutions
// of the ODE dynamics (forward conv and data gradient), with the GroupNorm that
// follows every conv of ODEfunc (model.py:343-347) -- or, in the backward, the ReLU
// mask + GroupNorm backward that follows every dgrad -- fused in the epilogue.
//
//   out[m, co] = sum_{tap, ci} A[pix(m) + tap, ci] * W[tap, ci, co]      M = N*H*W, K = 9*C
//
// What is different from a textbook LDS-tiled GEMM, and why (gfx950):
//  * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32: dopri5's embedded
//    error estimate is ~tol*|y| and cannot tolerate bf16 noise).  At 64 cycles per MFMA
//    the matrix pipe, not LDS bandwidth, is the bound -- so the loop is organised to
//    keep that pipe issuing back to back rather than to maximise operand reuse.
//  * M tiles are WHOLE SAMPLES and N tiles WHOLE GroupNorm groups, so the normalisation
//    statistics are tile-local and the epilogue needs no second kernel.
//  * The activation chunk (S samples x 32 channels) is staged ONCE per K chunk into a
//    zero-haloed LDS image; the nine taps are nine constant LDS offsets into it.
//  * MFMA step j of a 32-deep K chunk multiplies channels {j, 16 + j} (lane half hi
//    takes 16*hi + j), so one ds_read_b128 per operand feeds FOUR MFMA steps: a lane
//    reads channels 16*hi + 4g .. 4g+3 of its pixel (A) / its output column (B).
//    The B tile is packed [col][k] in HBM for that, and both LDS images use a 36-float
//    (144-B) row: 16-B aligned, and conflict-free for the B reads.
//  * Operands are register double-buffered one 4-step group ahead, ACROSS the piece
//    barrier too: B is triple-buffered in LDS and the next A chunk is written two taps
//    early, so the first group of piece q+1 is already in flight when the barrier of
//    piece q is reached -- the matrix pipe never waits for an LDS round trip.
//  * The constant-time channel of ConcatConv2d (model.py:321-322) is not carried
//    through K: its contribution is t * tmap[p, co] (border-aware tap sums), added
//    with the bias in the epilogue.
//  * Epilogue: accumulators -> LDS tile once; statistics with a lane<->pixel mapping
//    (conflict-free, no integer division in any loop); normalise + 16-B stores.
#include "node_internal.h"
#include <cstdlib>

namespace node {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_conv_bm = -1;

constexpr int AST2 = 36;          // floats per halo slot of the A image
constexpr int BST2 = 36;          // floats per output column of the B tile
constexpr int BBUF2 = BN * BST2;  // one B piece in LDS
constexpr int CT2 = BN + 1;       // epilogue tile stride

// 64-lane sum on the DPP cross-lane path (8 VALU ops) instead of six LDS-crossbar shuffles:
// quad swaps, row mirrors, then the two row broadcasts; the total lands in lane 63.
__device__ inline float wave_sum_p(float v) {
#define DPP_ADD(CTRL, RM)                                                                              \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, true))
  DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
  DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
  DPP_ADD(0x141, 0xf);  // row_half_mirror
  DPP_ADD(0x140, 0xf);  // row_mirror: every lane of a 16-lane row holds the row sum
  DPP_ADD(0x142, 0xa);  // row_bcast15 into rows 1 and 3
  DPP_ADD(0x143, 0xc);  // row_bcast31 into rows 2 and 3
#undef DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ inline int slot_of_p(int p, int W, int Wp) {
  const int h = p / W;
  return (h + 1) * Wp + (p - h * W) + 1;
}

#ifdef NODE_STAMPS
#define PSTAMP(buf, slot, INS)                                                                   \
  do {                                                                                           \
    if ((buf) != nullptr && (threadIdx.x & 63) == 0) {                                           \
      unsigned long long _t;                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      asm volatile(INS " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                       \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      (buf)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (slot)] = _t; \
    }                                                                                            \
  } while (0)
#define ABL(bit) (a.ablate & (bit))   /* timing-only ablations: 1 no B stream, 2 no barrier, 4 no operand reads, 8 no A stream */
#else
#define PSTAMP(buf, slot, INS) do { } while (0)
#define ABL(bit) 0
#endif

// ----------------------------------------------------------------------------
// Shared epilogue tail.  On entry the pre-normalisation tile Ct[BM][CT2] (conv output + bias + t*tmap,
// or the raw data gradient) is complete in LDS and the workgroup is synchronised.  Forward: GroupNorm
// statistics with a lane<->pixel mapping (conflict-free, no integer division in any loop, DPP wave
// reductions), normalise (+ReLU), 16-B stores of the activation and of xhat / rstd for the backward.
// Backward: ReLU mask, (dgamma, dbeta) tile partials, GroupNorm backward, 16-B stores.
// ----------------------------------------------------------------------------
template <int THREADS, int BM>
__device__ inline void conv_epilogue_tail(const ConvArgs& a, const Dims& d, float* smem, int n0, int c0, int nsamp,
                                          int ncols, int mtile) {
  constexpr int NWAVES = THREADS / 64;
  constexpr int RL = THREADS / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool fwd = a.mode != CM_BWD_RELU_GN;
  float* Ct = smem;                   // [BM][CT2]
  float* Xt = smem + BM * CT2;        // [BM][CT2]   (bwd only)
  float* st0 = smem + 2 * BM * CT2;   // [S*BN] mean / m1
  float* st1 = st0 + d.S * BN;        // [S*BN] rstd / m2
  float* cred = st1 + d.S * BN;       // [RL][64][2]
  const int GT = ncols / d.cpg;  // whole groups in this tile
  const int npairs = nsamp * GT;
  const float inv_m = 1.0f / (float)(d.HW * d.cpg);
  // thread <-> (column quad, row lane) mapping of the store passes
  const int colq = (tid & 15) * 4, rr = tid >> 4;
  const bool vec_ok = ((c0 & 3) == 0) && ((ncols & 3) == 0);
  int glq[4];
  bool okq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    okq[i] = (colq + i) < ncols;
    glq[i] = okq[i] ? (colq + i) / d.cpg : 0;
  }

  if (fwd) {
    for (int pair = wave; pair < npairs; pair += NWAVES) {
      const int s = pair / GT, gl = pair - s * GT;
      const float* base = Ct + (s * d.HW) * CT2 + gl * d.cpg;
      float sum = 0.f;
      for (int p = lane; p < d.HW; p += 64) {
#pragma unroll 8
        for (int cc = 0; cc < d.cpg; ++cc) sum += base[p * CT2 + cc];
      }
      const float mean = wave_sum_p(sum) * inv_m;
      float s2 = 0.f;
      for (int p = lane; p < d.HW; p += 64) {
#pragma unroll 8
        for (int cc = 0; cc < d.cpg; ++cc) {
          const float dv = base[p * CT2 + cc] - mean;
          s2 += dv * dv;
        }
      }
      const float var = wave_sum_p(s2) * inv_m;
      const float rstd = 1.0f / sqrtf(var + d.eps);
      if (lane == 0) {
        st0[pair] = mean;
        st1[pair] = rstd;
        if (a.rstd_out) a.rstd_out[(size_t)(n0 + s) * d.G + c0 / d.cpg + gl] = rstd;
      }
    }
    __syncthreads();
    PSTAMP(a.stamps, 7, "s_memtime");
    float gm[4], bt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gm[i] = okq[i] ? a.gamma[c0 + colq + i] : 0.f;
      bt[i] = okq[i] ? a.beta[c0 + colq + i] : 0.f;
    }
    const bool relu = a.mode == CM_FWD_GN_RELU;
    for (int s = 0; s < nsamp; ++s) {
      float mean[4], rstd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { mean[i] = st0[s * GT + glq[i]]; rstd[i] = st1[s * GT + glq[i]]; }
      for (int p = rr; p < d.HW; p += RL) {
        const float* src = Ct + (s * d.HW + p) * CT2 + colq;
        float xh[4], o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[i] = (src[i] - mean[i]) * rstd[i];
          float v = xh[i] * gm[i] + bt[i];
          if (relu) v = fmaxf(v, 0.f);
          o[i] = a.osign * v;
        }
        const size_t off = ((size_t)(n0 + s) * d.HW + p) * d.C + c0 + colq;
        if (vec_ok) {
          if (okq[0]) {
            *reinterpret_cast<float4*>(a.out + off) = make_float4(o[0], o[1], o[2], o[3]);
            if (a.xhat_out) *reinterpret_cast<float4*>(a.xhat_out + off) = make_float4(xh[0], xh[1], xh[2], xh[3]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (okq[i]) {
              a.out[off + i] = o[i];
              if (a.xhat_out) a.xhat_out[off + i] = xh[i];
            }
        }
      }
    }
  } else {
    // ReLU mask, dxhat = du * gamma, column partials of (dgamma, dbeta)
    float gm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gm[i] = okq[i] ? a.gamma[c0 + colq + i] : 0.f;
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsamp; ++s) {
      for (int p = rr; p < d.HW; p += RL) {
        const int row = s * d.HW + p;
        const size_t off = ((size_t)(n0 + s) * d.HW + p) * d.C + c0 + colq;
        float x[4], ac[4];
        if (vec_ok) {
          float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), av = xv;
          if (okq[0]) {
            xv = *reinterpret_cast<const float4*>(a.xhat + off);
            av = *reinterpret_cast<const float4*>(a.act + off);
          }
          x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w;
          ac[0] = av.x; ac[1] = av.y; ac[2] = av.z; ac[3] = av.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[i] = okq[i] ? a.xhat[off + i] : 0.f;
            ac[i] = okq[i] ? a.act[off + i] : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float du = (okq[i] && ac[i] > 0.f) ? Ct[row * CT2 + colq + i] : 0.f;
          dg[i] += du * x[i];
          db[i] += du;
          Ct[row * CT2 + colq + i] = du * gm[i];
          Xt[row * CT2 + colq + i] = x[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cred[(rr * 64 + colq + i) * 2] = dg[i];
      cred[(rr * 64 + colq + i) * 2 + 1] = db[i];
    }
    __syncthreads();
    if (tid < 128) {
      const int col = tid & 63, which = tid >> 6;
      if (col < ncols) {
        float v = 0.f;
#pragma unroll 8
        for (int r = 0; r < RL; ++r) v += cred[(r * 64 + col) * 2 + which];
        a.gpart[((size_t)mtile * 2 + which) * d.C + c0 + col] = v;
      }
    }
    for (int pair = wave; pair < npairs; pair += NWAVES) {
      const int s = pair / GT, gl = pair - s * GT;
      const int base = (s * d.HW) * CT2 + gl * d.cpg;
      float s1 = 0.f, s2 = 0.f;
      for (int p = lane; p < d.HW; p += 64) {
#pragma unroll 8
        for (int cc = 0; cc < d.cpg; ++cc) {
          const float dxh = Ct[base + p * CT2 + cc];
          s1 += dxh;
          s2 += dxh * Xt[base + p * CT2 + cc];
        }
      }
      s1 = wave_sum_p(s1) * inv_m;
      s2 = wave_sum_p(s2) * inv_m;
      if (lane == 0) { st0[pair] = s1; st1[pair] = s2; }
    }
    __syncthreads();
    for (int s = 0; s < nsamp; ++s) {
      float m1[4], m2[4], rs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m1[i] = st0[s * GT + glq[i]];
        m2[i] = st1[s * GT + glq[i]];
        rs[i] = okq[i] ? a.rstd[(size_t)(n0 + s) * d.G + c0 / d.cpg + glq[i]] : 0.f;
      }
      for (int p = rr; p < d.HW; p += RL) {
        const int row = s * d.HW + p;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[i] = a.osign * (rs[i] * (Ct[row * CT2 + colq + i] - m1[i] - Xt[row * CT2 + colq + i] * m2[i]));
        const size_t off = ((size_t)(n0 + s) * d.HW + p) * d.C + c0 + colq;
        if (vec_ok) {
          if (okq[0]) *reinterpret_cast<float4*>(a.out + off) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (okq[i]) a.out[off + i] = o[i];
        }
      }
    }
  }
}

template <int WM, int MT>
__global__ __launch_bounds__(WM * 128) void k_conv3x3(ConvArgs a, Dims d) {
  PSTAMP(a.stamps, 0, "s_memrealtime");
  PSTAMP(a.stamps, 1, "s_memtime");
  constexpr int THREADS = WM * 128;        // WM waves in M x 2 in N
  constexpr int NB = 512 / THREADS;        // float4 of one B tile per thread
  constexpr int BM = WM * 32 * MT;
  constexpr int NA = 2 * MT;  // float4 staging units per thread for one A chunk
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
  const int ABUF = AROWS * AST2;
  float* Abuf = smem;             // 2 x ABUF
  float* Bbuf = smem + 2 * ABUF;  // 3 x BBUF2

  // ---- zero both A images (halo, margins, channel padding) ----
  for (int i = tid * 4; i < 2 * ABUF; i += THREADS * 4)
    *reinterpret_cast<float4*>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- per-thread staging descriptors for the A chunk ----
  size_t gofs[NA];
  int lofs[NA];
  bool aval[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tid + i * THREADS;
    const int row = u >> 3, q4 = u & 7;
    aval[i] = row < rows_valid;
    const int rr = aval[i] ? row : 0;
    const int s = rr / d.HW, p = rr - s * d.HW;
    gofs[i] = ((size_t)(n0 + s) * d.HW + p) * d.C + q4 * 4;
    lofs[i] = (d.MARGIN + s * d.SLOTS + slot_of_p(p, d.W, d.Wp)) * AST2 + q4 * 4;
  }
  const int q4t = tid & 7;
  // ---- per-lane operand offsets ----
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = wm * (32 * MT) + mt * 32 + l31;
    int slot = 0;
    if (row < d.S * d.HW) {
      const int s = row / d.HW, p = row - s * d.HW;
      slot = s * d.SLOTS + slot_of_p(p, d.W, d.Wp);
    }
    arow[mt] = (d.MARGIN + slot) * AST2 + 16 * hi;
  }
  const int boff = (wn * 32 + l31) * BST2 + 16 * hi;
  int bwr[NB];   // where this thread's float4s of a B tile land
#pragma unroll
  for (int j = 0; j < NB; ++j) bwr[j] = ((tid + j * THREADS) >> 3) * BST2 + q4t * 4;

  const float* wbase = a.wpacked + (size_t)nt * d.nchunk * 9 * (KCH * BN);
  const int Q = d.nchunk * 9;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const bool fwd = a.mode != CM_BWD_RELU_GN;
  const int ncols = min(d.BNE, d.C - c0);

  float4 areg[NA];
  float4 breg[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) breg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();  // zero fill visible

  // B tile of tap T lives in LDS slot T % 3 (nine taps per chunk: the rotation is static)
#define BSLOT(T) ((T) % 3)
  constexpr int A_LD = 4;   // tap at whose start the next chunk's activations are requested
  constexpr int A_WR = 7;   // tap at whose end they are written to the other A image

  // ---- prologue: A chunk 0 and the B tiles of taps 0 and 1 ----
  {
    float4 bpro[2][NB];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        bpro[j][b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < Q) bpro[j][b] = *reinterpret_cast<const float4*>(wbase + (size_t)j * (KCH * BN) + (tid + b * THREADS) * 4);
      }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (aval[i] && q4t * 4 < d.C) areg[i] = *reinterpret_cast<const float4*>(a.in + gofs[i]);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (aval[i] && q4t * 4 < d.C) *reinterpret_cast<float4*>(Abuf + lofs[i]) = areg[i];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int b = 0; b < NB; ++b) *reinterpret_cast<float4*>(Bbuf + BSLOT(j) * BBUF2 + bwr[b]) = bpro[j][b];
  }
  __syncthreads();
  PSTAMP(a.stamps, 2, "s_memtime");

  // time-channel map values of this lane's 16 x MT output elements: requested now, used in the
  // epilogue, so their latency hides behind the whole main loop
  float tmv[MT][16];
  {
    const int col = wn * 32 + l31;
    const bool cok = col < ncols;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int pb = (wm * (32 * MT) + mt * 32) % d.HW;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int p = pb + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (d.HW >= 32) p = p >= d.HW ? p - d.HW : p;
        else p = p % d.HW;
        tmv[mt][r] = (fwd && cok) ? a.tmap[(size_t)p * d.C + c0 + col] : 0.f;
      }
    }
  }


  // operand register sets (group g of a tap = MFMA steps 4g..4g+3)
  float4 pa0[MT], pa1[MT], pb0, pb1;
#ifdef NODE_STAMPS
  pb1 = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int mt = 0; mt < MT; ++mt) pa1[mt] = make_float4(1.f, 2.f, 3.f, 4.f);
#endif
#define LOADG(PA, PB, AB, BB, G)                                                          \
  do {                                                                                    \
    if (!ABL(4)) {                                                                        \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                    \
          PA[mt] = *reinterpret_cast<const float4*>((AB) + arow[mt] + 4 * (G));           \
      PB = *reinterpret_cast<const float4*>((BB) + boff + 4 * (G));                       \
    }                                                                                     \
  } while (0)
#define MFMA4(PA, PB)                                                                     \
  do {                                                                                    \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[mt].x, PB.x, acc[mt], 0, 0, 0); \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[mt].y, PB.y, acc[mt], 0, 0, 0); \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[mt].z, PB.z, acc[mt], 0, 0, 0); \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[mt].w, PB.w, acc[mt], 0, 0, 0); \
  } while (0)
  // operand reads run one group ahead of the MFMAs that consume them; the scheduling
  // barriers keep hipcc from sinking the reads back down to their first use
#define SB __builtin_amdgcn_sched_barrier(0)
#define TAPHEAD(AC, BC)                                            \
  do {                                                             \
    LOADG(pa1, pb1, AC, BC, 1); SB; MFMA4(pa0, pb0); SB;           \
    LOADG(pa0, pb0, AC, BC, 2); SB; MFMA4(pa1, pb1); SB;           \
    LOADG(pa1, pb1, AC, BC, 3); SB; MFMA4(pa0, pb0); SB;           \
  } while (0)
#define TAPTAIL(AN, BNX)                                           \
  do {                                                             \
    LOADG(pa0, pb0, AN, BNX, 0); SB; MFMA4(pa1, pb1); SB;          \
  } while (0)

  // tap offsets into the haloed A image (wave-uniform, live in SGPRs)
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * d.Wp + (t % 3 - 1)) * AST2;

  LOADG(pa0, pb0, Abuf + toff[0], Bbuf, 0);

  // One tap of one K chunk (T compile-time) between two barriers.  At the end of the tap, in this
  // order: (1) the staged B tile of tap T+2 (requested one tap ago; its slot was last read one tap
  // ago) and, at tap A_WR, the next chunk's activations are written to LDS; (2) the B tile of tap
  // T+3 is requested; (3) the first operand group of tap T+1 is prefetched -- its data was made
  // visible by an earlier barrier; (4) the last MFMA group issues; (5) barrier, waiting only for the
  // staging writes (LDS ops retire in order, so lgkmcnt(MT + 1) leaves exactly the prefetch reads
  // in flight): the matrix pipe never waits for an LDS round trip, not even across the barrier.
#define BLOAD(TQ)                                                                                  \
  {                                                                                                \
    const int pq = qbase + (TQ);                                                                   \
    if (pq < Q && !ABL(1)) {                                                                       \
      _Pragma("unroll") for (int b = 0; b < NB; ++b)                                               \
        breg[b] = *reinterpret_cast<const float4*>(wbase + (size_t)pq * (KCH * BN) + (tid + b * THREADS) * 4); \
    }                                                                                              \
  }
#define PIECE(T)                                                                                   \
  {                                                                                                \
    constexpr int TN = ((T) + 1) % 9;                                                              \
    if constexpr ((T) == A_LD) {                                                                   \
      if (more_chunks && !ABL(8)) {                                                                \
        const int cbase = (chunk + 1) * KCH;                                                       \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                           \
          areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                               \
          if (aval[i] && cbase + q4t * 4 < d.C)                                                    \
            areg[i] = *reinterpret_cast<const float4*>(a.in + gofs[i] + cbase);                    \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
    TAPHEAD(Acur + toff[T], Bbuf + BSLOT(T) * BBUF2);                                              \
    if (qbase + (T) + 2 < Q && !ABL(1)) {                                                          \
      _Pragma("unroll") for (int b = 0; b < NB; ++b)                                               \
        *reinterpret_cast<float4*>(Bbuf + BSLOT((T) + 2) * BBUF2 + bwr[b]) = breg[b];              \
    }                                                                                              \
    if constexpr ((T) == A_WR) {                                                                   \
      if (more_chunks && !ABL(8)) {                                                                \
        const int cbase = (chunk + 1) * KCH;                                                       \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                             \
          if (aval[i] && cbase + q4t * 4 < d.C) *reinterpret_cast<float4*>(Anxt + lofs[i]) = areg[i]; \
      }                                                                                            \
    }                                                                                              \
    BLOAD((T) + 3)                                                                                 \
    SB;                                                                                            \
    TAPTAIL(((T) == 8 ? Anxt : Acur) + toff[TN], Bbuf + BSLOT(TN) * BBUF2);                        \
    if (!ABL(2)) {                                                                                 \
      if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                    \
      else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");                                      \
      __builtin_amdgcn_s_barrier();                                                                \
      SB;                                                                                          \
    }                                                                                              \
  }

  {  // B tile of tap 2: written at the end of tap 0
    const int qbase = 0;
    BLOAD(2)
  }
  for (int chunk = 0; chunk < d.nchunk; ++chunk) {
    const bool more_chunks = (chunk + 1) < d.nchunk;
    const int qbase = chunk * 9;
    float* Acur = Abuf + (chunk & 1) * ABUF;
    float* Anxt = more_chunks ? Abuf + ((chunk + 1) & 1) * ABUF : Acur;
    PIECE(0) PIECE(1) PIECE(2) PIECE(3) PIECE(4) PIECE(5) PIECE(6) PIECE(7) PIECE(8)
  }
#undef PIECE
#undef BLOAD
#undef TAPHEAD
#undef TAPTAIL
#undef SB
#undef LOADG
#undef MFMA4
#undef BSLOT
  PSTAMP(a.stamps, 3, "s_memtime");

  // ==========================================================================
  // epilogue: accumulators -> LDS tile -> GroupNorm (fwd or bwd) -> HBM
  // ==========================================================================
  float* Ct = smem;                   // [BM][CT2] pre-normalisation tile (the tail lays out the rest of the LDS)

  {
    const float tval = fwd ? eval_time(a.et) : 0.f;
    const int col = wn * 32 + l31;
    const int c = c0 + col;
    const bool cok = col < ncols;
    const float bias = (fwd && cok) ? a.bias[c] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int rbase = wm * (32 * MT) + mt * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = (r & 3) + 8 * (r >> 2) + 4 * hi;
        Ct[(rbase + off) * CT2 + col] = acc[mt][r] + (bias + tval * tmv[mt][r]);
      }
    }
  }
  __syncthreads();
  PSTAMP(a.stamps, 6, "s_memtime");

  conv_epilogue_tail<THREADS, BM>(a, d, smem, n0, c0, nsamp, ncols, mtile);
  PSTAMP(a.stamps, 4, "s_memtime");
  PSTAMP(a.stamps, 5, "s_memrealtime");
}

size_t conv_lds_bytes(const Dims& d, int /*mode*/) {
  const size_t arows = (size_t)d.S * d.SLOTS + 2 * d.MARGIN;
  const size_t main_loop = 2 * arows * AST2 + 3 * (size_t)BBUF2;
  const size_t epi = 2 * (size_t)d.BM * CT2 + 2 * (size_t)d.S * BN + 32 * 64 * 2;
  return (main_loop > epi ? main_loop : epi) * sizeof(float);
}

template <int WM, int MT>
static void launch_conv_t(const Dims& d, const ConvArgs& a, hipStream_t s) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv3x3<WM, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL((k_conv3x3<WM, MT>), dim3(d.mtiles, d.ntile), dim3(WM * 128), conv_lds_bytes(d, a.mode), s, a, d);
}

// d.BM (chosen by make_dims): 64 = four-wave workgroups, two of which share a CU and cover each other's
// barriers / prologue / epilogue when the grid is small; 128 / 256 = eight waves.
void launch_conv(const Dims& d, const ConvArgs& a, hipStream_t s) {
  if (d.BM == 64) launch_conv_t<2, 1>(d, a, s);
  else if (d.BM == 128) launch_conv_t<4, 1>(d, a, s);   // (four waves x (64 x 32) per wave measured slower: 94 vs 91 us at cfg 2)
  else launch_conv_t<4, 2>(d, a, s);
}

}  // namespace node
