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
#include <cstring>
#include <cstdlib>

namespace node {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_conv_bm = -1;
int g_conv_wino = -1;

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
      (buf)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = _t; \
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
        if (a.spart) {   // keep the finished tile for the column sums below (same thread read these four entries)
#pragma unroll
          for (int i = 0; i < 4; ++i) Ct[row * CT2 + colq + i] = okq[i] ? o[i] : 0.f;
        }
      }
    }
    if (a.spart) {
      // masked column sums of the data gradient this tile just produced (it is the next layer's dz): saves a
      // k_colsum launch and its pass over the tensor.  Xt is free now: border flags, then the reduction scratch.
      __syncthreads();
      unsigned char* flg = reinterpret_cast<unsigned char*>(Xt);
      float* red9 = Xt + 64;   // HW <= BM <= 256 bytes of flags
      for (int p = tid; p < d.HW; p += THREADS) {
        const int h = p / d.W, x = p - h * d.W;
        flg[p] = (unsigned char)((h == 0 ? 1 : 0) | (h == d.H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == d.W - 1 ? 8 : 0));
      }
      __syncthreads();
      for (int s = 0; s < nsamp; ++s)
        masked_colsum_tile(Ct + (s * d.HW) * CT2, CT2, d.HW, flg, ncols, max(1, min(min(THREADS / ncols, 8), (BM * CT2 - 64) / (9 * ncols))), tid, red9,
                           a.spart + (size_t)(n0 + s) * 9 * d.C + c0, d.C);
    }
  }
}

template <int WM, int MT>
__global__ __launch_bounds__(WM * 128) void k_conv3x3(ConvArgs a, Dims d) {
  if (a.et.ctrl != nullptr && a.et.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)

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

// ============================================================================
// k_conv3x3_w -- the same convolution through a 1-D Winograd F(2,3) transform along the image rows:
//   out[h, 2t + {0,1}] from in[h + kh - 1, 2t - 1 .. 2t + 2]:   12 instead of 18 multiplies per output pair
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3          (input,  at staging time)
//     U0 = g0   U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2          (filter, once per solve: k_pack_weights_w)
//     M_j[h, t, co] = sum_{kh, ci} V_j[h + kh - 1, t, ci] * U_j[kh, ci, co]     (MFMA: 4 components x 3 row taps)
//     y0 = M0 + M1 + M2          y1 = M1 - M2 - M3                         (output, in the epilogue)
// 1.5 x fewer MFMAs for fp32-benign coefficients (+-1, 1/2).  The GEMM rows are (sample, row, column
// pair) "tile-rows"; a workgroup owns 32*MT tile-rows = 64*MT pixels = whole samples x 64 output channels;
// its eight waves are 4 components x 2 column halves, MT accumulators each.  K chunks are 16 channels
// (MFMA step s multiplies channels {s, 8 + s}: one ds_read_b128 per operand feeds four steps), a piece is
// (chunk, kh): 2 operand groups of 4 steps.  Pipeline, staging discipline and barrier are those of
// k_conv3x3; the epilogue folds the four component tiles into the pixel tile in LDS (four passes of
// read-modify-write, one per component), adds bias + t*tmap, then runs the shared tail.
// Requires even W (the direct kernel serves odd widths).
// ============================================================================
constexpr int KCW = 16;            // channels per K chunk
constexpr int ASTW = 20;           // floats per (slot, component) row of the A image: 16 channels + 16-B pad
constexpr int SSTW = 4 * ASTW + 4; // floats per slot: 21 16-B units, odd, so the ds_read_b128 of 16 consecutive tile-rows
                                   // (= 16 consecutive slots) hit 16 different bank quads -- 80 floats gave a 4-way conflict
constexpr int BSTW = 20;           // floats per (component, column) row of a B tile
constexpr int BBUFW = 4 * BN * BSTW;

template <int MT>
__global__ __launch_bounds__(512) void k_conv3x3_w(ConvArgs a, Dims d) {
  if (a.et.ctrl != nullptr && a.et.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)

  PSTAMP(a.stamps, 0, "s_memrealtime");
  PSTAMP(a.stamps, 1, "s_memtime");
  constexpr int THREADS = 512;
  constexpr int TR = 32 * MT;        // tile-rows per workgroup
  constexpr int BM = 64 * MT;        // pixels per workgroup
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int jc = wave >> 1, wn = wave & 1;   // Winograd component, column half
  const int mtile = blockIdx.x, nt = blockIdx.y;
  const int n0 = mtile * d.S;
  const int c0 = nt * d.BNE;
  const int nsamp = min(d.S, d.N - n0);
  const int NT = d.W >> 1;           // column pairs per image row
  const int TRS = d.H * NT;          // tile-rows per sample
  const int SLW = (d.H + 2) * NT;    // slots per sample: one zero halo row above and below
  const int tr_valid = nsamp * TRS;
  const bool fwd = a.mode != CM_BWD_RELU_GN;
  const int ncols = min(d.BNE, d.C - c0);

  const int ABUF = d.S * SLW * SSTW;
  float* Abuf = smem;             // 2 x ABUF
  float* Bbuf = smem + 2 * ABUF;  // 3 x BBUFW

  for (int i = tid * 4; i < 2 * ABUF; i += THREADS * 4)
    *reinterpret_cast<float4*>(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- staging descriptor: thread u < 4 * TR handles (tile-row u >> 2, channel quad u & 3) ----
  const int q4s = tid & 3;
  bool sval = false;
  size_t sgofs = 0;       // global offset of pixel (h, 2t) of the tile-row, channel quad q4s
  int slofs = 0;          // LDS offset of its slot, component 0
  int sx0 = 0;
  if (tid < 4 * TR) {
    const int m = tid >> 2;
    if (m < tr_valid) {
      const int s = m / TRS, rem = m - s * TRS;
      const int h = rem / NT, t = rem - h * NT;
      sval = true;
      sx0 = 2 * t;
      sgofs = ((size_t)(n0 + s) * d.HW + h * d.W + 2 * t) * d.C + q4s * 4;
      slofs = (s * SLW + (h + 1) * NT + t) * SSTW + q4s * 4;
    }
  }
  // ---- tile-row tables (one thread per tile-row does the two integer divisions; lanes read LDS) ----
  int* stab = reinterpret_cast<int*>(Bbuf + 3 * BBUFW);   // [TR] A-image slot of tile-row m (row tap kh adds kh * NT)
  int* ptab = stab + TR;                                   // [TR] pixel row of output pixel (h, 2t) in the tile, -1 if none
  int* qtab = ptab + TR;                                   // [TR] the same pixel's index inside its sample
  if (tid < TR) {
    int slot = 0, pr = -1, q = 0;
    if (tid < d.S * TRS) {
      const int s = tid / TRS, rem = tid - s * TRS;
      const int h = rem / NT, t = rem - h * NT;
      slot = s * SLW + h * NT + t;
      q = h * d.W + 2 * t;
      pr = s * d.HW + q;
    }
    stab[tid] = slot;
    ptab[tid] = pr;
    qtab[tid] = q;
  }
  const int boff = ((jc * BN) + wn * 32 + l31) * BSTW + 8 * hi;
  int bwr[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int f = tid + b * THREADS;          // float4 index inside the packed tile [j][col][16]
    bwr[b] = (f >> 2) * BSTW + (f & 3) * 4;   // (j * 64 + col) = f >> 2
  }

  const int nchunk = (d.C + KCW - 1) / KCW;
  const int Q = nchunk * 3;
  const float* wbase = a.wpacked + (size_t)nt * Q * (4 * BN * KCW);

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  float4 areg[4];
  float4 breg[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) breg[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();  // zero fill + tables visible
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = stab[mt * 32 + l31] * SSTW + jc * ASTW + 8 * hi;

  // load the four pixels 2t-1 .. 2t+2 of a tile-row (zero outside the row) for channels cbase + 4*q4s ..
#define ALOAD(CBASE)                                                                       \
  {                                                                                        \
    const bool cv = sval && (CBASE) + q4s * 4 < d.C;                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
      const int x = sx0 - 1 + i;                                                           \
      areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                           \
      if (cv && x >= 0 && x < d.W)                                                         \
        areg[i] = *reinterpret_cast<const float4*>(a.in + sgofs + (ptrdiff_t)(i - 1) * d.C + (CBASE)); \
    }                                                                                      \
  }
  // input transform + write of the four components
#define AWRITE(ABASE)                                                                      \
  if (sval) {                                                                              \
    const float4 d0 = areg[0], d1 = areg[1], d2 = areg[2], d3 = areg[3];                   \
    float* dst = (ABASE) + slofs;                                                          \
    *reinterpret_cast<float4*>(dst) = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);             \
    *reinterpret_cast<float4*>(dst + ASTW) = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);      \
    *reinterpret_cast<float4*>(dst + 2 * ASTW) = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);  \
    *reinterpret_cast<float4*>(dst + 3 * ASTW) = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);  \
  }

  // ---- prologue: A chunk 0, B tiles of pieces 0 and 1 ----
  {
    float4 bpro[2][2];
#pragma unroll
    for (int jq = 0; jq < 2; ++jq)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        bpro[jq][b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jq < Q) bpro[jq][b] = *reinterpret_cast<const float4*>(wbase + (size_t)jq * (4 * BN * KCW) + (tid + b * THREADS) * 4);
      }
    ALOAD(0)
    AWRITE(Abuf)
#pragma unroll
    for (int jq = 0; jq < 2; ++jq)
#pragma unroll
      for (int b = 0; b < 2; ++b) *reinterpret_cast<float4*>(Bbuf + jq * BBUFW + bwr[b]) = bpro[jq][b];
  }
  __syncthreads();
  PSTAMP(a.stamps, 2, "s_memtime");

  // bias + t * tmap of the pixel-tile elements this thread finalises in the output transform, requested
  // now so their latency hides behind the main loop.  MT <= 2: column tid & 63, tile-rows (tid >> 6) + 8i,
  // both pixels of the pair; MT == 4: column tid & 63, pixel rows (tid >> 6) + 8i.
  float tmv[BM / 8];
  int ptr_[MT <= 2 ? TR / 8 : 1];   // pixel rows of this thread's tile-rows (single-pass transform)
  {
    const int col = tid & 63;
    const bool cok = fwd && col < ncols;
    const float tval = fwd ? eval_time(a.et) : 0.f;
    const float bias = cok ? a.bias[c0 + col] : 0.f;
    if constexpr (MT <= 2) {
#pragma unroll
      for (int i = 0; i < TR / 8; ++i) {
        const int m = (tid >> 6) + 8 * i;
        ptr_[i] = ptab[m];
        const int q = qtab[m];
        tmv[2 * i] = cok ? bias + tval * a.tmap[(size_t)q * d.C + c0 + col] : 0.f;
        tmv[2 * i + 1] = cok ? bias + tval * a.tmap[(size_t)(q + 1) * d.C + c0 + col] : 0.f;
      }
    } else {
      ptr_[0] = 0;
      int p = tid >> 6;
      while (p >= d.HW) p -= d.HW;
#pragma unroll
      for (int i = 0; i < BM / 8; ++i) {
        tmv[i] = cok ? bias + tval * a.tmap[(size_t)p * d.C + c0 + col] : 0.f;
        p += 8;
        while (p >= d.HW) p -= d.HW;
      }
    }
  }

  float4 pa0[MT], pa1[MT], pb0, pb1;
#define LOADG(PA, PB, AB, BB, G)                                                          \
  do {                                                                                    \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        PA[mt] = *reinterpret_cast<const float4*>((AB) + arow[mt] + 4 * (G));             \
    PB = *reinterpret_cast<const float4*>((BB) + boff + 4 * (G));                         \
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
#define SB __builtin_amdgcn_sched_barrier(0)

  const int khoff = NT * SSTW;   // one image row of slots
  LOADG(pa0, pb0, Abuf, Bbuf, 0);
  {  // B tile of piece 2: written at the end of piece 0
    if (2 < Q) {
#pragma unroll
      for (int b = 0; b < 2; ++b) breg[b] = *reinterpret_cast<const float4*>(wbase + (size_t)2 * (4 * BN * KCW) + (tid + b * THREADS) * 4);
    }
  }
  // One piece = (chunk, kh): same discipline as k_conv3x3's PIECE (write B tile q+2, request q+3,
  // prefetch the first group of piece q+1, last MFMA group, counted wait, barrier).
#define WPIECE(KH)                                                                                 \
  {                                                                                                \
    constexpr int KN = ((KH) + 1) % 3;                                                             \
    if constexpr ((KH) == 0) {                                                                     \
      if (more_chunks) ALOAD((chunk + 1) * KCW)                                                    \
    }                                                                                              \
    LOADG(pa1, pb1, Acur + (KH) * khoff, Bbuf + (KH) * BBUFW, 1); SB;                              \
    MFMA4(pa0, pb0); SB;                                                                           \
    if (qbase + (KH) + 2 < Q) {                                                                    \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                \
        *reinterpret_cast<float4*>(Bbuf + (((KH) + 2) % 3) * BBUFW + bwr[b]) = breg[b];            \
    }                                                                                              \
    if constexpr ((KH) == 1) {                                                                     \
      if (more_chunks) AWRITE(Anxt)                                                                \
    }                                                                                              \
    if (qbase + (KH) + 3 < Q) {                                                                    \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                \
        breg[b] = *reinterpret_cast<const float4*>(wbase + (size_t)(qbase + (KH) + 3) * (4 * BN * KCW) + (tid + b * THREADS) * 4); \
    }                                                                                              \
    SB;                                                                                            \
    LOADG(pa0, pb0, ((KH) == 2 ? Anxt : Acur) + KN * khoff, Bbuf + KN * BBUFW, 0); SB;             \
    MFMA4(pa1, pb1); SB;                                                                           \
    if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                      \
    else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");                 \
    else asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                                  \
    SB;                                                                                            \
  }

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const bool more_chunks = (chunk + 1) < nchunk;
    const int qbase = chunk * 3;
    float* Acur = Abuf + (chunk & 1) * ABUF;
    float* Anxt = more_chunks ? Abuf + ((chunk + 1) & 1) * ABUF : Acur;
    WPIECE(0) WPIECE(1) WPIECE(2)
  }
#undef WPIECE
#undef SB
#undef LOADG
#undef MFMA4
#undef ALOAD
#undef AWRITE
  PSTAMP(a.stamps, 3, "s_memtime");

  // ---- output transform into the pixel tile: y0 = M0 + M1 + M2, y1 = M1 - M2 - M3 ----
  float* Ct = smem;  // [BM][CT2]
  if constexpr (MT <= 2) {
    // single pass: the four component tiles go to LDS side by side (behind the region the tail uses),
    // then every thread folds the pairs of its (column, tile-rows) and adds bias + t * tmap
    float* Mt = smem + 2 * BM * CT2 + 2 * d.S * BN + 32 * 64 * 2;   // [4][TR][CT2]
    {
      const int col = wn * 32 + l31;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Mt[(jc * TR + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * CT2 + col] = acc[mt][r];
    }
    __syncthreads();
    PSTAMP(a.stamps, 8, "s_memtime");
    {
      const int col = tid & 63;
#pragma unroll
      for (int i = 0; i < TR / 8; ++i) {
        const int m = (tid >> 6) + 8 * i;
        const float m0 = Mt[m * CT2 + col], m1 = Mt[(TR + m) * CT2 + col];
        const float m2 = Mt[(2 * TR + m) * CT2 + col], m3 = Mt[(3 * TR + m) * CT2 + col];
        if (ptr_[i] >= 0) {
          Ct[ptr_[i] * CT2 + col] = ((m0 + m1) + m2) + tmv[2 * i];
          Ct[(ptr_[i] + 1) * CT2 + col] = ((m1 - m2) - m3) + tmv[2 * i + 1];
        }
      }
    }
    __syncthreads();
  } else {
    // LDS cannot hold four 128-row component tiles next to the 256-row pixel tile: three passes of
    // read-modify-write (component 0 sets y0 and component 3 sets y1; component 1 adds to both; component 2
    // adds to y0 and subtracts from y1), each lane's reads batched ahead of its writes
    int prw[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) prw[mt][r] = ptab[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
    __syncthreads();   // every wave holds its table values: the tile may now overwrite the LDS
    const int col = wn * 32 + l31;
    if (jc == 0 || jc == 3) {
      const int o = jc == 0 ? 0 : CT2;
      const float sg = jc == 0 ? 1.f : -1.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (prw[mt][r] >= 0) Ct[prw[mt][r] * CT2 + col + o] = sg * acc[mt][r];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 1; pass <= 2; ++pass) {
      if (jc == pass) {
        const float s1 = pass == 1 ? 1.f : -1.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float o0[16], o1[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int pr = prw[mt][r] >= 0 ? prw[mt][r] : 0;
            o0[r] = Ct[pr * CT2 + col];
            o1[r] = Ct[pr * CT2 + col + CT2];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (prw[mt][r] >= 0) {
              Ct[prw[mt][r] * CT2 + col] = o0[r] + acc[mt][r];
              Ct[prw[mt][r] * CT2 + col + CT2] = o1[r] + s1 * acc[mt][r];
            }
        }
      }
      __syncthreads();
    }
    if (fwd) {
      const int colf = tid & 63;
      float o[BM / 8];
#pragma unroll
      for (int i = 0; i < BM / 8; ++i) o[i] = Ct[((tid >> 6) + 8 * i) * CT2 + colf];
#pragma unroll
      for (int i = 0; i < BM / 8; ++i) Ct[((tid >> 6) + 8 * i) * CT2 + colf] = o[i] + tmv[i];
      __syncthreads();
    }
  }
  PSTAMP(a.stamps, 6, "s_memtime");
  conv_epilogue_tail<THREADS, BM>(a, d, smem, n0, c0, nsamp, ncols, mtile);
  PSTAMP(a.stamps, 4, "s_memtime");
  PSTAMP(a.stamps, 5, "s_memrealtime");
}

static size_t conv_w_lds_bytes(const Dims& d) {
  const size_t abuf = (size_t)d.S * (d.H + 2) * (d.W / 2) * SSTW;
  const size_t main_loop = 2 * abuf + 3 * (size_t)BBUFW + 3 * (size_t)(d.BM / 2);   // + the three tile-row tables
  size_t epi = 2 * (size_t)d.BM * CT2 + 2 * (size_t)d.S * BN + 32 * 64 * 2;
  if (d.BM <= 128) epi += 4 * (size_t)(d.BM / 2) * CT2;                               // + the four component tiles
  return (main_loop > epi ? main_loop : epi) * sizeof(float);
}

// ============================================================================
// k_conv3x3_w2 -- 2-D Winograd F(2x2, 3x3): each 2x2 output tile comes from a 4x4 input patch through
// sixteen component products instead of 36 multiplies per channel pair (2.25 x fewer MFMAs than the
// direct kernel, 1.5 x fewer than k_conv3x3_w):
//     V = B^T d B  (4x4, at staging time)    U = G g G^T  (4x4, once per solve: k_pack_weights_w2)
//     M_c[tile, co] = sum_ci V_c[tile, ci] * U_c[ci, co],  c = (xi, nu)           (MFMA, K = channels only)
//     Y = A^T M A  (2x2, in the epilogue)
// Workgroup = 32 tiles (128 pixels = whole samples) x 64 output channels x 16 components; wave w owns
// components 2w, 2w+1 for both column halves (four accumulators).  No two waves share a filter
// operand, so U never touches LDS: two pieces ahead it goes from L2 straight into registers (eight
// float4 per lane and K chunk).  The A image [tile][component][16 channels] is triple-buffered in LDS,
// which leaves ONE barrier per K chunk of 32 MFMAs per wave.  The 4x4 input transform is split over
// the four lanes of a quad: each lane loads one patch row, transforms it along x, and gets the one other
// row it needs for the transform along y through a DPP quad permute.  The epilogue folds the sixteen
// component tiles in two rounds (eight at a time through LDS) and then runs the shared tail.
// Requires even H and W and 128-pixel tiles; other geometries use k_conv3x3_w / k_conv3x3.
// ============================================================================
constexpr int SST2 = 16 * ASTW + 4;   // floats per tile of the A image: 16 components x 20, + 4: 81 16-B units (odd)

__global__ __launch_bounds__(512) void k_conv3x3_w2(ConvArgs a, Dims d) {
  if (a.et.ctrl != nullptr && a.et.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)

  PSTAMP(a.stamps, 0, "s_memrealtime");
  PSTAMP(a.stamps, 1, "s_memtime");
  constexpr int THREADS = 512;
  constexpr int TT = 32;             // tiles per workgroup
  constexpr int BM = 128;            // pixels per workgroup
  constexpr int ABUF = TT * SST2;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // XCD-aware tile order (speed-neutral, traffic only): blocks b and b + 8 share an XCD and its 4 MB L2.  In launch
  // order an XCD sees every column tile, i.e. the whole packed filter (4.2 MB at C = 256: 8 copies = 33.6 MB of
  // the 50.9 MB this kernel fetched per launch); giving each XCD two column tiles x a quarter of the pixel tiles
  // fetches the filter 4 x and the activations 2 x instead: the minimum over such partitions.
  int mtile = blockIdx.x, nt = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y, L = blockIdx.x + gx * blockIdx.y;
    if (gy == 4 && (gx & 3) == 0) {
      const int xcd = L & 7, slot = L >> 3;      // slot 0 .. gx/2 - 1
      nt = (xcd & 1) * 2 + (slot & 1);
      mtile = (xcd >> 1) * (gx >> 2) + (slot >> 1);
    }
  }
  // whole samples per tile, or (Dims::csplit workgroups per sample, images larger than 128 pixels) one 32-tile
  // band of ONE sample: whole tile rows, 128 consecutive pixels; GroupNorm is then a separate pointwise pass and
  // this kernel stores its raw tile
  const int csp = d.csplit;
  const int n0 = csp ? mtile / csp : mtile * d.S;
  const int band = csp ? mtile - n0 * csp : 0;
  const int c0 = nt * d.BNE;
  const int nsamp = csp ? 1 : min(d.S, d.N - n0);
  const int TW = d.W >> 1, TH = d.H >> 1;
  const int TPS = TH * TW;           // tiles per sample
  const int tiles_valid = csp ? TT : nsamp * TPS;
  const bool fwd = a.mode != CM_BWD_RELU_GN;
  const int ncols = min(d.BNE, d.C - c0);

  float* Abuf = smem;                // 3 x ABUF: chunk c in buffer c % 3
  // the tables sit behind BOTH the activation buffers and the epilogue's transform region: the first waves out
  // of the main loop write that region while the last ones still read the tables
  const int epi_end = 2 * BM * CT2 + 2 * d.S * BN + 32 * 64 * 2 + 8 * TT * CT2;
  int* ptab = reinterpret_cast<int*>(smem + max(3 * ABUF, epi_end));   // [TT] pixel row (in the tile) of output pixel (2 th, 2 tw), -1 if none
  int* qtab = ptab + TT;                                  // [TT] the same pixel's index inside its sample
  if (tid < TT) {
    int pr = -1, q = 0;
    if (csp) {
      const int thl = tid / TW, tw = tid - thl * TW;
      pr = (2 * thl) * d.W + 2 * tw;
      q = band * BM + pr;
    } else if (tid < d.S * TPS) {
      const int s = tid / TPS, rem = tid - s * TPS;
      const int th = rem / TW, tw = rem - th * TW;
      q = (2 * th) * d.W + 2 * tw;
      pr = s * d.HW + q;
    }
    ptab[tid] = pr;
    qtab[tid] = q;
  }

  // ---- staging descriptor: thread = (tile, channel quad, patch row r).  The four patch pixels are fetched with
  //      global loads "scalar base + 32-bit lane offset": the base (workgroup's first sample + chunk) moves on
  //      the scalar unit, the lane offsets are fixed for the whole kernel, and a pixel outside the image (or a
  //      tile past the batch) points at the row of C zeros the host keeps behind the tensor -- so the requests
  //      are branch-free (exact vmcnt bookkeeping; with exec-masked loads the compiler waited for the youngest
  //      request at every use) and cost no VALU per chunk.  Range-checked buffer loads did the same but issued
  //      ~100 cycles slower each (measured). ----
  const int sr = tid & 3, sq4 = (tid >> 2) & 3, stile = tid >> 4;
  // component (xi = sr, nu = 0).  ds_write_b128 is serviced in groups of 8 contiguous lanes with banks (a/4) % 32:
  // the four patch rows of a quad sit 80 floats apart (= 16 banks), so rows 0/2 and 1/3 collided (2-way, measured
  // as 31 % of all LDS cycles); the channel quad of rows 2 and 3 is stored at position sq4 ^ 2 instead, and the
  // readers of those components (xi = 2, 3) swap their lane halves to match.
  const int slofs = stile * SST2 + (sr * 4) * ASTW + ((sq4 ^ (sr & 2)) * 4);
  const float* abase = a.in + (size_t)n0 * d.HW * d.C;
  const unsigned zoff = (unsigned)(((size_t)(d.N - n0) * d.HW * d.C + sq4 * 4) * sizeof(float));   // the zero row
  unsigned svoff[4] = {zoff, zoff, zoff, zoff};   // byte offsets of the four patch pixels of row sr
  if (stile < tiles_valid) {
    const int s = csp ? 0 : stile / TPS, rem = csp ? band * TT + stile : stile - s * TPS;
    const int th = rem / TW, tw = rem - th * TW;
    const int sy = 2 * th - 1 + sr;
    if (sy >= 0 && sy < d.H) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = 2 * tw - 1 + i;
        if (x >= 0 && x < d.W) svoff[i] = (unsigned)((((s * d.HW) + sy * d.W + x) * d.C + sq4 * 4) * 4);
      }
    }
  }

  // ---- operand offsets: wave w = the four components (xi = w >> 1, nu = 0..3) of column half w & 1, so the
  //      nu half of the output transform happens in registers before anything goes through LDS ----
  const int wxi = wave >> 1, wnn = wave & 1;
  const int arow = l31 * SST2 + (4 * wxi) * ASTW + 8 * (hi ^ (wxi >> 1));   // + nu * ASTW, + 4 g  (xi >= 2: quads 0,1 <-> 2,3, see slofs)
  const int nchunk = (d.C + KCW - 1) / KCW;
  const float* wbase = a.wpacked + (size_t)nt * nchunk * (16 * BN * KCW);
  const int bofs = ((4 * wxi) * BN + wnn * 32 + l31) * KCW + 8 * hi;   // [comp][col][16]: + nu * BN*KCW, + 4 g

  f32x16 acc[4];   // [nu]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  // no zero fill: every (tile, component, channel) slot is rewritten for each chunk and the pads are never read

  float4 areg[2][4];   // patch rows of chunks q+1 and q+2 (set = chunk parity): requested two chunks ahead, the
                       // activations come from HBM / MALL and one chunk (~2500 cycles) did not cover them
  // CBASE is clamped by the callers (requests run up to three chunks ahead; C % 32 == 0 here: no ragged chunk)
#define ALOAD1(DST, CBASE, I)                                                              \
  DST[I] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(abase + (CBASE)) + svoff[I]);
  // x transform of this lane's patch row, then the y transform with ONE other row of the quad:
  //   xi = 0: e(0) - e(2)   xi = 1: e(1) + e(2)   xi = 2: e(2) - e(1)   xi = 3: e(3) - e(1)    (lane r = xi owns e(r))
  // Row xi = 3 is the NEGATIVE of the textbook B^T row (e(1) - e(3)); k_pack_weights_w2 negates the same row of the
  // filter transform, so the products are unchanged.  That makes every lane's result "own + (+/-) the other row":
  // two instructions per element -- a DPP-fused XOR that fetches the other row and sets its sign, and an add --
  // instead of four (mov_dpp, two multiplies by +/-1, fma).  The transform is the main loop's VALU bill, and a
  // VALU instruction costs ~11 cycles of this wave's issue time next to the partner's MFMA stream.
#define QPX(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), 0x5A /* quad_perm [2,2,1,1] */, 0xf, 0xf, true) ^ spmask)
  const int spmask = sr == 1 ? 0 : (int)0x80000000;   // the other row enters negated except for xi = 1
  // one transformed column nu of the patch row: 4 channels, one 16-B LDS write
#define AWRITE1(SRC, ABASE, NU)                                                            \
  {                                                                                        \
    const float4 pl = (NU) == 0 ? SRC[0] : (NU) == 2 ? SRC[2] : SRC[1];                    \
    const float4 pr = (NU) == 0 ? SRC[2] : (NU) == 2 ? SRC[1] : (NU) == 1 ? SRC[2] : SRC[3]; \
    const float sg = (NU) == 1 ? 1.f : -1.f;                                               \
    const float e0 = pl.x + sg * pr.x, e1 = pl.y + sg * pr.y, e2 = pl.z + sg * pr.z, e3 = pl.w + sg * pr.w; \
    *reinterpret_cast<float4*>((ABASE) + slofs + (NU) * ASTW) =                            \
        make_float4(e0 + QPX(e0), e1 + QPX(e1), e2 + QPX(e2), e3 + QPX(e3));              \
  }

  // filter operands [register set][nu 4 x group 2]: the set of chunk q+1 fills while
  // chunk q computes
  float4 pb[2][8];
#define BLOAD1(SET, PQ, G, J)                                                               \
  pb[SET][(J) * 2 + (G)] = *reinterpret_cast<const float4*>(wbase + (size_t)(PQ) * (16 * BN * KCW) + bofs + 4 * (G) + (J) * (BN * KCW));

#define SB __builtin_amdgcn_sched_barrier(0)
  __syncthreads();  // zero fill + tables visible
  // Prologue requests in the steady state's order (filter group 0, activations, filter group 1), pinned: the
  // compiler merges the wait state of the loop entry into every iteration, so a different order here would
  // cost a wait for the youngest request in every chunk.
  {
    float4 areg0[4];
    const int cbl1 = min(KCW, d.C - KCW), cbl2 = min(2 * KCW, d.C - KCW);
    ALOAD1(areg0, 0, 0) ALOAD1(areg0, 0, 1) ALOAD1(areg0, 0, 2) ALOAD1(areg0, 0, 3)
    ALOAD1(areg[1], cbl1, 0) ALOAD1(areg[1], cbl1, 1) ALOAD1(areg[1], cbl1, 2) ALOAD1(areg[1], cbl1, 3)
    SB;
    BLOAD1(0, 0, 0, 0) BLOAD1(0, 0, 0, 1) BLOAD1(0, 0, 0, 2) BLOAD1(0, 0, 0, 3)
    SB;
    BLOAD1(0, 0, 1, 0) BLOAD1(0, 0, 1, 1) BLOAD1(0, 0, 1, 2) BLOAD1(0, 0, 1, 3)
    SB;
    ALOAD1(areg[0], cbl2, 0) ALOAD1(areg[0], cbl2, 1) ALOAD1(areg[0], cbl2, 2) ALOAD1(areg[0], cbl2, 3)
    SB;
    AWRITE1(areg0, Abuf, 0) AWRITE1(areg0, Abuf, 1) AWRITE1(areg0, Abuf, 2) AWRITE1(areg0, Abuf, 3)
    SB;
  }
  __syncthreads();
  PSTAMP(a.stamps, 2, "s_memtime");

  float4 pa0[4], pa1[4];
#define LOADA2(PA, AB, G)                                                                  \
  do {                                                                                     \
    _Pragma("unroll") for (int nu_ = 0; nu_ < 4; ++nu_)                                    \
      PA[nu_] = *reinterpret_cast<const float4*>((AB) + arow + nu_ * ASTW + 4 * (G));      \
  } while (0)
  // Half a k step: the two MFMAs of components nu = 2 CC, 2 CC + 1.  MFMA intrinsics carry no chain, so
  // instruction selection may float them past a sched_barrier and the memory operations behind it; the empty
  // asm that "uses" the accumulators ties the pair to its place.
#define HSTEP(CC, PA, SET, G, E)                                                                                             \
  acc[2 * (CC)] = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[2 * (CC)].E, pb[SET][(2 * (CC)) * 2 + (G)].E, acc[2 * (CC)], 0, 0, 0);                 \
  acc[2 * (CC) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[2 * (CC) + 1].E, pb[SET][(2 * (CC) + 1) * 2 + (G)].E, acc[2 * (CC) + 1], 0, 0, 0); \
  asm volatile("" : "+v"(acc[2 * (CC)]), "+v"(acc[2 * (CC) + 1]) :: "memory");

  LOADA2(pa0, Abuf, 0);
#ifdef NODE_STAMPS
  unsigned long long tk_prev, tk_acc[4] = {0, 0, 0, 0};
#define TICK0 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_prev)::"memory")
#define TICK(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tk_acc[k] += t_ - tk_prev; tk_prev = t_; } while (0)
#else
#define TICK0 do { } while (0)
#define TICK(k) do { } while (0)
#endif
  // One piece = one K chunk (16 channels): 16 half steps of two MFMAs, each followed by ONE staging item, so
  // that neither the vector-memory issue (16 cycles of address path per 16-B request), the LDS writes nor the
  // transform VALU come as a burst in which the matrix pipe idles:
  //   half steps  0- 3 (group 0, k = 0,1): the four group-0 filter requests of chunk q+1 (other register set)
  //   half steps  4- 7 (group 0, k = 2,3): transform + LDS write of the four columns of chunk q+1's patch row
  //   half steps  8-11 (group 1, k = 0,1): the four activation requests of chunk q+3 (into the registers just freed)
  //   -- lgkmcnt(0) + barrier (the LDS writes have had four half steps to drain), group-0 operands of q+1 --
  //   half steps 12-15 (group 1, k = 2,3): the four group-1 filter requests of chunk q+1
  // Every request is unconditional and at least half a chunk ahead of its use; the in-order vmcnt waits the
  // compiler derives are exact (8 / 8 / 4 younger requests).  Buffer (q + 1) % 3 was last read in chunk q-2,
  // which every wave had left before anyone passed the previous barrier.  The pieces alternate between the two
  // filter register sets; an odd chunk count gets a phantom piece (activations out of range = zero).
#ifdef W2_NO_BLOAD
#define LB(...)
#else
#define LB(...) BLOAD1(__VA_ARGS__)
#endif
#ifdef W2_NO_ALOAD
#define LA(...)
#else
#define LA(...) ALOAD1(__VA_ARGS__)
#endif
#ifdef W2_NO_AWRITE
#define LW(...)
#else
#define LW(...) AWRITE1(__VA_ARGS__)
#endif
#define PIECE2(SET, NSET)                                                                  \
  {                                                                                        \
    float* Anxt = Abuf + abuf_n * ABUF;                                                    \
    const int q1 = min(chunk + 1, nchunk - 1);                                             \
    const int cb2 = min((chunk + 3) * KCW, d.C - KCW);                                                   \
    LOADA2(pa1, Acur, 1); SB;                                                              \
    HSTEP(0, pa0, SET, 0, x) LB(NSET, q1, 0, 0) SB;                                    \
    HSTEP(1, pa0, SET, 0, x) LB(NSET, q1, 0, 1) SB;                                    \
    HSTEP(0, pa0, SET, 0, y) LB(NSET, q1, 0, 2) SB;                                    \
    HSTEP(1, pa0, SET, 0, y) LB(NSET, q1, 0, 3) SB;                                    \
    TICK(0);                                                                               \
    HSTEP(0, pa0, SET, 0, z) LW(areg[NSET], Anxt, 0) SB;                                    \
    HSTEP(1, pa0, SET, 0, z) LW(areg[NSET], Anxt, 1) SB;                                    \
    HSTEP(0, pa0, SET, 0, w) LW(areg[NSET], Anxt, 2) SB;                                    \
    HSTEP(1, pa0, SET, 0, w) LW(areg[NSET], Anxt, 3) SB;                                    \
    TICK(1);                                                                               \
    HSTEP(0, pa1, SET, 1, x) LB(NSET, q1, 1, 0) SB;                                        \
    HSTEP(1, pa1, SET, 1, x) LB(NSET, q1, 1, 1) SB;                                        \
    HSTEP(0, pa1, SET, 1, y) LB(NSET, q1, 1, 2) SB;                                        \
    HSTEP(1, pa1, SET, 1, y) LB(NSET, q1, 1, 3) SB;                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
    __builtin_amdgcn_s_barrier();                                                          \
    SB;                                                                                    \
    TICK(2);                                                                               \
    LOADA2(pa0, Anxt, 0); SB;                                                              \
    HSTEP(0, pa1, SET, 1, z) LA(areg[NSET], cb2, 0) SB;                                    \
    HSTEP(1, pa1, SET, 1, z) LA(areg[NSET], cb2, 1) SB;                                    \
    HSTEP(0, pa1, SET, 1, w) LA(areg[NSET], cb2, 2) SB;                                    \
    HSTEP(1, pa1, SET, 1, w) LA(areg[NSET], cb2, 3) SB;                                    \
    TICK(3);                                                                               \
    Acur = Anxt;                                                                           \
    abuf_n = abuf_n == 2 ? 0 : abuf_n + 1;                                                 \
    ++chunk;                                                                               \
  }
  {
    int abuf_n = 1;
    float* Acur = Abuf;
    TICK0;
    for (int chunk = 0; chunk < nchunk;) {
      PIECE2(0, 1)
      PIECE2(1, 0)
#ifdef NODE_STAMPS
      if (chunk == 2) PSTAMP(a.stamps, 11, "s_memtime");   // end of the first loop iteration (cold instruction cache)
#endif
    }
  }
#undef PIECE2
#undef LB
#undef LA
#undef LW
#ifdef NODE_STAMPS
  if (a.stamps != nullptr && (threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k)
      a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + 12 + k] = tk_acc[k];
#endif
#undef TICK0
#undef TICK
#undef SB
#undef HSTEP
#undef LOADA2
#undef ALOAD1
#undef AWRITE1
#undef BLOAD1
#undef QPX
  PSTAMP(a.stamps, 3, "s_memtime");

  // bias + t * tmap of the pixel-tile elements this thread finalises (column tid & 63, tiles (tid >> 6) + 8 i, 2x2
  // pixels each): requested here, consumed after the two transform rounds (the kernel is at its register limit,
  // so they are not held across the main loop)
  float tmv[16];
  int ptl[4];
  {
    const int col = tid & 63;
    const int ccol = c0 + min(col, ncols - 1);   // clamped: the requests below stay unconditional (independent, one wait)
    const float tval = fwd ? eval_time(a.et) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ptl[i] = ptab[(tid >> 6) + 8 * i];
    if (fwd) {   // wave-uniform
      const float bias = a.bias[ccol];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = qtab[(tid >> 6) + 8 * i];
#pragma unroll
        for (int e = 0; e < 4; ++e) tmv[4 * i + e] = a.tmap[(size_t)(q + (e >> 1) * d.W + (e & 1)) * d.C + ccol];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) tmv[k] = bias + tval * tmv[k];
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) tmv[k] = 0.f;
    }
  }

  // ---- output transform Y = A^T M A: the nu half in registers, the xi half through LDS ----
  //   T[xi][0] = M[xi][0] + M[xi][1] + M[xi][2]     T[xi][1] = M[xi][1] - M[xi][2] - M[xi][3]
  //   Y[0][j] = T[0][j] + T[1][j] + T[2][j]         Y[1][j] = T[1][j] - T[2][j] - T[3][j]
  float* Ct = smem;  // [BM][CT2]
  float* Mt = smem + 2 * BM * CT2 + 2 * d.S * BN + 32 * 64 * 2;   // T [xi 4][j 2][TT][CT2], behind the region the tail uses
  float y[4][4];     // [tile i][pixel e = 2 * row + col]
  {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      Mt[((wxi * 2 + 0) * TT + row) * CT2 + wnn * 32 + l31] = acc[0][r] + m12;
      Mt[((wxi * 2 + 1) * TT + row) * CT2 + wnn * 32 + l31] = d12 - acc[3][r];
    }
  }
  __syncthreads();
  {
    const int col = tid & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tl = (tid >> 6) + 8 * i;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float t0 = Mt[((0 * 2 + j) * TT + tl) * CT2 + col], t1 = Mt[((1 * 2 + j) * TT + tl) * CT2 + col];
        const float t2 = Mt[((2 * 2 + j) * TT + tl) * CT2 + col], t3 = Mt[((3 * 2 + j) * TT + tl) * CT2 + col];
        y[i][j] = (t0 + t1) + t2;
        y[i][2 + j] = (t1 - t2) - t3;
      }
    }
  }
  {
    const int col = tid & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ptl[i] >= 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Ct[(ptl[i] + (e >> 1) * d.W + (e & 1)) * CT2 + col] = y[i][e] + tmv[4 * i + e];
      }
  }
  __syncthreads();
  PSTAMP(a.stamps, 6, "s_memtime");
  if (a.raw_out) {   // split mode: the band's 128 pixels x ncols, as they are (conv + bias + t * tmap, or the raw data gradient)
    const int colq = (tid & 15) * 4, rr = tid >> 4;
    const bool vec_ok = ((c0 & 3) == 0) && ((ncols & 3) == 0);
    for (int p = rr; p < BM; p += THREADS / 16) {
      const float* src = Ct + p * CT2 + colq;
      const size_t off = ((size_t)n0 * d.HW + (size_t)band * BM + p) * d.C + c0 + colq;
      if (vec_ok) {
        if (colq < ncols) *reinterpret_cast<float4*>(a.raw_out + off) = make_float4(src[0], src[1], src[2], src[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (colq + i < ncols) a.raw_out[off + i] = src[i];
      }
    }
    PSTAMP(a.stamps, 4, "s_memtime");
    PSTAMP(a.stamps, 5, "s_memrealtime");
    return;
  }
  conv_epilogue_tail<THREADS, BM>(a, d, smem, n0, c0, nsamp, ncols, mtile);
  PSTAMP(a.stamps, 4, "s_memtime");
  PSTAMP(a.stamps, 5, "s_memrealtime");
}

static size_t conv_w2_lds_bytes(const Dims& d) {
  const size_t main_loop = 3 * (size_t)32 * SST2;
  const size_t epi = 2 * (size_t)128 * CT2 + 2 * (size_t)d.S * BN + 32 * 64 * 2 + 8 * (size_t)32 * CT2;
  return ((main_loop > epi ? main_loop : epi) + 64) * sizeof(float);   // + the two tile tables
}

// 2-D filter transform + packing: packed[nt][chunk16][comp = xi*4 + nu][col 64][k 16],  U = G g G^T
struct PackJobs { const float* w[4]; float* packed[4]; int dgrad[4]; };
__global__ __launch_bounds__(256) void k_pack_weights_w2(PackJobs jobs, int C, int BNE, int ntile, int nchunk) {
  // blockIdx.y = job: the forward / data-gradient packings of both conv layers leave in one launch per solve
  const float* __restrict__ w = jobs.w[blockIdx.y];
  float* __restrict__ packed = jobs.packed[blockIdx.y];
  const int dgrad = jobs.dgrad[blockIdx.y];
  // thread = (nt, chunk, col, k): reads the nine taps of its (co, ci) pair ONCE and writes all sixteen components
  // (consecutive threads -> consecutive k, col: every component's store is contiguous across the wave)
  const size_t total = (size_t)ntile * nchunk * BN * KCW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = idx % KCW;
    size_t r = idx / KCW;
    const int col = r % BN; r /= BN;
    const int ch = r % nchunk;
    const int nt = r / nchunk;
    const int kidx = ch * KCW + kk, nidx = nt * BNE + col;
    float g[3][3];
    const bool on = col < BNE && kidx < C && nidx < C;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        g[kh][kw] = !on ? 0.f
                        : dgrad ? w[(((size_t)kidx * (C + 1) + 1 + nidx) * 3 + (2 - kh)) * 3 + (2 - kw)]
                                : w[(((size_t)nidx * (C + 1) + 1 + kidx) * 3 + kh) * 3 + kw];
    auto G = [](int a, int b) -> float {   // rows of G: [1,0,0], [.5,.5,.5], [.5,-.5,.5], [0,0,1]
      return a == 0 ? (b == 0 ? 1.f : 0.f) : a == 1 ? 0.5f : a == 2 ? (b == 1 ? -0.5f : 0.5f) : (b == 2 ? 1.f : 0.f);
    };
    float* dst = packed + (((size_t)(nt * nchunk + ch) * 16) * BN + col) * KCW + kk;   // + comp * BN * KCW
#pragma unroll
    for (int comp = 0; comp < 16; ++comp) {
      const int xi = comp >> 2, nu = comp & 3;
      float v = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) v += G(xi, kh) * g[kh][kw] * G(nu, kw);
      if (xi == 3) v = -v;   // the kernel's input transform uses the negated row xi = 3 (see AWRITE1)
      dst[(size_t)comp * BN * KCW] = v;
    }
  }
}
void launch_pack_weights_w2_multi(const Dims& d, const float* const* w, float* const* packed, const int* dgrad, int count,
                                 hipStream_t s) {
  PackJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  for (int i = 0; i < count; ++i) { jobs.w[i] = w[i]; jobs.packed[i] = packed[i]; jobs.dgrad[i] = dgrad[i]; }
  const int nchunk = (d.C + KCW - 1) / KCW;
  const size_t total = (size_t)d.ntile * nchunk * BN * KCW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_weights_w2, dim3(blocks, count), dim3(256), 0, s, jobs, d.C, d.BNE, d.ntile, nchunk);
}
void launch_pack_weights_w2(const Dims& d, const float* w, float* packed, int dgrad, hipStream_t s) {
  launch_pack_weights_w2_multi(d, &w, &packed, &dgrad, 1, s);
}

static size_t conv_d_lds_bytes(const Dims& d) {
  const size_t arows = (size_t)d.S * d.SLOTS + 2 * d.MARGIN;
  const size_t main_loop = 2 * arows * AST2 + 3 * (size_t)BBUF2;
  const size_t epi = 2 * (size_t)d.BM * CT2 + 2 * (size_t)d.S * BN + 32 * 64 * 2;
  return (main_loop > epi ? main_loop : epi) * sizeof(float);
}
size_t conv_lds_bytes(const Dims& d, int /*mode*/) { return d.wino == 2 ? conv_w2_lds_bytes(d) : d.wino ? conv_w_lds_bytes(d) : conv_d_lds_bytes(d); }

// packed-weight elements of one conv layer (forward or dgrad operand)
size_t conv_packed_elems(const Dims& d) {
  if (d.wino == 2) return (size_t)d.ntile * ((d.C + KCW - 1) / KCW) * (16 * BN * KCW);
  return d.wino ? (size_t)d.ntile * ((d.C + KCW - 1) / KCW) * 3 * (4 * BN * KCW) : (size_t)d.ntile * d.nchunk * 9 * (KCH * BN);
}

// Winograd filter transform + packing: packed[nt][chunk16][kh][j][col 64][k 16]
//   forward: g_kw = W[co = nt*BNE + col][1 + ci = chunk*16 + k][kh][kw];  dgrad: flipped and transposed
__global__ __launch_bounds__(256) void k_pack_weights_w(const float* __restrict__ w, float* __restrict__ packed,
                                                        int C, int BNE, int ntile, int nchunk, int dgrad) {
  const size_t total = (size_t)ntile * nchunk * 3 * 4 * BN * KCW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = idx % KCW;
    size_t r = idx / KCW;
    const int col = r % BN; r /= BN;
    const int j = r % 4; r /= 4;
    const int kh = r % 3; r /= 3;
    const int ch = r % nchunk;
    const int nt = r / nchunk;
    const int kidx = ch * KCW + kk, nidx = nt * BNE + col;
    float v = 0.f;
    if (col < BNE && kidx < C && nidx < C) {
      float g[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        g[kw] = dgrad ? w[(((size_t)kidx * (C + 1) + 1 + nidx) * 3 + (2 - kh)) * 3 + (2 - kw)]
                      : w[(((size_t)nidx * (C + 1) + 1 + kidx) * 3 + kh) * 3 + kw];
      v = j == 0 ? g[0] : j == 1 ? 0.5f * (g[0] + g[1] + g[2]) : j == 2 ? 0.5f * (g[0] - g[1] + g[2]) : g[2];
    }
    packed[idx] = v;
  }
}
void launch_pack_weights_w(const Dims& d, const float* w, float* packed, int dgrad, hipStream_t s) {
  const int nchunk = (d.C + KCW - 1) / KCW;
  const size_t total = (size_t)d.ntile * nchunk * 3 * 4 * BN * KCW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_weights_w, dim3(blocks), dim3(256), 0, s, w, packed, d.C, d.BNE, d.ntile, nchunk, dgrad);
}

template <int MT>
static void launch_conv_w_t(const Dims& d, const ConvArgs& a, hipStream_t s) {
  static bool attr[MAX_DEVICES];
  allow_full_lds((const void*)k_conv3x3_w<MT>, attr);
  hipLaunchKernelGGL((k_conv3x3_w<MT>), dim3(d.mtiles, d.ntile), dim3(512), conv_w_lds_bytes(d), s, a, d);
}

template <int WM, int MT>
static void launch_conv_t(const Dims& d, const ConvArgs& a, hipStream_t s) {
  static bool attr[MAX_DEVICES];
  allow_full_lds((const void*)k_conv3x3<WM, MT>, attr);
  hipLaunchKernelGGL((k_conv3x3<WM, MT>), dim3(d.mtiles, d.ntile), dim3(WM * 128), conv_lds_bytes(d, a.mode), s, a, d);
}

// d.BM (chosen by make_dims): 64 = four-wave workgroups, two of which share a CU and cover each other's
// barriers / prologue / epilogue when the grid is small; 128 / 256 = eight waves.
void launch_conv(const Dims& d, const ConvArgs& a, hipStream_t s) {
  if (d.wino == 2) {   // 2-D Winograd (128-pixel tiles, even H and W); weights packed by launch_pack_weights_w2
    static bool attr[MAX_DEVICES];
    allow_full_lds((const void*)k_conv3x3_w2, attr);
    hipLaunchKernelGGL(k_conv3x3_w2, dim3(d.mtiles, d.ntile), dim3(512), conv_w2_lds_bytes(d), s, a, d);
    return;
  }
  if (d.wino) {   // 1-D Winograd along the rows (even W); weights packed by launch_pack_weights_w
    if (d.BM == 64) launch_conv_w_t<1>(d, a, s);
    else if (d.BM == 128) launch_conv_w_t<2>(d, a, s);
    else launch_conv_w_t<4>(d, a, s);
    return;
  }
  if (d.BM == 64) launch_conv_t<2, 1>(d, a, s);
  else if (d.BM == 128) launch_conv_t<4, 1>(d, a, s);   // (four waves x (64 x 32) per wave measured slower: 94 vs 91 us at cfg 2)
  else launch_conv_t<4, 2>(d, a, s);
}


// ============================================================================
// k_conv3x3_small -- the LATENCY regime (evaluate.py:97-142: bs = 1, NFE per image; small inference batches).
// The tiles above are sized for throughput: whole samples x 64 columns per workgroup, so that GroupNorm fuses into
// the epilogue -- at [1, 256, 8, 8] that is a grid of FOUR workgroups and 75 MFLOP on four CUs (>= 36 us per conv
// from the MFMA rate alone; measured 88 us per function evaluation).  Here the work is cut for parallelism
// instead: a workgroup owns 32 pixels x 32 output channels (pixels are the flattened (sample, pixel) index, a
// tile may straddle samples), its four waves split K (a quarter of the input channels x all nine taps each) and
// meet through LDS once; GroupNorm runs as the pointwise pass behind it (k_combine_gn with an empty Butcher row,
// as for split images).  Direct convolution: each lane loads ONE float4 of activations (four consecutive input
// channels of its pixel, taken through a per-tap byte offset that points at the tensor's zero row outside the
// image -- branch-free) and ONE float4 of filter taps (packed [tap][ci/4][co][4] by k_pack_weights_small), and
// both feed four v_mfma_f32_32x32x2_f32 (lane half hi supplies channels +4..7).  No LDS in the loop, no barrier.
// Forward only (inference solves); used for single-digit grids only (Dims::small: measured crossover).
// ============================================================================
__global__ __launch_bounds__(256) void k_conv3x3_small(ConvArgs a, Dims d) {
  if (a.et.ctrl != nullptr && a.et.ctrl->done) return;
  __shared__ float red[4][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int P = d.N * d.HW;
  const int pix = blockIdx.x * 32 + l31;
  const int c0 = blockIdx.y * 32;
  const unsigned zoff = (unsigned)((size_t)P * d.C * sizeof(float));   // the row of C zeros behind the tensor
  unsigned off[9];
  {
    const int pp = pix < P ? pix : 0;
    const int n = pp / d.HW, q = pp - n * d.HW;
    const int h = q / d.W, x = q - h * d.W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = h + t / 3 - 1, xx = x + t % 3 - 1;
      const bool in = pix < P && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
      off[t] = in ? (unsigned)(((size_t)(n * d.HW + yy * d.W + xx) * d.C) * sizeof(float)) : zoff;
    }
  }
  const int C4 = d.C >> 2;
  const int groups = d.C >> 5;                 // 8-channel groups per wave (a quarter of C / 8)
  const int g0 = wave * groups;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const char* abase = reinterpret_cast<const char*>(a.in);
  const float* wbase = a.wpacked + ((size_t)hi * d.C + c0 + l31) * 4;   // + ((tap * C4 + c8 / 4) * C) * 4
#pragma unroll
  for (int t = 0; t < 9; ++t) {   // (fully unrolled: off[] stays in registers)
    const char* ap = abase + off[t] + (size_t)(4 * hi) * sizeof(float);
    const float* wp = wbase + (size_t)t * C4 * d.C * 4;
    for (int g = 0; g < groups; ++g) {
      const int c8 = (g0 + g) * 8;
      const float4 av = *reinterpret_cast<const float4*>(ap + (size_t)c8 * sizeof(float));
      const float4 bv = *reinterpret_cast<const float4*>(wp + (size_t)(c8 >> 2) * d.C * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * hi][l31] = acc[r];
  __syncthreads();
  const float tval = eval_time(a.et);
  for (int e = tid; e < 1024; e += 256) {
    const int r = e >> 5, c = e & 31;
    const int p = blockIdx.x * 32 + r;
    if (p >= P || c0 + c >= d.C) continue;
    const int q = p % d.HW;
    const float v = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
    a.raw_out[(size_t)p * d.C + c0 + c] = v + a.bias[c0 + c] + tval * a.tmap[(size_t)q * d.C + c0 + c];
  }
}

// packed[tap][ci / 4][co][ci % 4] = W[co][1 + ci][kh][kw]   (input channel 0 of the reference's layout is time)
__global__ __launch_bounds__(256) void k_pack_weights_small(const float* __restrict__ w, float* __restrict__ packed, int C) {
  const size_t total = (size_t)9 * C * C;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int kk = (int)(idx & 3);
    size_t r = idx >> 2;
    const int co = (int)(r % C); r /= C;
    const int cq = (int)(r % (C >> 2));
    const int tap = (int)(r / (C >> 2));
    packed[idx] = w[(((size_t)co * (C + 1) + 1 + cq * 4 + kk) * 3 + tap / 3) * 3 + tap % 3];
  }
}
void launch_pack_weights_small(const Dims& d, const float* w, float* packed, hipStream_t s) {
  size_t blocks = ((size_t)9 * d.C * d.C + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_weights_small, dim3((unsigned)blocks), dim3(256), 0, s, w, packed, d.C);
}
void launch_conv_small(const Dims& d, const ConvArgs& a, hipStream_t s) {
  const int P = d.N * d.HW;
  hipLaunchKernelGGL(k_conv3x3_small, dim3((P + 31) / 32, d.C / 32), dim3(256), 0, s, a, d);
}

}  // namespace node
