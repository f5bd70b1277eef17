// Winograd F(4x4, 3x3) as a three-stage pipeline (gfx950 only): the 3x3 convolutions of an 8x8 or 16x16 state whose
// solver tolerance leaves room for the transform's rounding (see Solver::choose_w4, node_api.hip).  A 16x16 image is
// four 8x8 quadrants, each a "virtual sample" of the layouts below (kernels_w4s.hip has the quadrant logic): wherever
// this file says N or "sample" for V, U, M, Z and the GEMM-side kernels, a 16x16 solve passes 4 N.
//
//   V = B^T d B   6x6 input transform of every 4x4 output tile's 6x6 patch -- written by the PRODUCER of the conv
//                 input (the GroupNorm passes, kernels_w4s.hip), not by the conv
//   M_c = V_c U_c 36 independent [rows x C] x [C x C] products, rows = samples x 4 tiles: k_w4_gemm64 (fp32 MFMA,
//                 operands straight from L2 into registers in MFMA-ready blocks, no LDS, no transform in the loop)
//   Y = A^T M A   4x4 output transform -- done by the CONSUMER (the GroupNorm pass behind the conv, kernels_w4s.hip)
//
// 36 multiplies per 16 outputs = 0.25 of the direct convolution's (F(2x2,3x3): 0.444).  Interpolation points
// (0, 1, -1, 1/2, -2, inf): max error 3.2e-6 of max|y| at C = 256 against an fp64 direct convolution (the textbook
// points (0, +-1, +-2, inf): 8.4e-6; F(2x2,3x3): 4.9e-7; tools/wino_error.py).
//
// Layouts (H = W = 8: T = 4 tiles per sample, R = 4 N rows, RB = N / 8 row blocks of 32; G8 = C / 8):
//   V  [comp 36][rb][g G8][s 8][hi 2][t 4][e 4]   channel = 8 g + 4 hi + e, sample = 8 rb + s: one contiguous 1 KB
//      block per (comp, rb, g) = ONE 16-B load per lane for four MFMA k-steps; a producer wave (one sample,
//      16 channels) writes whole 128-B lines
//   U  [comp 36][cb C/32][g G8][hi 2][col 32][e 4]   the same for the filter operand (k_w4_pack, once per solve)
//   M  [n][C/32][comp 36][t 4][c 32]   what one pass workgroup (sample, 32 channels) reads is ONE contiguous 18 KB
//      block; a GEMM store instruction (32 columns x rows r, r + 4 of an accumulator) writes two whole 128-B lines
#pragma once
#include "node_internal.h"

namespace node {

constexpr int W4_COMPS = 36;

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
// ----------------------------------------------------------------------------
// fp16-PAIR operand format of the component GEMMs (round 6; k_w4_gemm64h, k_w4_wgrad64h).
// An fp32 value x times a power-of-two scale s is held as TWO fp16 numbers, h = fp16(s x) and l = fp16(s x - h): 22 significand
// bits in the 4 bytes an fp32 takes, and a product of two such operands is THREE v_mfma_f32_32x32x16_f16 (hl, lh, hh; the
// dropped ll is 2^-22 of the result) with NO vector arithmetic in the loop -- against six bf16 MFMAs plus the in-register
// three-way split of the bf16-triple form (wino4.h above, k_w4_gemm64b).  fp16's narrow exponent is what the scales are for:
// an element of magnitude >= 2^-3 after scaling keeps all 22 bits, a smaller one an ABSOLUTE error of 2^-25 (subnormal l; MFMA
// and the conversions keep fp16 subnormals), 65504 is the ceiling.  tools/f16pair_error.py: at cfg 2's conv shape the convolution
// error against fp64 is that of the fp32 chain (3.1e-6 of max|y|) for any scale that puts the operand's maximum in [2^3, 2^16).
//   V pairs  [comp 36][rb][g2 C/16][part 2 (h, l)][s 8][hi 2][t 4][8 halves: (gp 2, e 4)]   channel = 16 g2 + 8 gp + 4 hi + e:
//            the bytes of the fp32 layout's two g blocks; a GEMM lane's two 16-B loads (h, l: 1 KB apart) are the A operands
//            of one K = 16 step; a producer lane swaps with its neighbour (one DPP) and stores one dword {h, h'} or {l, l'}
//   U pairs  [comp 36][cb C/32][g2][part 2][hi 2][col 32][8 halves]                          (k_w4_pack, once per solve)
// Scales (W4Scales, device memory): powers of two, held as exponents.  Filters: from max|w| (|U| <= 3.49 max|w|).  Forward row
// operands: from the GroupNorm in front -- |relu(gamma xhat + beta)| <= sqrt(m - 1) max|gamma| + max|beta| over a group of m
// values, |B^T d B| <= 49 max|d| -- so no data-dependent quantity is needed and no overflow is possible.
// ----------------------------------------------------------------------------
// exponent e with bound * 2^e <= 2^top (bound > 0, finite), clamped to +-60; bound == 0 -> 0
__host__ __device__ inline int w4_scale_exp(float bound, int top) {
  if (!(bound > 0.f)) return 0;
  int ex;
  (void)frexpf(bound, &ex);          // bound = m 2^ex, m in [0.5, 1)  =>  bound <= 2^ex
  int e = top - ex;
  return e > 60 ? 60 : e < -60 ? -60 : e;
}

// COTANGENT-side operands (the data gradients' row operands and Z = A dz A^T of the weight gradient) have no bound a priori:
// their scale 2^e[W4_E_G] follows the data.  Every pass that forms such a tensor adds max|dz| of its waves to gmax[] (atomicMax
// on the bit pattern, one slot per wave); the step controller (k_step_controller) and k_w4_gscale turn the maximum of the finished step
// into the next step's exponent, max|dz| 2^e in (2^4, 2^5]: |B^T dz B| <= 49 |dz| and |A dz A^T| <= 225 |dz| then stay under
// 2^13 -- a cotangent may grow EIGHT-fold from one step to the next before a value could pass fp16's 65504.  A pass that sees
// max|dz| 2^e > 2^8 raises `ovf`: the controller then REPEATS the step (nothing accepted, dt unchanged, not counted as a step of
// the solver) at the exponent the recorded maximum asks for.  The first evaluations of an interval (f0, before any maximum is
// known) run the bf16-triple kernels on fp32 operands and only record.
struct W4Scales {
  int e[8];            // scale = 2^e: [0] conv1 filters, [1] conv2 filters, [2] conv1's forward row operand, [3] conv2's, [4] cotangents
  unsigned mx[8];      // fp32 bit patterns of the maxima the exponents are derived from: |w1|, |w2|, |gamma1|, |beta1|, |gamma2|, |beta2|
  unsigned arrived;    // k_w4_scales: blocks that have added their maxima
  unsigned ovf;        // a cotangent-side value left the range of its scale in the current step
  int n_retry;         // steps repeated for that reason (diagnostics)
  unsigned pad[13];
  unsigned gmax[2048]; // max|dz| of the current step's passes, fp32 bit patterns (slot = wave of the pass % 2048: a slot is hit by one wave
                       // per pass at cfg 2 -- 32 waves per slot serialised their atomics for ~2.5 us behind every pass that records)
};
constexpr int W4_GSLOTS = 2048;
constexpr int W4_G_TOP = 5;          // max|dz| 2^e <= 2^5
constexpr float W4_G_LIMIT = 256.f;  // ... and a pass that meets more than 2^8 asks for the step to be repeated
// A whole workgroup of 256 threads (`red`: 4 floats of LDS): the exponent of the cotangent-side operands from the recorded maxima
// (then cleared); every thread returns whether a pass overflowed.
__device__ inline bool w4_gscale_update(W4Scales* sc, int tid, float* red) {
  const bool ovf = __hip_atomic_load(&sc->ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  float g = 0.f;
#pragma unroll
  for (int i = 0; i < W4_GSLOTS / 256; ++i) {
    g = fmaxf(g, __builtin_bit_cast(float, __hip_atomic_load(&sc->gmax[i * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    __hip_atomic_store(&sc->gmax[i * 256 + tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int o = 32; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = g;
  __syncthreads();                    // (also: every thread has read `ovf` before thread 0 clears it)
  if (tid == 0) {
    g = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __hip_atomic_store(&sc->ovf, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g > 0.f && g < INFINITY) sc->e[4] = w4_scale_exp(g, W4_G_TOP);
    if (ovf) sc->n_retry += 1;
  }
  return ovf;
}
void launch_w4_gscale(W4Scales* sc, hipStream_t s, int skew = 0);   // skew (diagnostics): the exponent lands `skew` too high      // the same as a one-workgroup launch (behind an interval's first evaluation)
constexpr int W4_E_U1 = 0, W4_E_U2 = 1, W4_E_V1 = 2, W4_E_V2 = 3, W4_E_G = 4;
struct W4ScaleJobs {
  const float* w[2];        // conv weights [C][CI][3][3] (w[1] nullable)
  size_t wn;                // elements of each
  const float* gb[4];       // gamma1, beta1, gamma2, beta2 [C] (nullable: exponents 2, 3 are then left alone)
  size_t vn[2];             // elements of gb[1] / gb[3] when they are not [C] vectors (diagnostics: a whole tensor as "beta"); 0 = C
  int C, gn_m;              // channels; values per GroupNorm group (cpg x pixels)
  W4Scales* sc;
  int bigjob[4];            // (filled by launch_w4_scales)
};
// one launch: maxima by atomics (sc->mx, sc->arrived must be zero), the last block derives the exponents and zeroes them again
void launch_w4_scales(const W4ScaleJobs& j, hipStream_t s);
constexpr int W4_SLACK = 16 * 256;         // floats behind V and U that k_w4_gemm's operand ring may read (never uses)
__host__ __device__ inline size_t w4_v_elems(int N, int C) { return (size_t)W4_COMPS * 4 * N * C + W4_SLACK; }
__host__ __device__ inline size_t w4_u_elems(int C) { return (size_t)W4_COMPS * C * C + W4_SLACK; }

// ----------------------------------------------------------------------------
// The GroupNorm passes around the component GEMMs (kernels_w4s.hip): wave-independent, register-resident.
// ----------------------------------------------------------------------------
struct W4sHead {          // the conv result in front of this pass
  const float* M;         // component products [N][C/32][36][4][32]
  const float* bias;      // forward: conv bias [C]
  const float* tmapS;     // forward: border-aware time-channel map in the W4S blocking (launch_w4s_tmap)
  EvalTime et;            // forward: time of the evaluation (tmap multiplier)
  const float* gamma;     // GroupNorm weight of this pass [C]
  const float* beta;      // forward: GroupNorm bias; backward: bias of the layer's ReLU mask fma(xhat, gamma, beta) > 0
  float osign;            // output multiplier
  int relu;               // forward: ReLU behind the affine
  float* out_s;           // W4S output, nullable (forward GroupNorm-3: k; backward GroupNorm-1: k_a)
  float* out_nhwc;        // NHWC output, nullable (forward GroupNorm-2: act2; backward GroupNorm-2: dz1): the weight gradient's operand
  float* xhat_s;          // forward: xhat out (W4S, nullable); backward: xhat in
  float* rstd;            // forward: 1/sigma out [N][G] (nullable); backward: in
  float* gpart;           // backward: [N][2][C] per-sample (dgamma, dbeta) partials
  float* spart;           // backward, nullable: [N][9][C] masked column sums of the output (masked_colsum_tile)
  float* z_out;           // backward, nullable: Z = A out A^T for the F(4x4,3x3)-domain weight gradient (dz1)
};
struct W4sTail {          // what follows the head inside the same launch
  Comb comb;              // stage combine (tail 1) / adjoint combine (tail 2), W4S tensors
  int self;               // tail 1 behind a forward head: comb.k[comb.nk - 1] IS the head's output (taken from registers)
  float csign;            // tail 2: cotangent = csign * combine
  float* y_out;           // the combined state, nullable (W4S: y1 / a1)
  const float* gamma;     // tail 1: GroupNorm-1 weight, bias
  const float* beta;
  float* act_nhwc;        // tail 1: act1 (nullable); tail 2: dz2 -- NHWC, the weight gradient's operands
  float* xhat_s;          // tail 1: xhat-1 out (W4S, nullable)
  float* rstd;            // tail 1: 1/sigma-1 out (nullable)
  float* gpart;           // tail 2: GroupNorm-3's (dgamma, dbeta) partials
  float* spart;           // tail 2: masked column sums of dz2
  float* z_out;           // tail 2, nullable: Z = A dz2 A^T for the F(4x4,3x3)-domain weight gradient
};
struct W4sArgs {
  const Ctrl* ctrl;       // nullable: the launch returns at once when ctrl->done
  int N, C, cpg;
  int Q;                  // 1: 8x8 images; 4: 16x16 images, four quadrant waves per (sample, channel block)
  int Nv;                 // Q N rounded up to the component GEMMs' row block (8 samples): strides of V, Z (rows of the padding
                          // samples are never written and never read by anything that leaves the GEMM's own rows)
  float eps;
  W4sHead h;
  W4sTail t;
  float* V;               // nullable: blocked input transform of the tensor this pass hands to the next conv
  const int* v_exp;       // non-null: V leaves as fp16 pairs at scale 2^*v_exp (W4Scales::e; same bytes, layout "V pairs" above)
  const int* z_exp;       // non-null: z_out (head or tail) leaves as fp16 pairs in V's layout at scale 2^*z_exp
  W4Scales* gstat;        // non-null: this pass forms a cotangent (dz1 / dz2): record max|dz|, raise ovf past the scale's range
};
// head: 0 none / 1 forward / 2 backward; tail: 0 none / 1 stage combine + GroupNorm-1 + ReLU / 2 adjoint combine + GroupNorm-3 backward
void launch_w4s_pass(int head, int tail, const W4sArgs& a, hipStream_t s);
void launch_w4s_from_nchw(const float* src_nchw, float* dst_w4s, int N, int C, int Q, hipStream_t s);
void launch_w4s_to_nchw(const float* src_w4s, float* dst_nchw, int N, int C, int Q, hipStream_t s);
void launch_w4s_tmap(const float* tmap0, const float* tmap1, float* out0, float* out1, int C, int Q, hipStream_t s);
void launch_w4s_emit_outputs(const Dims& d, const EmitArgs& a, hipStream_t s);

// F(4x4,3x3)-domain weight gradient of both conv layers (k_w4_wgrad, kernels_w4.hip).
//   Z  [comp 36][co/32][sample N][co%32][tile 4]   Z = A dz A^T of the conv output's cotangent: a lane's 16 B are four
//      rows of the reduction for one output channel; what a wave of the producing pass writes per component is 256
//      contiguous bytes
//   dU [layer 2][comp 36][ci][co]                  every element written once (no split-K slabs); dW = G^T dU G in
//      k_theta_finalize
struct W4WgradArgs {
  const float* V1; const float* Z1;   // conv1: V of act1, Z of dz1
  const float* V2; const float* Z2;   // conv2: V of act2, Z of dz2 (nullable: ONE layer -- the stem's -- half the grid)
  float* dU;
  const Ctrl* ctrl;                   // nullable: the launch returns at once when ctrl->done
  int N, C;
  int sharev;                         // set by launch_w4_wgrad (NODE_TUNE_W4_SHAREV): the waves of a workgroup share one component's V stream
};
void launch_w4_wgrad(const W4WgradArgs& a, hipStream_t s);
// fp16-pair operands (k_w4_wgrad64h): V pairs and Z pairs in V's layout; dU as above (unscaled in the epilogue)
bool w4_wgrad_f16_fits(int N, int C);
void launch_w4_wgrad_f16(const unsigned* V1, const unsigned* Z1, const unsigned* V2, const unsigned* Z2, float* dU, const Ctrl* ctrl, int N, int C,
                         const int* v1_exp, const int* v2_exp, const int* z_exp, hipStream_t s);
void w4_refresh_tuning();     // re-read the NODE_TUNE_W4_* switches (once per C-ABI call; kernels_w4.hip)
__host__ __device__ inline size_t w4_z_elems(int N, int C) { return (size_t)W4_COMPS * 4 * N * C; }
__host__ __device__ inline size_t w4_du_elems(int C) { return (size_t)2 * W4_COMPS * C * C; }

// launchers (kernels_w4.hip)
// smallest rtol / atol of an adaptive solve that takes the pipeline (Solver::choose_w4 has the error budget); a hair under
// 1e-5 so that a tolerance that went through a float keeps comparing equal
constexpr float W4_MIN_TOL = 0.99e-5f;
// per job ONE of u (fp32, k_w4_gemm / k_w4_gemm64) and ub (exact bf16 triples, k_w4_gemm64b) is written: ub when non-null
// plain[i] != 0: a [C][C][3][3] filter (the stem's conv3x3, model.py:255-258) instead of ConcatConv2d's [C][C + 1][3][3]
// uh[i] non-null: the fp16-pair form ("U pairs" above) at scale 2^*uh_exp[i] instead (k_w4_gemm64h)
struct W4PackJobs { const float* w[4]; float* u[4]; unsigned short* ub[4]; int dgrad[4]; int plain[4]; unsigned* uh[4]; const int* uh_exp[4]; };
// which form launch_w4_gemm will read for this batch (NODE_TUNE_W4_BF16X3, read on every call)
bool w4_uses_bf16(int N, int C);
__host__ __device__ inline size_t w4_ub_elems(int C) { return (size_t)W4_COMPS * C * C * 3 + 8 * 1536; }   // bf16 values (+ ring slack)
void launch_w4_pack(const W4PackJobs& jobs, int count, int C, hipStream_t s);
void launch_w4_gemm(const float* V, const float* U, float* M, const Ctrl* ctrl, int N, int C, hipStream_t s,
                    const unsigned short* Ub = nullptr);
// fp16-pair operands (V pairs at 2^*v_exp, U pairs at 2^*u_exp): M = the same fp32 products, unscaled in the epilogue
bool w4_f16_fits(int N, int C);     // N % 16 == 0, C % 64 == 0, NODE_TUNE_W4_F16 != 0
void launch_w4_gemm_f16(const unsigned* Vh, const unsigned* Uh, float* M, const Ctrl* ctrl, int N, int C, const int* v_exp, const int* u_exp,
                        hipStream_t s);
// diagnostics (node_w4_split3): out[3 i .. 3 i + 2] = the three bf16 parts of x[i] as the GEMM kernels split it; n % 8 == 0
void launch_w4_split_check(const float* x, float* out, size_t n, hipStream_t s);
// stand-alone transforms (diagnostics: node_conv3x3_w4; kernels_w4s.hip): W4S tensor -> V, M -> W4S tensor
void launch_w4_input(const float* x_w4s, float* V, int N, int C, int Q, int Nv, hipStream_t s, const int* v_exp = nullptr);
void launch_w4_output(const float* M, float* y_w4s, int N, int C, int Q, hipStream_t s);
// the stem's 8x8 filters -> filters convolution through the pipeline (kernels_w4s.hip, stem_api.hip): transforms that read / write
// the stem's NHWC tensors (and the caller's NCHW tensors at the stem's boundary) directly
void launch_w4s_stem_in(const float* h_nhwc, const float* gamma, const float* beta, float eps, int cpg, float* xhat_s, float* rstd, float* V,
                        int N, int C, int Nv, hipStream_t s);
void launch_w4s_stem_out(const float* M, const float* res_nhwc, float* out_nchw, int N, int C, hipStream_t s);
void launch_w4s_stem_gin(const float* g_nchw, float* V, float* Z, int N, int C, int Nv, hipStream_t s);
void launch_w4_du_to_dw(const float* dU, float* dW, int C, hipStream_t s);

}  // namespace node
