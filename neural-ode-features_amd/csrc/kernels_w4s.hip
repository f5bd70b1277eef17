// Winograd F(4x4,3x3) pipeline, stages 1 and 3: the GroupNorm passes around the component GEMMs (wino4.h), as
// WAVE-INDEPENDENT kernels: no LDS, no barrier, every value of a (sample, channel) stays in registers from the
// conv's component products to the next conv's row operand.  gfx950 (MI355X / CDNA4) only.
//
//   lane = (tile t of the 8x8 image, channel c15 of a 16-channel block): 64 lanes = 4 tiles x 16 channels.  A thread
//   owns the 4x4 pixels of its tile for its channel:
//     * output transform Y = A^T M A: 36 loads of its own (tile, channel) element, all arithmetic in registers;
//     * GroupNorm statistics: a group (cpg <= 16 consecutive channels x 4 tiles) lies inside ONE wave -- DPP + two
//       cross-row shuffles, no workgroup reduction;
//     * the 6x6 patch of the input transform V = B^T d B needs one pixel ring around the tile: an 8x8 image is 2x2
//       tiles, so the ring is nine values held by the lanes t^1, t^2, t^3 of the same wave -- nine shuffles;
//     * Butcher combines, adjoint combines and every solver-state tensor are read / written as 16-B vectors in the
//       tile-blocked state layout (W4S, below), so a wave instruction moves 1 KB of contiguous memory.
//
// The passes of one dynamics evaluation (model.py:339-348) and of its VJP, with what round 2 launched separately
// merged where the data is local to a (sample, channel) anyway:
//   C     combine (Butcher row) -> GroupNorm-1 -> ReLU -> V                                         <0,1>
//   P2    M -> +bias + t*tmap -> GroupNorm-2 -> ReLU -> V                                           <1,0>
//   P3    M -> +bias + t*tmap -> GroupNorm-3 -> k                                                   <1,0>
//   P3C   P3, then the NEXT stage's C with k taken from registers (forward solves)                  <1,1>
//   P3B3  P3, then the top of the backward chain: adjoint combine -> GroupNorm-3 backward -> V      <1,2>
//         (xhat-3 and 1/sigma-3 never leave the registers)
//   PB2   M (data gradient) -> ReLU mask -> GroupNorm-2 backward -> dz1 -> V                        <2,0>
//   PB1   M (data gradient) -> ReLU mask -> GroupNorm-1 backward -> k_a                             <2,0>
//   PB1C  PB1, then the NEXT stage's C (augmented solves)                                           <2,1>
// The ReLU mask is recomputed from the saved xhat (fma(xhat, gamma, beta) > 0, the producer's own expression), so the
// backward passes do not read the activations.
//
// State layout W4S (every solver-state tensor of an F(4x4,3x3) solve: Y, Y1, KY[], A, A1, KA[], the saved xhat):
//   float4 index (((n * C/16 + cb) * 4 + i) * 64 + lane), components = the four pixels of tile row i
//   (pixel (4 ty + i, 4 tx + j), t = 2 ty + tx, channel 16 cb + c15, lane = 16 t + c15).
// Element-wise kernels (error norm, commit, axpy, ...) do not care; NCHW <-> W4S happens at the solve boundary.
// M (component products) is [n][C/32][36][4 t][32 c]: what a wave pair reads is one contiguous 18 KB block.
//
// 16x16 images (Q = 4; cfg 5's [n, 1024, 16, 16] and the one-shot stem's [n, 256, 16, 16] states): the image is cut into
// four 8x8 QUADRANTS q = 2 QY + QX, and quadrant q of sample n is "virtual sample" nv = 4 n + q of every layout above
// (W4S, V, M, Z) -- the component GEMMs and the weight gradient see 4 N samples and do not change.  What changes is here:
// a workgroup holds the four quadrant waves of a (sample, GroupNorm group set) -- 4 waves for cpg <= 16, 8 for cpg = 32
// -- the GroupNorm sums cross the waves through LDS (one barrier per sum, two alternating slots), and the pixel ring of
// the input transform comes out of an LDS copy of the workgroup's tiles instead of nine shuffles.
#include "wino4.h"
#include <cstring>

namespace node {

#ifndef W4S_THREADS_DEF
#define W4S_THREADS_DEF 128
#endif
constexpr int W4S_THREADS = W4S_THREADS_DEF;   // two waves = 32 channels of one sample: whole 128-B lines of M, V and the NHWC copies

// V and Z -- written once here, read once by the next GEMM -- leave write-through (see st_wt in kernels_w4.hip)
#ifndef NODE_WT_STORES
#define NODE_WT_STORES 1
#endif
__device__ __forceinline__ void w4s_st_wt(float* p, float v) {
#if NODE_WT_STORES
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

template <int CTRL>
__device__ __forceinline__ float w4s_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the four tiles of a channel (lanes l, l^16, l^32, l^48); every lane gets the result
__device__ __forceinline__ float w4s_tile_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// sum over a GroupNorm group: cpg (1, 2, 4, 8, 16) consecutive channels x the four tiles; every lane gets the result
__device__ __forceinline__ float w4s_group_sum(float v, int cpg) {
  if (cpg >= 2) v += w4s_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  if (cpg >= 4) v += w4s_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  if (cpg >= 8) v += w4s_dpp<0x141>(v);   // row_half_mirror
  if (cpg >= 16) v += w4s_dpp<0x140>(v);  // row_mirror
  return w4s_tile_sum(v);
}

// Where a wave sits.  Q = 1: the wave holds the whole 8x8 image of its 16 channels.  Q = 4: one 8x8 quadrant of a 16x16 image.
template <int Q>
struct W4Wave {
  int lane, t, c15, ty, tx;   // lane = 16 t + c15, tile t = 2 ty + tx of the quadrant
  int n, q, nv, cb, c;        // sample, quadrant, virtual sample Q n + q, 16-channel block, channel 16 cb + c15
  int TY, TX, TM;             // tile coordinates in the image, their maximum (1 or 3)
  int w, nw;                  // Q = 4: wave of the workgroup (w = 4 cbi + q), waves per workgroup (4 or 8)
  float* red;                 // Q = 4: LDS [2 slots][8 waves][16 channels][2]
  float* tiles;               // Q = 4: LDS [8 waves][16 pixels][64 lanes]
  int slot;
};
constexpr int W4Q_RED = 2 * 8 * 16 * 2;
constexpr int W4Q_TILES = 8 * 16 * 64;

// sum over a GroupNorm group of the IMAGE; every lane gets the result.  Q = 4: the in-wave sums of the workgroup's waves
// (all of them hold the lane's group: cpg <= 16 -> the four quadrants of one channel block, cpg = 32 -> two blocks) meet
// in LDS and every wave adds them in the same order
template <int Q>
__device__ __forceinline__ float w4s_gsum(float v, int cpg, W4Wave<Q>& wv) {
  v = w4s_group_sum(v, cpg);
  if (Q == 4) {
    float* r = wv.red + (wv.slot & 1) * (W4Q_RED / 2);
    wv.slot++;
    if (wv.t == 0) r[(wv.w * 16 + wv.c15) * 2] = v;
    __syncthreads();
    float s = r[wv.c15 * 2];
    for (int k = 1; k < wv.nw; ++k) s += r[(k * 16 + wv.c15) * 2];
    v = s;
  }
  return v;
}
template <int Q>
__device__ __forceinline__ void w4s_gsum2(float& a, float& b, int cpg, W4Wave<Q>& wv) {
  a = w4s_group_sum(a, cpg);
  b = w4s_group_sum(b, cpg);
  if (Q == 4) {
    float* r = wv.red + (wv.slot & 1) * (W4Q_RED / 2);
    wv.slot++;
    if (wv.t == 0) { r[(wv.w * 16 + wv.c15) * 2] = a; r[(wv.w * 16 + wv.c15) * 2 + 1] = b; }
    __syncthreads();
    float sa = r[wv.c15 * 2], sb = r[wv.c15 * 2 + 1];
    for (int k = 1; k < wv.nw; ++k) { sa += r[(k * 16 + wv.c15) * 2]; sb += r[(k * 16 + wv.c15) * 2 + 1]; }
    a = sa; b = sb;
  }
}

// A^T x for the points (0, 1, -1, 1/2, -2, inf)   (W4_AT)
__device__ __forceinline__ void w4s_at6(float m0, float m1, float m2, float m3, float m4, float m5, float& o0, float& o1,
                                        float& o2, float& o3) {
  const float a = m1 + m2, b = m1 - m2;
  o0 = (m0 + a) + (m3 + m4);
  o1 = b + (0.5f * m3 - 2.f * m4);
  o2 = a + (0.25f * m3 + 4.f * m4);
  o3 = (b + m5) + (0.125f * m3 - 8.f * m4);
}
// B^T x   (W4_BT)
__device__ __forceinline__ void w4s_bt6(float d0, float d1, float d2, float d3, float d4, float d5, float& r0, float& r1,
                                        float& r2, float& r3, float& r4, float& r5) {
  const float p = d3 - d1, q = d4 - d2;
  r0 = (d0 + (d4 - 2.f * d2)) + 1.5f * p;
  r1 = (d4 - d1) + (0.5f * d2 + 2.5f * d3);
  r2 = (d4 + d1) + (0.5f * d3 - 2.5f * d2);
  r3 = q + 2.f * p;
  r4 = q - 0.5f * p;
  r5 = (d5 + (d1 - 2.f * d3)) + 1.5f * q;
}

// Y = A^T M A of the thread's (tile, channel): `mp` points at its element of component 0, components 128 floats apart
__device__ __forceinline__ void w4s_out_transform(const float* __restrict__ mp, float y[4][4]) {
  float m[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) m[q] = mp[q * 128];
  float z[4][6];
#pragma unroll
  for (int nu = 0; nu < 6; ++nu)
    w4s_at6(m[nu], m[6 + nu], m[12 + nu], m[18 + nu], m[24 + nu], m[30 + nu], z[0][nu], z[1][nu], z[2][nu], z[3][nu]);
#pragma unroll
  for (int i = 0; i < 4; ++i) w4s_at6(z[i][0], z[i][1], z[i][2], z[i][3], z[i][4], z[i][5], y[i][0], y[i][1], y[i][2], y[i][3]);
}

// (Measured and rejected, round 5 -- review item 3, "16-byte stores for V": the 36 component loads of M and the 36 component stores of
//  V as NINE 16-byte accesses per lane, a 4 x 4 transpose between the four lanes of a quad (two DPP butterfly stages) behind / in front
//  of each -- same bytes, a quarter of the vector-memory instructions, +288 VALU per thread; parity green, passes 1.58 -> 1.63 ms
//  per step, 24 390 -> 24 240 images/s (profiles/r05_pass_vec_ab.txt, commits 69ea3c6 + 5b5dc33).  The counters of the same
//  profile say why nothing was to win there: the texture path is 25 % busy, the waves wait 48 % of their life -- the launch is one
//  generation of waves (2 048 of 8 192 slots) that all load, then all compute, then all store; per launch it moves its bytes at
//  ~90 % of what a streaming copy of the same size reaches once the ~1.5 us of launch ramp are taken out.)
// V = B^T d B of the 6x6 patch d (the thread's tile + one pixel ring) -> the blocked row operand of the component GEMMs.
// `vp`: the thread's element of component 0; components `cstride` apart.
// fp16-pair form (wino4.h, "V pairs"; pscale != 0): the value times pscale as h = fp16(x), l = fp16(x - h); the lane exchanges with
// its channel neighbour (lane ^ 1) and stores {h, h'} (even channel -> part h) or {l', l} (odd -> part l): one dword per lane and
// component, as in the fp32 form, whole 128-B lines per wave and part
// (two values at a time: v_cvt_pk_f16_f32 converts a pair, ONE DPP exchange moves the dword the neighbour needs -- {h, h} of the odd
//  lane's two values to the even lane, {l, l} the other way -- and one v_perm_b32 per value with a lane-dependent selector puts the
//  halves in channel order: ~5.5 instead of ~10 vector instructions per value on passes that wait for them before they store)
typedef _Float16 w4s_h2 __attribute__((ext_vector_type(2)));
typedef float w4s_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void w4s_pair_words2(float& x0, float& x1, float pscale, bool odd) {
  const w4s_f2 xs = {x0 * pscale, x1 * pscale};
  const w4s_h2 h = __builtin_convertvector(xs, w4s_h2);
  const w4s_f2 hf = __builtin_convertvector(h, w4s_f2);
  const w4s_f2 r = {xs.x - hf.x, xs.y - hf.y};
  const w4s_h2 l = __builtin_convertvector(r, w4s_h2);
  const unsigned H = __builtin_bit_cast(unsigned, h), L = __builtin_bit_cast(unsigned, l);
  const unsigned keep = odd ? L : H, send = odd ? H : L;
  const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]: lane ^ 1
  // v_perm_b32: bytes 0-3 = `keep`, 4-7 = `got`; even lane {keep.lo, got.lo} / {keep.hi, got.hi}, odd lane {got.lo, keep.lo} / {got.hi, keep.hi}
  const unsigned sel0 = odd ? 0x01000504u : 0x05040100u, sel1 = odd ? 0x03020706u : 0x07060302u;
  x0 = __builtin_bit_cast(float, __builtin_amdgcn_perm(got, keep, sel0));
  x1 = __builtin_bit_cast(float, __builtin_amdgcn_perm(got, keep, sel1));
}
__device__ __forceinline__ void w4s_store_v(const float d[6][6], float* __restrict__ vp, size_t cstride, float pscale = 0.f, bool odd = false) {
  float w[6][6];   // w[j][l] = sum_k B^T[l][k] d[j][k]
#pragma unroll
  for (int j = 0; j < 6; ++j) w4s_bt6(d[j][0], d[j][1], d[j][2], d[j][3], d[j][4], d[j][5], w[j][0], w[j][1], w[j][2], w[j][3], w[j][4], w[j][5]);
#pragma unroll
  for (int l = 0; l < 6; ++l) {
    float v0, v1, v2, v3, v4, v5;   // V[xi][l] = sum_j B^T[xi][j] w[j][l]
    w4s_bt6(w[0][l], w[1][l], w[2][l], w[3][l], w[4][l], w[5][l], v0, v1, v2, v3, v4, v5);
    if (pscale != 0.f) {
      w4s_pair_words2(v0, v1, pscale, odd);
      w4s_pair_words2(v2, v3, pscale, odd);
      w4s_pair_words2(v4, v5, pscale, odd);
    }
    w4s_st_wt(vp + (size_t)(0 * 6 + l) * cstride, v0);
    w4s_st_wt(vp + (size_t)(1 * 6 + l) * cstride, v1);
    w4s_st_wt(vp + (size_t)(2 * 6 + l) * cstride, v2);
    w4s_st_wt(vp + (size_t)(3 * 6 + l) * cstride, v3);
    w4s_st_wt(vp + (size_t)(4 * 6 + l) * cstride, v4);
    w4s_st_wt(vp + (size_t)(5 * 6 + l) * cstride, v5);
  }
}
// 8x8 image: the ring is nine values held by the three other tiles' lanes of the same wave (zero outside the image)
__device__ __forceinline__ void w4s_emit_v(const float a[4][4], int ty, int tx, float* __restrict__ vp, size_t cstride, float pscale = 0.f,
                                           bool odd = false) {
  float sh[4], sv[4], rh[4], rv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sh[i] = tx ? a[i][0] : a[i][3];   // my edge column facing the horizontal neighbour
#pragma unroll
  for (int j = 0; j < 4; ++j) sv[j] = ty ? a[0][j] : a[3][j];   // my edge row facing the vertical neighbour
  const float sc = ty ? (tx ? a[0][0] : a[0][3]) : (tx ? a[3][0] : a[3][3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) rh[i] = __shfl_xor(sh[i], 16, 64);
#pragma unroll
  for (int j = 0; j < 4; ++j) rv[j] = __shfl_xor(sv[j], 32, 64);
  const float rc = __shfl_xor(sc, 48, 64);
  float d[6][6];   // patch row r <-> image row 4 ty - 1 + r, column k <-> image column 4 tx - 1 + k
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[1 + i][1 + j] = a[i][j];
    d[1 + i][0] = tx ? rh[i] : 0.f;
    d[1 + i][5] = tx ? 0.f : rh[i];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d[0][1 + j] = ty ? rv[j] : 0.f;
    d[5][1 + j] = ty ? 0.f : rv[j];
  }
  d[0][0] = (ty && tx) ? rc : 0.f;
  d[0][5] = (ty && !tx) ? rc : 0.f;
  d[5][0] = (!ty && tx) ? rc : 0.f;
  d[5][5] = (!ty && !tx) ? rc : 0.f;
  w4s_store_v(d, vp, cstride, pscale, odd);
}
// 16x16 image: the workgroup's tiles go through LDS ([wave][pixel][lane]: conflict-free both ways), the ring's twenty
// values come from up to eight neighbour tiles -- other lanes of this wave or of the other quadrants' waves
__device__ __forceinline__ void w4s_emit_v16(const float a[4][4], const W4Wave<4>& wv, float* __restrict__ vp, size_t cstride, float pscale = 0.f,
                                             bool odd = false) {
  float* mine = wv.tiles + (size_t)wv.w * 1024 + wv.lane;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) mine[(i * 4 + j) * 64] = a[i][j];
  __syncthreads();
  const int cb4 = (wv.w >> 2) << 2;
  // neighbour tile (TY + dy, TX + dx): LDS offset of its pixel 0 for this lane's channel, or -1 outside the image
  int nb[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int Yn = wv.TY + dy, Xn = wv.TX + dx;
      const bool in = (unsigned)Yn < 4u && (unsigned)Xn < 4u;
      const int wq = cb4 + ((Yn >> 1) << 1) + (Xn >> 1), tt = ((Yn & 1) << 1) + (Xn & 1);
      nb[dy + 1][dx + 1] = in ? wq * 1024 + tt * 16 + wv.c15 : -1;
    }
  float d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (r >= 1 && r <= 4 && k >= 1 && k <= 4) { d[r][k] = a[r - 1][k - 1]; continue; }
      const int dy = r == 0 ? 0 : r == 5 ? 2 : 1, dx = k == 0 ? 0 : k == 5 ? 2 : 1;
      const int ii = r == 0 ? 3 : r == 5 ? 0 : r - 1, jj = k == 0 ? 3 : k == 5 ? 0 : k - 1;
      const int base = nb[dy][dx];
      d[r][k] = base >= 0 ? wv.tiles[base + (ii * 4 + jj) * 64] : 0.f;
    }
  w4s_store_v(d, vp, cstride, pscale, odd);
}

// Z = A dz A^T of the thread's tile (no halo: the transform of a conv OUTPUT's cotangent) for the F(4x4,3x3)-domain
// weight gradient (k_w4_wgrad): [comp][co/32][sample][co%32][tile], i.e. 256 contiguous bytes per wave and component.
__device__ __forceinline__ void w4s_a6(float d0, float d1, float d2, float d3, float& o0, float& o1, float& o2, float& o3,
                                       float& o4, float& o5) {
  const float s02 = d0 + d2, s13 = d1 + d3;
  o0 = d0;
  o1 = s02 + s13;
  o2 = s02 - s13;
  o3 = (d0 + 0.25f * d2) + (0.5f * d1 + 0.125f * d3);
  o4 = (d0 + 4.f * d2) - (2.f * d1 + 8.f * d3);
  o5 = d3;
}
__device__ __forceinline__ void w4s_emit_z(const float a[4][4], int n, int N, int C, int c, int t, float* __restrict__ Z) {
  float w[4][6];   // w[i][nu] = sum_j a[i][j] A^T[j][nu]
#pragma unroll
  for (int i = 0; i < 4; ++i) w4s_a6(a[i][0], a[i][1], a[i][2], a[i][3], w[i][0], w[i][1], w[i][2], w[i][3], w[i][4], w[i][5]);
  float* zp = Z + ((size_t)(c >> 5) * N + n) * 128 + (c & 31) * 4 + t;
  const size_t cs = (size_t)4 * N * C;
#pragma unroll
  for (int nu = 0; nu < 6; ++nu) {
    float z0, z1, z2, z3, z4, z5;   // Z[xi][nu] = sum_i A^T[i][xi] w[i][nu]
    w4s_a6(w[0][nu], w[1][nu], w[2][nu], w[3][nu], z0, z1, z2, z3, z4, z5);
    w4s_st_wt(zp + (size_t)(0 * 6 + nu) * cs, z0);
    w4s_st_wt(zp + (size_t)(1 * 6 + nu) * cs, z1);
    w4s_st_wt(zp + (size_t)(2 * 6 + nu) * cs, z2);
    w4s_st_wt(zp + (size_t)(3 * 6 + nu) * cs, z3);
    w4s_st_wt(zp + (size_t)(4 * 6 + nu) * cs, z4);
    w4s_st_wt(zp + (size_t)(5 * 6 + nu) * cs, z5);
  }
}

// ... as fp16 pairs in V's layout (wino4.h, "V pairs" with co in the place of ci): `zp` = the thread's dword of component 0
__device__ __forceinline__ void w4s_emit_zh(const float a[4][4], float* __restrict__ zp, size_t cstride, float pscale, bool odd) {
  float w[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i) w4s_a6(a[i][0], a[i][1], a[i][2], a[i][3], w[i][0], w[i][1], w[i][2], w[i][3], w[i][4], w[i][5]);
#pragma unroll
  for (int nu = 0; nu < 6; ++nu) {
    float z[6];
    w4s_a6(w[0][nu], w[1][nu], w[2][nu], w[3][nu], z[0], z[1], z[2], z[3], z[4], z[5]);
    w4s_pair_words2(z[0], z[1], pscale, odd);
    w4s_pair_words2(z[2], z[3], pscale, odd);
    w4s_pair_words2(z[4], z[5], pscale, odd);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) w4s_st_wt(zp + (size_t)(xi * 6 + nu) * cstride, z[xi]);
  }
}
// max|dz| of the wave -> W4Scales::gmax (and `ovf` when the fp16-pair scale 2^*g_exp cannot hold it); see wino4.h
__device__ __forceinline__ void w4s_record_max(const float v[4][4], W4Scales* sc, const int* g_exp) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float av = fabsf(v[i][j]);
      m = av < INFINITY ? fmaxf(m, av) : m;      // (finite values only: behind an overflow the rest of the step computes on infinities,
    }                                            //  and the maximum must stay the one the repeated step is scaled by)
  m = fmaxf(m, w4s_dpp<0xB1>(m));
  m = fmaxf(m, w4s_dpp<0x4E>(m));
  m = fmaxf(m, w4s_dpp<0x141>(m));
  m = fmaxf(m, w4s_dpp<0x140>(m));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    atomicMax(&sc->gmax[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (W4_GSLOTS - 1)], __builtin_bit_cast(unsigned, m));
    // (a non-finite cotangent is not a question of scale: it goes on to the error norm, which stops the solve)
    if (g_exp != nullptr && m < INFINITY && m * ldexpf(1.f, *g_exp) > W4_G_LIMIT) __hip_atomic_store(&sc->ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// masked column sums of the thread's channel (node_internal.h, masked_colsum_tile): out[tap * ld] for the nine taps.
// fr / lr / fc / lc: the thread's tile touches the image's first / last row / column.  The sum runs over the wave's four
// tiles: the whole image (Q = 1) or one quadrant of it (Q = 4: k_theta_finalize adds the quadrants' rows like samples').
__device__ __forceinline__ void w4s_colsums(const float v[4][4], int t, bool fr, bool lr, bool fc, bool lc, float* __restrict__ out, int ld) {
  float R[4], Cc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) R[i] = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) Cc[j] = (v[0][j] + v[1][j]) + (v[2][j] + v[3][j]);
  float T = (R[0] + R[1]) + (R[2] + R[3]);
  float rf = fr ? R[0] : 0.f, rl = lr ? R[3] : 0.f;
  float cf = fc ? Cc[0] : 0.f, cl = lc ? Cc[3] : 0.f;
  float k00 = (fr && fc) ? v[0][0] : 0.f, k01 = (fr && lc) ? v[0][3] : 0.f, k10 = (lr && fc) ? v[3][0] : 0.f, k11 = (lr && lc) ? v[3][3] : 0.f;
  T = w4s_tile_sum(T); rf = w4s_tile_sum(rf); rl = w4s_tile_sum(rl); cf = w4s_tile_sum(cf); cl = w4s_tile_sum(cl);
  k00 = w4s_tile_sum(k00); k01 = w4s_tile_sum(k01); k10 = w4s_tile_sum(k10); k11 = w4s_tile_sum(k11);
  if (t == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {   // tap (kh, kw) excludes the first (k == 0) / last (k == 2) row and column
      const int kh = tap / 3, kw = tap % 3;
      float o = T;
      if (kh == 0) o -= rf;
      if (kh == 2) o -= rl;
      if (kw == 0) o -= cf;
      if (kw == 2) o -= cl;
      if (kh == 0 && kw == 0) o += k00;
      if (kh == 0 && kw == 2) o += k01;
      if (kh == 2 && kw == 0) o += k10;
      if (kh == 2 && kw == 2) o += k11;
      out[(size_t)tap * ld] = o;
    }
  }
}
template <int Q>
__device__ __forceinline__ void w4s_colsums(const float v[4][4], const W4Wave<Q>& wv, float* __restrict__ spart, int C) {
  w4s_colsums(v, wv.t, wv.TY == 0, wv.TY == wv.TM, wv.TX == 0, wv.TX == wv.TM, spart + (size_t)wv.nv * 9 * C + wv.c, C);
}

__device__ __forceinline__ float4 w4s_ld4(const float* p, size_t f4) { return reinterpret_cast<const float4*>(p)[f4]; }
// state tensors (k, k_a, y1, xhat): plain 16-B stores.  (Measured, round 3: as two 8-byte write-through stores per vector the
// step was 3 % SLOWER -- these tensors are read again by the kernels right behind, partly out of the same L2s; the
// write-through stores of M / V / Z / dU, read once by a kernel with another workgroup -> XCD map, gave + 1.4 %.)
#ifndef NODE_WT_STATE
#define NODE_WT_STATE 0
#endif
__device__ __forceinline__ void w4s_st4(float* p, size_t f4, const float r[4]) {
#if NODE_WT_STORES && NODE_WT_STATE
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p + 4 * f4);
  const f32x2_t lo = {r[0], r[1]}, hi = {r[2], r[3]};
  __hip_atomic_store(q, __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  reinterpret_cast<float4*>(p)[f4] = make_float4(r[0], r[1], r[2], r[3]);
#endif
}

// y + sum_j cf[j] k[j] on the thread's 16 pixels, every request issued before the first use (NK is a compile-time
// count: a loop over a run-time count makes the compiler wait for each tensor before it requests the next).
// SELF: the last term is `self` (the stage derivative the head of this pass just produced), not a load.
// Summation order as in k_combine_gn: s = cf0 k0; s += cfj kj; y += s.
template <int NK, bool SELF>
__device__ __forceinline__ void w4s_comb(const Comb& c, const float* cf, size_t f4, const float self[4][4], float y[4][4]) {
  constexpr int NL = SELF ? NK - 1 : NK;
  float4 yv[4], kv[NL > 0 ? NL : 1][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) yv[i] = w4s_ld4(c.y, f4 + i * 64);
#pragma unroll
  for (int j = 0; j < NL; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) kv[j][i] = w4s_ld4(c.k[j], f4 + i * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float yy[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w};
    if (NK > 0) {
      float s[4];
      if (NL > 0) {
        s[0] = cf[0] * kv[0][i].x; s[1] = cf[0] * kv[0][i].y; s[2] = cf[0] * kv[0][i].z; s[3] = cf[0] * kv[0][i].w;
#pragma unroll
        for (int j = 1; j < NL; ++j) {
          s[0] += cf[j] * kv[j][i].x; s[1] += cf[j] * kv[j][i].y; s[2] += cf[j] * kv[j][i].z; s[3] += cf[j] * kv[j][i].w;
        }
        if (SELF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += cf[NK - 1] * self[i][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = cf[0] * self[i][e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) yy[e] += s[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) y[i][e] = yy[e];
  }
}
template <bool SELF>
__device__ __forceinline__ void w4s_comb_any(const Comb& c, const float* cf, size_t f4, const float self[4][4], float y[4][4]) {
  switch (c.nk) {
    case 0: w4s_comb<0, false>(c, cf, f4, self, y); break;
    case 1: w4s_comb<1, SELF>(c, cf, f4, self, y); break;
    case 2: w4s_comb<2, SELF>(c, cf, f4, self, y); break;
    case 3: w4s_comb<3, SELF>(c, cf, f4, self, y); break;
    case 4: w4s_comb<4, SELF>(c, cf, f4, self, y); break;
    case 5: w4s_comb<5, SELF>(c, cf, f4, self, y); break;
    case 6: w4s_comb<6, SELF>(c, cf, f4, self, y); break;
    default: w4s_comb<7, SELF>(c, cf, f4, self, y); break;
  }
}

// the thread's 16 pixels -> an NHWC tensor (the weight-gradient kernel's operand layout)
template <int Q>
__device__ __forceinline__ void w4s_store_nhwc(float* __restrict__ dst, const W4Wave<Q>& wv, int C, const float v[4][4]) {
  constexpr int WI = Q == 4 ? 16 : 8;   // image side
  float* base = dst + ((size_t)wv.n * (WI * WI) + (4 * wv.TY) * WI + 4 * wv.TX) * C + wv.c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) base[(size_t)(i * WI + j) * C] = v[i][j];
}

// GroupNorm statistics of the thread's group (two passes over the registers, like the reference's kernels)
template <int Q>
__device__ __forceinline__ void w4s_gn_stats(const float z[4][4], int cpg, float inv_m, float eps, W4Wave<Q>& wv, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += (z[i][0] + z[i][1]) + (z[i][2] + z[i][3]);
  mean = w4s_gsum(s, cpg, wv) * inv_m;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float dv = z[i][j] - mean; s2 += dv * dv; }
  const float var = w4s_gsum(s2, cpg, wv) * inv_m;
  rstd = 1.0f / sqrtf(var + eps);
}

// GroupNorm backward on the thread's pixels: g = d L / d (affine output); returns dz = osign * rstd (g gamma - m1 - xhat m2)
// and leaves the per-(virtual-)sample (dgamma, dbeta) partials of the channel in gpart ([Q N][2][C]).
template <int Q>
__device__ __forceinline__ void w4s_gn_bwd(const float g[4][4], const float xh[4][4], float gam, float rstd, int cpg, float inv_m,
                                           float osign, W4Wave<Q>& wv, int C, float* __restrict__ gpart, float dz[4][4]) {
  float dg = 0.f, db = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dg += g[i][j] * xh[i][j];
      db += g[i][j];
      const float dxh = g[i][j] * gam;
      s1 += dxh;
      s2 += dxh * xh[i][j];
    }
  dg = w4s_tile_sum(dg);
  db = w4s_tile_sum(db);
  if (wv.t == 0) {
    gpart[((size_t)wv.nv * 2 + 0) * C + wv.c] = dg;
    gpart[((size_t)wv.nv * 2 + 1) * C + wv.c] = db;
  }
  w4s_gsum2(s1, s2, cpg, wv);
  s1 *= inv_m;
  s2 *= inv_m;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dz[i][j] = osign * (rstd * (g[i][j] * gam - s1 - xh[i][j] * s2));
}

// the wave's place in the launch (see W4Wave): Q = 1 -> independent waves, W4S_THREADS / 64 per workgroup;
// Q = 4 -> workgroup = (sample, NB consecutive 16-channel blocks) x four quadrants, NB = blockDim / 256
template <int Q>
__device__ __forceinline__ void w4s_place(W4Wave<Q>& wv, int C, float* red, float* tiles) {
  const int CB = C >> 4;
  wv.lane = threadIdx.x & 63;
  wv.t = wv.lane >> 4; wv.c15 = wv.lane & 15; wv.ty = wv.t >> 1; wv.tx = wv.t & 1;
  wv.w = threadIdx.x >> 6; wv.slot = 0; wv.red = red; wv.tiles = tiles;
  if (Q == 1) {
    const int unit = blockIdx.x * (blockDim.x >> 6) + wv.w;
    wv.n = unit / CB; wv.cb = unit - wv.n * CB;
    wv.q = 0; wv.nv = wv.n; wv.nw = 1;
    wv.TY = wv.ty; wv.TX = wv.tx; wv.TM = 1;
  } else {
    const int NB = blockDim.x >> 8, GS = CB / NB;
    wv.n = blockIdx.x / GS;
    wv.cb = (blockIdx.x - wv.n * GS) * NB + (wv.w >> 2);
    wv.q = wv.w & 3; wv.nv = 4 * wv.n + wv.q; wv.nw = 4 * NB;
    wv.TY = 2 * (wv.q >> 1) + wv.ty; wv.TX = 2 * (wv.q & 1) + wv.tx; wv.TM = 3;
  }
  wv.c = wv.cb * 16 + wv.c15;
}
// pointers of the thread's (virtual sample, tile, channel) element in M, V (component 0) and its first W4S vector
template <int Q>
__device__ __forceinline__ const float* w4s_m_ptr(const float* M, const W4Wave<Q>& wv, int C) {
  return M + (((size_t)wv.nv * (C >> 5) + (wv.cb >> 1)) * 36) * 128 + wv.t * 32 + (wv.cb & 1) * 16 + wv.c15;
}
template <int Q>
__device__ __forceinline__ float* w4s_v_ptr(float* V, const W4Wave<Q>& wv, int C) {
  const int g8 = wv.cb * 2 + (wv.c15 >> 3), hi = (wv.c15 >> 2) & 1, e = wv.c15 & 3;
  return V + ((size_t)((wv.nv >> 3) * (C >> 3) + g8) * 256 + (wv.nv & 7) * 32 + hi * 16 + wv.t * 4 + e);
}
// the thread's dword of component 0 in the fp16-pair form: part = channel parity (see w4s_pair_word)
template <int Q>
__device__ __forceinline__ float* w4s_vh_ptr(float* V, const W4Wave<Q>& wv, int C) {
  const int gp = wv.c15 >> 3, hi = (wv.c15 >> 2) & 1, e = wv.c15 & 3;
  return V + ((size_t)(((wv.nv >> 3) * (C >> 4) + wv.cb) * 2 + (e & 1)) * 256 + ((wv.nv & 7) * 8 + hi * 4 + wv.t) * 4 + gp * 2 + (e >> 1));
}
template <int Q>
__device__ __forceinline__ void w4s_put_v(const float v[4][4], const W4Wave<Q>& wv, float* V, int C, int Nv, const int* v_exp = nullptr) {
  const float pscale = v_exp != nullptr ? ldexpf(1.f, *v_exp) : 0.f;
  float* vp = v_exp != nullptr ? w4s_vh_ptr(V, wv, C) : w4s_v_ptr(V, wv, C);
  if constexpr (Q == 1) w4s_emit_v(v, wv.ty, wv.tx, vp, (size_t)4 * Nv * C, pscale, (wv.c15 & 1) != 0);
  else w4s_emit_v16(v, wv, vp, (size_t)4 * Nv * C, pscale, (wv.c15 & 1) != 0);
}

// HEAD: 0 none, 1 forward (conv result -> GroupNorm), 2 backward (data gradient -> ReLU mask -> GroupNorm backward)
// TAIL: 0 none, 1 stage combine -> GroupNorm-1 -> ReLU, 2 adjoint combine -> GroupNorm-3 backward (needs HEAD 1)
// Q:    1 8x8 images, 4 16x16 images (file header)
template <int HEAD, int TAIL, int Q>
__global__ __launch_bounds__(Q == 1 ? W4S_THREADS : 512) void k_w4s_pass(W4sArgs a) {
  if (a.ctrl != nullptr && a.ctrl->done) return;   // a step enqueued past the end of the interval (Ctrl::done)
  __shared__ float s_red[Q == 4 ? W4Q_RED : 1];
  __shared__ float s_tiles[Q == 4 ? W4Q_TILES : 1];
  W4Wave<Q> wv;
  w4s_place(wv, a.C, s_red, s_tiles);
  const int lane = wv.lane, CB = a.C >> 4, n = wv.n, cb = wv.cb, t = wv.t, c = wv.c;
  const int cpg = a.cpg, G = a.C / cpg, grp = c / cpg;
  const float inv_m = 1.0f / (float)(64 * Q * cpg);
  const size_t f4 = ((size_t)wv.nv * CB + cb) * 256 + lane;   // float4 index of tile row 0 in a W4S tensor (rows 64 apart)
  const bool stat_lane = t == 0 && wv.q == 0 && c % cpg == 0;   // the one lane of the launch that owns (sample, group)

  float v[4][4];     // the conv input whose transform leaves at the end
  float hx[4][4];    // HEAD 1: xhat of the head's GroupNorm
  float ho[4][4];    // the head's output (a stage derivative)
  float hrstd = 0.f;

  if (HEAD == 1) {
    const W4sHead& h = a.h;
    float z[4][4];
    w4s_out_transform(w4s_m_ptr(h.M, wv, a.C), z);
    const float tval = eval_time(h.et), bv = h.bias[c];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 tm = w4s_ld4(h.tmapS, ((size_t)wv.q * CB + cb) * 256 + i * 64 + lane);
      z[i][0] += fmaf(tval, tm.x, bv); z[i][1] += fmaf(tval, tm.y, bv); z[i][2] += fmaf(tval, tm.z, bv); z[i][3] += fmaf(tval, tm.w, bv);
    }
    float mean;
    w4s_gn_stats(z, cpg, inv_m, a.eps, wv, mean, hrstd);
    const float gam = h.gamma[c], bet = h.beta[c];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hx[i][j] = (z[i][j] - mean) * hrstd;
        float vv = fmaf(hx[i][j], gam, bet);
        if (h.relu) vv = fmaxf(vv, 0.f);
        ho[i][j] = h.osign * vv;
      }
    if (h.out_s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) w4s_st4(h.out_s, f4 + i * 64, ho[i]);
    }
    if (h.out_nhwc) w4s_store_nhwc(h.out_nhwc, wv, a.C, ho);
    if (h.xhat_s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) w4s_st4(h.xhat_s, f4 + i * 64, hx[i]);
    }
    if (h.rstd && stat_lane) h.rstd[(size_t)n * G + grp] = hrstd;
    if (TAIL == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = ho[i][j];
    }
  }
  if (HEAD == 2) {
    const W4sHead& h = a.h;
    float4 xq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xq[i] = w4s_ld4(h.xhat_s, f4 + i * 64);
    const float gam = h.gamma[c], bet = h.beta[c], rs = h.rstd[(size_t)n * G + grp];
    float g[4][4], xh[4][4];
    w4s_out_transform(w4s_m_ptr(h.M, wv, a.C), g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xh[i][0] = xq[i].x; xh[i][1] = xq[i].y; xh[i][2] = xq[i].z; xh[i][3] = xq[i].w;
#pragma unroll
      for (int j = 0; j < 4; ++j) g[i][j] = fmaf(xh[i][j], gam, bet) > 0.f ? g[i][j] : 0.f;   // ReLU mask of this layer's output
    }
    w4s_gn_bwd(g, xh, gam, rs, cpg, inv_m, h.osign, wv, a.C, h.gpart, ho);
    if (h.out_s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) w4s_st4(h.out_s, f4 + i * 64, ho[i]);
    }
    if (h.out_nhwc) w4s_store_nhwc(h.out_nhwc, wv, a.C, ho);
    if (h.spart) w4s_colsums(ho, wv, h.spart, a.C);
    if (a.gstat != nullptr && (h.z_out != nullptr || (TAIL == 0 && a.V != nullptr))) w4s_record_max(ho, a.gstat, a.z_exp != nullptr ? a.z_exp : (TAIL == 0 ? a.v_exp : nullptr));
    if (h.z_out) {
      if (a.z_exp != nullptr) w4s_emit_zh(ho, w4s_vh_ptr(h.z_out, wv, a.C), (size_t)4 * a.Nv * a.C, ldexpf(1.f, *a.z_exp), (wv.c15 & 1) != 0);
      else w4s_emit_z(ho, wv.nv, a.Nv, a.C, c, t, h.z_out);
    }
    if (TAIL == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = ho[i][j];
    }
  }

  if (TAIL != 0) {
    const W4sTail& tl = a.t;
    const float scale = comb_scale(tl.comb, a.ctrl);
    float cf[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cf[j] = scale * tl.comb.coef[j];
    float y[4][4];
    if (HEAD == 1 && TAIL == 1) {
      if (tl.self) w4s_comb_any<true>(tl.comb, cf, f4, ho, y);
      else w4s_comb_any<false>(tl.comb, cf, f4, ho, y);
    } else {
      w4s_comb_any<false>(tl.comb, cf, f4, ho, y);
    }
    if (tl.y_out) {
#pragma unroll
      for (int i = 0; i < 4; ++i) w4s_st4(tl.y_out, f4 + i * 64, y[i]);
    }
    if (TAIL == 1) {   // GroupNorm-1 -> ReLU (model.py:341-342)
      float mean, rstd;
      w4s_gn_stats(y, cpg, inv_m, a.eps, wv, mean, rstd);
      const float gam = tl.gamma[c], bet = tl.beta[c];
      float xh[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (y[i][j] - mean) * rstd;
          v[i][j] = fmaxf(fmaf(xh[i][j], gam, bet), 0.f);
        }
      if (tl.act_nhwc) w4s_store_nhwc(tl.act_nhwc, wv, a.C, v);
      if (tl.xhat_s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w4s_st4(tl.xhat_s, f4 + i * 64, xh[i]);
      }
      if (tl.rstd && stat_lane) tl.rstd[(size_t)n * G + grp] = rstd;
    } else {   // cotangent g = csign * (adjoint combine) through GroupNorm-3's backward (xhat-3, 1/sigma-3 from the head)
      float g[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = tl.csign * y[i][j];
      w4s_gn_bwd(g, hx, a.h.gamma[c], hrstd, cpg, inv_m, 1.f, wv, a.C, tl.gpart, v);
      if (tl.act_nhwc) w4s_store_nhwc(tl.act_nhwc, wv, a.C, v);
      if (tl.spart) w4s_colsums(v, wv, tl.spart, a.C);
      if (a.gstat != nullptr) w4s_record_max(v, a.gstat, a.z_exp != nullptr ? a.z_exp : a.v_exp);
      if (tl.z_out) {
        if (a.z_exp != nullptr) w4s_emit_zh(v, w4s_vh_ptr(tl.z_out, wv, a.C), (size_t)4 * a.Nv * a.C, ldexpf(1.f, *a.z_exp), (wv.c15 & 1) != 0);
        else w4s_emit_z(v, wv.nv, a.Nv, a.C, c, t, tl.z_out);
      }
    }
  }

  if (a.V != nullptr) w4s_put_v(v, wv, a.V, a.C, a.Nv, a.v_exp);
}

// workgroups of the Q = 4 kernels: (sample, NB consecutive 16-channel blocks) x four quadrants, NB = 2 when a GroupNorm
// group spans two blocks (cpg = 32)
static inline void w4s_grid(int N, int C, int cpg, int Q, dim3& grid, dim3& block) {
  if (Q == 1) {
    block = dim3(W4S_THREADS);
    grid = dim3(N * (C >> 4) / (W4S_THREADS / 64));
  } else {
    const int NB = cpg > 16 ? cpg / 16 : 1;
    block = dim3(256 * NB);
    grid = dim3(N * ((C >> 4) / NB));
  }
}

void launch_w4s_pass(int head, int tail, const W4sArgs& a, hipStream_t s) {
  dim3 grid, block;
  w4s_grid(a.N, a.C, a.cpg, a.Q, grid, block);
#define W4S_GO(H, T)                                                                 \
  {                                                                                  \
    if (a.Q == 1) hipLaunchKernelGGL((k_w4s_pass<H, T, 1>), grid, block, 0, s, a);   \
    else hipLaunchKernelGGL((k_w4s_pass<H, T, 4>), grid, block, 0, s, a);            \
  }
  if (head == 0 && tail == 1) W4S_GO(0, 1)
  else if (head == 1 && tail == 0) W4S_GO(1, 0)
  else if (head == 1 && tail == 1) W4S_GO(1, 1)
  else if (head == 1 && tail == 2) W4S_GO(1, 2)
  else if (head == 2 && tail == 0) W4S_GO(2, 0)
  else if (head == 2 && tail == 1) W4S_GO(2, 1)
#undef W4S_GO
}

// ----------------------------------------------------------------------------
// NCHW <-> W4S at the solve boundary: a (sample, 16-channel block) is the same 4 KB (16 KB for 16x16 images: four
// quadrant blocks of the W4S tensor) on both sides, the 16-B vectors (four pixels of an image row inside one tile) permuted.
// ----------------------------------------------------------------------------
// NCHW float4 position `idx` of an [N, C, 8 sqrt(Q), 8 sqrt(Q)] tensor -> its float4 index in the W4S tensor
template <int Q>
__device__ __forceinline__ size_t w4s_of_nchw(size_t idx, int CB) {
  constexpr int LB = Q == 4 ? 10 : 8;              // log2 float4s per (sample, 16-channel block)
  const size_t blk = idx >> LB;                    // = n * CB + cb
  const int f = (int)(idx & ((1 << LB) - 1));
  if (Q == 1) {
    const int c15 = f >> 4, y = (f >> 1) & 7, tx = f & 1;
    return (blk << 8) + (y & 3) * 64 + ((y >> 2) * 2 + tx) * 16 + c15;
  }
  const int c15 = f >> 6, y = (f >> 2) & 15, x4 = f & 3;
  const int q = 2 * (y >> 3) + (x4 >> 1), t = 2 * ((y >> 2) & 1) + (x4 & 1);
  const size_t n = blk / CB, cb = blk - n * CB;
  return (((4 * n + q) * CB + cb) << 8) + (y & 3) * 64 + t * 16 + c15;
}
template <int Q>
__global__ __launch_bounds__(256) void k_w4s_layout(const float4* __restrict__ src, float4* __restrict__ dst, size_t total, int CB, int to_nchw) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t g = w4s_of_nchw<Q>(idx, CB);
  if (to_nchw) dst[idx] = src[g];      // coalesced writes
  else dst[g] = src[idx];              // coalesced reads
}
static void w4s_layout(const float* src, float* dst, int N, int C, int Q, int to_nchw, hipStream_t s) {
  const size_t total = (size_t)N * C * 16 * Q;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (Q == 1) hipLaunchKernelGGL(k_w4s_layout<1>, grid, block, 0, s, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), total, C >> 4, to_nchw);
  else hipLaunchKernelGGL(k_w4s_layout<4>, grid, block, 0, s, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), total, C >> 4, to_nchw);
}
void launch_w4s_from_nchw(const float* src, float* dst, int N, int C, int Q, hipStream_t s) { w4s_layout(src, dst, N, C, Q, 0, s); }
void launch_w4s_to_nchw(const float* src, float* dst, int N, int C, int Q, hipStream_t s) { w4s_layout(src, dst, N, C, Q, 1, s); }

// border-aware time-channel map [HW][C] -> the W4S blocking ([Q quadrants][C/16][4 i][64 lanes][4 j]) the forward passes read
__global__ __launch_bounds__(256) void k_w4s_tmap(const float* __restrict__ tmap0, const float* __restrict__ tmap1, float* __restrict__ out0,
                                                  float* __restrict__ out1, int C, int Q) {
  const float* tm = blockIdx.y ? tmap1 : tmap0;
  float* out = blockIdx.y ? out1 : out0;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 64 * Q * C) return;
  const int CB = C >> 4, WI = Q == 4 ? 16 : 8;
  const int j = idx & 3, lane = (idx >> 2) & 63, i = (idx >> 8) & 3, rest = idx >> 10;
  const int q = rest / CB, cb = rest - q * CB;
  const int t = lane >> 4, c = cb * 16 + (lane & 15);
  const int p = (8 * (q >> 1) + 4 * (t >> 1) + i) * WI + 8 * (q & 1) + 4 * (t & 1) + j;
  out[idx] = tm[(size_t)p * C + c];
}
void launch_w4s_tmap(const float* tmap0, const float* tmap1, float* out0, float* out1, int C, int Q, hipStream_t s) {
  hipLaunchKernelGGL(k_w4s_tmap, dim3((64 * Q * C + 255) / 256, 2), dim3(256), 0, s, tmap0, tmap1, out0, out1, C, Q);
}

// Dense output of the forward solve (k_emit_outputs) for W4S state: element-wise on the 16-B vectors, written at their
// NCHW position.
template <int Q>
__global__ __launch_bounds__(256) void k_w4s_emit_outputs(EmitArgs a, size_t total /* float4s per tensor */, int CB) {
  const Ctrl* c = a.ctrl;
  const int j0 = c->j0, j1 = c->j1;
  if (j1 <= j0) return;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // NCHW float4 position
  if (idx >= total) return;
  const size_t src = w4s_of_nchw<Q>(idx, CB);
  const float dt = (float)c->dt_used;
  const float t0f = (float)c->t_prev, t1f = (float)c->t;
  const float4 y0 = w4s_ld4(a.y0, src), y1 = w4s_ld4(a.y1, src);
  float4 kq[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) kq[q] = (q == 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : w4s_ld4(a.k[q], src);
  for (int j = j0; j < j1; ++j) {
    const float x = ((float)a.targets[j] - t0f) / (t1f - t0f);
    float kk[7];
    float4 o;
#pragma unroll
    for (int q = 0; q < 7; ++q) kk[q] = kq[q].x;
    o.x = interp_one(y0.x, y1.x, kk, dt, x);
#pragma unroll
    for (int q = 0; q < 7; ++q) kk[q] = kq[q].y;
    o.y = interp_one(y0.y, y1.y, kk, dt, x);
#pragma unroll
    for (int q = 0; q < 7; ++q) kk[q] = kq[q].z;
    o.z = interp_one(y0.z, y1.z, kk, dt, x);
#pragma unroll
    for (int q = 0; q < 7; ++q) kk[q] = kq[q].w;
    o.w = interp_one(y0.w, y1.w, kk, dt, x);
    reinterpret_cast<float4*>(a.y_out)[(size_t)j * total + idx] = o;
  }
}
void launch_w4s_emit_outputs(const Dims& d, const EmitArgs& a, hipStream_t s) {
  const size_t total = d.numel / 4;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (d.w4q == 1) hipLaunchKernelGGL(k_w4s_emit_outputs<1>, grid, block, 0, s, a, total, d.C >> 4);
  else hipLaunchKernelGGL(k_w4s_emit_outputs<4>, grid, block, 0, s, a, total, d.C >> 4);
}

// ----------------------------------------------------------------------------
// Stand-alone transforms around the GEMM (diagnostics / tests: node_conv3x3_w4): W4S tensor -> V, M -> W4S tensor
// ----------------------------------------------------------------------------
template <int Q>
__global__ __launch_bounds__(Q == 1 ? W4S_THREADS : 512) void k_w4s_input(const float* __restrict__ x, float* __restrict__ V, int C, int Nv,
                                                                          const int* v_exp) {
  __shared__ float s_tiles[Q == 4 ? W4Q_TILES : 1];
  W4Wave<Q> wv;
  w4s_place(wv, C, nullptr, s_tiles);
  const size_t f4 = ((size_t)wv.nv * (C >> 4) + wv.cb) * 256 + wv.lane;
  float v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 q = w4s_ld4(x, f4 + i * 64);
    v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
  }
  w4s_put_v(v, wv, V, C, Nv, v_exp);
}
template <int Q>
__global__ __launch_bounds__(Q == 1 ? W4S_THREADS : 512) void k_w4s_output(const float* __restrict__ M, float* __restrict__ y, int C) {
  W4Wave<Q> wv;
  w4s_place(wv, C, nullptr, nullptr);
  const size_t f4 = ((size_t)wv.nv * (C >> 4) + wv.cb) * 256 + wv.lane;
  float z[4][4];
  w4s_out_transform(w4s_m_ptr(M, wv, C), z);
#pragma unroll
  for (int i = 0; i < 4; ++i) w4s_st4(y, f4 + i * 64, z[i]);
}
// N: samples; Nv: virtual samples (Q N) rounded up to the GEMMs' row block
void launch_w4_input(const float* x_w4s, float* V, int N, int C, int Q, int Nv, hipStream_t s, const int* v_exp) {
  dim3 grid, block;
  w4s_grid(N, C, 1, Q, grid, block);
  if (Q == 1) hipLaunchKernelGGL(k_w4s_input<1>, grid, block, 0, s, x_w4s, V, C, Nv, v_exp);
  else hipLaunchKernelGGL(k_w4s_input<4>, grid, block, 0, s, x_w4s, V, C, Nv, v_exp);
}
void launch_w4_output(const float* M, float* y_w4s, int N, int C, int Q, hipStream_t s) {
  dim3 grid, block;
  w4s_grid(N, C, 1, Q, grid, block);
  if (Q == 1) hipLaunchKernelGGL(k_w4s_output<1>, grid, block, 0, s, M, y_w4s, C);
  else hipLaunchKernelGGL(k_w4s_output<4>, grid, block, 0, s, M, y_w4s, C);
}

// ----------------------------------------------------------------------------
// The residual stem's last convolution (model.py:284-310: conv2 of ResBlock(64, filters, stride 2) -- 3x3, stride 1, pad 1,
// filters -> filters on an 8x8 image, the ODE conv's own shape) through the F(4x4,3x3) pipeline: the same component GEMMs
// and weight-gradient kernel as the ODE block (stem_api.hip), with these transforms around them.  The stem is NHWC fp32,
// so the transforms read / write NHWC (or the caller's NCHW at the stem's boundary) directly -- no W4S detour.
// ----------------------------------------------------------------------------
// h (NHWC, the GroupNorm's input) -> xhat (W4S, kept for the backward), 1/sigma, V = B^T relu(GN(h)) B
__global__ __launch_bounds__(W4S_THREADS) void k_w4s_stem_in(const float* __restrict__ h, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int cpg, float* __restrict__ xhat_s,
                                                             float* __restrict__ rstd, float* __restrict__ V, int C, int Nv) {
  W4Wave<1> wv;
  w4s_place(wv, C, nullptr, nullptr);
  const float* base = h + ((size_t)wv.n * 64 + (4 * wv.TY) * 8 + 4 * wv.TX) * C + wv.c;
  float y[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) y[i][j] = base[(size_t)(i * 8 + j) * C];
  float mean, rs;
  w4s_gn_stats(y, cpg, 1.0f / (float)(64 * cpg), eps, wv, mean, rs);
  const float gam = gamma[wv.c], bet = beta[wv.c];
  float xh[4][4], v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[i][j] = (y[i][j] - mean) * rs;
      v[i][j] = fmaxf(fmaf(xh[i][j], gam, bet), 0.f);
    }
  const size_t f4 = ((size_t)wv.nv * (C >> 4) + wv.cb) * 256 + wv.lane;
#pragma unroll
  for (int i = 0; i < 4; ++i) w4s_st4(xhat_s, f4 + i * 64, xh[i]);
  if (wv.t == 0 && wv.c % cpg == 0) rstd[(size_t)wv.n * (C / cpg) + wv.c / cpg] = rs;
  w4s_put_v(v, wv, V, C, Nv);
}
// M -> A^T M A + shortcut (NHWC) -> the stem's output, NCHW
__global__ __launch_bounds__(W4S_THREADS) void k_w4s_stem_out(const float* __restrict__ M, const float* __restrict__ res, float* __restrict__ out, int C) {
  W4Wave<1> wv;
  w4s_place(wv, C, nullptr, nullptr);
  float z[4][4];
  w4s_out_transform(w4s_m_ptr(M, wv, C), z);
  const float* rb = res + ((size_t)wv.n * 64 + (4 * wv.TY) * 8 + 4 * wv.TX) * C + wv.c;
  float* ob = out + ((size_t)wv.n * C + wv.c) * 64 + (4 * wv.TY) * 8 + 4 * wv.TX;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 o;
    o.x = z[i][0] + rb[(size_t)(i * 8 + 0) * C];
    o.y = z[i][1] + rb[(size_t)(i * 8 + 1) * C];
    o.z = z[i][2] + rb[(size_t)(i * 8 + 2) * C];
    o.w = z[i][3] + rb[(size_t)(i * 8 + 3) * C];
    *reinterpret_cast<float4*>(ob + i * 8) = o;
  }
}
// dL/d out (NCHW) -> its row operand V (for the data gradient) and Z = A g A^T (for the weight gradient)
__global__ __launch_bounds__(W4S_THREADS) void k_w4s_stem_gin(const float* __restrict__ g, float* __restrict__ V, float* __restrict__ Z, int C, int Nv) {
  W4Wave<1> wv;
  w4s_place(wv, C, nullptr, nullptr);
  const float* gb = g + ((size_t)wv.n * C + wv.c) * 64 + (4 * wv.TY) * 8 + 4 * wv.TX;
  float v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 q = *reinterpret_cast<const float4*>(gb + i * 8);
    v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
  }
  w4s_emit_z(v, wv.nv, Nv, C, wv.c, wv.t, Z);
  w4s_put_v(v, wv, V, C, Nv);
}
// dU [36][ci][co] (k_w4_wgrad) -> dW [co][ci][3][3] = G^T dU G in PyTorch's layout
__global__ __launch_bounds__(256) void k_w4_du_to_dw(const float* __restrict__ dU, float* __restrict__ dW, int C) {
  const size_t CC = (size_t)C * C;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;    // ci * C + co (co fastest: the reads are coalesced)
  if (idx >= CC) return;
  const int ci = (int)(idx / C), co = (int)(idx - (size_t)ci * C);
  float tq[6][3];   // t[xi][kw] = sum_nu dU[xi][nu] G[nu][kw]
#pragma unroll
  for (int xi = 0; xi < 6; ++xi) {
    float u[6];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) u[nu] = dU[(size_t)(xi * 6 + nu) * CC + idx];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float sacc = 0.f;
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        const float gq = (float)W4_G[nu][kw];
        if (gq != 0.f) sacc += gq * u[nu];
      }
      tq[xi][kw] = sacc;
    }
  }
  float* o = dW + ((size_t)co * C + ci) * 9;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float sacc = 0.f;
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        const float gq = (float)W4_G[xi][kh];
        if (gq != 0.f) sacc += gq * tq[xi][kw];
      }
      o[kh * 3 + kw] = sacc;
    }
}
void launch_w4s_stem_in(const float* h_nhwc, const float* gamma, const float* beta, float eps, int cpg, float* xhat_s, float* rstd, float* V,
                        int N, int C, int Nv, hipStream_t s) {
  dim3 grid, block;
  w4s_grid(N, C, cpg, 1, grid, block);
  hipLaunchKernelGGL(k_w4s_stem_in, grid, block, 0, s, h_nhwc, gamma, beta, eps, cpg, xhat_s, rstd, V, C, Nv);
}
void launch_w4s_stem_out(const float* M, const float* res_nhwc, float* out_nchw, int N, int C, hipStream_t s) {
  dim3 grid, block;
  w4s_grid(N, C, 1, 1, grid, block);
  hipLaunchKernelGGL(k_w4s_stem_out, grid, block, 0, s, M, res_nhwc, out_nchw, C);
}
void launch_w4s_stem_gin(const float* g_nchw, float* V, float* Z, int N, int C, int Nv, hipStream_t s) {
  dim3 grid, block;
  w4s_grid(N, C, 1, 1, grid, block);
  hipLaunchKernelGGL(k_w4s_stem_gin, grid, block, 0, s, g_nchw, V, Z, C, Nv);
}
void launch_w4_du_to_dw(const float* dU, float* dW, int C, hipStream_t s) {
  const size_t CC = (size_t)C * C;
  hipLaunchKernelGGL(k_w4_du_to_dw, dim3((unsigned)((CC + 255) / 256)), dim3(256), 0, s, dU, dW, C);
}

}  // namespace node
