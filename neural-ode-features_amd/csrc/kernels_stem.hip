// The residual stem in front of the ODE block as hand-written gfx950 kernels (stem.h has the data formats):
//   reference: model.py:167-178 (ResDownsample), model.py:284-310 (ResBlock), model.py:255-265 (conv3x3 / conv1x1),
//   model.py:268-271 (GroupNorm(min(32, C), C)).
// Round 3 ran these seven convolutions through MIOpen: 1.3 ms of the 5.5 ms cfg-2 step, a third of it layout transposes
// and PyTorch reductions around the library's kernels.  Here the stem is NHWC from its first kernel to its last:
//   k_stem_conv      forward convolutions and data gradients of the 64 / 256-channel layers: a gather GEMM over
//                    (tap, input channel) on the bf16 matrix pipe at fp32 accuracy -- both operands arrive as exact bf16
//                    triples, staged global -> LDS by plain 16-B copies, six v_mfma_f32_32x32x16_bf16 per fp32 product block
//   k_stem_wgrad     weight gradients: the reduction runs over pixels, the lanes over channels, so NHWC rows are the
//                    MFMA operands as they lie in memory (v_mfma_f32_32x32x2_f32, no LDS in the loop), split-K slabs
//   k_stem_conv0_*   the first layer (K = 9 in_ch <= 27) on the NCHW input
//   k_stem_gn_*      GroupNorm + ReLU forward (writes the triples the next convolution reads) and backward
//   k_stem_prep / k_stem_from_nchw / k_stem_to_nchw / k_stem_reduce   filter splits, boundary layouts, slab sums
#include "stem.h"

namespace node {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = h + m + l exactly (3 x 8 significant bits; round-to-nearest-even at each level)
__device__ __forceinline__ void split3(float x, bf16_t& h, bf16_t& m, bf16_t& l) {
  const __bf16 bh = (__bf16)x;
  const float r = x - (float)bh;
  const __bf16 bm = (__bf16)r;
  const float r2 = r - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(bf16_t, bh);
  m = __builtin_bit_cast(bf16_t, bm);
  l = __builtin_bit_cast(bf16_t, bl);
}
// eight consecutive channels -> three 16-B vectors (v_cvt_pk_bf16_f32 pairs)
__device__ __forceinline__ void split8(const float4& p, const float4& q, u32x4& hh, u32x4& mm, u32x4& ll) {
  const f32x2 v[4] = {{p.x, p.y}, {p.z, p.w}, {q.x, q.y}, {q.z, q.w}};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 h = __builtin_convertvector(v[i], bf16x2);
    const f32x2 r = v[i] - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    const f32x2 r2 = r - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hh[i] = __builtin_bit_cast(unsigned, h);
    mm[i] = __builtin_bit_cast(unsigned, m);
    ll[i] = __builtin_bit_cast(unsigned, l);
  }
}
__device__ __forceinline__ float bf16_f(bf16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }

// ============================================================================
// k_stem_conv<NB>: out[row][co] = sum_{tap, ci} in[row @ tap][ci] * w[tap][co][ci]
//   tile: 128 rows x 32 NB columns, four waves, wave w = rows 32 w .. 32 w + 31 x all columns (NB accumulator blocks)
//   K chunk: one tap x 32 input channels = two MFMA K steps; operands double-buffered in LDS, one barrier per chunk:
//     A [plane 3][row 128][32 ch] bf16: 64-B rows, 16-B piece p of row r at p ^ ((r >> 2) & 3) -- a ds_read_b128 of 16
//       lanes (rows r .. r + 3 of four row quads) then touches every bank once
//     B [plane 3][col 32 NB][32 ch] likewise
//   a thread stages the same A row for every chunk (row -> pixel decode once); a tap outside the image reads the zero row
// ============================================================================
__device__ __forceinline__ int swz(int row, int piece) { return (piece ^ ((row >> 2) & 3)) << 4; }

template <int NB>
__global__ __launch_bounds__(256) void k_stem_conv(const SConvArgs a) {
  constexpr int BM = 128, BN = 32 * NB;
  constexpr int A_BYTES = 3 * BM * 64, B_BYTES = 3 * BN * 64, BUF = A_BYTES + B_BYTES;
  constexpr int BJ = NB / 2;     // 16-B pieces of B per thread and plane
  extern __shared__ __align__(16) unsigned char smem[];
  int* out_off = reinterpret_cast<int*>(smem + 2 * BUF);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntn = a.Cout / BN;
  int b = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // blocks b, b + 8, .. share an XCD: neighbouring tiles (the same A rows)
  const int tile_m = b / ntn, tile_n = b - tile_m * ntn;
  // class of this tile (static indices only: a dynamically indexed kernel-argument array would be copied to scratch)
  int tile0 = 0, ch = a.cls_h[0], cw = a.cls_w[0], cpy = a.cls_py[0], cpx = a.cls_px[0], ntap = a.cls_ntap[0];
  unsigned long long taps = a.cls_taps[0];
#pragma unroll
  for (int c = 1; c < 4; ++c)
    if (c < a.nclass && tile_m >= a.cls_tile0[c]) {
      tile0 = a.cls_tile0[c]; ch = a.cls_h[c]; cw = a.cls_w[c]; cpy = a.cls_py[c]; cpx = a.cls_px[c]; ntap = a.cls_ntap[c];
      taps = a.cls_taps[c];
    }
  const int row0 = (tile_m - tile0) * BM;
  const int Mc = a.N * ch * cw;
  // the A row this thread stages
  const int ra = t >> 1, pa = (t & 1) * 2;
  const int r = row0 + ra;
  const bool rvalid = r < Mc;
  int n = 0, oy = 0, ox = 0;
  if (rvalid) {
    n = r / (ch * cw);
    const int rem = r - n * ch * cw;
    const int cy = rem / cw;
    oy = cy * a.step + cpy;
    ox = (rem - cy * cw) * a.step + cpx;
  }
  if ((t & 1) == 0) out_off[ra] = rvalid ? (n * a.OH + oy) * a.OW + ox : -1;
  const int ncc = a.Cin >> 5;
  const int nchunk = ntap * ncc;

  u32x4 ga[3][2], gb[3][BJ];
  auto fetch = [&](int q) {
    const int ti = q / ncc, c0 = (q - ti * ncc) << 5;
    const int tap = (int)((taps >> (4 * ti)) & 15);
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    int iy, ix;
    bool ok = rvalid;
    if (a.mode == 0) {
      iy = (oy << a.sshift) + ky - a.pad;
      ix = (ox << a.sshift) + kx - a.pad;
    } else {
      const int ty = oy + a.pad - ky, tx = ox + a.pad - kx;   // divisible by the stride: the class' tap list guarantees it
      ok = ok && ty >= 0 && tx >= 0;
      iy = ty >> a.sshift;
      ix = tx >> a.sshift;
    }
    ok = ok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
    const int irow = ok ? (n * a.IH + iy) * a.IW + ix : a.zero_row;
    const bf16_t* src = a.in + (size_t)irow * a.Cin + c0 + pa * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4* s4 = reinterpret_cast<const u32x4*>(src + p * a.in_plane);
      ga[p][0] = s4[0];
      ga[p][1] = s4[1];
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int item = t + 256 * j, col = item >> 2, piece = item & 3;
      const bf16_t* ws = a.w + ((size_t)tap * a.Cout + tile_n * BN + col) * a.Cin + c0 + piece * 8;
#pragma unroll
      for (int p = 0; p < 3; ++p) gb[p][j] = *reinterpret_cast<const u32x4*>(ws + p * a.w_plane);
    }
  };
  auto stash = [&](int buf) {
    unsigned char* A = smem + buf * BUF;
    unsigned char* B = A + A_BYTES;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      *reinterpret_cast<u32x4*>(A + p * (BM * 64) + ra * 64 + swz(ra, pa)) = ga[p][0];
      *reinterpret_cast<u32x4*>(A + p * (BM * 64) + ra * 64 + swz(ra, pa + 1)) = ga[p][1];
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const int item = t + 256 * j, col = item >> 2, piece = item & 3;
        *reinterpret_cast<u32x4*>(B + p * (BN * 64) + col * 64 + swz(col, piece)) = gb[p][j];
      }
    }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  const int li = lane & 31, lg = lane >> 5;
  const int arow = 32 * wave + li;
  for (int q = 0; q < nchunk; ++q) {
    if (q + 1 < nchunk) fetch(q + 1);
    const unsigned char* A = smem + (q & 1) * BUF;
    const unsigned char* B = A + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fa[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(A + p * (BM * 64) + arow * 64 + swz(arow, 2 * ks + lg)));
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const int brow = 32 * c + li;
        bf16x8 fb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(B + p * (BN * 64) + brow * 64 + swz(brow, 2 * ks + lg)));
        // smallest products first
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[c], 0, 0, 0);
      }
    }
    if (q + 1 < nchunk) stash((q + 1) & 1);
    __syncthreads();
  }

  // epilogue: register r of a 32 x 32 block = row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32: 128-B row stores
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int col = tile_n * BN + 32 * c + li;
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rl = 32 * wave + 8 * (i >> 2) + 4 * lg + (i & 3);
      const int off = out_off[rl];
      if (off < 0) continue;
      const size_t o = (size_t)off * a.Cout + col;
      float v = acc[c][i] + bv;
      if (a.res) v += a.res[o];
      if (a.accumulate) v += a.out[o];
      a.out[o] = v;
    }
  }
}

// ============================================================================
// k_stem_wgrad<T, EXTRA>: dW[tap][co][ci] = sum_rows dy[row][co] * in[row @ tap][ci]   (T = KH * KW taps: 9 or 1)
//   a wave owns one (32 co x 32 ci) block for ALL taps over its share of the rows: T (+1) accumulators; lanes run over
//   channels, so both operands are 128-B runs of NHWC rows, straight from L2 (the activation is rebuilt from its bf16
//   triple: h + m + l is exact).  A K step is two consecutive pixels (lane halves); their (n, oy, ox) advance incrementally.
//   The four waves of a workgroup split the workgroup's rows and meet through LDS; a workgroup writes one slab.
//   EXTRA: the shortcut's 1x1 stride-s filter (no padding) reads input pixel (s oy, s ox) -- the centre tap of the 3x3
//   pad-1 filter -- so its weight gradient is one more accumulator fed by the same activation operand.
// ============================================================================
template <int T, bool EXTRA>
__global__ __launch_bounds__(256) void k_stem_wgrad(const SWgradArgs a) {
  constexpr int NA = T + (EXTRA ? 1 : 0);
  extern __shared__ __align__(16) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int ncib = a.Cin >> 5, ncob = a.Cout >> 5;
  const int pair = blockIdx.x % (ncib * ncob), split = blockIdx.x / (ncib * ncob);
  const int cob = pair / ncib, cib = pair - cob * ncib;
  const int co0 = cob * 32, ci0 = cib * 32;
  const int rows = a.N * a.OH * a.OW;
  const int per_wave = a.rows_per_split >> 2;
  const int rbeg = split * a.rows_per_split + wave * per_wave;
  const int rend = min(rbeg + per_wave, rows);
  f32x16 acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  int r = rbeg + hh;
  int n = 0, oy = 0, ox = 0;
  if (r < rows) {
    n = r / (a.OH * a.OW);
    const int rem = r - n * a.OH * a.OW;
    oy = rem / a.OW;
    ox = rem - oy * a.OW;
  }
  const bf16_t* in0 = a.in + ci0 + li;
  const bf16_t* in1 = in0 + a.in_plane;
  const bf16_t* in2 = in1 + a.in_plane;
  for (; r - hh < rend; r += 2) {
    const bool valid = r < rend;
    const float av = valid ? a.dy[(size_t)r * a.Cout + co0 + li] : 0.f;
    float av2 = 0.f;
    if (EXTRA) av2 = valid ? a.dy2[(size_t)r * a.Cout + co0 + li] : 0.f;
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    const int base = (n * a.IH + iy0) * a.IW + ix0;
    float bv[T];
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int ky = T == 9 ? tap / 3 : 0, kx = T == 9 ? tap % 3 : 0;
      const bool ok = valid && (unsigned)(iy0 + ky) < (unsigned)a.IH && (unsigned)(ix0 + kx) < (unsigned)a.IW;
      const size_t e = (size_t)(ok ? base + ky * a.IW + kx : a.zero_row) * a.Cin;
      bv[tap] = (bf16_f(in0[e]) + bf16_f(in1[e])) + bf16_f(in2[e]);
    }
#pragma unroll
    for (int tap = 0; tap < T; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[tap], acc[tap], 0, 0, 0);
    if (EXTRA) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(av2, bv[T == 9 ? 4 : 0], acc[T], 0, 0, 0);
    ox += 2;
    if (ox >= a.OW) {
      ox -= a.OW;
      oy += 1;
      if (oy >= a.OH) { oy = 0; n += 1; }
    }
  }
  // waves 1..3 hand their blocks to wave 0, one after the other (NA * 4 KB of LDS)
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(j * 16 + i) * 64 + lane] = acc[j][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] += red[(j * 16 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave != 0) return;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    float* dst = j < T ? a.slab + ((size_t)split * T + j) * a.Cout * a.Cin : a.slab2 + (size_t)split * a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + 8 * (i >> 2) + 4 * hh + (i & 3);
      dst[(size_t)co * a.Cin + ci0 + li] = acc[j][i];
    }
  }
}

// ============================================================================
// first layer: nn.Conv2d(in_ch, 64, 3, 1), no padding, bias; x is NCHW.  K = 9 in_ch (<= 27, padded to 28 by zero filter rows)
// ============================================================================
__global__ __launch_bounds__(256) void k_stem_conv0_fwd(const float* __restrict__ x, const float* __restrict__ w0t,
                                                        const float* __restrict__ bias, float* __restrict__ h0, int N, int Cin, int H,
                                                        int W) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hh = lane >> 5;
  const int OH = H - 2, OW = W - 2, rows = N * OH * OW, K = 9 * Cin;
  const int row0 = (blockIdx.x * 4 + wave) * 32;
  if (row0 >= rows) return;
  const int r = row0 + li;
  const bool valid = r < rows;
  int base = 0;
  if (valid) {
    const int n = r / (OH * OW), rem = r - n * OH * OW, oy = rem / OW, ox = rem - oy * OW;
    base = (n * Cin * H + oy) * W + ox;
  }
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const int steps = (K + 1) >> 1;
  for (int s = 0; s < steps; ++s) {
    const int k = 2 * s + hh, kk = min(k, K - 1);
    const int ci = kk / 9, rem = kk - 9 * ci, ky = rem / 3, kx = rem - 3 * ky;
    const float av = valid ? x[base + (ci * H + ky) * W + kx] : 0.f;     // (k >= K meets a zero filter row)
    const float b0 = w0t[k * 64 + li], b1 = w0t[k * 64 + 32 + li];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
  }
  const float bv0 = bias[li], bv1 = bias[32 + li];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rr = row0 + 8 * (i >> 2) + 4 * hh + (i & 3);
    if (rr >= rows) continue;
    h0[(size_t)rr * 64 + li] = acc0[i] + bv0;
    h0[(size_t)rr * 64 + 32 + li] = acc1[i] + bv1;
  }
}

// dW0[co][k] = sum_rows dh0[row][co] * patch(row)[k]; column K carries the bias gradient (a constant-one operand)
__global__ __launch_bounds__(256) void k_stem_conv0_wgrad(const float* __restrict__ x, const float* __restrict__ dh0,
                                                          float* __restrict__ slab, int N, int Cin, int H, int W, int rows_per_split) {
  __shared__ float red[2 * 16 * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, hh = lane >> 5;
  const int OH = H - 2, OW = W - 2, rows = N * OH * OW, K = 9 * Cin;
  const int per_wave = rows_per_split >> 2;
  const int rbeg = blockIdx.x * rows_per_split + wave * per_wave;
  const int rend = min(rbeg + per_wave, rows);
  int koff = 0;
  if (li < K) {
    const int ci = li / 9, rem = li - 9 * ci, ky = rem / 3;
    koff = (ci * H + ky) * W + (rem - 3 * ky);
  }
  int r = rbeg + hh;
  int n = 0, oy = 0, ox = 0;
  if (r < rows) {
    n = r / (OH * OW);
    const int rem = r - n * OH * OW;
    oy = rem / OW;
    ox = rem - oy * OW;
  }
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (; r - hh < rend; r += 2) {
    const bool valid = r < rend;
    const float a0 = valid ? dh0[(size_t)r * 64 + li] : 0.f;
    const float a1 = valid ? dh0[(size_t)r * 64 + 32 + li] : 0.f;
    float bv = 0.f;
    if (valid) bv = li < K ? x[(n * Cin * H + oy) * W + ox + koff] : (li == K ? 1.f : 0.f);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc1, 0, 0, 0);
    ox += 2;
    if (ox >= OW) {
      ox -= OW;
      oy += 1;
      if (oy >= OH) { oy = 0; n += 1; }
    }
  }
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { red[i * 64 + lane] = acc0[i]; red[(16 + i) * 64 + lane] = acc1[i]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] += red[i * 64 + lane]; acc1[i] += red[(16 + i) * 64 + lane]; }
    }
    __syncthreads();
  }
  if (wave != 0) return;
  float* dst = slab + (size_t)blockIdx.x * 64 * 32;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int co = 8 * (i >> 2) + 4 * hh + (i & 3);
    dst[co * 32 + li] = acc0[i];
    dst[(32 + co) * 32 + li] = acc1[i];
  }
}

// ============================================================================
// GroupNorm + ReLU.  One workgroup per (sample, CB channels); the block [HW][CB] lives in LDS.
//   thread -> channel t % CB (fixed), pixels t / CB, t / CB + 256 / CB, ...; group sums through LDS
// ============================================================================
__device__ __forceinline__ void group_sums(float v, float* red, float* grp, int CB, int cpg, int t) {
  // red[256] <- v; grp[g] <- sum of the threads whose channel lies in group g.  Two barriers.
  red[t] = v;
  __syncthreads();
  const int ngrp = CB / cpg;
  if (t < ngrp) {
    float s = 0.f;
    for (int j = 0; j < 256 / CB; ++j)
      for (int k = 0; k < cpg; ++k) s += red[j * CB + t * cpg + k];
    grp[t] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_stem_gn_fwd(const SGnArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int t = threadIdx.x;
  const int CB = a.CB, HW = a.HW, C = a.C;
  // (no static __shared__ next to a dynamic region that may need the whole LDS: hipFuncSetAttribute refuses their sum)
  float* tile = reinterpret_cast<float*>(smem);
  float* red = tile + (size_t)HW * CB;      // [256]
  float* gmean = red + 256;                 // [128]
  float* grstd = gmean + 128;               // [128]
  const int nblk = C / CB;
  const int n = blockIdx.x / nblk, c0 = (blockIdx.x - n * nblk) * CB;
  const int q4 = CB >> 2;
  const float* src = a.h + ((size_t)n * HW) * C + c0;
  for (int idx = t; idx < HW * q4; idx += 256) {
    const int px = idx / q4, q = idx - px * q4;
    *reinterpret_cast<float4*>(tile + px * CB + 4 * q) = *reinterpret_cast<const float4*>(src + (size_t)px * C + 4 * q);
  }
  __syncthreads();
  const int c = t & (CB - 1), pg = t / CB, PG = 256 / CB;
  const float inv_m = 1.f / (float)(a.cpg * HW);
  float s = 0.f;
  for (int px = pg; px < HW; px += PG) s += tile[px * CB + c];
  group_sums(s, red, gmean, CB, a.cpg, t);
  const float mean = gmean[c / a.cpg] * inv_m;
  float s2 = 0.f;
  for (int px = pg; px < HW; px += PG) {
    const float d = tile[px * CB + c] - mean;
    s2 += d * d;
  }
  group_sums(s2, red, grstd, CB, a.cpg, t);
  const int ngrp = CB / a.cpg;
  if (t < ngrp) {
    const float m = gmean[t] * inv_m, rs = rsqrtf(grstd[t] * inv_m + a.eps);
    float* st = a.stats + ((size_t)n * (C / a.cpg) + c0 / a.cpg + t) * 2;
    st[0] = m;
    st[1] = rs;
  }
  __syncthreads();
  if (t < ngrp) {     // (second barrier of group_sums is behind every read of grstd above)
    const float m = gmean[t] * inv_m, rs = rsqrtf(grstd[t] * inv_m + a.eps);
    gmean[t] = m;
    grstd[t] = rs;
  }
  __syncthreads();
  const int q8 = CB >> 3;
  for (int idx = t; idx < HW * q8; idx += 256) {
    const int px = idx / q8, q = idx - px * q8;
    const float4 p0 = *reinterpret_cast<const float4*>(tile + px * CB + 8 * q);
    const float4 p1 = *reinterpret_cast<const float4*>(tile + px * CB + 8 * q + 4);
    float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cc = 8 * q + k, g = cc / a.cpg;
      const float y = (v[k] - gmean[g]) * grstd[g] * a.gamma[c0 + cc] + a.beta[c0 + cc];
      v[k] = fmaxf(y, 0.f);
    }
    u32x4 hh, mm, ll;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hh, mm, ll);
    bf16_t* dst = a.a3 + ((size_t)n * HW + px) * C + c0 + 8 * q;
    *reinterpret_cast<u32x4*>(dst) = hh;
    *reinterpret_cast<u32x4*>(dst + a.a_plane) = mm;
    *reinterpret_cast<u32x4*>(dst + 2 * a.a_plane) = ll;
  }
}

// backward: da = dL/d relu(GN(h)).  dy = da where the activation is positive; dgamma_c = sum dy xhat, dbeta_c = sum dy;
// dh = rstd (dy gamma - (s1 + xhat s2) / m) with s1 = sum_group gamma dbeta, s2 = sum_group gamma dgamma
__global__ __launch_bounds__(256) void k_stem_gn_bwd(const SGnArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int CB = a.CB, HW = a.HW, C = a.C;
  float* txh = reinterpret_cast<float*>(smem);          // h, then xhat
  float* tdy = txh + (size_t)HW * CB;                     // da, then dy
  float* redA = tdy + (size_t)HW * CB;                    // [256]
  float* redB = redA + 256;                               // [256]
  float* gmean = redB + 256;                              // [128] each
  float* grstd = gmean + 128;
  float* gs1 = grstd + 128;
  float* gs2 = gs1 + 128;
  const int t = threadIdx.x;
  const int nblk = C / CB;
  const int n = blockIdx.x / nblk, c0 = (blockIdx.x - n * nblk) * CB;
  const int q4 = CB >> 2;
  const float* sh = a.h + ((size_t)n * HW) * C + c0;
  const float* sd = a.da + ((size_t)n * HW) * C + c0;
  for (int idx = t; idx < HW * q4; idx += 256) {
    const int px = idx / q4, q = idx - px * q4;
    *reinterpret_cast<float4*>(txh + px * CB + 4 * q) = *reinterpret_cast<const float4*>(sh + (size_t)px * C + 4 * q);
    *reinterpret_cast<float4*>(tdy + px * CB + 4 * q) = *reinterpret_cast<const float4*>(sd + (size_t)px * C + 4 * q);
  }
  const int ngrp = CB / a.cpg;
  if (t < ngrp) {
    const float* st = a.stats + ((size_t)n * (C / a.cpg) + c0 / a.cpg + t) * 2;
    gmean[t] = st[0];
    grstd[t] = st[1];
  }
  __syncthreads();
  const int c = t & (CB - 1), pg = t / CB, PG = 256 / CB;
  const float mean = gmean[c / a.cpg], rstd = grstd[c / a.cpg], gam = a.gamma[c0 + c], bet = a.beta[c0 + c];
  float sA = 0.f, sB = 0.f;
  for (int px = pg; px < HW; px += PG) {
    const float xh = (txh[px * CB + c] - mean) * rstd;
    const float y = xh * gam + bet;
    const float dy = y > 0.f ? tdy[px * CB + c] : 0.f;
    txh[px * CB + c] = xh;
    tdy[px * CB + c] = dy;
    sA += dy;
    sB += dy * xh;
  }
  redA[t] = sA;
  redB[t] = sB;
  __syncthreads();
  if (t < CB) {
    float A = 0.f, B = 0.f;
    for (int j = 0; j < PG; ++j) { A += redA[j * CB + t]; B += redB[j * CB + t]; }
    a.gpart[((size_t)n * 2 + 0) * C + c0 + t] = B;      // dgamma
    a.gpart[((size_t)n * 2 + 1) * C + c0 + t] = A;      // dbeta
    redA[t] = A * gam;      // (t < CB: c == t)
    redB[t] = B * gam;
  }
  __syncthreads();
  if (t < ngrp) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < a.cpg; ++k) { s1 += redA[t * a.cpg + k]; s2 += redB[t * a.cpg + k]; }
    gs1[t] = s1;
    gs2[t] = s2;
  }
  __syncthreads();
  const float inv_m = 1.f / (float)(a.cpg * HW);
  const int q8 = CB >> 3;
  for (int idx = t; idx < HW * q8; idx += 256) {
    const int px = idx / q8, q = idx - px * q8;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cc = 8 * q + k, g = cc / a.cpg;
      const float xh = txh[px * CB + cc], dy = tdy[px * CB + cc];
      v[k] = grstd[g] * (dy * a.gamma[c0 + cc] - (gs1[g] + xh * gs2[g]) * inv_m);
    }
    const size_t o = ((size_t)n * HW + px) * C + c0 + 8 * q;
    const float4 p0 = make_float4(v[0], v[1], v[2], v[3]), p1 = make_float4(v[4], v[5], v[6], v[7]);
    if (a.dh) {
      *reinterpret_cast<float4*>(a.dh + o) = p0;
      *reinterpret_cast<float4*>(a.dh + o + 4) = p1;
    }
    if (a.dh3) {
      u32x4 hh, mm, ll;
      split8(p0, p1, hh, mm, ll);
      *reinterpret_cast<u32x4*>(a.dh3 + o) = hh;
      *reinterpret_cast<u32x4*>(a.dh3 + o + a.dh_plane) = mm;
      *reinterpret_cast<u32x4*>(a.dh3 + o + 2 * a.dh_plane) = ll;
    }
  }
}

// ============================================================================
// boundary layouts: NCHW fp32 <-> NHWC (fp32 and / or triples); 64 channels x 64 pixels per workgroup through LDS
// ============================================================================
__global__ __launch_bounds__(256) void k_stem_from_nchw(const float* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dst3,
                                                        size_t plane, int C, int HW) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int npb = (HW + 63) / 64, ncb = C / 64;
  const int n = blockIdx.x / (npb * ncb), rem = blockIdx.x - n * npb * ncb, cb = rem / npb, pb = rem - cb * npb;
  const int c0 = cb * 64, p0 = pb * 64;
  for (int idx = t; idx < 64 * 64; idx += 256) {
    const int cc = idx >> 6, pp = idx & 63;
    tile[cc][pp] = p0 + pp < HW ? src[((size_t)n * C + c0 + cc) * HW + p0 + pp] : 0.f;
  }
  __syncthreads();
  for (int idx = t; idx < 64 * 8; idx += 256) {
    const int pp = idx >> 3, q = idx & 7;
    if (p0 + pp >= HW) continue;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tile[8 * q + k][pp];
    const size_t o = ((size_t)n * HW + p0 + pp) * C + c0 + 8 * q;
    const float4 a0 = make_float4(v[0], v[1], v[2], v[3]), a1 = make_float4(v[4], v[5], v[6], v[7]);
    if (dst) {
      *reinterpret_cast<float4*>(dst + o) = a0;
      *reinterpret_cast<float4*>(dst + o + 4) = a1;
    }
    if (dst3) {
      u32x4 hh, mm, ll;
      split8(a0, a1, hh, mm, ll);
      *reinterpret_cast<u32x4*>(dst3 + o) = hh;
      *reinterpret_cast<u32x4*>(dst3 + o + plane) = mm;
      *reinterpret_cast<u32x4*>(dst3 + o + 2 * plane) = ll;
    }
  }
}

__global__ __launch_bounds__(256) void k_stem_to_nchw(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int npb = (HW + 63) / 64, ncb = C / 64;
  const int n = blockIdx.x / (npb * ncb), rem = blockIdx.x - n * npb * ncb, cb = rem / npb, pb = rem - cb * npb;
  const int c0 = cb * 64, p0 = pb * 64;
  for (int idx = t; idx < 64 * 64; idx += 256) {
    const int pp = idx >> 6, cc = idx & 63;
    tile[cc][pp] = p0 + pp < HW ? src[((size_t)n * HW + p0 + pp) * C + c0 + cc] : 0.f;
  }
  __syncthreads();
  for (int idx = t; idx < 64 * 64; idx += 256) {
    const int cc = idx >> 6, pp = idx & 63;
    if (p0 + pp < HW) dst[((size_t)n * C + c0 + cc) * HW + p0 + pp] = tile[cc][pp];
  }
}

// ============================================================================
// preparation (once per forward): filters -> triples in both operand layouts, conv0's filter transposed, zero rows
// ============================================================================
__global__ __launch_bounds__(256) void k_stem_prep(const SPrepArgs a) {
  const int job = blockIdx.y;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (job < a.njobs) {
    const SPrepJob& j = a.job[job];
    const size_t total = (size_t)j.taps * j.Cout * j.Cin;
    if (idx >= total) return;
    const int ci = (int)(idx % j.Cin);
    const size_t r = idx / j.Cin;
    const int co = (int)(r % j.Cout), tap = (int)(r / j.Cout);
    const float v = j.w[((size_t)co * j.Cin + ci) * j.taps + tap];
    bf16_t h, m, l;
    split3(v, h, m, l);
    const size_t of = ((size_t)tap * j.Cout + co) * j.Cin + ci, od = ((size_t)tap * j.Cin + ci) * j.Cout + co;
    j.wf[of] = h; j.wf[of + total] = m; j.wf[of + 2 * total] = l;
    j.wd[od] = h; j.wd[od + total] = m; j.wd[od + 2 * total] = l;
    return;
  }
  if (job == a.njobs) {        // conv0: [64][k0] -> [28][64], zero rows behind k0
    if (a.w0t == nullptr || idx >= 28 * 64) return;
    const int k = (int)(idx / 64), co = (int)(idx % 64);
    a.w0t[idx] = k < a.k0 ? a.w0[co * a.k0 + k] : 0.f;
    return;
  }
  // zero rows of the triples tensors
  const int which = (int)(idx / 4096), e = (int)(idx % 4096);
  if (which >= a.nzero * 3) return;
  const int ten = which / 3, p = which - 3 * ten;
  if (e < a.zero_c[ten]) a.zero[ten][p * a.zero_plane[ten] + e] = 0;
}

// ============================================================================
// slab / partial sums into the caller's gradient tensors
// ============================================================================
__global__ __launch_bounds__(256) void k_stem_reduce(const SReduceArgs a) {
  const SReduceJob& j = a.job[blockIdx.y];
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j.kind == 0) {
    const size_t total = (size_t)j.taps * j.Co * j.Ci;
    if (idx >= total) return;
    float s = 0.f;
    for (int k = 0; k < j.ns; ++k) s += j.slab[(size_t)k * total + idx];
    const int ci = (int)(idx % j.Ci);
    const size_t r = idx / j.Ci;
    const int co = (int)(r % j.Co), tap = (int)(r / j.Co);
    j.out[((size_t)co * j.Ci + ci) * j.taps + tap] = s;
  } else if (j.kind == 1) {
    if (idx >= 64 * 32) return;
    float s = 0.f;
    for (int k = 0; k < j.ns; ++k) s += j.slab[(size_t)k * 64 * 32 + idx];
    const int co = (int)(idx >> 5), k = (int)(idx & 31);
    if (k < j.Ci) j.out[co * j.Ci + k] = s;
    else if (k == j.Ci) j.out2[co] = s;
  } else {
    if (idx >= (size_t)2 * j.Co) return;
    float s = 0.f;
    for (int k = 0; k < j.ns; ++k) s += j.slab[(size_t)k * 2 * j.Co + idx];
    if (idx < (size_t)j.Co) j.out[idx] = s;
    else j.out2[idx - j.Co] = s;
  }
}

}  // namespace

// ============================================================================
// launchers
// ============================================================================
void launch_stem_conv(const SConvArgs& a, hipStream_t s) {
  const int mt = a.cls_tile0[a.nclass];
  // 128-column tiles halve the A traffic per MFMA but need twice the LDS (one workgroup per CU): only where the grid
  // still covers the chip twice
  const bool wide = a.Cout % 128 == 0 && (size_t)mt * (a.Cout / 128) >= 512;
  if (wide) {
    static bool attr[MAX_DEVICES] = {};
    allow_full_lds(reinterpret_cast<const void*>(k_stem_conv<4>), attr);
    const size_t lds = 2 * (3 * 128 * 64 + 3 * 128 * 64) + 512;
    hipLaunchKernelGGL(k_stem_conv<4>, dim3(mt * (a.Cout / 128)), dim3(256), lds, s, a);
  } else {
    static bool attr[MAX_DEVICES] = {};
    allow_full_lds(reinterpret_cast<const void*>(k_stem_conv<2>), attr);
    const size_t lds = 2 * (3 * 128 * 64 + 3 * 64 * 64) + 512;
    hipLaunchKernelGGL(k_stem_conv<2>, dim3(mt * (a.Cout / 64)), dim3(256), lds, s, a);
  }
}

void launch_stem_wgrad(const SWgradArgs& a, hipStream_t s) {
  const int grid = (a.Cin / 32) * (a.Cout / 32) * a.nsplit;
  const int taps = a.KH * a.KW;
  const size_t lds = (size_t)(taps + (a.dy2 ? 1 : 0)) * 16 * 64 * sizeof(float);
#define STEM_WG(T, EX)                                                                   \
  {                                                                                       \
    static bool attr[MAX_DEVICES] = {};                                                   \
    allow_full_lds(reinterpret_cast<const void*>(k_stem_wgrad<T, EX>), attr);             \
    hipLaunchKernelGGL((k_stem_wgrad<T, EX>), dim3(grid), dim3(256), lds, s, a);           \
  }
  if (taps == 9 && a.dy2) STEM_WG(9, true)
  else if (taps == 9) STEM_WG(9, false)
  else STEM_WG(1, false)
#undef STEM_WG
}

void launch_stem_conv0_fwd(const float* x, const float* w0t, const float* bias, float* h0, int N, int Cin, int H, int W, hipStream_t s) {
  const int rows = N * (H - 2) * (W - 2);
  hipLaunchKernelGGL(k_stem_conv0_fwd, dim3((rows + 127) / 128), dim3(256), 0, s, x, w0t, bias, h0, N, Cin, H, W);
}
void launch_stem_conv0_wgrad(const float* x, const float* dh0, float* slab, int N, int Cin, int H, int W, int nsplit, int rows_per_split,
                             hipStream_t s) {
  hipLaunchKernelGGL(k_stem_conv0_wgrad, dim3(nsplit), dim3(256), 0, s, x, dh0, slab, N, Cin, H, W, rows_per_split);
}

// channels per workgroup of the GroupNorm passes: a power of two in [max(8, cpg), 128] whose [HW][CB] block fits the LDS
// twice (the backward holds xhat and dy); 0 if even the smallest does not
int stem_gn_cb(int HW, int C, int cpg) {
  const size_t budget = 150 * 1024;
  int cb = 8;
  while (cb < cpg) cb *= 2;
  if (cb > C || cb > 128 || (size_t)HW * cb * 2 * sizeof(float) > budget) return 0;
  while (cb * 2 <= C && cb * 2 <= 128 && (size_t)HW * cb * 4 * sizeof(float) <= budget) cb *= 2;
  return cb;
}
void launch_stem_gn_fwd(const SGnArgs& a, hipStream_t s) {
  static bool attr[MAX_DEVICES] = {};
  allow_full_lds(reinterpret_cast<const void*>(k_stem_gn_fwd), attr);
  hipLaunchKernelGGL(k_stem_gn_fwd, dim3(a.N * (a.C / a.CB)), dim3(256), ((size_t)a.HW * a.CB + 512) * sizeof(float), s, a);
}
void launch_stem_gn_bwd(const SGnArgs& a, hipStream_t s) {
  static bool attr[MAX_DEVICES] = {};
  allow_full_lds(reinterpret_cast<const void*>(k_stem_gn_bwd), attr);
  hipLaunchKernelGGL(k_stem_gn_bwd, dim3(a.N * (a.C / a.CB)), dim3(256), ((size_t)2 * a.HW * a.CB + 1024) * sizeof(float), s, a);
}

void launch_stem_prep(const SPrepArgs& a, hipStream_t s) {
  size_t most = 28 * 64;
  for (int i = 0; i < a.njobs; ++i) most = max(most, (size_t)a.job[i].taps * a.job[i].Cout * a.job[i].Cin);
  most = max(most, (size_t)a.nzero * 3 * 4096);
  hipLaunchKernelGGL(k_stem_prep, dim3((unsigned)((most + 255) / 256), a.njobs + 2), dim3(256), 0, s, a);
}
void launch_stem_from_nchw(const float* src, float* dst_nhwc, bf16_t* dst3, size_t plane, int N, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(k_stem_from_nchw, dim3(N * (C / 64) * ((HW + 63) / 64)), dim3(256), 0, s, src, dst_nhwc, dst3, plane, C, HW);
}
void launch_stem_to_nchw(const float* src_nhwc, float* dst, int N, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(k_stem_to_nchw, dim3(N * (C / 64) * ((HW + 63) / 64)), dim3(256), 0, s, src_nhwc, dst, C, HW);
}
void launch_stem_reduce(const SReduceArgs& a, hipStream_t s) {
  size_t most = 64 * 32;
  for (int i = 0; i < a.njobs; ++i) {
    const SReduceJob& j = a.job[i];
    most = max(most, j.kind == 0 ? (size_t)j.taps * j.Co * j.Ci : (size_t)2 * j.Co);
  }
  hipLaunchKernelGGL(k_stem_reduce, dim3((unsigned)((most + 255) / 256), a.njobs), dim3(256), 0, s, a);
}

}  // namespace node
