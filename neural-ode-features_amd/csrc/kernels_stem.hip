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
// LDS images of the two MFMA kernels below: [row][64 channels] bf16 = 128-B rows, one image per bf16 part, filled by
// LDS-DMA (global_load_lds_dwordx4: one wave instruction = 1 KB = eight whole rows; the destination is lane-linear, so the
// XOR swizzle sits on the per-lane SOURCE address): 16-B piece p of row r lives at slot p ^ ((r >> 1) & 7).  Both the
// row reads (ds_read_b128, forward / data gradient) and the transposed reads (ds_read_b64_tr_b16, weight gradient) of a
// 32-lane half then touch every bank once.
// ============================================================================
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int img_off(int row, int piece) { return row * 128 + ((piece ^ ((row >> 1) & 7)) << 4); }
// eight rows (row0 .. row0 + 7) of one image: lane L brings piece (L & 7) ^ swizzle of row row0 + (L >> 3)
__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_dst) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src), (lds_ptr_t)lds_dst, 16, 0, 0);
}

// ============================================================================
// k_stem_conv: out[row][co] = sum_{tap, ci} in[row @ tap][ci] * w[tap][co][ci]
//   tile: 128 rows x 64 columns, four waves, wave w = rows 32 w .. 32 w + 31 x both 32-column blocks
//   K chunk: one tap x 64 input channels = four MFMA K steps of six part products; both operands double-buffered in LDS
//   (2 x 72 KB), chunk q + 1 in flight (LDS-DMA) while chunk q is multiplied, one barrier per chunk
//   a lane stages the same four A rows for every chunk (row -> pixel decode once); a tap outside the image reads the zero row
// ============================================================================
__global__ __launch_bounds__(256) void k_stem_conv(const SConvArgs a) {
  constexpr int BM = 128, BN = 64;
  constexpr int A_PLANE = BM * 128, B_PLANE = BN * 128, A_BYTES = 3 * A_PLANE, BUF = A_BYTES + 3 * B_PLANE;
  extern __shared__ __align__(16) unsigned char smem[];
  int* out_off = reinterpret_cast<int*>(smem + 2 * BUF);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntn = a.Cout / BN;
  int b = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);   // blocks b, b + 8, .. share an XCD: neighbouring tiles (the same A rows)
  const int ntn_all = (a.mode == 0 && a.w2 != nullptr) ? 2 * ntn : ntn;
  const int tile_m = b / ntn_all;
  int tile_n = b - tile_m * ntn_all;
  const bool shortcut_tile = tile_n >= ntn;        // forward: the shortcut's column tiles
  if (shortcut_tile) tile_n -= ntn;
  // class of this tile (static indices only: a dynamically indexed kernel-argument array would be copied to scratch)
  int tile0 = 0, ch = a.cls_h[0], cw = a.cls_w[0], cpy = a.cls_py[0], cpx = a.cls_px[0], ntap = a.cls_ntap[0];
  unsigned long long taps = a.cls_taps[0];
#pragma unroll
  for (int c = 1; c < 4; ++c)
    if (c < a.nclass && tile_m >= a.cls_tile0[c]) {
      tile0 = a.cls_tile0[c]; ch = a.cls_h[c]; cw = a.cls_w[c]; cpy = a.cls_py[c]; cpx = a.cls_px[c]; ntap = a.cls_ntap[c];
      taps = a.cls_taps[c];
    }
  if (shortcut_tile) { ntap = 1; taps = (unsigned long long)((a.KH >> 1) * a.KW + (a.KW >> 1)); }   // the centre tap
  const bf16_t* wsel = shortcut_tile ? a.w2 : a.w;
  const size_t wsel_plane = shortcut_tile ? a.w2_plane : a.w_plane;
  const int wtaps_one = shortcut_tile ? 1 : 0;     // the shortcut's filter has ONE tap block
  // data gradient: the shortcut's K segment, for the (even, even) class
  const bool extra_k = a.mode == 1 && a.in2 != nullptr && cpy == 0 && cpx == 0 && a.step == 2;
  const int row0 = (tile_m - tile0) * BM;
  const int Mc = a.N * ch * cw;
  // the four A rows this lane stages: 32 wave + 8 j + lane / 8
  const int lrow = lane >> 3, lslot = lane & 7;
  int rn[4], roy[4], rox[4];
  bool rval[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rl = 32 * wave + 8 * j + lrow, r = row0 + rl;
    rval[j] = r < Mc;
    rn[j] = roy[j] = rox[j] = 0;
    if (rval[j]) {
      rn[j] = r / (ch * cw);
      const int rem = r - rn[j] * ch * cw;
      const int cy = rem / cw;
      roy[j] = cy * a.step + cpy;
      rox[j] = (rem - cy * cw) * a.step + cpx;
    }
    if (lslot == 0) out_off[rl] = rval[j] ? (rn[j] * a.OH + roy[j]) * a.OW + rox[j] : -1;
  }
  const int ncc = a.Cin >> 6;
  const int nmain = ntap * ncc;
  const int nchunk = nmain + (extra_k ? ncc : 0);

  auto issue = [&](int q, int buf) {
    unsigned char* A = smem + buf * BUF;
    unsigned char* B = A + A_BYTES;
    const bool xk = q >= nmain;                      // the shortcut's segment of a data gradient
    const int qq = xk ? q - nmain : q;
    const int ti = qq / ncc, c0 = (qq - ti * ncc) << 6;
    const int tap = xk ? 0 : (int)((taps >> (4 * ti)) & 15);
    const int ky = xk ? 0 : tap / a.KW, kx = xk ? 0 : tap - ky * a.KW;
    const int pad = xk ? 0 : a.pad;
    const bf16_t* in = xk ? a.in2 : a.in;
    const bf16_t* wq = xk ? a.w2 : wsel;
    const size_t wq_plane = xk ? a.w2_plane : wsel_plane;
    const int wtap = (xk || wtaps_one) ? 0 : tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int iy, ix;
      bool ok = rval[j];
      if (a.mode == 0) {
        iy = (roy[j] << a.sshift) + ky - pad;
        ix = (rox[j] << a.sshift) + kx - pad;
      } else {
        const int ty = roy[j] + pad - ky, tx = rox[j] + pad - kx;   // divisible by the stride: the class' tap list guarantees it
        ok = ok && ty >= 0 && tx >= 0;
        iy = ty >> a.sshift;
        ix = tx >> a.sshift;
      }
      ok = ok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const int irow = ok ? (rn[j] * a.IH + iy) * a.IW + ix : a.zero_row;
      const int rl = 32 * wave + 8 * j + lrow;
      const bf16_t* src = in + (size_t)irow * a.Cin + c0 + ((lslot ^ ((rl >> 1) & 7)) << 3);
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16(src + p * a.in_plane, A + p * A_PLANE + (32 * wave + 8 * j) * 128);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 16 * wave + 8 * j + lrow;
      const bf16_t* ws = wq + ((size_t)wtap * a.Cout + tile_n * BN + col) * a.Cin + c0 + ((lslot ^ ((col >> 1) & 7)) << 3);
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16(ws + p * wq_plane, B + p * B_PLANE + (16 * wave + 8 * j) * 128);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  issue(0, 0);
  const int li = lane & 31, lg = lane >> 5;
  const int arow = 32 * wave + li;
  for (int q = 0; q < nchunk; ++q) {
    __syncthreads();                 // chunk q has landed (the barrier's fence drains the LDS-DMA); buffer (q + 1) & 1 is free
    if (q + 1 < nchunk) issue(q + 1, (q + 1) & 1);
    const unsigned char* A = smem + (q & 1) * BUF;
    const unsigned char* B = A + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fa[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(A + p * A_PLANE + img_off(arow, 2 * ks + lg)));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int brow = 32 * c + li;
        bf16x8 fb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(B + p * B_PLANE + img_off(brow, 2 * ks + lg)));
        // smallest products first
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[c], 0, 0, 0);
      }
    }
  }

  // epilogue: register r of a 32 x 32 block = row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32: 128-B row stores
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = tile_n * BN + 32 * c + li;
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rl = 32 * wave + 8 * (i >> 2) + 4 * lg + (i & 3);
      const int off = out_off[rl];
      if (off < 0) continue;
      const size_t o = (size_t)off * a.Cout + col;
      if (shortcut_tile) { a.out2[o] = acc[c][i]; continue; }
      float v = acc[c][i] + bv;
      if (a.res) v += a.res[o];
      if (a.accumulate) v += a.out[o];
      a.out[o] = v;
    }
  }
}

// ============================================================================
// k_stem_wgrad<TG, EXTRA>: dW[tap][co][ci] = sum_pixels dy[pixel][co] * in[pixel @ tap][ci] on the bf16 matrix pipe at fp32
//   accuracy.  Both operands are triples that already exist (the data gradient reads dy, the forward conv read `in`).
//   The reduction runs over PIXELS, so an MFMA operand is eight consecutive pixels of one channel: the [pixel][channel]
//   LDS images above, read with ds_read_b64_tr_b16 (a 4-row x 16-column block delivered column-major: the transpose is free).
//   workgroup = (64 co x 64 ci tile, one kernel row ky = TG taps, one share of the pixels); wave w = the 32 x 32 block
//   (w / 2, w % 2) for the TG taps; K chunk = 16 pixels = ONE MFMA K step: a dy image (6 KB) + TG gathered input images,
//   double-buffered, filled by LDS-DMA while the previous chunk is multiplied.  Every workgroup writes its TG blocks of
//   one slab; k_stem_reduce sums the slabs.
//   EXTRA (ky == 1 only): the shortcut's 1x1 stride-s filter reads the pixel the centre tap reads -- one more accumulator,
//   fed by the centre tap's input image and a second dy image.
// ============================================================================
template <int TG, bool EXTRA>
__global__ __launch_bounds__(256) void k_stem_wgrad(const SWgradArgs a) {
  constexpr int IMG = 3 * 16 * 128;                 // one image: three parts x 16 pixels x 128 B
  constexpr int NIMG = 1 + TG + (EXTRA ? 1 : 0);    // dy, TG inputs, dy2
  constexpr int BUF = NIMG * IMG;
  extern __shared__ __align__(16) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nct = a.Cin >> 6, npair = nct * (a.Cout >> 6);
  const int ngrp = TG == 3 ? 3 : 1;
  int bb = blockIdx.x;
  const int pair = bb % npair; bb /= npair;
  const int ky = bb % ngrp;
  const int split = bb / ngrp;
  const int cot = pair / nct, cit = pair - cot * nct;
  const bool extra = EXTRA && ky == 1;
  const int rows = a.N * a.OH * a.OW;
  const int rbeg = split * a.rows_per_split;
  const int rend = min(rbeg + a.rows_per_split, rows);
  const int nchunk = rbeg < rend ? (rend - rbeg + 15) >> 4 : 0;

  // staging: one wave instruction = 8 pixel rows of one part of one image; lane -> pixel row lane / 8 (+ 8 for the second
  // half of the chunk), piece (lane & 7) ^ swizzle.  The pixel coordinates of both rows advance by 16 per chunk.
  const int lrow = lane >> 3, lslot = lane & 7;
  int pn[2], py[2], px[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = rbeg + 8 * h + lrow;
    pn[h] = r / (a.OH * a.OW);
    const int rem = r - pn[h] * a.OH * a.OW;
    py[h] = rem / a.OW;
    px[h] = rem - py[h] * a.OW;
  }
  auto issue = [&](int q, int buf) {
    unsigned char* base = smem + buf * BUF;
    const int r0 = rbeg + 16 * q;
    constexpr int NINST = NIMG * 6;      // (image, part, half)
#pragma unroll
    for (int ii = 0; ii < (NINST + 3) / 4; ++ii) {
      const int inst = wave + 4 * ii;
      if (inst >= NINST) break;
      const int im = inst / 6, part = (inst % 6) >> 1, h = inst & 1;
      if (EXTRA && !extra && im == NIMG - 1) continue;
      const int rl = 8 * h + lrow;
      const bool valid = r0 + rl < rend;
      const int sw = (lslot ^ ((rl >> 1) & 7)) << 3;
      const bf16_t* src;
      if (im == 0 || (EXTRA && im == NIMG - 1)) {
        const bf16_t* d = im == 0 ? a.dy3 : a.dy23;
        src = d + part * a.dy_plane + (size_t)(valid ? r0 + rl : a.dy_zero_row) * a.Cout + cot * 64 + sw;
      } else {
        const int kx = TG == 3 ? im - 1 : 0;
        const int iy = py[h] * a.stride - a.pad + (TG == 3 ? ky : 0), ix = px[h] * a.stride - a.pad + kx;
        const bool ok = valid && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        src = a.in + part * a.in_plane + (size_t)(ok ? (pn[h] * a.IH + iy) * a.IW + ix : a.zero_row) * a.Cin + cit * 64 + sw;
      }
      glds16(src, base + im * IMG + part * 2048 + h * 1024);
    }
    // the next chunk's pixels
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      px[h] += 16;
      while (px[h] >= a.OW) { px[h] -= a.OW; py[h] += 1; }
      while (py[h] >= a.OH) { py[h] -= a.OH; pn[h] += 1; }
    }
  };

  constexpr int NA = TG + (EXTRA ? 1 : 0);
  f32x16 acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  // transposed fragment reads: lane l -> channel column (l & 31) of the wave's block, pixels 8 (l >> 5) .. + 7
  const int cob = wave >> 1, cib = wave & 1;
  const int kg = lane >> 5, g16 = (lane >> 4) & 1, m = lane & 15, tq = m >> 2, tp = m & 3;
  auto frag = [&](const unsigned char* img, int part, int cb0) -> bf16x8 {
    const int piece = ((cb0 + 16 * g16) >> 3) + (tp >> 1), inb = (tp & 1) << 3;
    const int r0 = 8 * kg + tq, r1 = r0 + 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + part * 2048 + img_off(r0, piece) + inb));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + part * 2048 + img_off(r1, piece) + inb));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  if (nchunk > 0) issue(0, 0);
  for (int q = 0; q < nchunk; ++q) {
    __syncthreads();
    if (q + 1 < nchunk) issue(q + 1, (q + 1) & 1);
    const unsigned char* base = smem + (q & 1) * BUF;
    bf16x8 fa[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = frag(base, p, 32 * cob);
#pragma unroll
    for (int j = 0; j < TG; ++j) {
      bf16x8 fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[p] = frag(base + (1 + j) * IMG, p, 32 * cib);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[j], 0, 0, 0);
      if (EXTRA && j == 1 && extra) {      // the centre tap's input image once more, against the shortcut's dy
        bf16x8 f2[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) f2[p] = frag(base + (NIMG - 1) * IMG, p, 32 * cob);
        acc[TG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[2], fb[0], acc[TG], 0, 0, 0);
        acc[TG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[0], fb[2], acc[TG], 0, 0, 0);
        acc[TG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[1], fb[1], acc[TG], 0, 0, 0);
        acc[TG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[1], fb[0], acc[TG], 0, 0, 0);
        acc[TG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[0], fb[1], acc[TG], 0, 0, 0);
        acc[TG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[0], fb[0], acc[TG], 0, 0, 0);
      }
    }
  }
  const int li = lane & 31, hh = lane >> 5;
  const int taps_total = TG == 3 ? 9 : 1;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    if (j == TG && !extra) continue;
    float* dst = j < TG ? a.slab + ((size_t)split * taps_total + (TG == 3 ? ky * 3 + j : 0)) * a.Cout * a.Cin
                        : a.slab2 + (size_t)split * a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cot * 64 + 32 * cob + 8 * (i >> 2) + 4 * hh + (i & 3);
      dst[(size_t)co * a.Cin + cit * 64 + 32 * cib + li] = acc[j][i];
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

// fp32 -> triples, element for element (n % 8 == 0)
__global__ __launch_bounds__(256) void k_stem_split(const float* __restrict__ src, bf16_t* __restrict__ dst3, size_t plane, size_t n8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const float4 a0 = reinterpret_cast<const float4*>(src)[2 * i], a1 = reinterpret_cast<const float4*>(src)[2 * i + 1];
  u32x4 hh, mm, ll;
  split8(a0, a1, hh, mm, ll);
  *reinterpret_cast<u32x4*>(dst3 + 8 * i) = hh;
  *reinterpret_cast<u32x4*>(dst3 + 8 * i + plane) = mm;
  *reinterpret_cast<u32x4*>(dst3 + 8 * i + 2 * plane) = ll;
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
  // a flat grid, every job its own run of workgroups (as [largest job] x [jobs] 18 k of the 19 k workgroups at cfg 2 came to leave at once)
  int seg = 0;                       // 0..5: the filter jobs (those past njobs are empty), 6: conv0's filter, 7: the zero rows
  while (seg < 7 && blockIdx.x >= a.blk0[seg + 1]) ++seg;
  const size_t idx = (size_t)(blockIdx.x - a.blk0[seg]) * 256 + threadIdx.x;
  const int job = seg < 6 ? seg : a.njobs + (seg - 6);
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
// slab / partial sums into the caller's gradient tensors.  kind 0: one workgroup per (output channel co, tap): thread =
// (ci % 64, quarter of the splits); the slabs are read in their own order (runs of 64 floats), the four quarters meet in
// LDS, and the sums go to PyTorch's [co][ci][tap] order.  (A first version gave every thread ALL splits of nine taps: chains
// of 500+ dependent loads, 120 us for 50 MB.)
// ============================================================================
__global__ __launch_bounds__(256) void k_stem_reduce(const SReduceArgs a) {
  __shared__ float part[4][64];
  // a flat grid, every job its own run of workgroups (as [largest job] x [jobs] two thirds of the 16 k workgroups at cfg 2 came to leave at once)
  int ji = 0;
  while (ji + 1 < a.njobs && blockIdx.x >= a.blk0[ji + 1]) ++ji;
  const SReduceJob& j = a.job[ji];
  const unsigned bx = blockIdx.x - a.blk0[ji];
  const int t = threadIdx.x;
  if (j.kind == 0) {
    const int co = bx / j.taps, tap = bx - co * j.taps;
    if (co >= j.Co) return;
    const size_t total = (size_t)j.taps * j.Co * j.Ci;
    const int cl = t & 63, sub = t >> 6;
    for (int c0 = 0; c0 < j.Ci; c0 += 64) {
      const float* src = j.slab + ((size_t)tap * j.Co + co) * j.Ci + c0 + cl;
      float s0 = 0.f, s1 = 0.f;
      int k = sub;
      // (eight slabs in flight per thread instead of two; the additions keep their order, so the sums are the same bits)
      for (; k + 28 < j.ns; k += 32) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(k + 4 * q) * total];
#pragma unroll
        for (int q = 0; q < 8; q += 2) { s0 += v[q]; s1 += v[q + 1]; }
      }
      for (; k + 4 < j.ns; k += 8) {
        s0 += src[(size_t)k * total];
        s1 += src[(size_t)(k + 4) * total];
      }
      if (k < j.ns) s0 += src[(size_t)k * total];
      part[sub][cl] = s0 + s1;
      __syncthreads();
      if (sub == 0) j.out[((size_t)co * j.Ci + c0 + cl) * j.taps + tap] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
      __syncthreads();
    }
  } else {
    // kinds 1 (conv0: 64 x 32 words per slab, 450 slabs at cfg 2) and 2 (per-sample partials, 2 Co words): a workgroup sums 32 outputs, its
    // eight 32-lane groups every eighth slab with eight loads in flight, the groups meet in LDS in a fixed order.  (One thread per output
    // and all slabs, four in flight: 113 dependent rounds at cfg 2 -- the 48 us this launch took.)
    const size_t nout = j.kind == 1 ? (size_t)64 * 32 : (size_t)2 * j.Co;
    const int o = t & 31, g = t >> 5;
    const size_t idx = (size_t)bx * 32 + o;
    if ((size_t)bx * 32 >= nout) return;
    const bool on = idx < nout;
    const float* src = j.slab + (on ? idx : 0);
    float acc = 0.f;
    int k = g;
    for (; k + 56 < j.ns; k += 64) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(k + 8 * q) * nout];
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; k < j.ns; k += 8) acc += src[(size_t)k * nout];
    part[g >> 1][(g & 1) * 32 + o] = acc;
    __syncthreads();
    if (g != 0 || !on) return;
    const float sum = ((part[0][o] + part[0][32 + o]) + (part[1][o] + part[1][32 + o])) + ((part[2][o] + part[2][32 + o]) + (part[3][o] + part[3][32 + o]));
    if (j.kind == 1) {
      const int co = (int)(idx >> 5), kk = (int)(idx & 31);
      if (kk < j.Ci) j.out[co * j.Ci + kk] = sum;
      else if (kk == j.Ci) j.out2[co] = sum;
    } else {
      if (idx < (size_t)j.Co) j.out[idx] = sum;
      else j.out2[idx - j.Co] = sum;
    }
  }
}

}  // namespace

// ============================================================================
// launchers
// ============================================================================
void launch_stem_conv(const SConvArgs& a, hipStream_t s) {
  const int mt = a.cls_tile0[a.nclass];
  static bool attr[MAX_DEVICES] = {};
  allow_full_lds(reinterpret_cast<const void*>(k_stem_conv), attr);
  const size_t lds = 2 * (3 * 128 * 128 + 3 * 64 * 128) + 512;
  const int ntn = (a.Cout / 64) * ((a.mode == 0 && a.w2 != nullptr) ? 2 : 1);
  hipLaunchKernelGGL(k_stem_conv, dim3(mt * ntn), dim3(256), lds, s, a);
}

void launch_stem_wgrad(const SWgradArgs& a, hipStream_t s) {
  const int taps = a.KH * a.KW;
  const int grid = (a.Cin / 64) * (a.Cout / 64) * (taps == 9 ? 3 : 1) * a.nsplit;
  const int nimg = 1 + (taps == 9 ? 3 : 1) + (a.dy23 ? 1 : 0);
  const size_t lds = (size_t)2 * nimg * 3 * 16 * 128;
#define STEM_WG(T, EX)                                                                   \
  {                                                                                       \
    static bool attr[MAX_DEVICES] = {};                                                   \
    allow_full_lds(reinterpret_cast<const void*>(k_stem_wgrad<T, EX>), attr);             \
    hipLaunchKernelGGL((k_stem_wgrad<T, EX>), dim3(grid), dim3(256), lds, s, a);           \
  }
  if (taps == 9 && a.dy23) STEM_WG(3, true)
  else if (taps == 9) STEM_WG(3, false)
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
  static int cap = -2;      // NODE_TUNE_STEM_GNCB: upper limit of the channel block (A/B measurements)
  if (cap == -2) { const char* e = getenv("NODE_TUNE_STEM_GNCB"); cap = e ? atoi(e) : -1; }
  while (cap > 0 && cb > cap && cb / 2 >= cpg && cb / 2 >= 8) cb /= 2;
  // the kernels index channels with `t & (CB - 1)`, run C / CB blocks per sample and CB / cpg whole groups per block:
  // a block that does not divide C leaves channels unwritten, a group that straddles blocks gets wrong statistics
  // (filters = 192, 384, ...: cpg = 6, 12).  Such shapes are refused (check_stem_shape) and run the module sequence.
  if (C % cb != 0 || cb % cpg != 0) return 0;
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

void launch_stem_prep(const SPrepArgs& a0, hipStream_t s) {
  SPrepArgs a = a0;
  unsigned at = 0;
  for (int i = 0; i < 6; ++i) {
    a.blk0[i] = at;
    if (i < a.njobs) at += (unsigned)(((size_t)a.job[i].taps * a.job[i].Cout * a.job[i].Cin + 255) / 256);
  }
  a.blk0[6] = at; at += (28 * 64 + 255) / 256;                              // conv0's filter
  a.blk0[7] = at; at += (unsigned)(((size_t)a.nzero * 3 * 4096 + 255) / 256);    // zero rows
  a.blk0[8] = at;
  hipLaunchKernelGGL(k_stem_prep, dim3(at), dim3(256), 0, s, a);
}
void launch_stem_from_nchw(const float* src, float* dst_nhwc, bf16_t* dst3, size_t plane, int N, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(k_stem_from_nchw, dim3(N * (C / 64) * ((HW + 63) / 64)), dim3(256), 0, s, src, dst_nhwc, dst3, plane, C, HW);
}
void launch_stem_split(const float* src, bf16_t* dst3, size_t plane, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_stem_split, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, src, dst3, plane, n / 8);
}
void launch_stem_to_nchw(const float* src_nhwc, float* dst, int N, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(k_stem_to_nchw, dim3(N * (C / 64) * ((HW + 63) / 64)), dim3(256), 0, s, src_nhwc, dst, C, HW);
}
void launch_stem_reduce(const SReduceArgs& a0, hipStream_t s) {
  SReduceArgs a = a0;
  unsigned at = 0;
  for (int i = 0; i < a.njobs; ++i) {
    const SReduceJob& j = a.job[i];
    a.blk0[i] = at;
    at += (unsigned)(j.kind == 0 ? j.Co * j.taps : j.kind == 1 ? 64 : (2 * j.Co + 31) / 32);
  }
  a.blk0[a.njobs] = at;
  if (at == 0) return;
  hipLaunchKernelGGL(k_stem_reduce, dim3(at), dim3(256), 0, s, a);
}

}  // namespace node
