// The residual stem in front of the ODE block (model.py:167-178, ResBlock model.py:284-310) as gfx950 kernels:
// internal declarations shared by kernels_stem.hip (kernels + launchers) and stem_api.hip (the C ABI entry points).
//
// Data inside the stem's workspace is NHWC.  A tensor that FEEDS a convolution (an activation behind GroupNorm + ReLU, a
// gradient in front of a data-gradient convolution) is stored as an exact three-way bf16 split ("triples"):
//     T[3][rows + 1][C]  bf16,   x = T[0] + T[1] + T[2]  exactly,   row `rows` of every plane is zero
// (the zero row is where taps that fall into the padding point: branch-free gathers).  The convolutions then are bf16
// MFMA GEMMs with six part products per fp32 product (hh, hm, mh, mm, hl, lh; fp32 accumulation) -- fp32 accuracy at
// 3/8 of the fp32 matrix instructions' time -- and their operands are staged by plain copies (no conversion in the
// loop).  Tensors that only pointwise kernels and the weight gradient read stay fp32.
#pragma once
#include "node_internal.h"

namespace node {

typedef unsigned short bf16_t;     // storage type of one bf16 part

// ----------------------------------------------------------------------------
// k_stem_conv: forward convolution / data gradient as a gather GEMM on triples
// ----------------------------------------------------------------------------
// M space: the pixels of the tensor being WRITTEN, cut into up to four classes (the stride-2 data gradient: pixels of
// one (row parity, column parity) class share the taps that reach them, so no MFMA multiplies zeros); a class is an
// [N][ch][cw] grid of pixels (oy, ox) = (step cy + py, step cx + px).  K space: (tap, input channel).
struct SConvArgs {
  const bf16_t* in;        // triples [3][in_rows + 1][Cin]
  size_t in_plane;         // elements per plane ((in_rows + 1) * Cin)
  int zero_row;            // = in_rows
  const bf16_t* w;         // triples [3][KH * KW][Cout][Cin]  (forward: [co][ci]; data gradient: [ci][co] -- "Cout" is always the
  size_t w_plane;          //          channel count of the tensor being written, "Cin" the reduction)
  float* out;              // fp32 NHWC [N * OH * OW][Cout]
  const float* res;        // nullable: added to the result (residual connection), same layout as out
  const float* bias;       // nullable [Cout]
  int accumulate;          // out += result
  // The block's shortcut convolution (1x1, same stride, no padding) rides in the 3x3's launch:
  //   forward (mode 0): it reads the pixel the 3x3's centre tap reads, so it is a second set of column tiles
  //     (tile_n >= Cout / 64) with ONE tap, filter w2, output out2 -- no launch of its own, the A rows already in L2;
  //   data gradient (mode 1): its gradient lands on the pixels of class (even row, even column) only: one more K segment of
  //     that class, gathered from in2 (the shortcut's output gradient, same shape as `in`) with filter w2
  const bf16_t* w2;        // nullable; triples [3][1][Cout][Cin] (forward) / [3][1][Cin_f][Cout_f] (data gradient)
  size_t w2_plane;
  float* out2;             // forward only
  const bf16_t* in2;       // data gradient only
  int N, IH, IW;           // spatial size of `in`
  int OH, OW;              // spatial size of `out`
  int Cin, Cout;
  int KH, KW, sshift, pad; // stride = 1 << sshift
  int mode;                // 0 forward: in pixel = o * stride + k - pad;  1 data gradient: in pixel = (o + pad - k) / stride
  int nclass, step;
  int cls_py[4], cls_px[4], cls_h[4], cls_w[4];
  int cls_tile0[5];        // first M tile of each class (prefix sums), [nclass] = total
  int cls_ntap[4];
  unsigned long long cls_taps[4];   // taps (ky * KW + kx) that reach the class, four bits each, first tap in the low bits
};
void launch_stem_conv(const SConvArgs& a, hipStream_t s);

// ----------------------------------------------------------------------------
// k_stem_wgrad: weight gradient dW[tap][co][ci] = sum_rows dy[row][co] * in[row @ tap][ci]; both operands triples, bf16
// MFMA with transposed LDS reads, split-K slabs
// ----------------------------------------------------------------------------
struct SWgradArgs {
  const bf16_t* dy3;       // triples [3][rows + 1][Cout], rows = N * OH * OW
  size_t dy_plane;
  int dy_zero_row;
  const bf16_t* dy23;      // nullable: the shortcut's output gradient (same shape) -- its 1x1 stride-s filter reads the pixel the
                           // centre tap of the 3x3 pad-1 filter reads, so it rides as one more accumulator of the ky = 1 workgroups
  const bf16_t* in;        // triples [3][N * IH * IW + 1][Cin]
  size_t in_plane;
  int zero_row;
  float* slab;             // [nsplit][KH * KW][Cout][Cin]
  float* slab2;            // [nsplit][Cout][Cin] (with dy23)
  int N, IH, IW, OH, OW, Cin, Cout, KH, KW, stride, pad;
  int nsplit, rows_per_split;   // rows_per_split % 16 == 0
};
void launch_stem_wgrad(const SWgradArgs& a, hipStream_t s);

// ----------------------------------------------------------------------------
// first layer: nn.Conv2d(in_ch, 64, 3, 1) on the NCHW input, K = 9 in_ch <= 27 (fp32 MFMA)
// ----------------------------------------------------------------------------
void launch_stem_conv0_fwd(const float* x, const float* w0t /*[28][64]*/, const float* bias, float* h0 /*NHWC*/, int N, int Cin,
                           int H, int W, hipStream_t s);
// slab [nsplit][64][32]: column k < 9 Cin = dW0[co][k], column 9 Cin = the bias gradient
void launch_stem_conv0_wgrad(const float* x, const float* dh0, float* slab, int N, int Cin, int H, int W, int nsplit,
                             int rows_per_split, hipStream_t s);

// ----------------------------------------------------------------------------
// GroupNorm + ReLU passes (one workgroup per (sample, block of CB channels), the block held in LDS)
// ----------------------------------------------------------------------------
struct SGnArgs {
  const float* h;          // fp32 NHWC [N][HW][C]: the GroupNorm's input
  const float* gamma;
  const float* beta;
  float* stats;            // [N][G][2] (mean, 1/sigma): written by the forward, read by the backward
  bf16_t* a3;              // forward: triples of relu(GN(h))
  size_t a_plane;
  // backward
  const float* da;         // fp32 NHWC gradient of the activation
  float* dh;               // nullable fp32 NHWC gradient of h
  bf16_t* dh3;             // nullable triples of the same
  size_t dh_plane;
  float* gpart;            // [N][2][C] per-sample (dgamma, dbeta)
  int N, HW, C, cpg, CB;
  float eps;
};
int stem_gn_cb(int HW, int C, int cpg);      // channel block: the largest power of two whose block fits the LDS budget
void launch_stem_gn_fwd(const SGnArgs& a, hipStream_t s);
void launch_stem_gn_bwd(const SGnArgs& a, hipStream_t s);

// ----------------------------------------------------------------------------
// layout / preparation / reductions
// ----------------------------------------------------------------------------
// filters -> triples in both operand layouts: wf[3][taps][co][ci], wd[3][taps][ci][co]; conv0's filter -> w0t[28][64] fp32;
// and the zero rows of the triples tensors
struct SPrepJob { const float* w; bf16_t* wf; bf16_t* wd; int Cout, Cin, taps; };
struct SPrepArgs {
  SPrepJob job[6];
  int njobs;
  const float* w0; float* w0t; int k0;            // conv0: [64][k0] -> [28][64], rows >= k0 zero (nullable)
  bf16_t* zero[10]; size_t zero_plane[10]; int zero_c[10]; int nzero;   // plane p of tensor i: zero[i] + p * zero_plane[i] .. + zero_c[i]
  unsigned blk0[9];                               // (filled by launch_stem_prep) first workgroup of job 0..5, of the conv0 job, of the zero job, and the end
};
void launch_stem_prep(const SPrepArgs& a, hipStream_t s);
// NCHW fp32 -> NHWC fp32 (nullable) + triples (nullable); NHWC fp32 -> NCHW fp32
void launch_stem_from_nchw(const float* src, float* dst_nhwc, bf16_t* dst3, size_t plane, int N, int C, int HW, hipStream_t s);
void launch_stem_to_nchw(const float* src_nhwc, float* dst, int N, int C, int HW, hipStream_t s);
void launch_stem_split(const float* src, bf16_t* dst3, size_t plane, size_t n /* % 8 == 0 */, hipStream_t s);   // fp32 -> triples
// sums over split-K slabs / per-sample partials into the caller's gradient tensors
//   kind 0: slab [ns][taps][Co][Ci] -> dW [Co][Ci][taps] (PyTorch layout); kind 1: conv0 slab [ns][64][32] -> dW0 [64][k0] + db [64];
//   kind 2: GroupNorm partials [ns = N][2][C] -> dgamma [C], dbeta [C]
struct SReduceJob { const float* slab; float* out; float* out2; int kind, ns, Co, Ci, taps; };
struct SReduceArgs { SReduceJob job[12]; int njobs; unsigned blk0[13]; };     // blk0: (filled by launch_stem_reduce) first workgroup of every job, and the end
void launch_stem_reduce(const SReduceArgs& a, hipStream_t s);

}  // namespace node
