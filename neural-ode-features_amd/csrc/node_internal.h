// Internal declarations shared by the HIP translation units of libnode_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace node {

// ----------------------------------------------------------------------------
// Geometry
// ----------------------------------------------------------------------------
constexpr int KCH = 32;          // input channels per K chunk of the implicit GEMM
constexpr int BN = 64;           // output-channel tile (two 32-wide MFMA column tiles)
constexpr int WG_THREADS = 256;    // wgrad / pointwise kernels

struct Dims {
  int N, C, H, W, HW, G, cpg;
  float eps;
  int Wp, Hp, SLOTS;   // halo-padded image: (H+2) x (W+2) slots
  int MARGIN;          // Wp + 1 zero slots before/after so every tap offset stays in bounds
  int BNE;             // effective N tile: largest multiple of cpg <= 64 (groups never straddle tiles)
  int ntile;           // ceil(C / BNE)
  int nchunk;          // ceil(C / 32)
  int BM;              // rows per M tile: 128 or 256
  int S;               // whole samples per M tile
  int wino;            // conv kernel: 0 direct, 1 Winograd F(2,3) along the rows (even W), 2 Winograd F(2x2,3x3) (even H, W; 128-pixel tiles)
  int wgrad_wino;      // weight gradient accumulated in the same Winograd domain (k_wgrad_w)
  int wut;          // tiles per unit of the 2-D Winograd wgrad kernel (wgrad_wino == 2)
  int tiny;         // latency path (kernels_tiny.hip): input channels per workgroup of k_tiny_conv_gn, or 0 -- forward solves of
                    // batches so small that an evaluation is bound by its dependent launches (bs = 1 census, evaluate.py:97-142)
  int small;        // the throughput tiles give a grid under 32 workgroups: inference solves run k_conv3x3_small (32 px x 32 columns
                    // per workgroup, four-way split K) + a GroupNorm pass instead (latency regime, evaluate.py:97-142)
  int wino4;        // geometry fits the F(4x4,3x3) pipeline (wino4.h): 8x8 images, C % 64 == 0, cpg | 16 -- or 16x16 images,
                    // C % 128 == 0, cpg | 16 or cpg == 32 (w4q = 4 quadrants per image, each a virtual sample of the GEMMs).
                    // Any batch: the component GEMMs run on N8 = w4q N rounded up to 8 samples (their rows are independent; the
                    // weight gradient, which sums over rows, sees zero rows for the padding samples)
  int w4q, N8;
                    // Whether a solve USES it depends on its tolerance (Solver::w4)
  int csplit;       // 2-D Winograd conv on images larger than its 128-pixel tile: workgroups per sample (0: whole samples per tile).
                    // The conv then writes its raw output and GroupNorm runs as a pointwise pass (k_combine_gn / k_gn_bwd)
  int mtiles;          // ceil(N / S)
  // pointwise slab (combine+GN kernels)
  int cs;              // channels per slab (multiple of lcm(cpg,4))
  int nslab;
  // wgrad
  int RB;              // image rows per staging band
  int nbands;          // bands per sample
  int nsplit;          // K-split factor
  int wgrad_pair;      // both layers' weight gradients in one k_wgrad_w2 launch (nsplit then fills the chip with two problems)
  size_t P;            // flat parameter count
  size_t numel;        // N*C*H*W
};

// theta-segment internal layout (a permutation of the PyTorch flat layout):
//   [g1 C][b1 C][Wc1 9*C*C as [tap][ci][co]][Wt1 9*C as [tap][co]][cb1 C]
//   [g2 C][b2 C][Wc2 ...][Wt2 ...][cb2 C][g3 C][b3 C]
struct ThetaLayout {
  size_t g[3], b[3], wc[2], wt[2], cb[2];
};
__host__ __device__ inline ThetaLayout theta_layout(int C) {
  ThetaLayout L;
  size_t o = 0, cc = (size_t)C;
  L.g[0] = o; o += cc; L.b[0] = o; o += cc;
  L.wc[0] = o; o += 9 * cc * cc; L.wt[0] = o; o += 9 * cc; L.cb[0] = o; o += cc;
  L.g[1] = o; o += cc; L.b[1] = o; o += cc;
  L.wc[1] = o; o += 9 * cc * cc; L.wt[1] = o; o += 9 * cc; L.cb[1] = o; o += cc;
  L.g[2] = o; o += cc; L.b[2] = o; o += cc;
  return L;
}

// ----------------------------------------------------------------------------
// Device-resident step controller state (one per solve, lives in the workspace)
// ----------------------------------------------------------------------------
struct Ctrl {
  double t;        // solver time at the start of the current step
  double dt;       // step size of the current step
  double t_prev;   // start of the step just finished (== t unless accepted)
  double dt_used;  // dt of the step just finished
  float ratio[4];  // mean squared error ratio per state segment (y, a, adj_t, adj_params)
  float h0;        // initial-step probe size
  float d0, d1;    // initial-step norms (max over segments)
  int accept;      // last step accepted?
  int status;      // 0 or NODE_ERR_*
  int n_acc, n_rej;
  // scalar (adj_t) segment of the augmented state
  float ts_cur;    // current value
  float ts_new;    // end-of-step value
  float ts_k[7];   // stage derivatives
  float ts_y0_prev; // value at the start of the last accepted step (dense output)
  float ts_f0_prev; // derivative at the start of the last accepted step
  float pad;
  // device-resident stepping: the host enqueues steps without reading anything back; once `done` is set every
  // kernel of a later step returns at once, so steps enqueued past the end of the interval cost launch time only
  int done;        // the interval reached its last target time (or stopped with `status`)
  int step_idx;    // steps tried in this interval (accepted + rejected)
  int j;           // next target time not yet passed
  int j0, j1;      // targets [j0, j1) were passed by the step just finished (dense output is emitted for them)
  int pad2;
  double first_dt; // step size of the interval's first step
};

enum TimeMode { TM_STAGE = 0, TM_PROBE = 1 };

// how an eval derives its time:  t_real = tsign * ((float)t + (mode==TM_STAGE ? alpha*(float)dt : h0))
struct EvalTime {
  const Ctrl* ctrl;
  float alpha;
  float tsign;
  int mode;
};
__device__ inline float eval_time(const EvalTime& et) {
  float tf = (float)et.ctrl->t;
  float inc = et.mode == TM_STAGE ? et.alpha * (float)et.ctrl->dt : et.ctrl->h0;
  return et.tsign * (tf + inc);
}

// linear combination  y + scale * sum_j coef[j] * k[j]
enum ScaleMode { SC_ABS = 0, SC_DT = 1, SC_H0 = 2 };
struct Comb {
  const float* y;
  const float* k[7];
  float coef[7];
  int nk;
  int scale_mode;
};
__device__ inline float comb_scale(const Comb& c, const Ctrl* ctrl) {
  return c.scale_mode == SC_ABS ? 1.0f : (c.scale_mode == SC_DT ? (float)ctrl->dt : ctrl->h0);
}

// Dense output: quartic through (y0, y1, y_mid, f0, f1) of the last accepted step (`_interp_fit_dopri5` +
// `_interp_evaluate`, power form like upstream); k[1] has weight zero everywhere.
__device__ constexpr float DP_CMID_F[7] = {
    (float)(6025192743.0 / 30085553152.0 / 2.0), 0.f, (float)(51252292925.0 / 65400821598.0 / 2.0),
    (float)(-2691868925.0 / 45128329728.0 / 2.0), (float)(187940372067.0 / 1594534317056.0 / 2.0),
    (float)(-1776094331.0 / 19743644256.0 / 2.0), (float)(11237099.0 / 235043384.0 / 2.0)};
__device__ inline float interp_one(float y0, float y1, const float* k, float dt, float x) {
  float s = (dt * DP_CMID_F[0]) * k[0];
#pragma unroll
  for (int j = 2; j < 7; ++j) s += (dt * DP_CMID_F[j]) * k[j];
  const float ymid = y0 + s;
  const float f0 = k[0], f1 = k[6];
  const float ca = (-2.f * dt) * f0 + (2.f * dt) * f1 + -8.f * y0 + -8.f * y1 + 16.f * ymid;
  const float cb = (5.f * dt) * f0 + (-3.f * dt) * f1 + 18.f * y0 + 14.f * y1 + -32.f * ymid;
  const float cc = (-4.f * dt) * f0 + dt * f1 + -11.f * y0 + -5.f * y1 + 16.f * ymid;
  const float cd = dt * f0;
  const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
  return ca * x4 + cb * x3 + cc * x2 + cd * x + y0;
}

// Kernels that stage more than 64 KB of LDS need the opt-in attribute, and the attribute is PER DEVICE: a process
// that drives several GPUs must set it on each (one flag per device ordinal; benign if two threads race to set it).
constexpr int MAX_DEVICES = 64;
inline void allow_full_lds(const void* fn, bool (&done)[MAX_DEVICES]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < MAX_DEVICES;
  if (tracked && done[dev]) return;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    (void)hipGetLastError();     // (a kernel with static __shared__ arrays: refused; do not leave the error for the next launch check)
  if (tracked) done[dev] = true;
}

// thread-local error text of the C ABI (node_last_error), set from any translation unit; returns `code`
int set_error(int code, const char* msg);

// ----------------------------------------------------------------------------
// kernel launchers (defined in the kernels_*.hip units)
// ----------------------------------------------------------------------------
// layout
void launch_nchw_to_nhwc(const Dims& d, const float* src, float* dst, hipStream_t s);
void launch_nhwc_to_nchw(const Dims& d, const float* src, float* dst, hipStream_t s);
void launch_pack_weights(const Dims& d, const float* w /*[C][C+1][3][3]*/, float* packed, int dgrad, hipStream_t s);
void launch_tmap(const Dims& d, const float* w, float* tmap /*[HW][C]*/, hipStream_t s);
void launch_wtime(const Dims& d, const float* w, float* wtime /*[9][C]*/, hipStream_t s);
// tmap of both layers (+ wtime of both when wtime1 != nullptr) in one launch
void launch_time_prep(const Dims& d, const float* w1, const float* w2, float* tmap1, float* tmap2, float* wtime1, float* wtime2,
                      float* const* zero, const size_t* zero_n, int nzero /*<= 8: regions to zero-fill*/, hipStream_t s);
void launch_theta_to_torch(const Dims& d, const float* theta_int, float* flat, hipStream_t s);

// pointwise / reductions
struct CombineGnArgs {
  Comb comb;
  const Ctrl* ctrl;
  float* y_out;        // nullable
  float* act_out;      // relu(GN(y_i))
  float* xhat_out;     // nullable
  float* rstd_out;     // nullable [N][G]
  const float* gamma;
  const float* beta;
  int relu;            // 1: act = relu(GN(y_i)); 0: act = GN(y_i)   (split-conv GroupNorm pass)
  float osign;         // output multiplier (1 for the stage combine)
};
void launch_combine_gn(const Dims& d, const CombineGnArgs& a, hipStream_t s);

// Masked column sums of one sample's [HW][cols] tile held in LDS (row stride `ld`):
//   out[tap][c] = sum of v[p][c] over the pixels p whose tap neighbour p + tap lies inside the image
// (the conv-bias gradient is tap 4; t x these are the time-channel weight gradients; their contraction with the
// time-channel weights is d f / d t).  Nine inclusion-exclusion terms from: total, first/last row, first/last
// column, four corners.  Producers of dz call it on the tile they are about to store, which saves a launch and
// a pass over dz per layer (k_colsum is what runs otherwise).  `flg[p]`: bit0 first row, bit1 last row, bit2
// first column, bit3 last column.  Threads = ncols x ngrp (column fastest); `red` holds ngrp * 9 * ncols floats.
// Call with all threads of the workgroup; contains two barriers.
__device__ inline void masked_colsum_tile(const float* tile, int ld, int HW, const unsigned char* flg, int ncols,
                                          int ngrp, int tid, float* red, float* out, int out_ld) {
  const int cl = tid % ncols, pg = tid / ncols;
  if (pg < ngrp) {
    float T = 0.f, rf = 0.f, rl = 0.f, cf = 0.f, cl_ = 0.f, k00 = 0.f, k01 = 0.f, k10 = 0.f, k11 = 0.f;
    for (int p = pg; p < HW; p += ngrp) {
      const float v = tile[p * ld + cl];
      const int f = flg[p];
      T += v;
      if (f & 1) rf += v;
      if (f & 2) rl += v;
      if (f & 4) cf += v;
      if (f & 8) cl_ += v;
      if ((f & 5) == 5) k00 += v;
      if ((f & 9) == 9) k01 += v;
      if ((f & 6) == 6) k10 += v;
      if ((f & 10) == 10) k11 += v;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {   // tap (kh, kw) excludes the first (k == 0) / last (k == 2) row and column
      const int kh = t / 3, kw = t % 3;
      float o = T;
      if (kh == 0) o -= rf;
      if (kh == 2) o -= rl;
      if (kw == 0) o -= cf;
      if (kw == 2) o -= cl_;
      if (kh == 0 && kw == 0) o += k00;
      if (kh == 0 && kw == 2) o += k01;
      if (kh == 2 && kw == 0) o += k10;
      if (kh == 2 && kw == 2) o += k11;
      red[(pg * 9 + t) * ncols + cl] = o;
    }
  }
  __syncthreads();
  if (pg == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float o = red[t * ncols + cl];
      for (int r = 1; r < ngrp; ++r) o += red[(r * 9 + t) * ncols + cl];
      out[(size_t)t * out_ld + cl] = o;
    }
  }
  __syncthreads();
}

struct GnBwdArgs {     // cotangent g = csign * (a + scale*sum coef*k);  dz = GN_bwd(g)
  Comb comb;
  const Ctrl* ctrl;
  float csign;
  float* a_out;        // nullable: combined adjoint state
  const float* xhat;   // [N,HW,C]
  const float* rstd;   // [N][G]
  const float* gamma;
  float* dz_out;
  float* gpart;        // [N][2][C] per-sample (dgamma, dbeta) partials
  float* spart;        // nullable: [N][9][C] masked column sums of dz_out (see masked_colsum_tile)
  const float* mask_act;   // nullable: g is zeroed where this activation is <= 0 (ReLU mask of a split-conv data gradient)
  float osign;         // output multiplier (1 for GroupNorm-3's backward)
};
void launch_gn_bwd(const Dims& d, const GnBwdArgs& a, hipStream_t s);

struct ErrSeg {
  const float* y0;
  float* y1;           // read, or written when compute_y1
  const float* k[7];
  size_t n;
  int compute_y1;
};
constexpr int ERR_BLOCKS = 512;
// all segments of the state in one launch (grid.y = segment)
void launch_error_norm(const ErrSeg* segs, float* const* partial /*[ERR_BLOCKS] each*/, int nseg, const Ctrl* ctrl, float rtol, float atol, hipStream_t s);
// ... with the step controller as the LAST-ARRIVING workgroup of the same launch (round 6: one launch and one kernel boundary less per
// step): `arrive` = a device word that is zero before the launch (the last workgroup zeroes it again)
struct StepCtlArgs;
void launch_error_norm_ctl(const ErrSeg* segs, float* const* partial, int nseg, const StepCtlArgs& ctl, unsigned* arrive, hipStream_t s);

struct StepCtlArgs {
  Ctrl* ctrl;
  const float* partial[3];
  double numel[3];
  int nseg;            // tensor segments (1 fwd, 3 aug: y, a, theta)
  int has_scalar;      // adj_t segment present
  float rtol, atol;
  const double* targets;   // device: times (in solver orientation) at which output is wanted, increasing
  int n_targets;
  const double* forced;    // device, nullable: replay mode -- step k takes forced[k] (the last one repeats)
  int n_forced;
  double* dt_log;          // device, nullable: dt tried at step k (negative: rejected)
  int dt_log_cap;
  int interp_scalar;       // aug: when the last target is passed, ts_cur <- dense output at that time
  struct W4Scales* w4sc;   // nullable: the solve's cotangent-side fp16-pair scale follows the data (wino4.h): update it, repeat a step that overflowed
  const float* gbuf;       // nullable: GLOBAL-NORM mode (include/node_hip.h): [0 .. nseg) sums of (err / tol)^2 over all ranks, [3] the sum of
  float gworld;            // the ranks' scalar-segment ratios, [4] > 0: some rank asks for the step to be repeated; gworld = ranks
};
// global-norm mode: this rank's sums -> gbuf (k_norm_pack), all-reduced by the caller's hook before the controller reads them
struct NormPackArgs {
  const Ctrl* ctrl;
  const float* partial[3];
  int nseg, has_scalar;
  int mode;                // 0: step (partial = [ERR_BLOCKS]); 1 / 2: initial step phase 0 / 1 (partial = [ERR_BLOCKS][2])
  float rtol, atol;
  const struct W4Scales* w4sc;
  float* gbuf;             // [8]
};
void launch_norm_pack(const NormPackArgs& a, hipStream_t s);
void launch_step_controller(const StepCtlArgs& a, hipStream_t s);

// After the controller, on the device (no host decision):
//   forward solve: dense output for the targets [j0, j1) of the step just finished, straight into y_out (NCHW);
//   then, if the step was accepted and the interval goes on, y <- y1 and k0 <- k6 (FSAL) for every segment.
struct EmitArgs {
  const Ctrl* ctrl;
  const double* targets;
  const float* y0; const float* y1; const float* k[7];   // NHWC
  float* y_out;        // [n_targets][N][C][HW] (NCHW), slot j <-> targets[j]
};
void launch_emit_outputs(const Dims& d, const EmitArgs& a, hipStream_t s);
void launch_emit_flat(const EmitArgs& a, size_t n, hipStream_t s);       // flat state: out[j][i], no layout change
void launch_flat_time(const EvalTime& et, float* out, hipStream_t s);
void launch_flat_scalar(Ctrl* ctrl, int which, const float* src, float scale, int accumulate, hipStream_t s);
struct CommitArgs {
  const Ctrl* ctrl;
  const double* targets;   // aug: the interval's end time (targets[0]) for the in-place dense output
  int nseg;
  float* y[3]; const float* y1[3]; float* k0[3]; const float* k6[3];
  const float* k[3][7];    // aug only: all stage derivatives (dense output at the interval's end)
  size_t n[3];
  int interp_final;        // aug: when the interval's end is passed, y[s] <- dense output at targets[0] (in place)
};
void launch_commit(const CommitArgs& a, hipStream_t s);
void launch_set_interval(Ctrl* ctrl, double t, double dt, hipStream_t s);   // new interval: t, dt, done = 0, counters of the interval
void launch_set_target(double* targets, double t_end, hipStream_t s);       // targets[0] = t_end without host staging
}  // namespace node
struct node_step_record;
namespace node {
void launch_export_record(const Ctrl* ctrl, node_step_record* rec, float* miss_flag, int expect_steps, hipStream_t s);

// initial step (Hairer)
struct InitSeg { const float* y0; const float* f0; const float* f1; size_t n; };
void launch_init_norms(const InitSeg* segs, float* const* partial /*[ERR_BLOCKS][2] each*/, int nseg, float rtol, float atol, int phase, hipStream_t s);
struct InitCtlArgs;
void launch_init_norms_ctl(const InitSeg* segs, float* const* partial, int nseg, const InitCtlArgs& ctl, unsigned* arrive, hipStream_t s);
struct InitCtlArgs {
  Ctrl* ctrl;
  const float* partial[3];
  double numel[3];
  int nseg, has_scalar, phase;
  float rtol, atol;
  const float* gbuf;       // nullable: global-norm mode -- phase 0: [2 seg], [2 seg + 1] the segment's two sums, [6], [7] the scalar segment's
  float gworld;            // squares; phase 1: [seg], [3]; all summed over the ranks
};
void launch_init_controller(const InitCtlArgs& a, hipStream_t s);
void launch_set_ctrl(Ctrl* ctrl, double t, double dt, int reset_counters, hipStream_t s);
void launch_set_scalar_state(Ctrl* ctrl, float v, int which, hipStream_t s);

void launch_axpy(float* y, const float* x, float alpha, size_t n, hipStream_t s);  // y += alpha*x
// ctrl->ts_cur -= sign * <a, b>;  *out_dot = sign * <a, b>
void launch_dot_sub_scalar(Ctrl* ctrl, const float* a, const float* b, size_t n, float sign, float* partial, float* out_dot, hipStream_t s);
void launch_fill(float* p, float v, size_t n, hipStream_t s);
// dst[q][i] += coef[q] * src[i] for q < nt: a cotangent fanned out over the tensors it feeds (backprop through a step)
struct ScatterArgs { const float* src; float* dst[8]; float coef[8]; int nt; size_t n; };
void launch_scatter_axpy(const ScatterArgs& a, hipStream_t s);
void launch_lincomb(const Comb& c, const Ctrl* ctrl, float* out, size_t n, hipStream_t s);
void launch_copy_scalar_out(const Ctrl* ctrl, float* dst, hipStream_t s);

// conv implicit GEMM
enum ConvMode { CM_FWD_GN_RELU = 0, CM_FWD_GN = 1, CM_BWD_RELU_GN = 2 };
struct ConvArgs {
  const float* in;        // [N,HW,C] NHWC: activation (fwd) or dz (dgrad)
  const float* wpacked;   // [ntile][nchunk][9][32][64]
  int mode;
  // fwd epilogue
  const float* bias;      // [C]
  const float* tmap;      // [HW][C]
  EvalTime et;
  const float* gamma;     // [C]
  const float* beta;      // [C]   (fwd)
  float osign;            // output multiplier
  float* out;             // fwd: post-affine (ReLU'd for CM_FWD_GN_RELU) ; bwd: dx
  float* xhat_out;        // fwd nullable
  float* rstd_out;        // fwd nullable [N][G]
  // bwd epilogue
  const float* act;       // [N,HW,C] post-ReLU activation of the layer below (mask)
  const float* xhat;      // [N,HW,C]
  const float* rstd;      // [N][G]
  float* gpart;           // [mtiles][2][C]
  float* spart;           // bwd, nullable: [N][9][C] masked column sums of the output (see masked_colsum_tile)
  float* raw_out;         // split mode (Dims::csplit): conv + bias + t*tmap (fwd) or the raw data gradient (bwd), no GroupNorm
  unsigned long long* stamps;  // diagnostics only (NODE_STAMPS builds); nullptr otherwise
  int ablate;                  // diagnostics only (NODE_STAMPS builds): timing-only ablation bits
};
void launch_conv(const Dims& d, const ConvArgs& a, hipStream_t s);
// tuning (tools/kbench.hip): kernel variant override (<0: production choice)
extern int g_wgrad_variant;
extern int g_conv_bm;      // force the conv M tile (64 / 128) where the geometry allows; <= 0: heuristic
extern int g_wgrad_wino;   // 0 / 1: force the direct / Winograd wgrad kernel; < 0: heuristic
extern int g_conv_wino;    // 0 / 1 / 2: force the direct / 1-D Winograd / 2-D Winograd conv kernel; < 0: heuristic
size_t conv_lds_bytes(const Dims& d, int mode);
size_t conv_packed_elems(const Dims& d);
void launch_pack_weights_w(const Dims& d, const float* w, float* packed, int dgrad, hipStream_t s);
void launch_pack_weights_w2(const Dims& d, const float* w, float* packed, int dgrad, hipStream_t s);
void launch_pack_weights_w2_multi(const Dims& d, const float* const* w, float* const* packed, const int* dgrad, int count /*<= 4*/,
                                 hipStream_t s);
void launch_pack_weights_small(const Dims& d, const float* w, float* packed /*[9][C/4][C][4]*/, hipStream_t s);
void launch_conv_small(const Dims& d, const ConvArgs& a, hipStream_t s);   // forward only; a.raw_out, a.wpacked = small packing

struct WgradArgs {
  const float* act;       // [N,HW,C] conv input activation
  const float* dz;        // [N,HW,C] cotangent of conv output
  const Ctrl* ctrl;       // nullable: the launch returns at once when ctrl->done (see Ctrl)
  float* wpart;           // [nsplit][9][C][C]
  unsigned long long* stamps;  // diagnostics only (NODE_STAMPS builds)
  // second problem of the same geometry served by the same launch (grid.z = 1): the two conv layers of an augmented
  // evaluation go out together, each with HALF the K splits (Dims::wgrad_pair) -- twice the K loop per workgroup,
  // half the slab bytes, one launch boundary less.  nullptr: single problem.  k_wgrad_w2 only.
  const float* act2;
  const float* dz2;
  float* wpart2;
};
void launch_wgrad(const Dims& d, const WgradArgs& a, hipStream_t s);
// masked column sums of dz per sample: spart[N][9][C] (conv-bias / time-channel-weight / d-dt terms)
void launch_colsum(const Dims& d, const float* dz, float* spart, hipStream_t s);
size_t wgrad_lds_bytes(const Dims& d);

struct ThetaFinalizeArgs {
  const float* dU;         // nullable: F(4x4,3x3)-domain weight gradients [2][36][C][C] (k_w4_wgrad) instead of wpart: dW = G^T dU G
  const float* wpart[2];   // conv1, conv2
  const float* spart[2];   // [N][9][C] each
  const float* gpart[3];   // GN1 (mtiles), GN2 (mtiles), GN3 (N)
  int gpart_rows[3];
  int spart_rows;          // rows of spart (0: N) -- the F(4x4,3x3) passes of 16x16 images leave one row per quadrant
  const float* wtime[2];   // time-channel weights [tap][co] gathered from the raw conv weights (launch_wtime)
  float* sred;             // [2][9][C] split-reduced masked column sums (scratch)
  EvalTime et;
  float osign;             // tsign
  float* theta_out;        // [P] internal layout
  Ctrl* ctrl;              // ts_k[kidx] <- osign * vjp_t   (when write_scalar)
  int kidx;
  int write_scalar;
  float* vjp_t_out;        // nullable device float
};
void launch_theta_finalize(const Dims& d, const ThetaFinalizeArgs& a, hipStream_t s);

// latency path (kernels_tiny.hip): a whole 3x3 convolution + bias + time map + GroupNorm (+ ReLU) of a tiny batch in one launch
struct TinyConvArgs {
  const float* act;            // [N][HW][C] NHWC
  const unsigned short* wq;    // launch_tiny_pack
  const float* bias; const float* tmap;
  EvalTime et;
  const float* gamma; const float* beta;
  float* out;                  // [N][HW][C] NHWC
  float* part; unsigned* counter;    // K-slice partial sums, arrival counters [N G] (zero between launches)
  const Ctrl* ctrl;
  int relu; float osign;
  // optional: the next evaluation's stage combine -> GroupNorm-1 -> ReLU behind this launch (its last convolution)
  int nx_on, nx_self;          // nx_self: index of the combine's term that IS this launch's output (taken from registers), or -1
  Comb nx; float* nx_y_out; const float* nx_gamma; const float* nx_beta; float* nx_act;
};
// latency path, resident form (kernels_tiny_solve.hip): a whole forward dopri5 solve of a tiny state in one launch
struct TinyResidentArgs {
  const float* y0;               // NCHW
  float* y_first;                // NCHW: the trajectory's first slot (<- y0), nullable
  float* y_out;                  // NCHW [n_targets][N][C][HW]
  const float* w[2];             // conv1 / conv2 filters as the model holds them: [C][C + 1][3][3]
  const float* bias[2];
  const float* gamma[3];
  const float* beta[3];
  void* handoff;                 // tiny_resident_handoff_words 8-byte tagged words (never zeroed per solve: see `nonce`)
  unsigned nonce;                // 28 bits, never repeated by this process within 2^28 solves
  Ctrl* ctrl;                    // the record in the workspace
  Ctrl* ctrl_host;               // the caller's pinned host copy (nullable)
  const double* targets; int n_targets;      // device array, or nullptr with the times in targets_inline (n_targets <= 8)
  double targets_inline[8];
  const double* forced; int n_forced;
  double* dt_log; int dt_log_cap;
  double t0;
  long long max_steps;
  float rtol, atol, tsign;
};
bool tiny_resident_ok(const Dims& d);
size_t tiny_resident_handoff_words(const Dims& d);
void launch_tiny_solve(const Dims& d, const TinyResidentArgs& a, hipStream_t s);
int tiny_slice_channels(const Dims& d);
size_t tiny_packed_elems(const Dims& d);
size_t tiny_part_elems(const Dims& d);
void launch_tiny_pack(const Dims& d, const float* w, unsigned short* wq, hipStream_t s);
void launch_tiny_conv_gn(const Dims& d, const TinyConvArgs& a, hipStream_t s);

// classifier head (kernels_head.hip); node_shape is the public struct of include/node_hip.h
}  // namespace node
struct node_shape;
namespace node {
int head_check(const node_shape* sh, char* why, size_t why_len);
void launch_head_gsum(const float* gpart, float* gsum, int N, int C, hipStream_t s);
void launch_head_fwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, const float* scale,
                     float* pooled, float* stats, hipStream_t s);
void launch_gn_relu_fwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, int relu, float* out,
                        float* stats, hipStream_t s);
void launch_gn_relu_bwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, const float* stats,
                        int relu, const float* gout, float* dz, float* gpart, hipStream_t s);
// fused multi-tensor SGD (kernels_optim.hip)
constexpr int SGD_TABLE = 64;
struct SgdEntry { float* p; const float* g; float* m; size_t n; };
struct SgdTable { SgdEntry e[SGD_TABLE]; };
void launch_sgd_multi(const SgdTable& tb, int count, size_t max_n, float lr, float momentum, float wd, float gscale,
                      const float* skip, hipStream_t s);
void launch_head_bwd(const node_shape& sh, const float* z, const float* gamma, const float* beta, const float* scale,
                     const float* stats, const float* gpool, float* dz, float* gpart, hipStream_t s);

}  // namespace node
