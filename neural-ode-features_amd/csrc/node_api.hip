// C ABI + host-side solver of libnode_hip.so.
//
// The host code below is the *driver* of the integrator: it owns no numerics.
// Every number (stage values, error norms, accept decisions, step sizes) is
// produced by device kernels and stays in device memory -- including the decisions of the adaptive step loop
// (accept / reject, which output times a step passed, dense output, FSAL commit, end of the interval).  The host
//   * enqueues whole steps on the caller's stream, as many as the previous solve of the same problem needed,
//   * reads one small `Ctrl` record back per solve to learn whether that was enough (and tops up if not), or --
//     deferred completion, node_solve_opts.blind_steps -- reads nothing back at all and leaves the verdict in a
//     device record on which the caller predicates whatever commits results.
//
// Algorithm: restated torchdiffeq dopri5 / rk4(3/8) / continuous adjoint, see
// SURVEY.md 8c and oracle/torchdiffeq_restated.py (the CPU checker).
#include "node_internal.h"
#include "wino4.h"
#include "../../include/node_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

using namespace node;

// ----------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// the same thread-local message for the entry points that live in other translation units (stem_api.hip)
namespace node {
int set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
}  // namespace node
#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return fail(NODE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)
#define TRY(expr)              \
  do {                         \
    int _rc = (expr);          \
    if (_rc != NODE_OK) return _rc; \
  } while (0)

// ----------------------------------------------------------------------------
// profiling (HIP events around the GEMM-class launches, on the caller's stream)
// ----------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  std::mutex mu;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
} g_prof;

struct ProfScope {
  bool active;
  ProfRec r;
  hipStream_t s;
  ProfScope(int cls, double flops, hipStream_t st) : active(g_prof.on), s(st) {
    if (!active) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    r.a = g_prof.get();
    r.b = g_prof.get();
    r.cls = cls;
    r.flops = flops;
    (void)hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(r.b, s);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.recs.push_back(r);
  }
};
}  // namespace

// ----------------------------------------------------------------------------
// geometry
// ----------------------------------------------------------------------------
static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

static int make_dims(const node_shape* sh, Dims* out) {
  if (!sh) return fail(NODE_ERR_NULL, "shape is NULL");
  Dims d;
  memset(&d, 0, sizeof(d));
  d.N = sh->n; d.C = sh->c; d.H = sh->h; d.W = sh->w; d.G = sh->groups; d.eps = sh->eps;
  if (d.N <= 0 || d.C <= 0 || d.H <= 0 || d.W <= 0 || d.G <= 0) return fail(NODE_ERR_SHAPE, "non-positive dimension");
  if (d.C % d.G != 0) return fail(NODE_ERR_SHAPE, "groups (%d) must divide channels (%d)", d.G, d.C);
  if (!(d.eps > 0.f)) return fail(NODE_ERR_SHAPE, "eps must be > 0");
  if (d.C % 4 != 0) return fail(NODE_ERR_UNSUPPORTED, "channels (%d) must be a multiple of 4", d.C);
  d.HW = d.H * d.W;
  d.cpg = d.C / d.G;
  d.Wp = d.W + 2; d.Hp = d.H + 2; d.SLOTS = d.Hp * d.Wp; d.MARGIN = d.Wp + 1;
  if (d.cpg > 64) return fail(NODE_ERR_UNSUPPORTED, "channels per group (%d) > 64", d.cpg);
  d.BNE = (64 / d.cpg) * d.cpg;
  d.ntile = (d.C + d.BNE - 1) / d.BNE;
  d.nchunk = (d.C + KCH - 1) / KCH;
  d.csplit = 0;
  // images larger than 256 pixels (the reference's one-shot / ODE stems on 64x64 inputs give 32x32 states, model.py:119-126,
  // 181-196, utils.py:168-195) run the 2-D Winograd conv in bands of 128 pixels (csplit below) with GroupNorm as a pass
  const bool banded = d.HW > 256 && d.HW <= 1024 && d.HW % 128 == 0 && d.H % 2 == 0 && d.W % 2 == 0 && 32 % (d.W / 2) == 0 && d.C % 32 == 0;
  if (d.HW <= 128) d.BM = 128;
  else if (d.HW <= 256) d.BM = 256;
  else if (banded) d.BM = 128;
  else return fail(NODE_ERR_UNSUPPORTED, "H*W = %d > 256: only even-sided images of up to 1024 pixels whose tile rows divide 32 "
                   "(16x16, 32x32, 16x32) with C %% 32 == 0 are tiled", d.HW);
  if (d.HW <= 64) {
    // grids that cannot fill the chip with 128-row tiles (MNIST-sized states, bs=1 census) use 64-row
    // tiles in four-wave workgroups: twice the workgroups (measured [32,64,7,7]: 31 -> 22.6 us).  Once
    // the 128-row grid reaches one workgroup per CU it is the faster one (cfg 2: 91.5 vs 95 us): two
    // co-resident 64-row workgroups run in lockstep and hide nothing of each other.
    const int s128 = 128 / d.HW < d.N ? 128 / d.HW : d.N;
    const long wg128 = (long)((d.N + s128 - 1) / s128) * d.ntile;
    static int bm_env = -2;   // NODE_TUNE_CONV_BM=64|128 forces the M tile (tests: the 2-D Winograd kernel on small batches)
    if (bm_env == -2) { const char* e = getenv("NODE_TUNE_CONV_BM"); bm_env = e ? atoi(e) : -1; }
    const int force_bm = g_conv_bm > 0 ? g_conv_bm : bm_env;
    const bool want64 = force_bm > 0 ? force_bm == 64 : wg128 < 256;
    if (want64) d.BM = 64;
  }
  if (d.W > 64) return fail(NODE_ERR_UNSUPPORTED, "W = %d > 64", d.W);
  {
    static int wino_env = -2;
    if (wino_env == -2) { const char* e = getenv("NODE_TUNE_CONV_WINO"); wino_env = e ? atoi(e) : -1; }
    const int want = g_conv_wino >= 0 ? g_conv_wino : wino_env;
    d.wino = (d.W % 2 == 0) ? (want < 0 ? 2 : want) : 0;   // even widths: Winograd kernels (2-D where the tile fits, else 1-D); odd: direct kernel
    if (d.wino == 2 && !(d.H % 2 == 0 && d.BM == 128 && 128 % d.HW == 0 && d.HW >= 16 && d.C % 32 == 0 &&
                      ((size_t)d.N * d.HW * d.C + d.C) * sizeof(float) < ((size_t)1 << 32)))
      d.wino = 1;   // 2-D variant: whole samples in 32 tiles
    // ... or, for 256-pixel images, two workgroups per sample (32 tiles = whole tile rows = 128 consecutive
    // pixels each) with the GroupNorm as a pointwise pass behind the conv: 16x16 at C = 256 runs the 1-D kernel
    // at 109 algorithmic TFLOP/s (one sample per 256-pixel tile), the 2-D kernel + pass is ~1.5x faster
    if ((want < 0 || want == 2 || banded) && (d.HW == 256 || banded) && d.H % 2 == 0 && d.W % 2 == 0 && 32 % (d.W / 2) == 0 && d.C % 32 == 0 &&
        ((size_t)d.N * d.HW * d.C + d.C) * sizeof(float) < ((size_t)1 << 32)) {
      d.wino = 2;
      d.BM = 128;
      d.csplit = d.HW / 128;
    }
  }
  if (d.HW > 256 && !d.csplit)   // banded geometry whose tensors pass 2^32 bytes: the 128-row tile holds no whole sample (S = 0)
    return fail(NODE_ERR_UNSUPPORTED, "H*W = %d with N*H*W*C = %zu elements: the banded convolution addresses at most 2^32 bytes per tensor",
                d.HW, (size_t)d.N * d.HW * d.C);
  d.S = d.csplit ? 1 : d.BM / d.HW;
  if (d.S > d.N) d.S = d.N;
  while (d.wino != 2 && d.S > 1 && conv_lds_bytes(d, 0) > 150 * 1024) d.S--;
  if (conv_lds_bytes(d, 0) > 160 * 1024) return fail(NODE_ERR_UNSUPPORTED, "conv tile does not fit LDS");
  d.mtiles = d.csplit ? d.N * d.csplit : (d.N + d.S - 1) / d.S;
  {
    // latency regime: the throughput tiles leave most of the chip idle (bs = 1 at C = 256: four workgroups)
    static int small_env = -2;   // NODE_TUNE_SMALL = 0 / 1 forces the choice (A/B measurements)
    if (small_env == -2) { const char* e = getenv("NODE_TUNE_SMALL"); small_env = e ? atoi(e) : -1; }
    const bool fits = d.C % 32 == 0 && d.C >= 128 && ((size_t)d.N * d.HW * d.C + d.C) * sizeof(float) < ((size_t)1 << 32);
    // measured at C = 256, 8x8 (tools/latency_bs1.py, us per function evaluation, small / throughput tiles): bs 1: 74.7 /
    // 87.6, bs 4: 99.3 / 90.8, bs 16: 99.3 / 93.0 -- the extra GroupNorm launches cost more than the parallelism
    // buys as soon as the throughput grid has eight workgroups, so only single-digit grids take the small kernel
    d.small = fits && (small_env >= 0 ? small_env != 0 : (long)d.mtiles * d.ntile < 8);
  }
  {
    // latency path (kernels_tiny.hip): forward solves of batches of up to 256 pixels (bs = 1 .. 4 at 8x8, bs = 1 at 16x16).  NODE_TUNE_TINY =
    // 0 never / 1 wherever the geometry fits (A/B measurements, tests); results are fp32-exact products either way
    static int tiny_env = -2;
    if (tiny_env == -2) { const char* e = getenv("NODE_TUNE_TINY"); tiny_env = e ? atoi(e) : -1; }
    d.numel = (size_t)d.N * d.C * d.HW;
    d.tiny = 0;
    if (tiny_env != 0 && (tiny_env == 1 || (size_t)d.N * d.HW <= 256)) d.tiny = tiny_slice_channels(d);
  }
  const int unit = d.cpg / gcd_i(d.cpg, 4) * 4;  // lcm(cpg, 4)
  static int slab_elems = -1;   // elements of one (sample, channel slab) workgroup of the combine / GN kernels
  if (slab_elems < 0) { const char* e = getenv("NODE_TUNE_SLAB"); slab_elems = e ? atoi(e) : 2048; }
  int mult = (slab_elems / d.HW) / unit;
  if (mult < 1) mult = 1;
  d.cs = unit * mult;
  if (d.cs > d.C) d.cs = d.C;
  if ((size_t)d.HW * d.cs > 16384) return fail(NODE_ERR_UNSUPPORTED, "GroupNorm slab does not fit LDS");
  d.nslab = (d.C + d.cs - 1) / d.cs;
  {
    // F(4x4,3x3) pipeline (wino4.h): geometry only; a solve uses it when its tolerance allows (Solver::w4)
    // NODE_TUNE_WINO4 = 0 never / 1 by tolerance (default) / 2 wherever the geometry fits.  Read on every call (unlike
    // the other switches) so that one test process can run both conv paths on the same inputs.
    const char* w4e = getenv("NODE_TUNE_WINO4");
    const int w4_env = w4e ? atoi(w4e) : 1;
    // 16x16 images: four 8x8 quadrants per image (w4q), each a virtual sample of the GEMM-side layouts; the passes
    // hold a GroupNorm group in one workgroup (cpg | 16 or cpg == 32) and only the F(4x4,3x3)-domain weight gradient
    // (C % 128 == 0) is wired behind them
    const bool fit8 = d.H == 8 && d.W == 8 && d.C % 64 == 0 && 16 % d.cpg == 0;
    const bool fit16 = d.H == 16 && d.W == 16 && d.C % 128 == 0 && (16 % d.cpg == 0 || d.cpg == 32);
    d.wino4 = (w4_env != 0 && (fit8 || fit16)) ? w4_env : 0;
    d.w4q = fit16 ? 4 : 1;
    d.N8 = (d.N * d.w4q + 7) & ~7;
  }
  d.RB = 64 / d.W;
  if (d.W >= 32) d.RB = 1;      // (the generic weight-gradient kernel stages RB + 2 rows in registers: 3 x 32 pixels is its limit)
  if (d.RB < 1) d.RB = 1;
  if (d.RB > d.H) d.RB = d.H;
  {
    // weight gradient in the Winograd domain where an instance of k_wgrad_w exists (W % 4 == 0)
    static int ww_env = -2;
    if (ww_env == -2) { const char* e = getenv("NODE_TUNE_WGRAD_WINO"); ww_env = e ? atoi(e) : -1; }
    const int want = g_wgrad_wino >= 0 ? g_wgrad_wino : (ww_env >= 0 ? ww_env : 2);
    d.wgrad_wino = 0;
    d.wut = 0;
    if (want && ((d.W == 8 && d.H % 8 == 0) || (d.W == 16 && d.H % 2 == 0) || (d.W == 4 && d.H % 4 == 0))) {
      d.wgrad_wino = 1;
      d.RB = d.W == 8 ? 8 : d.W == 16 ? 2 : 4;
    }
    // 2-D Winograd domain: units of 8 (or 4) tiles that are whole tile rows of one sample
    if (want >= 2 && d.H % 2 == 0 && d.W % 2 == 0 && ((size_t)d.N * d.HW * d.C + 2 * (size_t)d.C * (d.W + 2)) * sizeof(float) < ((size_t)1 << 32)) {
      const int TW = d.W / 2, TPS = (d.H / 2) * TW;
      const int ut = (TPS % 8 == 0 && 8 % TW == 0) ? 8 : (TPS % 4 == 0 && 4 % TW == 0) ? 4 : 0;
      if (ut) { d.wgrad_wino = 2; d.wut = ut; }
    }
  }
  d.nbands = (d.H + d.RB - 1) / d.RB;
  if ((d.RB + 2) * d.W * 16 > 6 * WG_THREADS || d.RB * d.W * 16 > 4 * WG_THREADS)
    return fail(NODE_ERR_UNSUPPORTED, "W = %d: wgrad staging does not fit its registers", d.W);
  if (wgrad_lds_bytes(d) > 160 * 1024) return fail(NODE_ERR_UNSUPPORTED, "wgrad tile does not fit LDS");
  {
    const int U = d.wgrad_wino == 2 ? d.N * ((d.H / 2) * (d.W / 2) / d.wut) : d.N * d.nbands;
    const int ntc = (d.C + 63) / 64;
    // both conv layers' weight gradients go out in ONE launch where the 2-D Winograd kernel serves them
    // (NODE_TUNE_WGRAD_PAIR=0: two launches with twice the splits each, the round-1 arrangement)
    static int pair_env = -2;
    if (pair_env == -2) { const char* e = getenv("NODE_TUNE_WGRAD_PAIR"); pair_env = e ? atoi(e) : 1; }
    d.wgrad_pair = (d.wgrad_wino == 2 && pair_env != 0) ? 1 : 0;
    int ns = (d.wgrad_pair ? 128 : 256) / (ntc * ntc);   // one workgroup per CU (307 VGPR+AGPR: one wave per SIMD)
    if (ns < 1) ns = 1;
    if (ns > 32) ns = 32;
    if (ns > U) ns = U;
    d.nsplit = ns;
  }
  d.P = 18 * (size_t)d.C * d.C + 26 * (size_t)d.C;
  d.numel = (size_t)d.N * d.C * d.HW;
  *out = d;
  return NODE_OK;
}

namespace node {
// internal (non-ABI) access to the tiling geometry for tools/kbench.hip
int dims_for(const node_shape* sh, Dims* out) { return make_dims(sh, out); }
}  // namespace node

// ----------------------------------------------------------------------------
// workspace plan
// ----------------------------------------------------------------------------
namespace {
struct Plan {
  // common
  Ctrl* ctrl;
  unsigned* arrive;         // arrival counter of the norm kernels whose last workgroup is the controller (zero between launches)
  float* partial[3];        // [ERR_BLOCKS][2] each
  float* wf[2];             // packed forward weights
  float* wd[2];             // packed dgrad weights (adjoint)
  float* tmap[2];
  float *Y, *Y1, *KY[7];
  float *act1, *act2;
  float* RAW;               // split-conv / small mode: the conv's raw output, consumed by the GroupNorm pass
  float* wsmall[2];         // small mode: filters packed for k_conv3x3_small
  unsigned short* wtiny[2]; // latency path (kernels_tiny.hip): filters as column-padded bf16 triples in fragment order
  float* tpart;             //   K-slice partial sums
  unsigned* tcount;         //   arrival counters [N G]
  void* thand;              // resident form (kernels_tiny_solve.hip): hand-off buffers of tagged words (activations, partial sums, decisions)
  float *W4V, *W4M;         // F(4x4,3x3) pipeline: the current conv's row operand and component products (wino4.h)
  float* w4u[4];            // its filter operands: forward conv1 / conv2, data gradient conv1 / conv2
  unsigned short* w4ub[4];  // the same as exact bf16 triples (k_w4_gemm64b)
  float* tmapS[2];          // the border maps in the W4S blocking (kernels_w4s.hip)
  W4Scales* w4sc;           // power-of-two scales of the fp16-pair operands (wino4.h)
  float* W4Va0b;            // second copy of W4Va[0]: the pass that ends an evaluation writes the NEXT one's conv-1 operand while this
                            // evaluation's weight gradient may still read its own (side stream, Solver::wgrad_side)
  float *W4Va[2], *W4Z[2], *W4dU;   // F(4x4,3x3)-domain weight gradient (C % 128 == 0): the forward convs' row operands
                                    // kept until it runs, Z = A dz A^T of both conv outputs' cotangents, the gradients
  float *act1b, *xh1b, *r1b;   // second set of GroupNorm-1's saved tensors: the pass that ends evaluation s also forms
                               // stage s + 1's conv input, while evaluation s's own set is still being read
  // adjoint
  float *A, *A1, *KA[7];
  float *TH, *TH1, *KT[7];
  float *xh1, *xh2, *xh3, *r1, *r2, *r3;
  float *dz1, *dz2, *G;
  float *wpart[2], *spart[2], *gpart[3], *sred, *wtime[2];
  float* dots;              // [n_t] time vjps scratch
  // device-resident stepping
  double* targets;          // [n_t] output times of the current interval (solver orientation)
  double* forced;           // [STEP_LIST_CAP] replay-mode step sizes
  double* dtlog;            // [STEP_LIST_CAP] dt tried per step of the current interval (negative: rejected)
  size_t bytes;
};
constexpr int STEP_LIST_CAP = 4096;   // replay lists / dt logs longer than this are refused / truncated

struct Bump {
  char* base;
  size_t off;
  explicit Bump(void* b) : base((char*)b), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = (T*)(base + off);
    off += count * sizeof(T);
    return p;
  }
};

Plan make_plan(const Dims& d, int adjoint, int n_t, void* base) {
  Plan p;
  memset(&p, 0, sizeof(p));
  Bump b(base);
  p.ctrl = b.take<Ctrl>(1);
  p.arrive = b.take<unsigned>(4);
  p.targets = b.take<double>((size_t)(n_t > 0 ? n_t : 1));
  p.forced = b.take<double>(STEP_LIST_CAP);
  p.dtlog = b.take<double>(STEP_LIST_CAP);
  for (int i = 0; i < 3; ++i) p.partial[i] = b.take<float>(ERR_BLOCKS * 2);
  const size_t wsz = conv_packed_elems(d);
  for (int i = 0; i < 2; ++i) p.wf[i] = b.take<float>(wsz);
  for (int i = 0; i < 2; ++i) p.tmap[i] = b.take<float>((size_t)d.HW * d.C);
  p.Y = b.take<float>(d.numel);
  p.Y1 = b.take<float>(d.numel);
  for (int i = 0; i < 7; ++i) p.KY[i] = b.take<float>(d.numel);
  // conv inputs carry a tail of C zeros: the 2-D Winograd kernel reads its zero halo there
  p.act1 = b.take<float>(d.numel + d.C);
  p.act2 = b.take<float>(d.numel + d.C);
  if (d.csplit || d.small) p.RAW = b.take<float>(d.numel);
  if (d.small && !adjoint)
    for (int i = 0; i < 2; ++i) p.wsmall[i] = b.take<float>((size_t)9 * d.C * d.C);
  if (d.tiny && !adjoint) {
    for (int i = 0; i < 2; ++i) p.wtiny[i] = b.take<unsigned short>(tiny_packed_elems(d));
    p.tpart = b.take<float>(tiny_part_elems(d));
    p.tcount = b.take<unsigned>((size_t)d.N * d.G);
    if (tiny_resident_ok(d)) p.thand = b.take<unsigned long long>(tiny_resident_handoff_words(d));
  }
  if (d.wino4) {
    p.W4V = b.take<float>(w4_v_elems(d.N8, d.C));
    p.W4M = b.take<float>(w4_v_elems(d.N8, d.C));
    for (int i = 0; i < (adjoint ? 4 : 2); ++i) p.w4u[i] = b.take<float>(w4_u_elems(d.C));
    for (int i = 0; i < (adjoint ? 4 : 2); ++i) p.w4ub[i] = b.take<unsigned short>(w4_ub_elems(d.C));
    for (int i = 0; i < 2; ++i) p.tmapS[i] = b.take<float>((size_t)d.HW * d.C);
    p.w4sc = b.take<W4Scales>(1);
    if (adjoint && d.C % 128 == 0) {
      for (int i = 0; i < 2; ++i) p.W4Va[i] = b.take<float>(w4_v_elems(d.N8, d.C));
      p.W4Va0b = b.take<float>(w4_v_elems(d.N8, d.C));
      for (int i = 0; i < 2; ++i) p.W4Z[i] = b.take<float>(w4_z_elems(d.N8, d.C));
      p.W4dU = b.take<float>(w4_du_elems(d.C));
    }
    if (adjoint) {
      p.act1b = b.take<float>(d.numel + d.C);
      p.xh1b = b.take<float>(d.numel);
      p.r1b = b.take<float>((size_t)d.N * d.G);
    }
  }
  const size_t prow = d.wino4 ? (size_t)d.N * d.w4q : (size_t)d.N;   // rows of the per-sample partials (F(4x4,3x3) passes: per quadrant)
  if (adjoint) {
    for (int i = 0; i < 2; ++i) p.wd[i] = b.take<float>(wsz);
    p.A = b.take<float>(d.numel);
    p.A1 = b.take<float>(d.numel);
    for (int i = 0; i < 7; ++i) p.KA[i] = b.take<float>(d.numel);
    p.TH = b.take<float>(d.P);
    p.TH1 = b.take<float>(d.P);
    for (int i = 0; i < 7; ++i) p.KT[i] = b.take<float>(d.P);
    p.xh1 = b.take<float>(d.numel);
    p.xh2 = b.take<float>(d.numel);
    p.xh3 = b.take<float>(d.numel);
    p.r1 = b.take<float>((size_t)d.N * d.G);
    p.r2 = b.take<float>((size_t)d.N * d.G);
    p.r3 = b.take<float>((size_t)d.N * d.G);
    p.dz1 = b.take<float>(d.numel + d.C);
    p.dz2 = b.take<float>(d.numel + d.C);
    p.G = b.take<float>(d.numel);
    for (int i = 0; i < 2; ++i) {
      p.wpart[i] = b.take<float>((size_t)d.nsplit * 9 * d.C * d.C);
      p.spart[i] = b.take<float>(prow * 9 * d.C);
    }
    p.sred = b.take<float>((size_t)2 * 9 * d.C + 2 * ((9 * (size_t)d.C + 63) / 64) + 4);   // + vjp_t partials + arrival counter
    for (int i = 0; i < 2; ++i) p.wtime[i] = b.take<float>((size_t)9 * d.C);
    const size_t grows = prow > (size_t)d.mtiles ? prow : (size_t)d.mtiles;
    p.gpart[0] = b.take<float>(grows * 2 * d.C);
    p.gpart[1] = b.take<float>(grows * 2 * d.C);
    p.gpart[2] = b.take<float>(prow * 2 * d.C);
    p.dots = b.take<float>((size_t)(n_t > 0 ? n_t : 1));
  }
  p.bytes = ((b.off + 255) & ~(size_t)255);
  return p;
}

// pinned host staging: the mirror of Ctrl for the read-backs and the small lists that travel to / from the device
// (target times, replay list, dt log).  One per host thread; every solve ends with a stream synchronisation, so
// the staging is free again when the next solve of this thread starts.
struct HostStage {
  Ctrl* ctrl = nullptr;
  double* lists = nullptr;
  size_t cap = 0;      // doubles
};
thread_local HostStage g_stage;

int get_stage(size_t doubles, HostStage** out) {
  if (!g_stage.ctrl) HIP_TRY(hipHostMalloc((void**)&g_stage.ctrl, sizeof(Ctrl), hipHostMallocDefault));
  if (g_stage.cap < doubles) {
    if (g_stage.lists) (void)hipHostFree(g_stage.lists);
    g_stage.lists = nullptr;
    g_stage.cap = 0;
    const size_t want = doubles < 16384 ? 16384 : doubles;
    HIP_TRY(hipHostMalloc((void**)&g_stage.lists, want * sizeof(double), hipHostMallocDefault));
    g_stage.cap = want;
  }
  *out = &g_stage;
  return NODE_OK;
}

// How many steps the last solve of the same problem took: the number enqueued blind before the first read-back.
struct StepGuess { int N, C, H, W, aug, forced; float rtol, atol; double t0, t1; int steps; };
thread_local std::vector<StepGuess> g_guess;
int guess_steps(const StepGuess& k) {
  for (const auto& g : g_guess)
    if (g.N == k.N && g.C == k.C && g.H == k.H && g.W == k.W && g.aug == k.aug && g.forced == k.forced && g.rtol == k.rtol &&
        g.atol == k.atol && g.t0 == k.t0 && g.t1 == k.t1)
      return g.steps;
  return 1;
}
void remember_steps(const StepGuess& k) {
  for (auto& g : g_guess)
    if (g.N == k.N && g.C == k.C && g.H == k.H && g.W == k.W && g.aug == k.aug && g.forced == k.forced && g.rtol == k.rtol &&
        g.atol == k.atol && g.t0 == k.t0 && g.t1 == k.t1) { g.steps = k.steps; return; }
  if (g_guess.size() >= 64) g_guess.erase(g_guess.begin());
  g_guess.push_back(k);
}

const double DP_ALPHA[6] = {1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
const double DP_CMID[7] = {6025192743.0 / 30085553152.0 / 2.0, 0.0, 51252292925.0 / 65400821598.0 / 2.0,
                           -2691868925.0 / 45128329728.0 / 2.0, 187940372067.0 / 1594534317056.0 / 2.0,
                           -1776094331.0 / 19743644256.0 / 2.0, 11237099.0 / 235043384.0 / 2.0};
const double DP_BETA[6][6] = {
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},
};

// A second stream for the weight gradient of an augmented evaluation (Solver::wgrad_side): one per host thread and device, created at
// first use and kept (stream creation costs milliseconds).
struct SideStream {
  int dev = -1;
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
thread_local SideStream g_side;
static bool get_side(SideStream** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (g_side.s != nullptr && g_side.dev != dev) {
    (void)hipEventDestroy(g_side.fork); (void)hipEventDestroy(g_side.join); (void)hipStreamDestroy(g_side.s);
    g_side = SideStream();
  }
  if (g_side.s == nullptr) {
    if (hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g_side = SideStream();
      return false;
    }
    g_side.dev = dev;
  }
  *out = &g_side;
  return true;
}
int g_w4_pair_stats[4] = {0, -1, 0, 0};   // node_w4_pair_stats (diagnostics; process-wide: a backward pass runs on autograd's thread)
thread_local Ctrl* g_blind_resident_ctrl = nullptr;   // pinned: where a DEFERRED resident solve leaves its record for this library itself --
                                                       // the caller reads the device record; a later call of this thread arms the cooldown from this one
std::atomic<int> g_resident_cooldown{0};     // solves left before the resident latency path is tried again (see Solver::choose_resident)

// ----------------------------------------------------------------------------
// The solver context: one solve (forward or augmented/adjoint) on one stream
// ----------------------------------------------------------------------------
struct Solver {
  Dims d;
  Plan p;
  node_params prm;
  hipStream_t st;
  bool aug = false;
  float tsign = 1.f;
  float rtol = 0.f, atol = 0.f;
  int nfe = 0;
  // F(4x4,3x3) pipeline for the convs of this solve (wino4.h).  Its rounding error (3.2e-6 of max|y| per conv, against
  // 4.9e-7 for F(2x2,3x3)) must stay far below what the step controller resolves.  dopri5's embedded estimate is
  // h * sum_i e_i k_i with sum_i |e_i| = 0.16, so conv noise moves it by <= 0.16 * 3.2e-6 * h |f| ~ 5e-7 |y| -- 5 % of
  // the tolerance at 1e-5, 50 % at 1e-6.  Adaptive solves with rtol, atol >= 1e-5 take the pipeline (measured at tol
  // 1e-5: same step sequences, gradients as close to fp64 as the fp32 oracle's -- tests/test_gpu_w4.py).
  bool w4 = false;
  void choose_w4(bool adaptive) {
    w4 = d.wino4 == 2 || (d.wino4 == 1 && adaptive && rtol >= W4_MIN_TOL && atol >= W4_MIN_TOL && !tiny_mode());
  }
  // latency path: forward solves of tiny batches run two fused direct-convolution launches per evaluation (kernels_tiny.hip)
  bool tiny_mode() const { return d.tiny != 0 && !aug && p.wtiny[0] != nullptr; }
  // ... and a free-running or replayed dopri5 forward solve of a state the chip can hold resident is ONE launch (kernels_tiny_solve.hip)
  bool resident = false;
  void choose_resident(bool dopri5) {
    resident = dopri5 && tiny_mode() && !w4 && p.thand != nullptr && tiny_resident_ok(d);
    if (g_blind_resident_ctrl != nullptr && g_blind_resident_ctrl->status == NODE_ERR_HIP) {
      // an earlier DEFERRED resident solve of this thread ran into its deadline (nobody read its record on this side): same cooldown
      g_blind_resident_ctrl->status = 0;
      g_resident_cooldown.store(64, std::memory_order_relaxed);
    }
    if (resident) {      // a captured launch would replay its nonce: words of the previous replay would pass for this one's
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
      if (cs != hipStreamCaptureStatusNone) resident = false;
    }
    // a grid that did not get the whole chip costs its 2 s deadline: after one, the next 64 solves of this process do not try
    if (resident && g_resident_cooldown.load(std::memory_order_relaxed) > 0) {
      g_resident_cooldown.fetch_sub(1, std::memory_order_relaxed);
      resident = false;
    }
  }
  // F(4x4,3x3) passes merged across evaluations (kernels_w4s.hip): the pass that ends evaluation s may already have
  // formed evaluation s + 1's conv input (Butcher combine -> GroupNorm-1 -> ReLU -> V)
  bool w4_b16 = false;     // the component GEMMs read the filters as exact bf16 triples (k_w4_gemm64b), decided in prepare()
  bool w4_f16 = false;     // ... both operands as fp16 pairs (k_w4_gemm64h; wino4.h), decided in prepare()
  bool w4_f16_aug = false; // set by the caller before prepare(): an augmented solve may use them (adaptive dopri5 solves: the cotangent-side
                           // scale follows the data through the step controller)
  // The fp16-pair weight gradient (k_w4_wgrad64h: two small workgroups per CU, <= 128 registers) on a SIDE stream beside the data
  // gradient of conv 1 and the pass behind it (k_w4_gemm64h: one 332-register wave per SIMD, which leaves it room): forked behind the
  // pass that wrote Z1, joined in front of k_theta_finalize.  NODE_TUNE_W4_WGRAD_SIDE = 0 / 1 (read per solve in prepare()).
  SideStream* side = nullptr;
  bool side_pending = false;
  float* va0_of(int set) const { return (set && p.W4Va0b != nullptr) ? p.W4Va0b : p.W4Va[0]; }
  bool g_ready = false;    // the cotangent-side scale is known (behind an interval's first evaluation, launch_w4_gscale)
  // the format of the evaluation being enqueued: forward solves always pairs; augmented ones once the cotangent scale is known
  bool f16_now() const { return w4_f16 && (!aug || g_ready); }
  bool v_ready = false;    // the next evaluation's first pass has run
  int cur = 0;             // which set of GroupNorm-1's saved tensors (act1, xhat-1, 1/sigma-1) the current evaluation owns
  float* act1_of(int i) const { return i ? p.act1b : p.act1; }
  float* xh1_of(int i) const { return i ? p.xh1b : p.xh1; }
  float* r1_of(int i) const { return i ? p.r1b : p.r1; }
  void to_state(const float* nchw, float* dst) {    // NCHW -> the solve's internal state layout
    if (w4) launch_w4s_from_nchw(nchw, dst, d.N, d.C, d.w4q, st);
    else launch_nchw_to_nhwc(d, nchw, dst, st);
  }
  void from_state(const float* src, float* nchw) {
    if (w4) launch_w4s_to_nchw(src, nchw, d.N, d.C, d.w4q, st);
    else launch_nhwc_to_nchw(d, src, nchw, st);
  }
  struct NextComb { Comb cy; float* y_out; };
  // global-norm mode of a data-parallel solve (node_solve_opts::norm_reduce): the caller's hook adds the ranks' sums before each decision
  void (*nr_fn)(void*, float*, int32_t, void*) = nullptr;
  void* nr_ctx = nullptr;
  float* nr_buf = nullptr;
  float nr_world = 1.f;
  void take_norm_hook(const node_solve_opts* o) {
    if (o != nullptr && o->norm_reduce != nullptr && o->norm_buf != nullptr && o->norm_world >= 1) {
      nr_fn = o->norm_reduce; nr_ctx = o->norm_reduce_ctx; nr_buf = o->norm_buf; nr_world = (float)o->norm_world;
    }
  }
  // this rank's sums of the coming decision -> nr_buf, summed over the ranks by the hook (both enqueued on the solve's stream)
  void norm_exchange(int mode, int nseg) {
    NormPackArgs np;
    memset(&np, 0, sizeof(np));
    np.ctrl = p.ctrl; np.partial[0] = p.partial[0]; np.partial[1] = p.partial[1]; np.partial[2] = p.partial[2];
    np.nseg = nseg; np.has_scalar = aug ? 1 : 0; np.mode = mode; np.rtol = rtol; np.atol = atol;
    np.w4sc = (aug && w4_f16) ? p.w4sc : nullptr; np.gbuf = nr_buf;
    launch_norm_pack(np, st);
    nr_fn(nr_ctx, nr_buf, 8, (void*)st);
  }
  // NODE_TUNE_FOLD_CTL = 1: the controllers as the LAST-ARRIVING workgroup of the norm kernels (k_error_norm_ctl, k_init_norms_ctl: eight
  // launches less per training step).  Built for the round-5 review's item 5 and measured: 27 940 against 28 110 images/s at cfg 2 -- the
  // last workgroup's coherent re-read of the partial sums behind the arrival chain costs what the launch boundary did.  Off by default.
  static bool fold_ctl() {
    const char* e = getenv("NODE_TUNE_FOLD_CTL");
    return e != nullptr && atoi(e) != 0;
  }
  bool count_nfe = true;   // off while steps are enqueued blind: those evaluations are counted from the device's step counter
  Ctrl* hctrl = nullptr;

  double conv_flops() const { return 2.0 * 9.0 * d.C * d.C * (double)d.N * d.HW; }

  int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return NODE_OK;
  }

  // inference solves on grids the throughput tiles cannot spread over the chip (Dims::small)
  bool small_mode() const { return d.small && !aug && !w4 && p.wsmall[0] != nullptr && !tiny_mode(); }

  int prepare() {
    auto pack = d.wino == 2 ? launch_pack_weights_w2 : d.wino ? launch_pack_weights_w : launch_pack_weights;
    // zero fills of the solve, folded into the preparation launch below: the stage-2 parameter derivative (read with
    // weight zero by the error norm / dense output, never written by dopri5 steps -- the initial-step probe does
    // write it -- so it must hold finite values), the arrival counter of k_theta_finalize, and the zero rows behind
    // the conv inputs (see make_plan)
    float* zr[12];
    size_t zn[12];
    int nz = 0;
    zr[nz] = reinterpret_cast<float*>(p.arrive); zn[nz++] = 4;
    if (aug) {
      zr[nz] = p.KT[1]; zn[nz++] = d.P;
      zr[nz] = p.sred + (size_t)2 * 9 * d.C + 2 * ((9 * (size_t)d.C + 63) / 64); zn[nz++] = 1;
    }
    if (d.wino == 2 || d.wgrad_wino == 2 || small_mode()) {
      zr[nz] = p.act1 + d.numel; zn[nz++] = d.C;
      zr[nz] = p.act2 + d.numel; zn[nz++] = d.C;
      if (aug) {
        zr[nz] = p.dz1 + d.numel; zn[nz++] = d.C;
        zr[nz] = p.dz2 + d.numel; zn[nz++] = d.C;
        if (p.act1b) { zr[nz] = p.act1b + d.numel; zn[nz++] = d.C; }
      }
    }
    if (small_mode()) {
      launch_pack_weights_small(d, prm.conv1_w, p.wsmall[0], st);
      launch_pack_weights_small(d, prm.conv2_w, p.wsmall[1], st);
    }
    if (tiny_mode() && !w4) {
      launch_tiny_pack(d, prm.conv1_w, p.wtiny[0], st);
      launch_tiny_pack(d, prm.conv2_w, p.wtiny[1], st);
      launch_fill(reinterpret_cast<float*>(p.tcount), 0.f, (size_t)d.N * d.G, st);     // (0.f is the all-zero word)
    }
    if (w4) {
      w4_b16 = w4_uses_bf16(d.N8, d.C);
      w4_f16 = w4_b16 && w4_f16_fits(d.N8, d.C) && (!aug || (w4_f16_aug && w4_wgrad_on() && w4_wgrad_f16_fits(d.N8, d.C)));
      g_ready = false;
      side = nullptr; side_pending = false;
      if (w4_f16 && aug) {
        const char* e = getenv("NODE_TUNE_W4_WGRAD_SIDE");
        if ((e ? atoi(e) : 0) != 0 && !get_side(&side)) side = nullptr;
      }
      if (w4_f16) { zr[nz] = reinterpret_cast<float*>(p.w4sc); zn[nz++] = sizeof(W4Scales) / sizeof(float); }
    }
    // (first: it carries the solve's zero fills, among them the scratch words of k_w4_scales)
    launch_time_prep(d, prm.conv1_w, prm.conv2_w, p.tmap[0], p.tmap[1], aug ? p.wtime[0] : nullptr, aug ? p.wtime[1] : nullptr, zr, zn,
                     nz, st);
    if (w4) {
      W4PackJobs jobs;
      memset(&jobs, 0, sizeof(jobs));
      const float* ws_[4] = {prm.conv1_w, prm.conv2_w, prm.conv1_w, prm.conv2_w};
      if (w4_f16) {       // the scales of the fp16-pair operands: filters from max|w|, forward row operands from the GroupNorm in front
        W4ScaleJobs sj;
        memset(&sj, 0, sizeof(sj));
        sj.w[0] = prm.conv1_w; sj.w[1] = prm.conv2_w; sj.wn = (size_t)d.C * (d.C + 1) * 9;
        sj.gb[0] = prm.norm1_w; sj.gb[1] = prm.norm1_b; sj.gb[2] = prm.norm2_w; sj.gb[3] = prm.norm2_b;
        sj.C = d.C; sj.gn_m = d.cpg * d.HW; sj.sc = p.w4sc;
        launch_w4_scales(sj, st);
      }
      for (int i = 0; i < (aug ? 4 : 2); ++i) {
        jobs.w[i] = ws_[i]; jobs.u[i] = p.w4u[i]; jobs.dgrad[i] = i >= 2;
        if (w4_f16) { jobs.uh[i] = reinterpret_cast<unsigned*>(p.w4u[i]); jobs.uh_exp[i] = &p.w4sc->e[(i & 1) ? W4_E_U2 : W4_E_U1]; }
        if (!w4_f16 || aug) jobs.ub[i] = w4_b16 ? p.w4ub[i] : nullptr;
      }
      launch_w4_pack(jobs, aug ? 4 : 2, d.C, st);
    } else if (d.wino == 2) {   // every packing of the solve in one launch
      const float* ws_[4] = {prm.conv1_w, prm.conv2_w, prm.conv1_w, prm.conv2_w};
      float* dst_[4] = {p.wf[0], p.wf[1], p.wd[0], p.wd[1]};
      const int dg_[4] = {0, 0, 1, 1};
      launch_pack_weights_w2_multi(d, ws_, dst_, dg_, aug ? 4 : 2, st);
    } else {
      pack(d, prm.conv1_w, p.wf[0], 0, st);
      pack(d, prm.conv2_w, p.wf[1], 0, st);
      if (aug) {
        pack(d, prm.conv1_w, p.wd[0], 1, st);
        pack(d, prm.conv2_w, p.wd[1], 1, st);
      }
    }
    if (w4) launch_w4s_tmap(p.tmap[0], p.tmap[1], p.tmapS[0], p.tmapS[1], d.C, d.w4q, st);
    if (w4 && aug && d.N * d.w4q != d.N8 && p.W4dU != nullptr) {
      // the weight gradient SUMS over the GEMM rows: the rows of the padding samples (never written by a pass) must be zero
      for (int i = 0; i < 2; ++i) {
        launch_fill(p.W4Va[i], 0.f, w4_v_elems(d.N8, d.C), st);
        launch_fill(p.W4Z[i], 0.f, w4_z_elems(d.N8, d.C), st);
      }
      launch_fill(p.W4Va0b, 0.f, w4_v_elems(d.N8, d.C), st);
    }
    v_ready = false;
    cur = 0;
    return check_launch("prepare");
  }

  static Comb make_comb(const float* y, float* const* k, const double* coef, int ncoef, int scale_mode) {
    Comb c;
    memset(&c, 0, sizeof(c));
    c.y = y;
    c.scale_mode = scale_mode;
    int nk = 0;
    for (int j = 0; j < ncoef; ++j) {
      if (coef[j] == 0.0) continue;
      c.k[nk] = k[j];
      c.coef[nk] = (float)coef[j];
      ++nk;
    }
    c.nk = nk;
    return c;
  }

  // f(t, y_i) with y_i = comb; writes k_out = tsign * f  (and y_i to y_out if asked)
  // split-conv mode: GroupNorm (+ReLU) of the conv's raw output as a pointwise pass (k_combine_gn with an empty
  // Butcher row), and the ReLU-mask + GroupNorm backward of a raw data gradient (k_gn_bwd)
  void gn_pass_fwd(const float* gamma, const float* beta, int relu, float osign, float* out, float* xhat_out, float* rstd_out) {
    CombineGnArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.comb.y = p.RAW; ca.comb.nk = 0; ca.comb.scale_mode = SC_ABS; ca.ctrl = p.ctrl;
    ca.act_out = out; ca.xhat_out = xhat_out; ca.rstd_out = rstd_out;
    ca.gamma = gamma; ca.beta = beta; ca.relu = relu; ca.osign = osign;
    launch_combine_gn(d, ca, st);
  }
  void gn_pass_bwd(const float* act, const float* xhat, const float* rstd, const float* gamma, float osign, float* out,
                   float* gpart, float* spart) {
    GnBwdArgs g;
    memset(&g, 0, sizeof(g));
    g.comb.y = p.RAW; g.comb.nk = 0; g.comb.scale_mode = SC_ABS; g.ctrl = p.ctrl; g.csign = 1.f;
    g.xhat = xhat; g.rstd = rstd; g.gamma = gamma; g.dz_out = out; g.gpart = gpart; g.spart = spart;
    g.mask_act = act; g.osign = osign;
    launch_gn_bwd(d, g, st);
  }

  // F(4x4,3x3) pipeline: one conv = component GEMMs on the row operand its producer left in W4V; the GroupNorm pass
  // behind it reads the products (output transform, + bias + t * tmap for a forward conv) and, when another conv
  // follows, leaves that conv's row operand in W4V again
  void w4_gemm(int which, const float* V = nullptr) {
    ProfScope ps(2, conv_flops(), st);
    if (f16_now()) {     // which: 0 / 1 forward conv1 / conv2, 2 / 3 their data gradients (row operand = a cotangent)
      launch_w4_gemm_f16(reinterpret_cast<const unsigned*>(V ? V : p.W4V), reinterpret_cast<const unsigned*>(p.w4u[which]), p.W4M, p.ctrl, d.N8, d.C,
                         &p.w4sc->e[which >= 2 ? W4_E_G : which ? W4_E_V2 : W4_E_V1], &p.w4sc->e[(which & 1) ? W4_E_U2 : W4_E_U1], st);
      return;
    }
    launch_w4_gemm(V ? V : p.W4V, p.w4u[which], p.W4M, p.ctrl, d.N8, d.C, st, w4_b16 ? p.w4ub[which] : nullptr);
  }
  // the weight gradients of an augmented evaluation in the F(4x4,3x3) domain (k_w4_wgrad): needs the forward convs'
  // row operands alive behind the data-gradient convs, so they get buffers of their own
  bool w4_wgrad_on() const { return w4 && aug && p.W4dU != nullptr; }
  W4sArgs w4_args() const {
    W4sArgs a;
    memset(&a, 0, sizeof(a));
    a.ctrl = p.ctrl; a.N = d.N; a.Q = d.w4q; a.Nv = d.N8; a.C = d.C; a.cpg = d.cpg; a.eps = d.eps;
    return a;
  }
  // tail 1 of a pass: stage combine -> GroupNorm-1 -> ReLU -> V (+ act1, xhat-1, 1/sigma-1 of set `set` when training)
  void w4_tail_combine(W4sArgs& a, const Comb& cy, float* y_out, bool train, int set, int self) {
    a.t.comb = cy; a.t.self = self; a.t.y_out = y_out; a.t.gamma = prm.norm1_w; a.t.beta = prm.norm1_b;
    if (train) { a.t.act_nhwc = w4_wgrad_on() ? nullptr : act1_of(set); a.t.xhat_s = xh1_of(set); a.t.rstd = r1_of(set); }
    a.V = (train && w4_wgrad_on()) ? va0_of(set) : p.W4V;
    if (f16_now()) a.v_exp = &p.w4sc->e[W4_E_V1];
  }
  // launch one pass; under node_profile_begin() with HIP events around it and its algorithmic bytes (every tensor it
  // must read or write, once) in the record
  void w4_pass(int head, int tail, const W4sArgs& a) {
    double bytes = 0.0;
    if (g_prof.on) {
      const double state = (double)d.numel * sizeof(float), comp = 36.0 * 4.0 * d.N * d.w4q * d.C * sizeof(float);
      int tensors = 0;
      if (head) {
        tensors += (a.h.out_s != nullptr) + (a.h.out_nhwc != nullptr);
        tensors += a.h.xhat_s != nullptr;      // written (forward) or read (backward)
        bytes += comp;
      }
      if (tail) {
        tensors += 1 + a.t.comb.nk - (a.t.self ? 1 : 0);
        tensors += (a.t.y_out != nullptr) + (a.t.act_nhwc != nullptr) + (a.t.xhat_s != nullptr);
      }
      if (a.V) bytes += comp;
      bytes += tensors * state;
    }
    const int cls = head == 0 ? 3 : head == 1 ? 4 + tail : 7 + tail;
    ProfScope ps(cls, bytes, st);
    launch_w4s_pass(head, tail, a, st);
  }
  // One dynamics evaluation (ca == nullptr) or one augmented evaluation on the F(4x4,3x3) pipeline.  `next`: the
  // evaluation that follows takes its conv input from this one's last pass (v_ready).
  int eval_w4(const Comb& cy, float* y_out, const EvalTime& et, float* kY_out, bool train, const Comb* ca, float* a_out,
              float* kA_out, float* kT_out, int kidx, float csign, float* vjp_t_out, bool need_theta, const NextComb* next) {
    const bool do_aug = ca != nullptr;
    if (next != nullptr)   // a combine that reads this evaluation's own derivative anywhere but as its last term cannot merge
      for (int j = 0; j + 1 < next->cy.nk; ++j)
        if (next->cy.k[j] == kY_out || (do_aug && next->cy.k[j] == kA_out)) next = nullptr;
    if (!v_ready) {
      W4sArgs a = w4_args();
      w4_tail_combine(a, cy, y_out, train, cur, 0);
      w4_pass(0, 1, a);
    }
    v_ready = false;
    const bool wg4 = do_aug && w4_wgrad_on();
    w4_gemm(0, wg4 ? va0_of(cur) : nullptr);
    {   // P2
      W4sArgs a = w4_args();
      a.h.M = p.W4M; a.h.bias = prm.conv1_b; a.h.tmapS = p.tmapS[0]; a.h.et = et; a.h.gamma = prm.norm2_w; a.h.beta = prm.norm2_b;
      a.h.osign = 1.f; a.h.relu = 1;
      if (train) { a.h.out_nhwc = wg4 ? nullptr : p.act2; a.h.xhat_s = p.xh2; a.h.rstd = p.r2; }
      a.V = wg4 ? p.W4Va[1] : p.W4V;
      if (f16_now()) a.v_exp = &p.w4sc->e[W4_E_V2];
      w4_pass(1, 0, a);
    }
    w4_gemm(1, wg4 ? p.W4Va[1] : nullptr);
    W4sArgs a3 = w4_args();
    a3.h.M = p.W4M; a3.h.bias = prm.conv2_b; a3.h.tmapS = p.tmapS[1]; a3.h.et = et; a3.h.gamma = prm.norm3_w; a3.h.beta = prm.norm3_b;
    a3.h.osign = et.tsign; a3.h.relu = 0; a3.h.out_s = kY_out;
    if (!do_aug) {
      if (next != nullptr) {   // P3C
        const int self = next->cy.nk > 0 && next->cy.k[next->cy.nk - 1] == kY_out;
        w4_tail_combine(a3, next->cy, next->y_out, false, cur, self);
        w4_pass(1, 1, a3);
        v_ready = true;
      } else {
        w4_pass(1, 0, a3);
      }
      if (count_nfe) nfe += 1;
      return check_launch("odefunc forward (F(4x4,3x3))");
    }
    // P3B3: GroupNorm-3, then the adjoint combine through its backward
    a3.t.comb = *ca; a3.t.csign = csign; a3.t.y_out = a_out; a3.t.gpart = p.gpart[2]; a3.t.spart = p.spart[1];
    if (wg4) a3.t.z_out = need_theta ? p.W4Z[1] : nullptr;
    else a3.t.act_nhwc = p.dz2;
    a3.V = p.W4V;
    if (w4_f16) a3.gstat = p.w4sc;
    if (f16_now()) { a3.v_exp = &p.w4sc->e[W4_E_G]; a3.z_exp = &p.w4sc->e[W4_E_G]; }
    w4_pass(1, 2, a3);
    if (count_nfe) nfe += 1;
    w4_gemm(3);   // data gradient of conv2
    {   // PB2
      W4sArgs a = w4_args();
      a.h.M = p.W4M; a.h.gamma = prm.norm2_w; a.h.beta = prm.norm2_b; a.h.xhat_s = p.xh2; a.h.rstd = p.r2; a.h.osign = 1.f;
      a.h.gpart = p.gpart[1]; a.h.spart = p.spart[0];
      if (wg4) a.h.z_out = need_theta ? p.W4Z[0] : nullptr;
      else a.h.out_nhwc = p.dz1;
      a.V = p.W4V;
      if (w4_f16) a.gstat = p.w4sc;
      if (f16_now()) { a.v_exp = &p.w4sc->e[W4_E_G]; a.z_exp = &p.w4sc->e[W4_E_G]; }
      w4_pass(2, 0, a);
    }
    if (need_theta && wg4 && f16_now()) {
      hipStream_t ws = st;
      if (side != nullptr) {        // fork: the weight gradient runs beside the data gradient of conv 1 and its pass
        (void)hipEventRecord(side->fork, st);
        (void)hipStreamWaitEvent(side->s, side->fork, 0);
        ws = side->s;
      }
      {
        ProfScope ps(1, 2.0 * conv_flops(), ws);
        launch_w4_wgrad_f16(reinterpret_cast<const unsigned*>(va0_of(cur)), reinterpret_cast<const unsigned*>(p.W4Z[0]), reinterpret_cast<const unsigned*>(p.W4Va[1]),
                            reinterpret_cast<const unsigned*>(p.W4Z[1]), p.W4dU, p.ctrl, d.N8, d.C, &p.w4sc->e[W4_E_V1], &p.w4sc->e[W4_E_V2], &p.w4sc->e[W4_E_G], ws);
      }
      if (side != nullptr) { (void)hipEventRecord(side->join, side->s); side_pending = true; }
    } else if (need_theta && wg4) {
      W4WgradArgs wa;
      memset(&wa, 0, sizeof(wa));
      wa.V1 = va0_of(cur); wa.Z1 = p.W4Z[0]; wa.V2 = p.W4Va[1]; wa.Z2 = p.W4Z[1]; wa.dU = p.W4dU; wa.ctrl = p.ctrl; wa.N = d.N8; wa.C = d.C;
      ProfScope ps(1, 2.0 * conv_flops(), st);
      launch_w4_wgrad(wa, st);
    } else if (need_theta) {
      WgradArgs w1;
      memset(&w1, 0, sizeof(w1));
      w1.act = act1_of(cur); w1.dz = p.dz1; w1.wpart = p.wpart[0]; w1.ctrl = p.ctrl;
      if (d.wgrad_pair) { w1.act2 = p.act2; w1.dz2 = p.dz2; w1.wpart2 = p.wpart[1]; }
      { ProfScope ps(1, (d.wgrad_pair ? 2.0 : 1.0) * conv_flops(), st); launch_wgrad(d, w1, st); }
      if (!d.wgrad_pair) {
        WgradArgs w2 = w1;
        w2.act = p.act2; w2.dz = p.dz2; w2.wpart = p.wpart[1];
        { ProfScope ps(1, conv_flops(), st); launch_wgrad(d, w2, st); }
      }
    }
    w4_gemm(2);   // data gradient of conv1
    {   // PB1 (+ the next evaluation's combine)
      W4sArgs a = w4_args();
      a.h.M = p.W4M; a.h.gamma = prm.norm1_w; a.h.beta = prm.norm1_b; a.h.xhat_s = xh1_of(cur); a.h.rstd = r1_of(cur);
      a.h.osign = et.tsign; a.h.out_s = kA_out; a.h.gpart = p.gpart[0];
      if (next != nullptr) {
        w4_tail_combine(a, next->cy, next->y_out, true, cur ^ 1, 0);
        w4_pass(2, 1, a);
        cur ^= 1;
        v_ready = true;
      } else {
        w4_pass(2, 0, a);
      }
    }
    if (!need_theta) return check_launch("augmented dynamics (F(4x4,3x3))");
    if (side_pending) { (void)hipStreamWaitEvent(st, side->join, 0); side_pending = false; }      // join: dU is complete
    ThetaFinalizeArgs tf;
    memset(&tf, 0, sizeof(tf));
    tf.dU = wg4 ? p.W4dU : nullptr;
    tf.wpart[0] = p.wpart[0]; tf.wpart[1] = p.wpart[1];
    tf.spart[0] = p.spart[0]; tf.spart[1] = p.spart[1];
    tf.gpart[0] = p.gpart[0]; tf.gpart[1] = p.gpart[1]; tf.gpart[2] = p.gpart[2];
    tf.gpart_rows[0] = tf.gpart_rows[1] = tf.gpart_rows[2] = tf.spart_rows = d.N * d.w4q;   // per-sample (per-quadrant) partials from the GroupNorm passes
    tf.wtime[0] = p.wtime[0]; tf.wtime[1] = p.wtime[1]; tf.sred = p.sred;
    tf.et = et; tf.osign = et.tsign; tf.theta_out = kT_out;
    tf.ctrl = p.ctrl; tf.kidx = kidx; tf.write_scalar = kidx >= 0 ? 1 : 0; tf.vjp_t_out = vjp_t_out;
    launch_theta_finalize(d, tf, st);
    return check_launch("augmented dynamics (F(4x4,3x3))");
  }

  int eval_fwd(const Comb& cy, float* y_out, const EvalTime& et, float* k_out, bool train, const NextComb* next = nullptr) {
    if (w4) return eval_w4(cy, y_out, et, k_out, train, nullptr, nullptr, nullptr, nullptr, -1, 0.f, nullptr, false, next);
    const bool tiny = tiny_mode() && !train;
    if (!(tiny && v_ready)) {      // (latency path: the previous evaluation's last launch may have formed this conv input already)
      CombineGnArgs ca;
      memset(&ca, 0, sizeof(ca));
      ca.comb = cy; ca.ctrl = p.ctrl; ca.y_out = y_out; ca.act_out = p.act1;
      ca.xhat_out = train ? p.xh1 : nullptr; ca.rstd_out = train ? p.r1 : nullptr;
      ca.gamma = prm.norm1_w; ca.beta = prm.norm1_b; ca.relu = 1; ca.osign = 1.f;
      launch_combine_gn(d, ca, st);
    }
    v_ready = false;

    if (tiny) {     // latency path: conv + bias + t * tmap + GroupNorm (+ ReLU) per launch
      TinyConvArgs t1;
      memset(&t1, 0, sizeof(t1));
      t1.act = p.act1; t1.wq = p.wtiny[0]; t1.bias = prm.conv1_b; t1.tmap = p.tmap[0]; t1.et = et;
      t1.gamma = prm.norm2_w; t1.beta = prm.norm2_b; t1.out = p.act2; t1.part = p.tpart; t1.counter = p.tcount; t1.ctrl = p.ctrl;
      t1.relu = 1; t1.osign = 1.f;
      { ProfScope ps(0, conv_flops(), st); launch_tiny_conv_gn(d, t1, st); }
      TinyConvArgs t2 = t1;
      t2.act = p.act2; t2.wq = p.wtiny[1]; t2.bias = prm.conv2_b; t2.tmap = p.tmap[1];
      t2.gamma = prm.norm3_w; t2.beta = prm.norm3_b; t2.out = k_out; t2.relu = 0; t2.osign = et.tsign;
      if (next != nullptr) {     // the next evaluation's combine -> GroupNorm-1 -> ReLU rides in this launch
        t2.nx_on = 1; t2.nx = next->cy; t2.nx_self = -1;
        for (int j = 0; j < next->cy.nk; ++j)
          if (next->cy.k[j] == k_out) t2.nx_self = j;
        t2.nx_y_out = next->y_out; t2.nx_gamma = prm.norm1_w; t2.nx_beta = prm.norm1_b; t2.nx_act = p.act1;
        v_ready = true;
      }
      { ProfScope ps(0, conv_flops(), st); launch_tiny_conv_gn(d, t2, st); }
      if (count_nfe) nfe += 1;
      return check_launch("odefunc forward (latency path)");
    }

    ConvArgs c1;
    memset(&c1, 0, sizeof(c1));
    c1.in = p.act1; c1.wpacked = p.wf[0]; c1.mode = CM_FWD_GN_RELU;
    c1.bias = prm.conv1_b; c1.tmap = p.tmap[0]; c1.et = et;
    c1.gamma = prm.norm2_w; c1.beta = prm.norm2_b; c1.osign = 1.f;
    c1.out = p.act2; c1.xhat_out = train ? p.xh2 : nullptr; c1.rstd_out = train ? p.r2 : nullptr;
    const bool small = small_mode() && !train;
    c1.raw_out = (d.csplit || small) ? p.RAW : nullptr;
    if (small) c1.wpacked = p.wsmall[0];
    { ProfScope ps(0, conv_flops(), st); if (small) launch_conv_small(d, c1, st); else launch_conv(d, c1, st); }
    if (d.csplit || small) gn_pass_fwd(prm.norm2_w, prm.norm2_b, 1, 1.f, p.act2, train ? p.xh2 : nullptr, train ? p.r2 : nullptr);

    ConvArgs c2 = c1;
    c2.in = p.act2; c2.wpacked = small ? p.wsmall[1] : p.wf[1]; c2.mode = CM_FWD_GN;
    c2.bias = prm.conv2_b; c2.tmap = p.tmap[1];
    c2.gamma = prm.norm3_w; c2.beta = prm.norm3_b; c2.osign = et.tsign;
    c2.out = k_out; c2.xhat_out = train ? p.xh3 : nullptr; c2.rstd_out = train ? p.r3 : nullptr;
    { ProfScope ps(0, conv_flops(), st); if (small) launch_conv_small(d, c2, st); else launch_conv(d, c2, st); }
    if (d.csplit || small) gn_pass_fwd(prm.norm3_w, prm.norm3_b, 0, et.tsign, k_out, train ? p.xh3 : nullptr, train ? p.r3 : nullptr);
    if (count_nfe) nfe += 1;
    return check_launch("odefunc forward");
  }

  // augmented dynamics: (f, csign*a^T df/dy, csign*a^T df/dt, csign*a^T df/dtheta) * tsign
  //   upstream adjoint: csign = -1.   kT_out / scalar ts_k[kidx] optional.
  // need_theta = false: the parameter / time components of this stage derivative are never consumed (dopri5
  // stage 2: b_2 = b^_2 = c_mid,2 = 0 and no stage STATE of those segments is ever formed, since f does not depend
  // on them), so the two weight-gradient GEMMs and the finalize are skipped; kT_out / ts_k[kidx] keep their
  // (finite, zero-weighted) contents.
  int eval_aug(const Comb& cy, const Comb& ca, float* y_out, float* a_out, const EvalTime& et,
               float* kY_out, float* kA_out, float* kT_out, int kidx, float csign, float* vjp_t_out,
               bool need_theta = true, const NextComb* next = nullptr) {
    if (w4) return eval_w4(cy, y_out, et, kY_out, true, &ca, a_out, kA_out, kT_out, kidx, csign, vjp_t_out, need_theta, next);
    TRY(eval_fwd(cy, y_out, et, kY_out, true));

    GnBwdArgs g;
    memset(&g, 0, sizeof(g));
    g.comb = ca; g.ctrl = p.ctrl; g.csign = csign; g.a_out = a_out;
    g.xhat = p.xh3; g.rstd = p.r3; g.gamma = prm.norm3_w; g.dz_out = p.dz2; g.gpart = p.gpart[2]; g.osign = 1.f;
    static int fuse_colsum = -1;   // NODE_TUNE_FUSE_COLSUM=0: separate k_colsum launches (A/B measurements)
    if (fuse_colsum < 0) { const char* e = getenv("NODE_TUNE_FUSE_COLSUM"); fuse_colsum = e ? atoi(e) : 1; }
    g.spart = fuse_colsum ? p.spart[1] : nullptr;   // masked column sums of dz2, fused (k_colsum otherwise)
    launch_gn_bwd(d, g, st);
    if (!fuse_colsum) launch_colsum(d, p.dz2, p.spart[1], st);

    if (need_theta && !d.wgrad_pair) {
      WgradArgs w2;
      memset(&w2, 0, sizeof(w2));
      w2.act = p.act2; w2.dz = p.dz2; w2.wpart = p.wpart[1]; w2.ctrl = p.ctrl;
      { ProfScope ps(1, conv_flops(), st); launch_wgrad(d, w2, st); }
    }

    ConvArgs b2;
    memset(&b2, 0, sizeof(b2));
    b2.in = p.dz2; b2.wpacked = p.wd[1]; b2.mode = CM_BWD_RELU_GN; b2.et = et;
    b2.gamma = prm.norm2_w; b2.osign = 1.f; b2.out = p.dz1;
    b2.act = p.act2; b2.xhat = p.xh2; b2.rstd = p.r2; b2.gpart = p.gpart[1];
    b2.spart = fuse_colsum ? p.spart[0] : nullptr;   // masked column sums of dz1, fused into the epilogue
    b2.raw_out = d.csplit ? p.RAW : nullptr;
    { ProfScope ps(0, conv_flops(), st); launch_conv(d, b2, st); }
    if (d.csplit) gn_pass_bwd(p.act2, p.xh2, p.r2, prm.norm2_w, 1.f, p.dz1, p.gpart[1], fuse_colsum ? p.spart[0] : nullptr);
    if (!fuse_colsum) launch_colsum(d, p.dz1, p.spart[0], st);

    if (need_theta) {   // dz1 exists now: with pairing, conv2's weight gradient rides in the same launch (grid.z = 1)
      WgradArgs w1;
      memset(&w1, 0, sizeof(w1));
      w1.act = p.act1; w1.dz = p.dz1; w1.wpart = p.wpart[0]; w1.ctrl = p.ctrl;
      if (d.wgrad_pair) { w1.act2 = p.act2; w1.dz2 = p.dz2; w1.wpart2 = p.wpart[1]; }
      { ProfScope ps(1, (d.wgrad_pair ? 2.0 : 1.0) * conv_flops(), st); launch_wgrad(d, w1, st); }
    }

    ConvArgs b1 = b2;
    b1.spart = nullptr;
    b1.in = p.dz1; b1.wpacked = p.wd[0];
    b1.gamma = prm.norm1_w; b1.osign = et.tsign; b1.out = kA_out;
    b1.act = p.act1; b1.xhat = p.xh1; b1.rstd = p.r1; b1.gpart = p.gpart[0];
    { ProfScope ps(0, conv_flops(), st); launch_conv(d, b1, st); }
    if (d.csplit) gn_pass_bwd(p.act1, p.xh1, p.r1, prm.norm1_w, et.tsign, kA_out, p.gpart[0], nullptr);

    if (!need_theta) return check_launch("augmented dynamics");
    ThetaFinalizeArgs tf;
    memset(&tf, 0, sizeof(tf));
    tf.wpart[0] = p.wpart[0]; tf.wpart[1] = p.wpart[1];
    tf.spart[0] = p.spart[0]; tf.spart[1] = p.spart[1];
    tf.gpart[0] = p.gpart[0]; tf.gpart[1] = p.gpart[1]; tf.gpart[2] = p.gpart[2];
    tf.gpart_rows[0] = tf.gpart_rows[1] = d.csplit ? d.N : d.mtiles;   // (split-conv mode: per-sample partials from the GroupNorm pass)
    tf.gpart_rows[2] = d.N;
    tf.wtime[0] = p.wtime[0]; tf.wtime[1] = p.wtime[1]; tf.sred = p.sred;
    tf.et = et; tf.osign = et.tsign; tf.theta_out = kT_out;
    tf.ctrl = p.ctrl; tf.kidx = kidx; tf.write_scalar = kidx >= 0 ? 1 : 0; tf.vjp_t_out = vjp_t_out;
    launch_theta_finalize(d, tf, st);
    return check_launch("augmented dynamics");
  }

  EvalTime et_stage(double alpha) const { EvalTime e; e.ctrl = p.ctrl; e.alpha = (float)alpha; e.tsign = tsign; e.mode = TM_STAGE; return e; }
  EvalTime et_probe() const { EvalTime e; e.ctrl = p.ctrl; e.alpha = 0.f; e.tsign = tsign; e.mode = TM_PROBE; return e; }

  // evaluate the system at (state + scale * sum coef_j k_j) into k[kout]
  // next_coef (F(4x4,3x3) solves): the Butcher row of the evaluation that follows -- its combine rides in this one's last pass
  int eval_sys(int kout, const double* coef, int ncoef, int scale_mode, const EvalTime& et, bool write_new,
               bool need_theta = true, const double* next_coef = nullptr, int next_ncoef = 0, bool next_write_new = false) {
    Comb cy = make_comb(p.Y, p.KY, coef, ncoef, scale_mode);
    NextComb nc;
    const NextComb* next = nullptr;
    if ((w4 || (tiny_mode() && !aug)) && next_coef != nullptr) {
      nc.cy = make_comb(p.Y, p.KY, next_coef, next_ncoef, scale_mode);
      nc.y_out = next_write_new ? p.Y1 : nullptr;
      next = &nc;
    }
    if (!aug) return eval_fwd(cy, write_new ? p.Y1 : nullptr, et, p.KY[kout], false, next);
    Comb ca = make_comb(p.A, p.KA, coef, ncoef, scale_mode);
    return eval_aug(cy, ca, write_new ? p.Y1 : nullptr, write_new ? p.A1 : nullptr, et,
                    p.KY[kout], p.KA[kout], p.KT[kout], kout, -1.f, nullptr, need_theta, next);
  }

  // (measured and not kept, profiles/r05_poll_readback_ab.txt: the record stored into coherent pinned memory by a one-thread launch and
  //  a host spin on its sequence number instead of copy + synchronise -- 0.884 vs 0.883 of the deferred rate: hipStreamSynchronize spins too)
  int readback() {
    HIP_TRY(hipMemcpyAsync(hctrl, p.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NODE_OK;
  }

  // Hairer initial step; leaves dt in ctrl.  Costs one probe eval (upstream: +1 NFE).
  int initial_step() {
    const int nseg = aug ? 3 : 1;
    InitSeg segs[3] = {{p.Y, p.KY[0], p.KY[1], d.numel}, {p.A, p.KA[0], p.KA[1], d.numel}, {p.TH, p.KT[0], p.KT[1], d.P}};
    InitCtlArgs ic;
    memset(&ic, 0, sizeof(ic));
    ic.ctrl = p.ctrl;
    for (int i = 0; i < nseg; ++i) { ic.partial[i] = p.partial[i]; ic.numel[i] = (double)segs[i].n; }
    ic.nseg = nseg; ic.has_scalar = aug ? 1 : 0; ic.phase = 0; ic.rtol = rtol; ic.atol = atol;
    const bool fold = nr_fn == nullptr && fold_ctl();     // norms + decision as one launch (the last workgroup decides)
    if (fold) launch_init_norms_ctl(segs, p.partial, nseg, ic, p.arrive, st);
    else {
      launch_init_norms(segs, p.partial, nseg, rtol, atol, 0, st);
      if (nr_fn != nullptr) { norm_exchange(1, nseg); ic.gbuf = nr_buf; ic.gworld = nr_world; }
      launch_init_controller(ic, st);
    }
    const double one[1] = {1.0};
    TRY(eval_sys(1, one, 1, SC_H0, et_probe(), false));
    ic.phase = 1;
    if (fold) launch_init_norms_ctl(segs, p.partial, nseg, ic, p.arrive, st);
    else {
      launch_init_norms(segs, p.partial, nseg, rtol, atol, 1, st);
      if (nr_fn != nullptr) norm_exchange(2, nseg);
      launch_init_controller(ic, st);
    }
    return check_launch("initial step");
  }

  // Where a run of dopri5 steps writes: the interval's target times, the replay list, the dt log (device arrays),
  // and -- forward solve -- the caller's trajectory.
  struct StepIO {
    int n_targets = 0;
    int n_forced = 0;        // > 0: replay mode
    int log_cap = 0;         // > 0: dt log wanted
    float* y_out = nullptr;  // forward: [n_targets][N][C][H][W], slot j <-> target j
  };

  // one dopri5 step, entirely on the device: six stages, error norms, controller (accept / dt / targets passed),
  // dense output, commit.  The host learns nothing here; see run_steps().
  int enqueue_step(const StepIO& io) {
    // stage 2 (s == 0): its parameter / time derivative has zero weight everywhere (see eval_aug)
    static int skip_k2 = -1;
    if (skip_k2 < 0) { const char* e = getenv("NODE_TUNE_SKIP_K2_THETA"); skip_k2 = e ? atoi(e) : 1; }
    const bool was_counting = count_nfe;
    count_nfe = false;
    for (int s = 0; s < 6; ++s) {
      const int rc = eval_sys(s + 1, DP_BETA[s], s + 1, SC_DT, et_stage(DP_ALPHA[s]), s == 5, !(skip_k2 && s == 0),
                              s < 5 ? DP_BETA[s + 1] : nullptr, s + 2, s + 1 == 5);
      if (rc != NODE_OK) { count_nfe = was_counting; return rc; }
    }
    count_nfe = was_counting;
    const int nseg = aug ? 3 : 1;
    ErrSeg es[3];
    es[0].y0 = p.Y; es[0].y1 = p.Y1; es[0].n = d.numel; es[0].compute_y1 = 0;
    for (int j = 0; j < 7; ++j) es[0].k[j] = p.KY[j];
    if (aug) {
      es[1] = es[0];
      es[1].y0 = p.A; es[1].y1 = p.A1;
      for (int j = 0; j < 7; ++j) es[1].k[j] = p.KA[j];
      es[2].y0 = p.TH; es[2].y1 = p.TH1; es[2].n = d.P; es[2].compute_y1 = 1;
      for (int j = 0; j < 7; ++j) es[2].k[j] = p.KT[j];
    }
    StepCtlArgs sc;
    memset(&sc, 0, sizeof(sc));
    sc.ctrl = p.ctrl;
    sc.partial[0] = p.partial[0]; sc.partial[1] = p.partial[1]; sc.partial[2] = p.partial[2];
    sc.numel[0] = (double)d.numel; sc.numel[1] = (double)d.numel; sc.numel[2] = (double)d.P;
    sc.nseg = nseg; sc.has_scalar = aug ? 1 : 0; sc.rtol = rtol; sc.atol = atol;
    sc.targets = p.targets; sc.n_targets = io.n_targets;
    sc.forced = io.n_forced > 0 ? p.forced : nullptr; sc.n_forced = io.n_forced;
    sc.dt_log = io.log_cap > 0 ? p.dtlog : nullptr; sc.dt_log_cap = io.log_cap;
    sc.interp_scalar = aug ? 1 : 0;
    sc.w4sc = (aug && w4_f16) ? p.w4sc : nullptr;
    if (nr_fn == nullptr && fold_ctl()) {
      launch_error_norm_ctl(es, p.partial, nseg, sc, p.arrive, st);     // the last-arriving workgroup of the error norm is the controller
    } else {
      launch_error_norm(es, p.partial, nseg, p.ctrl, rtol, atol, st);
      if (nr_fn != nullptr) { norm_exchange(0, nseg); sc.gbuf = nr_buf; sc.gworld = nr_world; }
      launch_step_controller(sc, st);
    }
    if (!aug) {
      EmitArgs ea;
      ea.ctrl = p.ctrl; ea.targets = p.targets; ea.y0 = p.Y; ea.y1 = p.Y1;
      for (int j = 0; j < 7; ++j) ea.k[j] = p.KY[j];
      ea.y_out = io.y_out;
      if (w4) launch_w4s_emit_outputs(d, ea, st);
      else launch_emit_outputs(d, ea, st);
    }
    CommitArgs cm;
    memset(&cm, 0, sizeof(cm));
    cm.ctrl = p.ctrl; cm.targets = p.targets; cm.nseg = nseg; cm.interp_final = aug ? 1 : 0;
    cm.y[0] = p.Y; cm.y1[0] = p.Y1; cm.k0[0] = p.KY[0]; cm.k6[0] = p.KY[6]; cm.n[0] = d.numel;
    for (int j = 0; j < 7; ++j) cm.k[0][j] = p.KY[j];
    if (aug) {
      // (the y segment is reloaded from the forward trajectory at every interval: its dense output is not needed,
      //  but it is cheap and keeps the three segments uniform)
      cm.y[1] = p.A; cm.y1[1] = p.A1; cm.k0[1] = p.KA[0]; cm.k6[1] = p.KA[6]; cm.n[1] = d.numel;
      cm.y[2] = p.TH; cm.y1[2] = p.TH1; cm.k0[2] = p.KT[0]; cm.k6[2] = p.KT[6]; cm.n[2] = d.P;
      for (int j = 0; j < 7; ++j) { cm.k[1][j] = p.KA[j]; cm.k[2][j] = p.KT[j]; }
    }
    launch_commit(cm, st);
    return check_launch("dopri5 step");
  }

  // Advance the current interval to its last target.  `guess` steps are enqueued before the first read-back (what the
  // previous solve of the same problem needed: one synchronisation per solve in steady state), then two at a time;
  // steps enqueued past the end return at once on the device (Ctrl::done).  On return *hctrl holds the final record.
  int run_steps(const StepIO& io, long long max_steps, int guess, int* status) {
    long long enq = 0;
    long long batch = guess > 0 ? guess : 1;
    for (;;) {
      if (batch > max_steps - enq) batch = max_steps - enq;
      for (long long i = 0; i < batch; ++i) TRY(enqueue_step(io));
      enq += batch;
      TRY(readback());
      if (hctrl->status != 0) { *status = hctrl->status; return NODE_OK; }
      if (hctrl->done) return NODE_OK;
      if (enq >= max_steps) { *status = NODE_ERR_MAX_STEPS; return NODE_OK; }
      batch = 2;
    }
  }

  // host -> device: the interval's target times (and, once per solve, the replay list)
  int upload(double* dst, const double* src, int n, double* stage) {
    for (int i = 0; i < n; ++i) stage[i] = src[i];
    HIP_TRY(hipMemcpyAsync(dst, stage, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    return NODE_OK;
  }

  // one interval of the fixed-grid RK4 (3/8 rule): state advanced in place
  int rk4_interval(double t0, double t1, const float* dot_with = nullptr, float* dot_out = nullptr) {
    // upstream keeps the fixed grid in the state dtype (fp32)
    const float t0f = (float)t0, t1f = (float)t1;
    launch_set_ctrl(p.ctrl, (double)t0f, (double)(t1f - t0f), 0, st);
    const double c2[1] = {1.0 / 3}, c3[2] = {-1.0 / 3, 1.0}, c4[3] = {1.0, -1.0, 1.0};
    const double cf[4] = {1.0 / 8, 3.0 / 8, 3.0 / 8, 1.0 / 8};
    TRY(eval_sys(0, nullptr, 0, SC_ABS, et_stage(0.0), false));
    if (dot_with) launch_dot_sub_scalar(p.ctrl, p.KY[0], dot_with, d.numel, tsign, p.partial[0], dot_out, st);   // adjoint: adj_t -= <f_i, g_i>
    TRY(eval_sys(1, c2, 1, SC_DT, et_stage(1.0 / 3), false));
    TRY(eval_sys(2, c3, 2, SC_DT, et_stage(2.0 / 3), false));
    TRY(eval_sys(3, c4, 3, SC_DT, et_stage(1.0), false));
    launch_lincomb(make_comb(p.Y, p.KY, cf, 4, SC_DT), p.ctrl, p.Y1, d.numel, st);
    std::swap(p.Y, p.Y1);
    if (aug) {
      launch_lincomb(make_comb(p.A, p.KA, cf, 4, SC_DT), p.ctrl, p.A1, d.numel, st);
      std::swap(p.A, p.A1);
      launch_lincomb(make_comb(p.TH, p.KT, cf, 4, SC_DT), p.ctrl, p.TH1, d.P, st);
      std::swap(p.TH, p.TH1);
      launch_set_scalar_state(p.ctrl, 0.f, 1, st);
    }
    return check_launch("rk4 interval");
  }
};

struct DtLog {
  const node_solve_opts* o;
  int n = 0;
  explicit DtLog(const node_solve_opts* opts) : o(opts) { if (o && o->n_dt_log) *o->n_dt_log = 0; }
  void add(double dt, bool accepted) {
    if (!o || o->record_dt <= 0 || !o->dt_log) return;
    if (n < o->record_dt) o->dt_log[n] = accepted ? dt : -dt;
    ++n;
    if (o->n_dt_log) *o->n_dt_log = n < o->record_dt ? n : o->record_dt;
  }
};

int check_common(const node_shape* shape, const node_params* params, void* ws, size_t ws_bytes, int adjoint, int n_t,
                 Dims* d, Plan* plan) {
  if (!params) return fail(NODE_ERR_NULL, "params is NULL");
  if (!ws) return fail(NODE_ERR_NULL, "workspace is NULL");
  TRY(make_dims(shape, d));
  const float* ptrs[10] = {params->norm1_w, params->norm1_b, params->conv1_w, params->conv1_b, params->norm2_w,
                           params->norm2_b, params->conv2_w, params->conv2_b, params->norm3_w, params->norm3_b};
  for (int i = 0; i < 10; ++i) {
    if (!ptrs[i]) return fail(NODE_ERR_NULL, "parameter pointer %d is NULL", i);
    if (((uintptr_t)ptrs[i]) & 15) return fail(NODE_ERR_ARG, "parameter pointer %d is not 16-byte aligned", i);
  }
  if (((uintptr_t)ws) & 255) return fail(NODE_ERR_ARG, "workspace must be 256-byte aligned");
  *plan = make_plan(*d, adjoint, n_t, ws);
  if (ws_bytes < plan->bytes) return fail(NODE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, plan->bytes);
  return NODE_OK;
}

int check_times(const float* t_pts, int n_t) {
  if (!t_pts) return fail(NODE_ERR_NULL, "t_pts is NULL");
  if (n_t < 2) return fail(NODE_ERR_ARG, "need at least two time points (got %d)", n_t);
  bool inc = true, dec = true;
  for (int i = 1; i < n_t; ++i) {
    if (!(t_pts[i] > t_pts[i - 1])) inc = false;
    if (!(t_pts[i] < t_pts[i - 1])) dec = false;
  }
  if (!inc && !dec) return fail(NODE_ERR_ARG, "t must be strictly increasing or strictly decreasing");
  return NODE_OK;
}

int status_to_rc(int status) {
  if (status == 0) return NODE_OK;
  if (status == NODE_ERR_NONFINITE) return fail(NODE_ERR_NONFINITE, "non-finite error norm / state");
  return fail(status, "solver stopped with status %d", status);
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
extern "C" {

int node_abi_version(void) { return NODE_ABI_VERSION; }
const char* node_last_error(void) { return g_err; }

size_t node_param_count(const node_shape* shape) {
  if (!shape) return 0;
  return 18 * (size_t)shape->c * shape->c + 26 * (size_t)shape->c;
}

size_t node_workspace_bytes(const node_shape* shape, int /*method*/, int adjoint, int n_t) {
  Dims d;
  if (make_dims(shape, &d) != NODE_OK) return 0;
  Plan p = make_plan(d, adjoint, n_t, nullptr);
  return p.bytes;
}

int node_solve_is_resident(const node_shape* shape) {
  Dims d;
  if (make_dims(shape, &d) != NODE_OK) return 0;
  return d.tiny != 0 && tiny_resident_ok(d) ? 1 : 0;
}

int node_odefunc_fwd(const node_shape* shape, const node_params* params, float t, const float* y, float* f,
                     void* ws, size_t ws_bytes, void* stream) {
  w4_refresh_tuning();     // (the NODE_TUNE_W4_* switches: once per call, not per launch)
  if (!y || !f) return fail(NODE_ERR_NULL, "y / f is NULL");
  Solver S;
  TRY(check_common(shape, params, ws, ws_bytes, 0, 2, &S.d, &S.p));
  S.prm = *params; S.st = (hipStream_t)stream; S.aug = false; S.tsign = 1.f;
  S.choose_w4(false);
  TRY(S.prepare());
  launch_set_ctrl(S.p.ctrl, (double)t, 0.0, 1, S.st);
  S.to_state(y, S.p.Y);
  TRY(S.eval_sys(0, nullptr, 0, SC_ABS, S.et_stage(0.0), false));
  S.from_state(S.p.KY[0], f);
  return S.check_launch("node_odefunc_fwd");
}

// Diagnostics: one bias-free 3x3 convolution (pad 1) of an [N, C, 8, 8] tensor through the F(4x4,3x3) pipeline with
// stand-alone transform kernels around the component GEMMs -- what the solver fuses into its GroupNorm passes.
size_t node_conv3x3_w4_workspace_bytes(const node_shape* shape) {
  if (!shape) return 0;
  const size_t numel = (size_t)shape->n * shape->c * shape->h * shape->w;
  const int nv = (shape->n * (shape->h == 16 ? 4 : 1) + 7) & ~7;
  return (2 * numel + 2 * w4_v_elems(nv, shape->c) + w4_u_elems(shape->c)) * sizeof(float) +
         w4_ub_elems(shape->c) * sizeof(unsigned short) + sizeof(W4Scales) + 8 * 256;
}
int node_conv3x3_w4(const node_shape* shape, const float* weight, int dgrad, const float* x, float* y, void* ws,
                    size_t ws_bytes, void* stream) {
  w4_refresh_tuning();     // (the NODE_TUNE_W4_* switches: once per call, not per launch)
  if (!shape || !weight || !x || !y || !ws) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  Dims d;
  TRY(make_dims(shape, &d));
  const bool sq8 = d.H == 8 && d.W == 8, sq16 = d.H == 16 && d.W == 16;
  if (!((sq8 || sq16) && d.C % 64 == 0 && (d.N * (sq16 ? 4 : 1)) % 8 == 0))
    return fail(NODE_ERR_UNSUPPORTED, "the F(4x4,3x3) pipeline takes 8x8 (N %% 8 == 0) or 16x16 (N %% 2 == 0) images, C %% 64 == 0");
  const int Q = sq16 ? 4 : 1, Nv = d.N * Q;
  if (ws_bytes < node_conv3x3_w4_workspace_bytes(shape)) return fail(NODE_ERR_ARG, "workspace too small");
  if (((uintptr_t)ws) & 255) return fail(NODE_ERR_ARG, "workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Bump b(ws);
  float* xn = b.take<float>(d.numel);
  float* yn = b.take<float>(d.numel);
  float* V = b.take<float>(w4_v_elems(Nv, d.C));
  float* M = b.take<float>(w4_v_elems(Nv, d.C));
  float* U = b.take<float>(w4_u_elems(d.C));
  unsigned short* Ub = b.take<unsigned short>(w4_ub_elems(d.C));
  W4Scales* sc = b.take<W4Scales>(1);
  W4PackJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  const bool b16 = w4_uses_bf16(Nv, d.C);
  const bool f16 = w4_f16_fits(Nv, d.C);     // fp16-pair operands (k_w4_gemm64h): the scales from max|w| and max|x|
  if (f16) {
    (void)hipMemsetAsync(sc, 0, sizeof(W4Scales), st);
    W4ScaleJobs sj;
    memset(&sj, 0, sizeof(sj));
    sj.w[0] = weight; sj.wn = (size_t)d.C * (d.C + 1) * 9; sj.gb[1] = x; sj.vn[0] = d.numel; sj.C = d.C; sj.gn_m = 1; sj.sc = sc;
    launch_w4_scales(sj, st);
  }
  jobs.w[0] = weight; jobs.u[0] = U; jobs.ub[0] = (b16 && !f16) ? Ub : nullptr; jobs.dgrad[0] = dgrad ? 1 : 0;
  if (f16) { jobs.uh[0] = reinterpret_cast<unsigned*>(U); jobs.uh_exp[0] = &sc->e[W4_E_U1]; }
  launch_w4_pack(jobs, 1, d.C, st);
  launch_w4s_from_nchw(x, xn, d.N, d.C, Q, st);
  launch_w4_input(xn, V, d.N, d.C, Q, Nv, st, f16 ? &sc->e[W4_E_V1] : nullptr);
  if (f16) launch_w4_gemm_f16(reinterpret_cast<const unsigned*>(V), reinterpret_cast<const unsigned*>(U), M, nullptr, Nv, d.C, &sc->e[W4_E_V1], &sc->e[W4_E_U1], st);
  else launch_w4_gemm(V, U, M, nullptr, Nv, d.C, st, b16 ? Ub : nullptr);
  launch_w4_output(M, yn, d.N, d.C, Q, st);
  launch_w4s_to_nchw(yn, y, d.N, d.C, Q, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_w4_pair_stats(int32_t* out4) {
  if (!out4) return fail(NODE_ERR_NULL, "out4 is NULL");
  for (int i = 0; i < 4; ++i) out4[i] = g_w4_pair_stats[i];
  return NODE_OK;
}

// Diagnostics: the exact three-way bf16 split the component GEMMs apply to their fp32 row operands (k_w4_gemm64b and its
// siblings), element by element: out[3 i + p] = part p of x[i] as a float.
int node_w4_split3(const float* x, float* out, size_t n, void* stream) {
  if (!x || !out) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  if (n == 0 || n % 8 != 0) return fail(NODE_ERR_ARG, "n must be a positive multiple of 8");
  launch_w4_split_check(x, out, n, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_odefunc_vjp(const node_shape* shape, const node_params* params, float t, const float* y, const float* cot,
                     float* f, float* vjp_y, float* vjp_t, float* vjp_params, void* ws, size_t ws_bytes, void* stream) {
  w4_refresh_tuning();     // (the NODE_TUNE_W4_* switches: once per call, not per launch)
  if (!y || !cot || !f || !vjp_y || !vjp_t || !vjp_params) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  Solver S;
  TRY(check_common(shape, params, ws, ws_bytes, 1, 2, &S.d, &S.p));
  S.prm = *params; S.st = (hipStream_t)stream; S.aug = true; S.tsign = 1.f;
  S.choose_w4(false);
  TRY(S.prepare());
  launch_set_ctrl(S.p.ctrl, (double)t, 0.0, 1, S.st);
  S.to_state(y, S.p.Y);
  S.to_state(cot, S.p.A);
  Comb cy = Solver::make_comb(S.p.Y, S.p.KY, nullptr, 0, SC_ABS);
  Comb ca = Solver::make_comb(S.p.A, S.p.KA, nullptr, 0, SC_ABS);
  TRY(S.eval_aug(cy, ca, nullptr, nullptr, S.et_stage(0.0), S.p.KY[0], S.p.KA[0], S.p.KT[0], -1, +1.f, vjp_t));
  S.from_state(S.p.KY[0], f);
  S.from_state(S.p.KA[0], vjp_y);
  launch_theta_to_torch(S.d, S.p.KT[0], vjp_params, S.st);
  return S.check_launch("node_odefunc_vjp");
}

int node_solve_fwd(const node_shape* shape, const node_params* params, const float* y0, const float* t_pts, int n_t,
                   float rtol, float atol, int method, const node_solve_opts* opts, float* y_out, node_stats* stats,
                   void* ws, size_t ws_bytes, void* stream) {
  w4_refresh_tuning();
  if (!y0 || !y_out) return fail(NODE_ERR_NULL, "y0 / y_out is NULL");
  if (method != NODE_METHOD_DOPRI5 && method != NODE_METHOD_RK4) return fail(NODE_ERR_ARG, "unknown method %d", method);
  TRY(check_times(t_pts, n_t));
  Solver S;
  TRY(check_common(shape, params, ws, ws_bytes, 0, n_t, &S.d, &S.p));
  S.prm = *params; S.st = (hipStream_t)stream; S.aug = false; S.rtol = rtol; S.atol = atol;
  const bool forced = method == NODE_METHOD_DOPRI5 && opts && opts->n_forced_dt > 0 && opts->forced_dt;
  if (forced && opts->n_forced_dt > STEP_LIST_CAP) return fail(NODE_ERR_ARG, "replay list longer than %d", STEP_LIST_CAP);
  HostStage* hs = nullptr;
  TRY(get_stage((size_t)n_t + 2 * STEP_LIST_CAP, &hs));
  S.hctrl = hs->ctrl;
  const bool decreasing = t_pts[1] < t_pts[0];
  S.tsign = decreasing ? -1.f : 1.f;
  std::vector<double> ts(n_t);
  for (int i = 0; i < n_t; ++i) ts[i] = (double)(decreasing ? -t_pts[i] : t_pts[i]);
  node_stats stt;
  memset(&stt, 0, sizeof(stt));
  DtLog dlog(opts);
  const size_t numel = S.d.numel;

  S.choose_w4(method == NODE_METHOD_DOPRI5);   // (a replay of recorded steps runs the numerics of the solve it replays)
  S.choose_resident(method == NODE_METHOD_DOPRI5);
  if (method == NODE_METHOD_DOPRI5) S.take_norm_hook(opts);
  if (S.nr_fn != nullptr) S.resident = false;      // (global-norm mode: the host's hook sits between the launches of a step)
  if (!S.resident) {      // (the resident solve packs its filters, forms its border maps and copies y0 inside its one launch)
    TRY(S.prepare());
    S.to_state(y0, S.p.Y);
    HIP_TRY(hipMemcpyAsync(y_out, y0, numel * sizeof(float), hipMemcpyDeviceToDevice, S.st));
  }

  if (method == NODE_METHOD_RK4) {
    launch_set_ctrl(S.p.ctrl, ts[0], 0.0, 1, S.st);
    for (int j = 1; j < n_t; ++j) {
      TRY(S.rk4_interval(ts[j - 1], ts[j]));
      S.from_state(S.p.Y, y_out + (size_t)j * numel);
      stt.accepted += 1;
      dlog.add(ts[j] - ts[j - 1], true);
    }
    HIP_TRY(hipStreamSynchronize(S.st));
    stt.nfe = S.nfe; stt.t_final = ts[n_t - 1]; stt.last_dt = ts[n_t - 1] - ts[n_t - 2];
    if (stats) *stats = stt;
    return S.check_launch("node_solve_fwd(rk4)");
  }

  // ---- dopri5: every decision of the step loop is taken on the device ----
  const long long max_steps = (opts && opts->max_num_steps > 0) ? opts->max_num_steps : 2147483647LL;
  Solver::StepIO io;
  io.n_targets = n_t - 1;
  io.n_forced = forced ? opts->n_forced_dt : 0;
  io.log_cap = (opts && opts->record_dt > 0 && opts->dt_log) ? (opts->record_dt < STEP_LIST_CAP ? opts->record_dt : STEP_LIST_CAP) : 0;
  io.y_out = y_out + numel;
  // deferred completion: exactly `blind_steps` steps, no read-back, no host staging (nothing of this call may be
  // touched by the host after it returns); the outcome goes to the caller's device record
  const int blind = (opts && opts->blind_steps > 0 && opts->record && n_t == 2 && !forced && io.log_cap == 0)
                        ? (opts->blind_steps < max_steps ? opts->blind_steps : (int)max_steps) : 0;
  const bool inline_targets = S.resident && io.n_targets <= 8;      // (the target times ride in the kernel arguments)
  if (inline_targets) {
    if (blind) HIP_TRY(hipMemcpyAsync(y_out + numel, y0, numel * sizeof(float), hipMemcpyDeviceToDevice, S.st));
  } else if (blind) {
    launch_set_target(S.p.targets, ts[1], S.st);
    // a MISSED blind solve never emits its output: leave y0 there, not uninitialised memory (the caller's loss of
    // such a step is then a finite number of a step whose update is skipped anyway)
    HIP_TRY(hipMemcpyAsync(y_out + numel, y0, numel * sizeof(float), hipMemcpyDeviceToDevice, S.st));
  } else {
    TRY(S.upload(S.p.targets, ts.data() + 1, n_t - 1, hs->lists));
  }
  if (forced) TRY(S.upload(S.p.forced, opts->forced_dt, opts->n_forced_dt, hs->lists + n_t));
  if (S.resident) {
    // the whole solve -- f0, the initial step, every step with its decision, dense output -- is one launch; it needs as many steps
    // as it needs (a deferred solve of this kind cannot miss), and the host reads the same record back
    TinyResidentArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.y0 = y0; ra.y_first = y_out; ra.y_out = io.y_out;
    ra.w[0] = params->conv1_w; ra.w[1] = params->conv2_w;
    ra.bias[0] = params->conv1_b; ra.bias[1] = params->conv2_b;
    ra.gamma[0] = params->norm1_w; ra.gamma[1] = params->norm2_w; ra.gamma[2] = params->norm3_w;
    ra.beta[0] = params->norm1_b; ra.beta[1] = params->norm2_b; ra.beta[2] = params->norm3_b;
    if (blind && g_blind_resident_ctrl == nullptr) {
      if (hipHostMalloc((void**)&g_blind_resident_ctrl, sizeof(Ctrl), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g_blind_resident_ctrl = nullptr; }
      else memset(g_blind_resident_ctrl, 0, sizeof(Ctrl));
    }
    ra.handoff = S.p.thand; ra.ctrl = S.p.ctrl; ra.ctrl_host = blind ? g_blind_resident_ctrl : S.hctrl;
    {
      // every word that crosses workgroups carries {nonce, version}: stale words of any earlier solve of this process never match, so
      // nothing is zeroed per solve.  The 28-bit nonce starts over every 2^28 solves: a hand-off buffer is zeroed the first time it
      // is used under a new generation of the counter (and the first time ever: whatever a fresh allocation holds is gone).
      // Three values are skipped: 0 (zeroed memory) and the two that, with the "solve is over" version, are the 0xFF / 0x7F byte patterns
      static std::mutex g_mu;
      static unsigned long long g_count = 0;
      static std::vector<std::pair<void*, unsigned long long>> g_seen;
      std::lock_guard<std::mutex> lock(g_mu);
      unsigned nn;
      do { nn = (unsigned)(++g_count & 0x0FFFFFFFull); } while (nn == 0u || nn == 0x0FFFFFFFu || nn == 0x07F7F7F7u);
      const unsigned long long gen = (g_count >> 28) + 1;
      bool known = false;
      for (auto& e : g_seen)
        if (e.first == S.p.thand) { known = e.second == gen; e.second = gen; goto found; }
      if (g_seen.size() >= 256) g_seen.erase(g_seen.begin());
      g_seen.emplace_back(S.p.thand, gen);
    found:
      if (!known) HIP_TRY(hipMemsetAsync(S.p.thand, 0, tiny_resident_handoff_words(S.d) * 8, S.st));
      ra.nonce = nn;
    }
    ra.targets = inline_targets ? nullptr : S.p.targets; ra.n_targets = io.n_targets;
    if (inline_targets) for (int j = 0; j < io.n_targets; ++j) ra.targets_inline[j] = ts[j + 1];
    ra.forced = forced ? S.p.forced : nullptr; ra.n_forced = io.n_forced;
    ra.dt_log = io.log_cap > 0 ? S.p.dtlog : nullptr; ra.dt_log_cap = io.log_cap;
    ra.t0 = ts[0]; ra.max_steps = max_steps;
    ra.rtol = rtol; ra.atol = atol; ra.tsign = S.tsign;
    if (!blind) { S.hctrl->done = 0; S.hctrl->status = NODE_ERR_HIP; }      // (overwritten by the launch: if it never ran, the record says so)
    launch_tiny_solve(S.d, ra, S.st);
    S.nfe = forced ? 1 : 2;
    if (blind) {
      launch_export_record(S.p.ctrl, opts->record, opts->miss_flag, 2147483647, S.st);
      stt.status = NODE_PENDING;
      stt.accepted = blind; stt.rejected = 0;
      stt.nfe = S.nfe + 6 * blind;
      stt.t_final = ts[1];
      if (stats) *stats = stt;
      return S.check_launch("node_solve_fwd(dopri5, resident, deferred)");
    }
    // the launch wrote the record into the pinned host copy itself: completion of the stream is all the host waits for
    HIP_TRY(hipStreamSynchronize(S.st));
    if (S.hctrl->status == NODE_ERR_HIP) {
      // a wait inside the launch ran into its deadline: the grid was not co-resident (other processes' resident grids on this
      // GPU can leave two launches each waiting for compute units the other holds).  Every workgroup has left; the solve runs
      // again on the launch-per-convolution path, which needs no co-residency.
      S.resident = false;
      g_resident_cooldown.store(64, std::memory_order_relaxed);
      S.nfe = 0;
      TRY(S.prepare());
      S.to_state(y0, S.p.Y);
      HIP_TRY(hipMemcpyAsync(y_out, y0, numel * sizeof(float), hipMemcpyDeviceToDevice, S.st));
      if (inline_targets) TRY(S.upload(S.p.targets, ts.data() + 1, n_t - 1, hs->lists));
    }
  }
  if (!S.resident) {
    launch_set_ctrl(S.p.ctrl, ts[0], forced ? opts->forced_dt[0] : 0.0, 1, S.st);
    TRY(S.eval_sys(0, nullptr, 0, SC_ABS, S.et_stage(0.0), false));  // f0 (FSAL seed)
    if (!forced) TRY(S.initial_step());
  }
  if (blind) {
    for (int i = 0; i < blind; ++i) TRY(S.enqueue_step(io));
    launch_export_record(S.p.ctrl, opts->record, opts->miss_flag, blind, S.st);
    stt.status = NODE_PENDING;
    stt.accepted = blind; stt.rejected = 0;            // predicted: true iff the record says no miss
    stt.nfe = S.nfe + 6 * blind;
    stt.t_final = ts[1];
    if (stats) *stats = stt;
    return S.check_launch("node_solve_fwd(dopri5, deferred)");
  }
  StepGuess key = {S.d.N, S.d.C, S.d.H, S.d.W, 0, forced ? 1 : 0, rtol, atol, ts[0], ts[n_t - 1], 0};
  int status = 0;
  if (S.resident) status = S.hctrl->status;
  else TRY(S.run_steps(io, max_steps, guess_steps(key), &status));
  const Ctrl& h = *S.hctrl;
  key.steps = h.step_idx;
  if (status == 0 && !S.resident) remember_steps(key);
  stt.status = status;
  stt.accepted = h.n_acc; stt.rejected = h.n_rej;
  stt.nfe = S.nfe + 6 * h.step_idx;     // f0 (+ the initial-step probe) + six stages per step tried (show.py:199)
  stt.first_dt = h.first_dt; stt.t_final = h.t; stt.last_dt = h.dt;
  if (io.log_cap > 0) {
    const int n = h.step_idx < io.log_cap ? h.step_idx : io.log_cap;
    double* stage = hs->lists;
    HIP_TRY(hipMemcpyAsync(stage, S.p.dtlog, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, S.st));
    HIP_TRY(hipStreamSynchronize(S.st));
    for (int i = 0; i < n; ++i) dlog.add(fabs(stage[i]), stage[i] > 0.0);
  }
  if (stats) *stats = stt;
  TRY(S.check_launch("node_solve_fwd(dopri5)"));
  if (stt.status == NODE_ERR_MAX_STEPS) return fail(NODE_ERR_MAX_STEPS, "max_num_steps exceeded");
  if (stt.status == NODE_ERR_DT_UNDERFLOW) return fail(NODE_ERR_DT_UNDERFLOW, "underflow in dt %g", h.dt);
  return status_to_rc(stt.status);
}

int node_solve_adjoint(const node_shape* shape, const node_params* params, const float* y_traj, const float* grad_out,
                       const float* t_pts, int n_t, float rtol, float atol, int method, const node_solve_opts* opts,
                       float* grad_y0, float* grad_params, float* grad_t, node_stats* stats, void* ws, size_t ws_bytes,
                       void* stream) {
  w4_refresh_tuning();
  if (!y_traj || !grad_out || !grad_y0 || !grad_params) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  if (method != NODE_METHOD_DOPRI5 && method != NODE_METHOD_RK4) return fail(NODE_ERR_ARG, "unknown method %d", method);
  TRY(check_times(t_pts, n_t));
  Solver S;
  TRY(check_common(shape, params, ws, ws_bytes, 1, n_t, &S.d, &S.p));
  S.prm = *params; S.st = (hipStream_t)stream; S.aug = true; S.rtol = rtol; S.atol = atol;
  const bool forced = opts && opts->n_forced_dt > 0 && opts->forced_dt && method == NODE_METHOD_DOPRI5;
  if (forced && opts->n_forced_dt > STEP_LIST_CAP) return fail(NODE_ERR_ARG, "replay list longer than %d", STEP_LIST_CAP);
  HostStage* hs = nullptr;
  TRY(get_stage((size_t)n_t + 2 * STEP_LIST_CAP, &hs));
  S.hctrl = hs->ctrl;
  node_stats stt;
  memset(&stt, 0, sizeof(stt));
  DtLog dlog(opts);
  const size_t numel = S.d.numel;
  const long long max_steps = (opts && opts->max_num_steps > 0) ? opts->max_num_steps : 2147483647LL;
  Solver::StepIO io;
  io.n_targets = 1;
  io.n_forced = forced ? opts->n_forced_dt : 0;
  io.log_cap = (opts && opts->record_dt > 0 && opts->dt_log) ? (opts->record_dt < STEP_LIST_CAP ? opts->record_dt : STEP_LIST_CAP) : 0;

  const int blind = (opts && opts->blind_steps > 0 && opts->record && n_t == 2 && !forced && io.log_cap == 0 &&
                     method == NODE_METHOD_DOPRI5)
                        ? (opts->blind_steps < max_steps ? opts->blind_steps : (int)max_steps) : 0;
  S.choose_w4(method == NODE_METHOD_DOPRI5);   // (a replay of recorded steps runs the numerics of the solve it replays)
  S.w4_f16_aug = method == NODE_METHOD_DOPRI5;
  if (method == NODE_METHOD_DOPRI5) S.take_norm_hook(opts);
  TRY(S.prepare());
  int w4_gskew = 0;           // diagnostics (include/node_hip.h, node_w4_pair_stats)
  bool w4_stats = false;
  { const char* e = getenv("NODE_TUNE_W4_GSKEW"); if (e != nullptr) w4_gskew = atoi(e); }
  { const char* e = getenv("NODE_TUNE_W4_STATS"); w4_stats = e != nullptr && atoi(e) != 0; }
  g_w4_pair_stats[0] = S.w4_f16 ? 1 : 0; g_w4_pair_stats[1] = -1; g_w4_pair_stats[2] = 0; g_w4_pair_stats[3] = 0;
  launch_set_ctrl(S.p.ctrl, 0.0, 0.0, 1, S.st);  // also zeroes the scalar segment (adj_time = 0)
  launch_fill(S.p.TH, 0.f, S.d.P, S.st);         // adj_params = 0
  // grad_last_only: `grad_out` is the last slice alone, every other slice of dL/dy_out is zero (node_solve_opts)
  const bool last_only = opts && opts->grad_last_only;
  const float* g_last = last_only ? grad_out : grad_out + (size_t)(n_t - 1) * numel;
  S.to_state(g_last, S.p.A);  // adj_y = grad_output[-1]
  if (forced) TRY(S.upload(S.p.forced, opts->forced_dt, opts->n_forced_dt, hs->lists + n_t));
  double cur_t = 0.0, cur_dt = 0.0;
  bool first = true;
  int steps_total = 0;

  for (int i = n_t - 1; i >= 1 && stt.status == 0; --i) {
    // the interval is integrated from t_i to t_{i-1}; upstream negates time when that is decreasing
    const bool decreasing = t_pts[i - 1] < t_pts[i];
    S.tsign = decreasing ? -1.f : 1.f;
    const double s0 = (double)(decreasing ? -t_pts[i] : t_pts[i]);
    const double s1 = (double)(decreasing ? -t_pts[i - 1] : t_pts[i - 1]);

    S.to_state(y_traj + (size_t)i * numel, S.p.Y);
    // grad_output_i in NHWC for the dot product below: in the first interval the adjoint state still IS it; a zero
    // slice (grad_last_only) contributes nothing
    const float* gdot = i == n_t - 1 ? S.p.A : (last_only ? nullptr : S.p.G);
    if (i != n_t - 1 && !last_only) S.to_state(grad_out + (size_t)i * numel, S.p.G);
    // func_i = f(t_i, y_i); adj_time -= <func_i, grad_output_i>.  Upstream evaluates f here and again as
    // the first stage of the augmented solve at the same (t_i, y_i); the stage-0 evaluation below
    // produces tsign * f bit-identically, so the dot product is taken from it (times tsign) and the
    // separate evaluation is only COUNTED (the reference's nfe counter, model.py:340, would have seen it).
    S.nfe += 1;
    float* dots_i = grad_t ? S.p.dots + i : nullptr;
    if (gdot == nullptr && dots_i) launch_fill(dots_i, 0.f, 1, S.st);

    if (method == NODE_METHOD_RK4) {
      TRY(S.rk4_interval(s0, s1, gdot, gdot ? dots_i : nullptr));
      stt.accepted += 1;
      dlog.add(s1 - s0, true);
      cur_t = s1; cur_dt = s1 - s0;
    } else {
      // replay list restarts per interval (one odeint call each upstream)
      launch_set_interval(S.p.ctrl, s0, forced ? opts->forced_dt[0] : 0.0, S.st);
      if (blind) launch_set_target(S.p.targets, s1, S.st);
      else TRY(S.upload(S.p.targets, &s1, 1, hs->lists + (n_t - 1 - i) % n_t));
      S.g_ready = false;      // (fp16-pair operands: the interval's first evaluation runs the triples and records max|dz|, wino4.h)
      TRY(S.eval_sys(0, nullptr, 0, SC_ABS, S.et_stage(0.0), false));
      if (S.w4_f16) { launch_w4_gscale(S.p.w4sc, S.st, w4_gskew); S.g_ready = true; }
      if (gdot) launch_dot_sub_scalar(S.p.ctrl, S.p.KY[0], gdot, numel, S.tsign, S.p.partial[0], dots_i, S.st);
      if (!forced) TRY(S.initial_step());
      if (blind) {   // deferred completion (one interval): the record says later whether these were the steps needed
        for (int q = 0; q < blind; ++q) TRY(S.enqueue_step(io));
        launch_export_record(S.p.ctrl, opts->record, opts->miss_flag, blind, S.st);
        steps_total += blind;
        stt.accepted = blind; stt.rejected = 0; stt.status = 0;
        cur_t = s1; cur_dt = 0.0;
        if (!last_only) {
          S.to_state(grad_out + (size_t)(i - 1) * numel, S.p.G);
          launch_axpy(S.p.A, S.p.G, 1.f, numel, S.st);
        }
        continue;
      }
      StepGuess key = {S.d.N, S.d.C, S.d.H, S.d.W, 1, forced ? 1 : 0, rtol, atol, s0, s1, 0};
      int status = 0;
      // the dense output of the adjoint, parameter and time segments at s1 happens on the device with the last step
      TRY(S.run_steps(io, max_steps, guess_steps(key), &status));
      const Ctrl& h = *S.hctrl;
      key.steps = h.step_idx;
      if (status == 0) remember_steps(key);
      stt.status = status;
      if (first) { stt.first_dt = h.first_dt; first = false; }
      steps_total += h.step_idx;
      stt.accepted = h.n_acc; stt.rejected = h.n_rej;     // cumulative over the intervals
      cur_t = h.t; cur_dt = h.dt;
      if (io.log_cap > 0) {
        const int n = h.step_idx < io.log_cap ? h.step_idx : io.log_cap;
        double* stage = hs->lists + n_t + STEP_LIST_CAP;
        HIP_TRY(hipMemcpyAsync(stage, S.p.dtlog, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, S.st));
        HIP_TRY(hipStreamSynchronize(S.st));
        for (int q = 0; q < n; ++q) dlog.add(fabs(stage[q]), stage[q] > 0.0);
      }
      if (stt.status != 0) break;
    }
    // adj_y += grad_output[i-1]
    if (!last_only) {
      S.to_state(grad_out + (size_t)(i - 1) * numel, S.p.G);
      launch_axpy(S.p.A, S.p.G, 1.f, numel, S.st);
    }
  }

  S.from_state(S.p.A, grad_y0);
  launch_theta_to_torch(S.d, S.p.TH, grad_params, S.st);
  if (grad_t) {
    // time_vjps = [adj_time, dLd_t1, ..., dLd_t_{T-1}]
    launch_copy_scalar_out(S.p.ctrl, S.p.dots, S.st);
    HIP_TRY(hipMemcpyAsync(grad_t, S.p.dots, (size_t)n_t * sizeof(float), hipMemcpyDeviceToDevice, S.st));
  }
  if (blind) {
    stt.status = NODE_PENDING;
    stt.nfe = S.nfe + 6 * steps_total;
    stt.t_final = cur_t;
    if (stats) *stats = stt;
    return S.check_launch("node_solve_adjoint(deferred)");
  }
  // (dopri5: every interval ended with a read-back, behind which nothing of this call is staged on the host -- the launches above are
  //  ordinary stream work the caller's next launches queue behind, and the host does not wait for them; rk4 has no read-back)
  if (method == NODE_METHOD_RK4) HIP_TRY(hipStreamSynchronize(S.st));
  if (w4_stats && S.w4_f16) {   // (diagnostics: two words of the scale block; the read-back above left the stream idle)
    W4Scales hsc;
    HIP_TRY(hipMemcpyAsync(&hsc, S.p.w4sc, offsetof(W4Scales, pad), hipMemcpyDeviceToHost, S.st));
    HIP_TRY(hipStreamSynchronize(S.st));
    g_w4_pair_stats[1] = hsc.n_retry;
    g_w4_pair_stats[2] = hsc.e[W4_E_G];
  }
  stt.nfe = S.nfe + 6 * steps_total;
  stt.t_final = cur_t; stt.last_dt = cur_dt;
  if (stats) *stats = stt;
  TRY(S.check_launch("node_solve_adjoint"));
  if (stt.status == NODE_ERR_MAX_STEPS) return fail(NODE_ERR_MAX_STEPS, "max_num_steps exceeded");
  if (stt.status == NODE_ERR_DT_UNDERFLOW) return fail(NODE_ERR_DT_UNDERFLOW, "underflow in dt %g", cur_dt);
  return status_to_rc(stt.status);
}

// ----------------------------------------------------------------------------
// Backward of the NON-adjoint `odeint` (model.py:359 with adjoint=False, the constructor default model.py:7):
// upstream differentiates through the solver's own operations.  Here: the accepted steps are replayed from y0 with
// the recorded step sizes (the kernels the forward solve ran -- Solver::choose_w4 on its tolerances -- so the stage
// values are the ones its output was computed from), every stage derivative is kept on a tape, and
// the cotangents walk the steps backwards -- one VJP of the dynamics per stage evaluation, the Butcher rows and the
// dense-output polynomial transposed.  Step sizes are treated as constants (upstream's 2019 controller is itself
// differentiable; that sensitivity is O(local error) and is not reproduced -- DESIGN.md).
// ----------------------------------------------------------------------------
namespace {
struct Tape {
  std::vector<float*> Y;   // state at the start of step n (n = 0..S), NHWC
  std::vector<float*> K;   // stage derivatives: dopri5 K[6 n + i], i = 0..6 (k6 of step n IS k0 of step n + 1); rk4 K[4 n + i]
  float* KB[7];            // cotangents of the stage derivatives of the step being processed
  float* YB;               // cotangent of the step's end state
  float* Y0B;              // cotangent of the step's start state (being assembled)
  float* G;                // one slice of grad_out, NHWC
};
size_t backprop_tensors(int method, int n_steps) {
  return (size_t)(method == NODE_METHOD_DOPRI5 ? 7 * n_steps + 2 : 5 * n_steps + 1) + 10;
}
}  // namespace

extern "C" size_t node_backprop_workspace_bytes(const node_shape* shape, int method, int n_t, int n_steps) {
  Dims d;
  if (make_dims(shape, &d) != NODE_OK || n_steps < 1) return 0;
  Plan p = make_plan(d, 1, n_t, nullptr);
  return p.bytes + backprop_tensors(method, n_steps) * (((d.numel * sizeof(float)) + 255) & ~(size_t)255) + 256;
}

extern "C" int node_solve_backprop(const node_shape* shape, const node_params* params, const float* y0, const float* t_pts,
                                   int n_t, const double* step_dt, int n_steps, float rtol, float atol, int method, const float* grad_out,
                                   float* grad_y0, float* grad_params, void* ws, size_t ws_bytes, void* stream) {
  w4_refresh_tuning();
  if (!y0 || !grad_out || !grad_y0 || !grad_params) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  if (method != NODE_METHOD_DOPRI5 && method != NODE_METHOD_RK4) return fail(NODE_ERR_ARG, "unknown method %d", method);
  TRY(check_times(t_pts, n_t));
  const bool dopri = method == NODE_METHOD_DOPRI5;
  if (dopri && (!step_dt || n_steps < 1)) return fail(NODE_ERR_ARG, "dopri5 backprop needs the forward solve's accepted step sizes");
  if (!dopri) n_steps = n_t - 1;
  Solver S;
  TRY(check_common(shape, params, ws, ws_bytes, 1, n_t, &S.d, &S.p));
  const size_t need = node_backprop_workspace_bytes(shape, method, n_t, n_steps);
  if (ws_bytes < need) return fail(NODE_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);
  S.prm = *params; S.st = (hipStream_t)stream; S.aug = true; S.rtol = rtol; S.atol = atol;
  S.choose_w4(dopri);   // the forward solve's decision (node_solve_fwd): the replay below runs the kernels it ran
  const Dims& d = S.d;
  const size_t numel = d.numel;
  const bool decreasing = t_pts[1] < t_pts[0];
  S.tsign = decreasing ? -1.f : 1.f;
  std::vector<double> ts(n_t);
  for (int i = 0; i < n_t; ++i) ts[i] = (double)(decreasing ? -t_pts[i] : t_pts[i]);

  Tape tp;
  {
    Bump b((char*)ws + S.p.bytes);
    tp.Y.resize(n_steps + 1);
    for (auto& q : tp.Y) q = b.take<float>(numel);
    tp.K.resize(dopri ? 6 * n_steps + 1 : 4 * n_steps);
    for (auto& q : tp.K) q = b.take<float>(numel);
    for (int i = 0; i < 7; ++i) tp.KB[i] = b.take<float>(numel);
    tp.YB = b.take<float>(numel);
    tp.Y0B = b.take<float>(numel);
    tp.G = b.take<float>(numel);
  }
  TRY(S.prepare());
  S.count_nfe = false;

  // ---- replay: the forward solve's accepted steps, every stage derivative kept ----
  std::vector<double> tn(n_steps + 1), dtn(n_steps);
  std::vector<int> out_step(n_t, -1);          // which step's dense output produced y_out[j]
  std::vector<float> out_x(n_t, 0.f);
  S.to_state(y0, tp.Y[0]);
  const double c2[1] = {1.0 / 3}, c3[2] = {-1.0 / 3, 1.0}, c4[3] = {1.0, -1.0, 1.0};
  const double* rk4_rows[3] = {c2, c3, c4};
  const double rk4_alpha[3] = {1.0 / 3, 2.0 / 3, 1.0};
  const double rk4_b[4] = {1.0 / 8, 3.0 / 8, 3.0 / 8, 1.0 / 8};
  auto comb_of = [&](const float* y, float* const* k, const double* coef, int nc) { return Solver::make_comb(y, k, coef, nc, SC_DT); };
  tn[0] = ts[0];
  if (dopri) {
    launch_set_ctrl(S.p.ctrl, ts[0], 0.0, 1, S.st);
    TRY(S.eval_fwd(Solver::make_comb(tp.Y[0], nullptr, nullptr, 0, SC_ABS), nullptr, S.et_stage(0.0), tp.K[0], false));
    int j = 1;
    for (int n = 0; n < n_steps; ++n) {
      dtn[n] = step_dt[n];
      if (!(dtn[n] > 0.0)) return fail(NODE_ERR_ARG, "step size %d is not positive", n);
      launch_set_ctrl(S.p.ctrl, tn[n], dtn[n], 0, S.st);
      for (int s = 0; s < 6; ++s)
        TRY(S.eval_fwd(comb_of(tp.Y[n], &tp.K[6 * n], DP_BETA[s], s + 1), s == 5 ? tp.Y[n + 1] : nullptr, S.et_stage(DP_ALPHA[s]),
                       tp.K[6 * n + s + 1], false));
      tn[n + 1] = tn[n] + dtn[n];
      while (j < n_t && !(ts[j] > tn[n + 1])) {
        out_step[j] = n;
        const float t0f = (float)tn[n], t1f = (float)tn[n + 1], tjf = (float)ts[j];
        out_x[j] = (tjf - t0f) / (t1f - t0f);
        ++j;
      }
    }
    if (j < n_t) return fail(NODE_ERR_ARG, "the %d recorded steps end at t = %g, before the last output time", n_steps, tn[n_steps]);
  } else {
    launch_set_ctrl(S.p.ctrl, ts[0], 0.0, 1, S.st);
    for (int n = 0; n < n_steps; ++n) {
      const float t0f = (float)ts[n], t1f = (float)ts[n + 1];
      tn[n] = (double)t0f; dtn[n] = (double)(t1f - t0f); tn[n + 1] = (double)t1f;
      launch_set_ctrl(S.p.ctrl, tn[n], dtn[n], 0, S.st);
      float* const* K = &tp.K[4 * n];
      TRY(S.eval_fwd(Solver::make_comb(tp.Y[n], nullptr, nullptr, 0, SC_ABS), nullptr, S.et_stage(0.0), K[0], false));
      for (int s = 0; s < 3; ++s)
        TRY(S.eval_fwd(comb_of(tp.Y[n], K, rk4_rows[s], s + 1), nullptr, S.et_stage(rk4_alpha[s]), K[s + 1], false));
      launch_lincomb(comb_of(tp.Y[n], K, rk4_b, 4), S.p.ctrl, tp.Y[n + 1], numel, S.st);
      out_step[n + 1] = n;
      out_x[n + 1] = 1.f;
    }
  }

  // ---- reverse ----
  launch_fill(S.p.TH, 0.f, d.P, S.st);
  launch_fill(tp.YB, 0.f, numel, S.st);
  launch_fill(tp.Y0B, 0.f, numel, S.st);
  for (int i = 0; i < 7; ++i) launch_fill(tp.KB[i], 0.f, numel, S.st);
  // VJP of the dynamics at (stage time, stage state) with cotangent `cot`: vjp_y -> p.KA[0], vjp_theta added to p.TH
  auto vjp = [&](const Comb& state, double alpha, const float* cot) -> int {
    Comb ca = Solver::make_comb(cot, nullptr, nullptr, 0, SC_ABS);
    TRY(S.eval_aug(state, ca, nullptr, nullptr, S.et_stage(alpha), S.p.KY[0], S.p.KA[0], S.p.KT[0], -1, +1.f, nullptr, true));
    launch_axpy(S.p.TH, S.p.KT[0], 1.f, d.P, S.st);
    return NODE_OK;
  };
  const int nst = dopri ? 6 : 4;     // stage derivatives per step that feed y1
  bool k6_live = false;              // dopri5: does KB[6] hold anything?
  for (int n = n_steps - 1; n >= 0; --n) {
    const float dtf = (float)dtn[n];
    float* const* K = dopri ? &tp.K[6 * n] : &tp.K[4 * n];
    launch_set_ctrl(S.p.ctrl, tn[n], dtn[n], 0, S.st);
    // dense-output contributions of the outputs this step produced (transposed quartic)
    for (int j = n_t - 1; j >= 1; --j) {
      if (out_step[j] != n) continue;
      S.to_state(grad_out + (size_t)j * numel, tp.G);
      ScatterArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.src = tp.G; sa.n = numel;
      const double x = (double)out_x[j];
      if (!dopri || x == 1.0) {        // the output IS the end state
        sa.dst[0] = tp.YB; sa.coef[0] = 1.f; sa.nt = 1;
      } else {
        const double x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        const double wy0 = 1 - 11 * x2 + 18 * x3 - 8 * x4, wy1 = -5 * x2 + 14 * x3 - 8 * x4;
        const double wf0 = x - 4 * x2 + 5 * x3 - 2 * x4, wf1 = x2 - 3 * x3 + 2 * x4, wm = 16 * x2 - 32 * x3 + 16 * x4;
        int q = 0;
        sa.dst[q] = tp.Y0B; sa.coef[q++] = (float)(wy0 + wm);
        sa.dst[q] = tp.YB; sa.coef[q++] = (float)wy1;
        for (int i = 0; i < 7; ++i) {
          double c = wm * DP_CMID[i];
          if (i == 0) c += wf0;
          if (i == 6) c += wf1;
          if (c == 0.0) continue;
          sa.dst[q] = tp.KB[i]; sa.coef[q++] = (float)(dtn[n] * c);
        }
        sa.nt = q;
        k6_live = true;
      }
      launch_scatter_axpy(sa, S.st);
    }
    if (dopri && k6_live) {   // k6 = f(t + dt, y1): the evaluation the next step reused as its k0 (FSAL)
      TRY(vjp(Solver::make_comb(tp.Y[n + 1], nullptr, nullptr, 0, SC_ABS), 1.0, tp.KB[6]));
      launch_axpy(tp.YB, S.p.KA[0], 1.f, numel, S.st);
    }
    {   // y1 = y0 + dt sum b_i k_i
      ScatterArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.src = tp.YB; sa.n = numel;
      int q = 0;
      sa.dst[q] = tp.Y0B; sa.coef[q++] = 1.f;
      for (int i = 0; i < nst; ++i) {
        const double bi = dopri ? DP_BETA[5][i] : rk4_b[i];
        if (bi == 0.0) continue;
        sa.dst[q] = tp.KB[i]; sa.coef[q++] = dtf * (float)bi;
      }
      sa.nt = q;
      launch_scatter_axpy(sa, S.st);
    }
    for (int i = nst - 1; i >= 1; --i) {   // stage i: state y0 + dt sum_{j<i} beta_ij k_j
      const double* row = dopri ? DP_BETA[i - 1] : rk4_rows[i - 1];
      const double alpha = dopri ? DP_ALPHA[i - 1] : rk4_alpha[i - 1];
      TRY(vjp(comb_of(tp.Y[n], K, row, i), alpha, tp.KB[i]));
      ScatterArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.src = S.p.KA[0]; sa.n = numel;
      int q = 0;
      sa.dst[q] = tp.Y0B; sa.coef[q++] = 1.f;
      for (int jj = 0; jj < i; ++jj) {
        if (row[jj] == 0.0) continue;
        sa.dst[q] = tp.KB[jj]; sa.coef[q++] = dtf * (float)row[jj];
      }
      sa.nt = q;
      launch_scatter_axpy(sa, S.st);
    }
    if (!dopri || n == 0) {   // k0 = f(t, y0) evaluated by this step itself
      TRY(vjp(Solver::make_comb(tp.Y[n], nullptr, nullptr, 0, SC_ABS), 0.0, tp.KB[0]));
      launch_axpy(tp.Y0B, S.p.KA[0], 1.f, numel, S.st);
    } else {                  // FSAL: it is the previous step's k6
      std::swap(tp.KB[0], tp.KB[6]);
      k6_live = true;
    }
    std::swap(tp.YB, tp.Y0B);
    launch_fill(tp.Y0B, 0.f, numel, S.st);
    for (int i = 0; i < (dopri ? 6 : 4); ++i) launch_fill(tp.KB[i], 0.f, numel, S.st);
  }
  // out[0] = y0
  S.to_state(grad_out, tp.G);
  launch_axpy(tp.YB, tp.G, 1.f, numel, S.st);
  S.from_state(tp.YB, grad_y0);
  launch_theta_to_torch(d, S.p.TH, grad_params, S.st);
  HIP_TRY(hipStreamSynchronize(S.st));
  return S.check_launch("node_solve_backprop");
}

}  // extern "C"

// ----------------------------------------------------------------------------
// Generic ("flat") solver: the library's device-resident step controller around dynamics that the CALLER evaluates
// (SURVEY.md 8b "Fallback": model.py:367 accepts any nn.Module; train.py:202 offers norm='batch').  The state is up to
// three flat fp32 tensors (+ one scalar kept in the controller: the adjoint's time cotangent); the caller owns every
// buffer, asks for a stage state + stage time, evaluates its function on them (PyTorch ops on the same stream), stores the
// result in the stage's derivative buffer and calls node_flat_finish_step: error norm, accept / reject, next step
// size, dense output and FSAL commit are the SAME kernels the fused solves run (k_error_norm, k_step_controller,
// k_emit_flat, k_commit) -- no decision is taken on the host, which reads the controller back whenever it wants.
// ----------------------------------------------------------------------------
namespace {
struct FlatPlan { Ctrl* ctrl; double* targets; float* partial[3]; size_t bytes; };
FlatPlan flat_plan(void* base, int n_targets) {
  FlatPlan f;
  Bump b(base);
  f.ctrl = b.take<Ctrl>(1);
  f.targets = b.take<double>((size_t)(n_targets > 0 ? n_targets : 1));
  for (int i = 0; i < 3; ++i) f.partial[i] = b.take<float>(ERR_BLOCKS * 2);
  f.bytes = (b.off + 255) & ~(size_t)255;
  return f;
}
int flat_check(const node_flat_solve* f, FlatPlan* plan) {
  if (!f) return fail(NODE_ERR_NULL, "node_flat_solve is NULL");
  if (f->nseg < 1 || f->nseg > 3) return fail(NODE_ERR_ARG, "nseg must be 1..3 (got %d)", f->nseg);
  if (!f->ws || (((uintptr_t)f->ws) & 255)) return fail(NODE_ERR_ARG, "workspace must be non-NULL and 256-byte aligned");
  if (f->n_targets < 1 || f->n_targets > STEP_LIST_CAP) return fail(NODE_ERR_ARG, "n_targets must be 1..%d", STEP_LIST_CAP);
  for (int i = 0; i < f->nseg; ++i) {
    const node_flat_seg& sg = f->seg[i];
    if (!sg.y || !sg.y1 || sg.n == 0) return fail(NODE_ERR_NULL, "segment %d: y / y1 is NULL or empty", i);
    for (int j = 0; j < 7; ++j)
      if (!sg.k[j]) return fail(NODE_ERR_NULL, "segment %d: stage derivative buffer %d is NULL", i, j);
    if ((((uintptr_t)sg.y) | ((uintptr_t)sg.y1)) & 15) return fail(NODE_ERR_ARG, "segment %d: buffers must be 16-byte aligned", i);
  }
  *plan = flat_plan(f->ws, f->n_targets);
  if (f->ws_bytes < plan->bytes) return fail(NODE_ERR_WORKSPACE, "workspace too small: %zu < %zu", f->ws_bytes, plan->bytes);
  return NODE_OK;
}
int flat_launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return NODE_OK;
}
}  // namespace

extern "C" size_t node_flat_workspace_bytes(int n_targets) { return flat_plan(nullptr, n_targets).bytes; }

extern "C" int node_flat_begin(const node_flat_solve* f, double t0, const double* targets, double first_dt, int new_solve, void* stream) {
  FlatPlan p;
  TRY(flat_check(f, &p));
  if (!targets) return fail(NODE_ERR_NULL, "targets is NULL");
  hipStream_t st = (hipStream_t)stream;
  HostStage* hs = nullptr;
  TRY(get_stage((size_t)f->n_targets + 2 * STEP_LIST_CAP, &hs));
  // (the staging is reused by the next call of this thread: wait for the copy -- a solve begins once per interval)
  for (int i = 0; i < f->n_targets; ++i) {
    if (i > 0 && !(targets[i] > targets[i - 1])) return fail(NODE_ERR_ARG, "targets must increase (solver orientation)");
    hs->lists[i] = targets[i];
  }
  HIP_TRY(hipMemcpyAsync(p.targets, hs->lists, (size_t)f->n_targets * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (new_solve) launch_set_ctrl(p.ctrl, t0, first_dt, 1, st);      // also zeroes the scalar segment
  else launch_set_interval(p.ctrl, t0, first_dt, st);               // keeps the scalar segment and the cumulative counters
  return flat_launch_ok("node_flat_begin");
}

extern "C" int node_flat_stage(const node_flat_solve* f, int method, int stage, float* const* y_stage, float* t_stage, void* stream) {
  FlatPlan p;
  TRY(flat_check(f, &p));
  hipStream_t st = (hipStream_t)stream;
  EvalTime et;
  et.ctrl = p.ctrl; et.tsign = f->tsign; et.mode = TM_STAGE; et.alpha = 0.f;
  const double* row = nullptr;
  int ncoef = 0, scale = SC_DT;
  static const double rk4_rows[3][3] = {{1.0 / 3, 0, 0}, {-1.0 / 3, 1.0, 0}, {1.0, -1.0, 1.0}};
  static const double rk4_alpha[3] = {1.0 / 3, 2.0 / 3, 1.0};
  static const double one[1] = {1.0};
  if (stage == NODE_FLAT_F0) {
    // f(t, y) at the current point: nothing to combine
  } else if (stage == NODE_FLAT_PROBE) {
    row = one; ncoef = 1; scale = SC_H0; et.mode = TM_PROBE;
  } else if (method == NODE_METHOD_DOPRI5 && stage >= 0 && stage < 6) {
    row = DP_BETA[stage]; ncoef = stage + 1; et.alpha = (float)DP_ALPHA[stage];
  } else if (method == NODE_METHOD_RK4 && stage >= 1 && stage <= 3) {
    row = rk4_rows[stage - 1]; ncoef = stage; et.alpha = (float)rk4_alpha[stage - 1];
  } else {
    return fail(NODE_ERR_ARG, "bad (method, stage) = (%d, %d)", method, stage);
  }
  if (row != nullptr) {
    if (!y_stage) return fail(NODE_ERR_NULL, "y_stage is NULL");
    for (int i = 0; i < f->nseg; ++i) {
      if (!y_stage[i]) return fail(NODE_ERR_NULL, "y_stage[%d] is NULL", i);
      float* ks[7];
      for (int j = 0; j < 7; ++j) ks[j] = f->seg[i].k[j];
      launch_lincomb(Solver::make_comb(f->seg[i].y, ks, row, ncoef, scale), p.ctrl, y_stage[i], f->seg[i].n, st);
    }
  }
  if (t_stage) launch_flat_time(et, t_stage, st);
  return flat_launch_ok("node_flat_stage");
}

extern "C" int node_flat_scalar(const node_flat_solve* f, int which, const float* src, float scale, int accumulate, void* stream) {
  FlatPlan p;
  TRY(flat_check(f, &p));
  if (!src) return fail(NODE_ERR_NULL, "src is NULL");
  if (which < -1 || which > 6) return fail(NODE_ERR_ARG, "which must be -1 (value) or 0..6 (stage derivative)");
  launch_flat_scalar(p.ctrl, which, src, scale, accumulate, (hipStream_t)stream);
  return flat_launch_ok("node_flat_scalar");
}

extern "C" int node_flat_initial_step(const node_flat_solve* f, int phase, void* stream) {
  FlatPlan p;
  TRY(flat_check(f, &p));
  if (phase != 0 && phase != 1) return fail(NODE_ERR_ARG, "phase must be 0 or 1");
  hipStream_t st = (hipStream_t)stream;
  InitSeg segs[3];
  for (int i = 0; i < f->nseg; ++i) segs[i] = {f->seg[i].y, f->seg[i].k[0], f->seg[i].k[1], f->seg[i].n};
  launch_init_norms(segs, p.partial, f->nseg, f->rtol, f->atol, phase, st);
  InitCtlArgs ic;
  memset(&ic, 0, sizeof(ic));
  ic.ctrl = p.ctrl;
  for (int i = 0; i < f->nseg; ++i) { ic.partial[i] = p.partial[i]; ic.numel[i] = (double)f->seg[i].n; }
  ic.nseg = f->nseg; ic.has_scalar = f->has_scalar ? 1 : 0; ic.phase = phase; ic.rtol = f->rtol; ic.atol = f->atol;
  launch_init_controller(ic, st);
  return flat_launch_ok("node_flat_initial_step");
}

extern "C" int node_flat_finish_step(const node_flat_solve* f, int method, float* y_out, void* stream) {
  FlatPlan p;
  TRY(flat_check(f, &p));
  hipStream_t st = (hipStream_t)stream;
  const int nseg = f->nseg, aug = f->has_scalar ? 1 : 0;
  if (method == NODE_METHOD_RK4) {     // y <- y + dt (k0 + 3 k1 + 3 k2 + k3) / 8 on the fixed grid; t advances by dt
    const double cf[4] = {1.0 / 8, 3.0 / 8, 3.0 / 8, 1.0 / 8};
    for (int i = 0; i < nseg; ++i) {
      float* ks[7];
      for (int j = 0; j < 7; ++j) ks[j] = f->seg[i].k[j];
      launch_lincomb(Solver::make_comb(f->seg[i].y, ks, cf, 4, SC_DT), p.ctrl, f->seg[i].y1, f->seg[i].n, st);
      HIP_TRY(hipMemcpyAsync(f->seg[i].y, f->seg[i].y1, f->seg[i].n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (aug) launch_set_scalar_state(p.ctrl, 0.f, 1, st);
    return flat_launch_ok("node_flat_finish_step(rk4)");
  }
  if (method != NODE_METHOD_DOPRI5) return fail(NODE_ERR_ARG, "unknown method %d", method);
  ErrSeg es[3];
  for (int i = 0; i < nseg; ++i) {
    es[i].y0 = f->seg[i].y; es[i].y1 = f->seg[i].y1; es[i].n = f->seg[i].n; es[i].compute_y1 = 0;   // (y1 = the sixth stage's state)
    for (int j = 0; j < 7; ++j) es[i].k[j] = f->seg[i].k[j];
  }
  launch_error_norm(es, p.partial, nseg, p.ctrl, f->rtol, f->atol, st);
  StepCtlArgs sc;
  memset(&sc, 0, sizeof(sc));
  sc.ctrl = p.ctrl;
  for (int i = 0; i < nseg; ++i) { sc.partial[i] = p.partial[i]; sc.numel[i] = (double)f->seg[i].n; }
  sc.nseg = nseg; sc.has_scalar = aug; sc.rtol = f->rtol; sc.atol = f->atol;
  sc.targets = p.targets; sc.n_targets = f->n_targets; sc.interp_scalar = aug;
  launch_step_controller(sc, st);
  if (y_out != nullptr) {
    EmitArgs ea;
    ea.ctrl = p.ctrl; ea.targets = p.targets; ea.y0 = f->seg[0].y; ea.y1 = f->seg[0].y1;
    for (int j = 0; j < 7; ++j) ea.k[j] = f->seg[0].k[j];
    ea.y_out = y_out;
    launch_emit_flat(ea, f->seg[0].n, st);
  }
  CommitArgs cm;
  memset(&cm, 0, sizeof(cm));
  cm.ctrl = p.ctrl; cm.targets = p.targets; cm.nseg = nseg; cm.interp_final = aug;
  for (int i = 0; i < nseg; ++i) {
    cm.y[i] = f->seg[i].y; cm.y1[i] = f->seg[i].y1; cm.k0[i] = f->seg[i].k[0]; cm.k6[i] = f->seg[i].k[6]; cm.n[i] = f->seg[i].n;
    for (int j = 0; j < 7; ++j) cm.k[i][j] = f->seg[i].k[j];
  }
  launch_commit(cm, st);
  return flat_launch_ok("node_flat_finish_step");
}

extern "C" int node_flat_status_read(const node_flat_solve* f, node_flat_status* out, void* stream) {
  FlatPlan p;
  TRY(flat_check(f, &p));
  if (!out) return fail(NODE_ERR_NULL, "out is NULL");
  hipStream_t st = (hipStream_t)stream;
  HostStage* hs = nullptr;
  TRY(get_stage(16, &hs));
  HIP_TRY(hipMemcpyAsync(hs->ctrl, p.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const Ctrl& c = *hs->ctrl;
  out->done = c.done; out->status = c.status; out->steps = c.step_idx; out->accepted = c.n_acc; out->rejected = c.n_rej;
  out->t = c.t; out->dt = c.dt; out->first_dt = c.first_dt; out->scalar = c.ts_cur;
  return NODE_OK;
}

extern "C" {
int node_gn_relu_fwd(const node_shape* shape, const float* z, const float* gamma, const float* beta, int relu, float* out,
                     float* stats, void* stream) {
  char why[200];
  const int rc = head_check(shape, why, sizeof(why));
  if (rc != NODE_OK && rc != NODE_ERR_UNSUPPORTED) return fail(rc, "%s", why);   // (the per-group kernels take any C)
  if (!z || !gamma || !beta || !out || !stats) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  launch_gn_relu_fwd(*shape, z, gamma, beta, relu, out, stats, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of node_gn_relu_fwd failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_gn_relu_bwd(const node_shape* shape, const float* z, const float* gamma, const float* beta, const float* stats,
                     int relu, const float* g_out, float* dz, float* gpart, float* gsum, void* stream) {
  char why[200];
  const int rc = head_check(shape, why, sizeof(why));
  if (rc != NODE_OK && rc != NODE_ERR_UNSUPPORTED) return fail(rc, "%s", why);
  if (!z || !gamma || !beta || !stats || !g_out || !dz || !gpart) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  launch_gn_relu_bwd(*shape, z, gamma, beta, stats, relu, g_out, dz, gpart, (hipStream_t)stream);
  if (gsum != nullptr) launch_head_gsum(gpart, gsum, shape->n, shape->c, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of node_gn_relu_bwd failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_head_fwd(const node_shape* shape, const float* z, const float* gamma, const float* beta, const float* scale,
                  float* pooled, float* stats, void* stream) {
  char why[200];
  const int rc = head_check(shape, why, sizeof(why));
  if (rc != NODE_OK) return fail(rc, "%s", why);
  if (!z || !gamma || !beta || !pooled || !stats) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  launch_head_fwd(*shape, z, gamma, beta, scale, pooled, stats, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of node_head_fwd failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_head_bwd(const node_shape* shape, const float* z, const float* gamma, const float* beta, const float* scale,
                  const float* stats, const float* g_pooled, float* dz, float* gpart, float* gsum, void* stream) {
  char why[200];
  const int rc = head_check(shape, why, sizeof(why));
  if (rc != NODE_OK) return fail(rc, "%s", why);
  if (!z || !gamma || !beta || !stats || !g_pooled || !dz || !gpart) return fail(NODE_ERR_NULL, "a required pointer is NULL");
  launch_head_bwd(*shape, z, gamma, beta, scale, stats, g_pooled, dz, gpart, (hipStream_t)stream);
  if (gsum != nullptr) launch_head_gsum(gpart, gsum, shape->n, shape->c, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of node_head_bwd failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_sgd_step(const node_sgd_tensor* tensors, int count, float lr, float momentum, float weight_decay, float grad_scale,
                  const float* skip_if_nonzero, void* stream) {
  if (count < 0) return fail(NODE_ERR_ARG, "count < 0");
  if (count == 0) return NODE_OK;
  if (!tensors) return fail(NODE_ERR_NULL, "tensors is NULL");
  if (!(lr >= 0.f) || !(momentum >= 0.f) || !(weight_decay >= 0.f)) return fail(NODE_ERR_ARG, "lr / momentum / weight_decay must be >= 0");
  for (int i = 0; i < count; ++i) {
    if (!tensors[i].param || !tensors[i].grad) return fail(NODE_ERR_NULL, "tensor %d: a pointer is NULL", i);
    if (!tensors[i].momentum_buf && momentum != 0.f) return fail(NODE_ERR_NULL, "tensor %d: momentum buffer is NULL with momentum %g", i, momentum);
    if ((((uintptr_t)tensors[i].param) | ((uintptr_t)tensors[i].grad) | ((uintptr_t)tensors[i].momentum_buf)) & 3)
      return fail(NODE_ERR_ARG, "tensor %d: pointers must be 4-byte aligned", i);
  }
  for (int base = 0; base < count; base += SGD_TABLE) {
    SgdTable tb;
    memset(&tb, 0, sizeof(tb));
    const int m = count - base < SGD_TABLE ? count - base : SGD_TABLE;
    size_t max_n = 0;
    for (int i = 0; i < m; ++i) {
      tb.e[i].p = tensors[base + i].param; tb.e[i].g = tensors[base + i].grad; tb.e[i].m = tensors[base + i].momentum_buf;
      tb.e[i].n = tensors[base + i].n;
      if (tb.e[i].n > max_n) max_n = tb.e[i].n;
    }
    launch_sgd_multi(tb, m, max_n, lr, momentum, weight_decay, grad_scale, skip_if_nonzero, (hipStream_t)stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NODE_ERR_HIP, "launch of node_sgd_step failed: %s", hipGetErrorString(e));
  return NODE_OK;
}

int node_profile_begin(void) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (auto& r : g_prof.recs) { g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b); }
  g_prof.recs.clear();
  g_prof.on = true;
  return NODE_OK;
}

int node_profile_end(node_profile* out) {
  if (!out) return fail(NODE_ERR_NULL, "out is NULL");
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = false;
  memset(out, 0, sizeof(*out));
  for (auto& r : g_prof.recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return fail(NODE_ERR_HIP, "hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return fail(NODE_ERR_HIP, "hipEventElapsedTime failed");
    out->launches[r.cls] += 1;
    out->total_ms[r.cls] += ms;
    out->flops[r.cls] += r.flops;
    g_prof.pool.push_back(r.a);
    g_prof.pool.push_back(r.b);
  }
  g_prof.recs.clear();
  return NODE_OK;
}

}  // extern "C"
