// C ABI of the residual stem (include/node_hip.h: node_stem_fwd / node_stem_bwd / node_stem_conv): host-side plan of
// the launches in kernels_stem.hip.  Reference: model.py:167-178 (ResDownsample), model.py:284-310 (ResBlock).
//
// forward (11 launches):  prep | conv0 | GN+ReLU | conv 3x3/2 | conv 1x1/2 | GN+ReLU | conv 3x3 (+ shortcut) | GN+ReLU |
//                         conv 3x3/2 | conv 1x1/2 | GN+ReLU | conv 3x3 (+ shortcut) | NHWC -> NCHW
// backward: NCHW -> NHWC (+ triples) | per convolution: data gradient (+ the shortcut's on top) and weight gradient (the
//           shortcut's rides in the 3x3's launch) | per GroupNorm: backward pass | ONE reduction launch for every slab
#include "stem.h"
#include "wino4.h"
#include "../../include/node_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

using namespace node;

namespace {

int failf(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return set_error(code, buf);
}

struct Bump {
  char* base;
  size_t off;
  explicit Bump(void* b) : base((char*)b), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

struct Trip {            // a triples tensor [3][rows + 1][C]
  bf16_t* p;
  size_t plane;
  int rows, C;
};
Trip take_trip(Bump& b, int rows, int C) {
  Trip t;
  t.rows = rows; t.C = C;
  t.plane = (size_t)(rows + 1) * C;
  t.p = b.take<bf16_t>(3 * t.plane);
  return t;
}
struct Filt {            // a filter as triples in both operand layouts
  bf16_t* wf; bf16_t* wd;
  size_t plane;
  int Cout, Cin, taps;
};
Filt take_filt(Bump& b, int Cout, int Cin, int taps) {
  Filt f;
  f.Cout = Cout; f.Cin = Cin; f.taps = taps;
  f.plane = (size_t)taps * Cout * Cin;
  f.wf = b.take<bf16_t>(3 * f.plane);
  f.wd = b.take<bf16_t>(3 * f.plane);
  return f;
}
struct Wg {              // split-K plan + slabs of one weight gradient
  int nsplit, rps;
  float* slab; float* slab2;
};
Wg take_wg(Bump& b, int rows, int Cout, int Cin, int taps, bool extra) {
  // workgroups = (64 x 64 tiles) x (kernel rows) x splits: aim at ~1.5 workgroups per CU, shares of >= 64 pixels,
  // and keep the slabs (written once, read once by k_stem_reduce) under ~16 MB
  Wg w;
  const int pairs = (Cout / 64) * (Cin / 64) * (taps == 9 ? 3 : 1);
  int ns = (384 + pairs - 1) / pairs;
  if (ns > rows / 64) ns = rows / 64;
  const size_t per = (size_t)(taps + (extra ? 1 : 0)) * Cout * Cin * sizeof(float);
  while (ns > 4 && ns * per > ((size_t)16 << 20)) ns = (ns + 1) / 2;
  if (ns < 1) ns = 1;
  int rps = (rows + ns - 1) / ns;
  rps = (rps + 15) & ~15;
  w.rps = rps;
  w.nsplit = (rows + rps - 1) / rps;
  w.slab = b.take<float>((size_t)w.nsplit * taps * Cout * Cin);
  w.slab2 = extra ? b.take<float>((size_t)w.nsplit * Cout * Cin) : nullptr;
  return w;
}

inline int down2(int h) { return (h - 1) / 2 + 1; }     // 3x3 stride 2 pad 1 and 1x1 stride 2 pad 0 alike

struct StemPlan {
  int N, Cin0, H, W, F;
  int H0, W0, H1, W1, H2, W2, R0, R1, R2;
  float eps;
  float* w0t;
  Filt c1, c2, d1, c3, c4, d2;
  float *h0, *h1, *s1, *x1, *h3, *s2, *outn;
  Trip a0, a1, a2, a3;
  float* stats[4];
  // backward
  float *da3, *da2, *da1, *da0, *dh0;
  Trip g3, dh3t, dx1t, dh1t;
  float* gpart[4];
  Wg w4, w3, w2, w1, w0;
  // the last convolution (3x3, filters -> filters on an 8x8 image: the ODE conv's own shape) through the F(4x4,3x3)
  // pipeline of the ODE block (wino4.h): component GEMMs k_w4_gemm64b, weight gradient k_w4_wgrad
  bool f4;
  bool f4_b16;
  float *xh4, *rs4, *Va, *Vg, *M4, *Z4, *dU4, *U4[2], *dh3n;
  unsigned short* Ub4[2];
  size_t bytes;
};

bool stem_takes_w4(const node_stem_shape* sh) {
  const char* e = getenv("NODE_TUNE_STEM_W4");       // 0: the stem's own gather-GEMM kernels for the last convolution too (A/B, tests)
  if (e && atoi(e) == 0) return false;
  const int h2 = ((sh->h - 2 - 1) / 2 + 1 - 1) / 2 + 1, w2 = ((sh->w - 2 - 1) / 2 + 1 - 1) / 2 + 1;
  return h2 == 8 && w2 == 8 && sh->filters % 128 == 0 && sh->n % 8 == 0 && 16 % (sh->filters / 32) == 0;
}

StemPlan make_stem_plan(const node_stem_shape* sh, void* base) {
  StemPlan p;
  memset(&p, 0, sizeof(p));
  p.N = sh->n; p.Cin0 = sh->in_ch; p.H = sh->h; p.W = sh->w; p.F = sh->filters; p.eps = sh->eps;
  p.H0 = p.H - 2; p.W0 = p.W - 2;
  p.H1 = down2(p.H0); p.W1 = down2(p.W0);
  p.H2 = down2(p.H1); p.W2 = down2(p.W1);
  p.R0 = p.N * p.H0 * p.W0; p.R1 = p.N * p.H1 * p.W1; p.R2 = p.N * p.H2 * p.W2;
  const int F = p.F, G64 = 32, GF = F < 32 ? F : 32;
  Bump b(base);
  p.w0t = b.take<float>(28 * 64);
  p.c1 = take_filt(b, 64, 64, 9);
  p.c2 = take_filt(b, 64, 64, 9);
  p.d1 = take_filt(b, 64, 64, 1);
  p.c3 = take_filt(b, F, 64, 9);
  p.c4 = take_filt(b, F, F, 9);
  p.d2 = take_filt(b, F, 64, 1);
  p.h0 = b.take<float>((size_t)p.R0 * 64);
  p.a0 = take_trip(b, p.R0, 64);
  p.h1 = b.take<float>((size_t)p.R1 * 64);
  p.s1 = b.take<float>((size_t)p.R1 * 64);
  p.a1 = take_trip(b, p.R1, 64);
  p.x1 = b.take<float>((size_t)p.R1 * 64);
  p.a2 = take_trip(b, p.R1, 64);
  p.h3 = b.take<float>((size_t)p.R2 * F);
  p.s2 = b.take<float>((size_t)p.R2 * F);
  p.a3 = take_trip(b, p.R2, F);
  p.outn = b.take<float>((size_t)p.R2 * F);
  p.stats[0] = b.take<float>((size_t)p.N * G64 * 2);
  p.stats[1] = b.take<float>((size_t)p.N * G64 * 2);
  p.stats[2] = b.take<float>((size_t)p.N * G64 * 2);
  p.stats[3] = b.take<float>((size_t)p.N * GF * 2);
  // backward
  p.g3 = take_trip(b, p.R2, F);
  p.da3 = b.take<float>((size_t)p.R2 * F);
  p.dh3t = take_trip(b, p.R2, F);
  p.da2 = b.take<float>((size_t)p.R1 * 64);
  p.dx1t = take_trip(b, p.R1, 64);
  p.da1 = b.take<float>((size_t)p.R1 * 64);
  p.dh1t = take_trip(b, p.R1, 64);
  p.da0 = b.take<float>((size_t)p.R0 * 64);
  p.dh0 = b.take<float>((size_t)p.R0 * 64);
  p.gpart[0] = b.take<float>((size_t)p.N * 2 * 64);
  p.gpart[1] = b.take<float>((size_t)p.N * 2 * 64);
  p.gpart[2] = b.take<float>((size_t)p.N * 2 * 64);
  p.gpart[3] = b.take<float>((size_t)p.N * 2 * F);
  p.w4 = take_wg(b, p.R2, F, F, 9, false);
  p.w3 = take_wg(b, p.R2, F, 64, 9, true);
  p.w2 = take_wg(b, p.R1, 64, 64, 9, false);
  p.w1 = take_wg(b, p.R1, 64, 64, 9, true);
  {   // conv0: K steps of two pixels over R0 rows, one 64 x 32 slab per workgroup
    int ns = p.R0 / 256;
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    int rps = ((p.R0 + ns - 1) / ns + 7) & ~7;
    p.w0.rps = rps;
    p.w0.nsplit = (p.R0 + rps - 1) / rps;
    p.w0.slab = b.take<float>((size_t)p.w0.nsplit * 64 * 32);
    p.w0.slab2 = nullptr;
  }
  p.f4 = stem_takes_w4(sh);
  if (p.f4) {
    const int Nv = p.N;
    p.f4_b16 = w4_uses_bf16(Nv, F);
    p.xh4 = b.take<float>((size_t)p.R2 * F);
    p.rs4 = b.take<float>((size_t)p.N * GF);
    p.Va = b.take<float>(w4_v_elems(Nv, F));
    p.Vg = b.take<float>(w4_v_elems(Nv, F));
    p.M4 = b.take<float>(w4_v_elems(Nv, F));
    p.Z4 = b.take<float>(w4_z_elems(Nv, F));
    p.dU4 = b.take<float>((size_t)W4_COMPS * F * F);
    for (int i = 0; i < 2; ++i) {
      p.U4[i] = b.take<float>(w4_u_elems(F));
      p.Ub4[i] = b.take<unsigned short>(w4_ub_elems(F));
    }
    p.dh3n = b.take<float>((size_t)p.R2 * F);
  }
  p.bytes = b.off + 256;
  return p;
}

int check_stem_shape(const node_stem_shape* sh) {
  if (!sh) return failf(NODE_ERR_NULL, "shape is NULL");
  if (sh->n <= 0 || sh->in_ch <= 0 || sh->h < 5 || sh->w < 5 || sh->filters <= 0) return failf(NODE_ERR_SHAPE, "bad stem shape");
  if (sh->in_ch > 3) return failf(NODE_ERR_UNSUPPORTED, "the stem's first layer takes in_ch <= 3 (got %d)", sh->in_ch);
  if (sh->filters % 64 != 0) return failf(NODE_ERR_UNSUPPORTED, "the stem's kernels take filters %% 64 == 0 (got %d)", sh->filters);
  const size_t biggest = (size_t)sh->n * (sh->h - 2) * (sh->w - 2) * 64;
  if (biggest >= ((size_t)1 << 31)) return failf(NODE_ERR_UNSUPPORTED, "stem tensors must stay under 2^31 elements");
  if (stem_gn_cb((sh->h - 2) * (sh->w - 2), 64, 2) == 0)
    return failf(NODE_ERR_UNSUPPORTED, "the stem's GroupNorm passes hold (sample, 8 channels) blocks in LDS: images up to %d pixels "
                 "behind the first layer (got %d x %d)", 150 * 1024 / 64, sh->h - 2, sh->w - 2);
  {   // the last block's GroupNorm passes run on `filters` channels in min(32, filters) groups (model.py:268-271)
    const int cpg = sh->filters / (sh->filters < 32 ? sh->filters : 32);
    const int h1 = (sh->h - 2 - 1) / 2 + 1, w1 = (sh->w - 2 - 1) / 2 + 1, h2 = (h1 - 1) / 2 + 1, w2 = (w1 - 1) / 2 + 1;
    if ((sh->filters & (sh->filters - 1)) != 0 || stem_gn_cb(h2 * w2, sh->filters, cpg) == 0)
      return failf(NODE_ERR_UNSUPPORTED, "the stem's GroupNorm passes take power-of-two filter counts (whole groups per "
                   "power-of-two channel block); got %d", sh->filters);
  }
  return NODE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// geometry -> SConvArgs
// ---------------------------------------------------------------------------------------------------------------
void set_single_class(SConvArgs& a, int taps) {
  a.nclass = 1; a.step = 1;
  a.cls_py[0] = a.cls_px[0] = 0;
  a.cls_h[0] = a.OH; a.cls_w[0] = a.OW;
  a.cls_ntap[0] = taps;
  a.cls_taps[0] = 0;
  for (int t = 0; t < taps; ++t) a.cls_taps[0] |= (unsigned long long)t << (4 * t);
  a.cls_tile0[0] = 0;
  a.cls_tile0[1] = (a.N * a.OH * a.OW + 127) / 128;
}
// forward convolution: in = activation triples [N, IH, IW, Cin], out [N, OH, OW, Cout]
SConvArgs conv_fwd_args(const Trip& in, const Filt& f, float* out, int N, int IH, int IW, int OH, int OW, int k, int stride, int pad) {
  SConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in.p; a.in_plane = in.plane; a.zero_row = in.rows;
  a.w = f.wf; a.w_plane = f.plane;
  a.out = out;
  a.N = N; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.Cin = f.Cin; a.Cout = f.Cout;
  a.KH = a.KW = k; a.sshift = stride == 2 ? 1 : 0; a.pad = pad; a.mode = 0;
  set_single_class(a, k * k);
  return a;
}
// data gradient: in = dy triples [N, YH, YW, Cout_f], out = dx [N, XH, XW, Cin_f]
SConvArgs conv_dgrad_args(const Trip& dy, const Filt& f, float* dx, int N, int YH, int YW, int XH, int XW, int k, int stride, int pad) {
  SConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = dy.p; a.in_plane = dy.plane; a.zero_row = dy.rows;
  a.w = f.wd; a.w_plane = f.plane;
  a.out = dx;
  a.N = N; a.IH = YH; a.IW = YW; a.OH = XH; a.OW = XW; a.Cin = f.Cout; a.Cout = f.Cin;
  a.KH = a.KW = k; a.sshift = stride == 2 ? 1 : 0; a.pad = pad; a.mode = 1;
  if (stride == 1) {
    set_single_class(a, k * k);
    return a;
  }
  // stride 2: one class per (row parity, column parity) of the pixel written; its taps are those with (o + pad - k) even
  a.step = 2;
  int nc = 0, tiles = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      int nt = 0, taps[9];
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
          if (((py + pad - ky) & 1) == 0 && ((px + pad - kx) & 1) == 0) taps[nt++] = ky * k + kx;
      const int ch = (XH - py + 1) / 2, cw = (XW - px + 1) / 2;
      if (nt == 0 || ch <= 0 || cw <= 0) continue;
      a.cls_py[nc] = py; a.cls_px[nc] = px; a.cls_h[nc] = ch; a.cls_w[nc] = cw;
      a.cls_ntap[nc] = nt;
      a.cls_taps[nc] = 0;
      for (int t = 0; t < nt; ++t) a.cls_taps[nc] |= (unsigned long long)taps[t] << (4 * t);
      a.cls_tile0[nc] = tiles;
      tiles += (N * ch * cw + 127) / 128;
      ++nc;
    }
  a.nclass = nc;
  a.cls_tile0[nc] = tiles;
  return a;
}
SWgradArgs wgrad_args(const Trip& dy, const Trip* dy2, const Trip& in, const Wg& w, int N, int IH, int IW, int OH, int OW, int Cin,
                      int Cout, int k, int stride, int pad) {
  SWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy3 = dy.p; a.dy_plane = dy.plane; a.dy_zero_row = dy.rows;
  a.dy23 = dy2 ? dy2->p : nullptr;
  a.in = in.p; a.in_plane = in.plane; a.zero_row = in.rows;
  a.slab = w.slab; a.slab2 = w.slab2;
  a.N = N; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.Cin = Cin; a.Cout = Cout; a.KH = a.KW = k; a.stride = stride; a.pad = pad;
  a.nsplit = w.nsplit; a.rows_per_split = w.rps;
  return a;
}
SGnArgs gn_args(const float* h, const float* gamma, const float* beta, float* stats, int N, int HW, int C, float eps) {
  SGnArgs a;
  memset(&a, 0, sizeof(a));
  a.h = h; a.gamma = gamma; a.beta = beta; a.stats = stats;
  a.N = N; a.HW = HW; a.C = C; a.cpg = C / (C < 32 ? C : 32); a.eps = eps;
  a.CB = stem_gn_cb(HW, C, a.cpg);
  return a;
}

int launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return failf(NODE_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return NODE_OK;
}

}  // namespace

extern "C" {

size_t node_stem_workspace_bytes(const node_stem_shape* shape) {
  if (check_stem_shape(shape) != NODE_OK) return 0;
  return make_stem_plan(shape, nullptr).bytes;
}

int node_stem_fwd(const node_stem_shape* shape, const node_stem_params* prm, const float* x, float* out, void* ws, size_t ws_bytes,
                  void* stream) {
  w4_refresh_tuning();     // (NODE_TUNE_W4_*: once per call)
  int rc = check_stem_shape(shape);
  if (rc != NODE_OK) return rc;
  if (!prm || !x || !out || !ws) return failf(NODE_ERR_NULL, "a required pointer is NULL");
  const float* const* pp = reinterpret_cast<const float* const*>(prm);
  for (int i = 0; i < 16; ++i)
    if (!pp[i]) return failf(NODE_ERR_NULL, "stem parameter %d is NULL", i);
  if (((uintptr_t)ws) & 255) return failf(NODE_ERR_ARG, "workspace must be 256-byte aligned");
  StemPlan p = make_stem_plan(shape, ws);
  if (ws_bytes < p.bytes) return failf(NODE_ERR_WORKSPACE, "stem workspace too small: %zu < %zu", ws_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int N = p.N, F = p.F;

  SPrepArgs pa;
  memset(&pa, 0, sizeof(pa));
  const Filt* fl[6] = {&p.c1, &p.c2, &p.d1, &p.c3, &p.c4, &p.d2};
  const float* fw[6] = {prm->b1_c1_w, prm->b1_c2_w, prm->b1_ds_w, prm->b2_c1_w, prm->b2_c2_w, prm->b2_ds_w};
  for (int i = 0; i < 6; ++i) pa.job[i] = {fw[i], fl[i]->wf, fl[i]->wd, fl[i]->Cout, fl[i]->Cin, fl[i]->taps};
  pa.njobs = 6;
  pa.w0 = prm->conv0_w; pa.w0t = p.w0t; pa.k0 = 9 * p.Cin0;
  const Trip* tz[8] = {&p.a0, &p.a1, &p.a2, &p.a3, &p.g3, &p.dh3t, &p.dx1t, &p.dh1t};
  for (int i = 0; i < 8; ++i) {
    pa.zero[i] = tz[i]->p + (size_t)tz[i]->rows * tz[i]->C;
    pa.zero_plane[i] = tz[i]->plane;
    pa.zero_c[i] = tz[i]->C;
  }
  pa.nzero = 8;
  if (F > 4096) return failf(NODE_ERR_UNSUPPORTED, "filters > 4096");
  launch_stem_prep(pa, st);
  if ((rc = launch_ok("stem_prep")) != NODE_OK) return rc;

  launch_stem_conv0_fwd(x, p.w0t, prm->conv0_b, p.h0, N, p.Cin0, p.H, p.W, st);
  if ((rc = launch_ok("stem_conv0_fwd")) != NODE_OK) return rc;
  {   // block 1: relu(norm1(h0)) -> a0
    SGnArgs g = gn_args(p.h0, prm->b1_n1_w, prm->b1_n1_b, p.stats[0], N, p.H0 * p.W0, 64, p.eps);
    g.a3 = p.a0.p; g.a_plane = p.a0.plane;
    launch_stem_gn_fwd(g, st);
    if ((rc = launch_ok("stem_gn_fwd")) != NODE_OK) return rc;
  }
  {   // conv1 (3x3 / 2) and the shortcut (1x1 / 2) in one launch
    SConvArgs c = conv_fwd_args(p.a0, p.c1, p.h1, N, p.H0, p.W0, p.H1, p.W1, 3, 2, 1);
    c.w2 = p.d1.wf; c.w2_plane = p.d1.plane; c.out2 = p.s1;
    launch_stem_conv(c, st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  }
  {
    SGnArgs g = gn_args(p.h1, prm->b1_n2_w, prm->b1_n2_b, p.stats[1], N, p.H1 * p.W1, 64, p.eps);
    g.a3 = p.a1.p; g.a_plane = p.a1.plane;
    launch_stem_gn_fwd(g, st);
    if ((rc = launch_ok("stem_gn_fwd")) != NODE_OK) return rc;
  }
  {
    SConvArgs c = conv_fwd_args(p.a1, p.c2, p.x1, N, p.H1, p.W1, p.H1, p.W1, 3, 1, 1);
    c.res = p.s1;
    launch_stem_conv(c, st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  }
  {   // block 2
    SGnArgs g = gn_args(p.x1, prm->b2_n1_w, prm->b2_n1_b, p.stats[2], N, p.H1 * p.W1, 64, p.eps);
    g.a3 = p.a2.p; g.a_plane = p.a2.plane;
    launch_stem_gn_fwd(g, st);
    if ((rc = launch_ok("stem_gn_fwd")) != NODE_OK) return rc;
  }
  {
    SConvArgs c = conv_fwd_args(p.a2, p.c3, p.h3, N, p.H1, p.W1, p.H2, p.W2, 3, 2, 1);
    c.w2 = p.d2.wf; c.w2_plane = p.d2.plane; c.out2 = p.s2;
    launch_stem_conv(c, st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  }
  if (p.f4) {
    W4PackJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    for (int i = 0; i < 2; ++i) {
      jobs.w[i] = prm->b2_c2_w; jobs.u[i] = p.U4[i]; jobs.ub[i] = p.f4_b16 ? p.Ub4[i] : nullptr; jobs.dgrad[i] = i; jobs.plain[i] = 1;
    }
    launch_w4_pack(jobs, 2, F, st);
    if ((rc = launch_ok("w4_pack")) != NODE_OK) return rc;
    launch_w4s_stem_in(p.h3, prm->b2_n2_w, prm->b2_n2_b, p.eps, F / 32, p.xh4, p.rs4, p.Va, N, F, N, st);
    if ((rc = launch_ok("w4s_stem_in")) != NODE_OK) return rc;
    launch_w4_gemm(p.Va, p.U4[0], p.M4, nullptr, N, F, st, p.f4_b16 ? p.Ub4[0] : nullptr);
    if ((rc = launch_ok("w4_gemm")) != NODE_OK) return rc;
    launch_w4s_stem_out(p.M4, p.s2, out, N, F, st);
    return launch_ok("node_stem_fwd");
  }
  {
    SGnArgs g = gn_args(p.h3, prm->b2_n2_w, prm->b2_n2_b, p.stats[3], N, p.H2 * p.W2, F, p.eps);
    g.a3 = p.a3.p; g.a_plane = p.a3.plane;
    launch_stem_gn_fwd(g, st);
    if ((rc = launch_ok("stem_gn_fwd")) != NODE_OK) return rc;
  }
  {
    SConvArgs c = conv_fwd_args(p.a3, p.c4, p.outn, N, p.H2, p.W2, p.H2, p.W2, 3, 1, 1);
    c.res = p.s2;
    launch_stem_conv(c, st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  }
  launch_stem_to_nchw(p.outn, out, N, F, p.H2 * p.W2, st);
  if ((rc = launch_ok("stem_to_nchw")) != NODE_OK) return rc;
  return launch_ok("node_stem_fwd");
}

int node_stem_bwd(const node_stem_shape* shape, const node_stem_params* prm, const float* x, const float* grad_out,
                  const node_stem_grads* gr, void* ws, size_t ws_bytes, void* stream) {
  w4_refresh_tuning();     // (NODE_TUNE_W4_*: once per call)
  int rc = check_stem_shape(shape);
  if (rc != NODE_OK) return rc;
  if (!prm || !x || !grad_out || !gr || !ws) return failf(NODE_ERR_NULL, "a required pointer is NULL");
  const float* const* pp = reinterpret_cast<const float* const*>(prm);
  float* const* gp = reinterpret_cast<float* const*>(gr);
  for (int i = 0; i < 16; ++i)
    if (!pp[i] || !gp[i]) return failf(NODE_ERR_NULL, "stem parameter / gradient %d is NULL", i);
  if (((uintptr_t)ws) & 255) return failf(NODE_ERR_ARG, "workspace must be 256-byte aligned");
  StemPlan p = make_stem_plan(shape, ws);
  if (ws_bytes < p.bytes) return failf(NODE_ERR_WORKSPACE, "stem workspace too small: %zu < %zu", ws_bytes, p.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int N = p.N, F = p.F;
  const int HW0 = p.H0 * p.W0, HW1 = p.H1 * p.W1, HW2 = p.H2 * p.W2;

  // dL/d out: also the gradient of the second block's shortcut s2
  launch_stem_from_nchw(grad_out, nullptr, p.g3.p, p.g3.plane, N, F, HW2, st);
  if ((rc = launch_ok("stem_from_nchw")) != NODE_OK) return rc;
  if (p.f4) {
    // block 2, conv2 through the F(4x4,3x3) pipeline: V and Z of dL/d out, data-gradient GEMM, then ONE pass for the ReLU mask and
    // GroupNorm's backward (k_w4s_pass<2, 0>, the ODE block's own), the weight gradient in the transform domain
    launch_w4s_stem_gin(grad_out, p.Vg, p.Z4, N, F, N, st);
    if ((rc = launch_ok("w4s_stem_gin")) != NODE_OK) return rc;
    launch_w4_gemm(p.Vg, p.U4[1], p.M4, nullptr, N, F, st, p.f4_b16 ? p.Ub4[1] : nullptr);
    if ((rc = launch_ok("w4_gemm")) != NODE_OK) return rc;
    W4sArgs a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.Q = 1; a.Nv = N; a.C = F; a.cpg = F / 32; a.eps = p.eps;
    a.h.M = p.M4; a.h.gamma = prm->b2_n2_w; a.h.beta = prm->b2_n2_b; a.h.xhat_s = p.xh4; a.h.rstd = p.rs4; a.h.osign = 1.f;
    a.h.gpart = p.gpart[3]; a.h.out_nhwc = p.dh3n;
    launch_w4s_pass(2, 0, a, st);
    if ((rc = launch_ok("w4s_pass")) != NODE_OK) return rc;
    launch_stem_split(p.dh3n, p.dh3t.p, p.dh3t.plane, (size_t)p.R2 * F, st);
    if ((rc = launch_ok("stem_split")) != NODE_OK) return rc;
    W4WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.V1 = p.Va; wa.Z1 = p.Z4; wa.dU = p.dU4; wa.N = N; wa.C = F;
    launch_w4_wgrad(wa, st);
    if ((rc = launch_ok("w4_wgrad")) != NODE_OK) return rc;
    launch_w4_du_to_dw(p.dU4, gr->b2_c2_w, F, st);
    if ((rc = launch_ok("w4_du_to_dw")) != NODE_OK) return rc;
  } else {
  // block 2, conv2 (3x3, F -> F) : weight gradient, data gradient -> da3
    launch_stem_wgrad(wgrad_args(p.g3, nullptr, p.a3, p.w4, N, p.H2, p.W2, p.H2, p.W2, F, F, 3, 1, 1), st);
    if ((rc = launch_ok("stem_wgrad")) != NODE_OK) return rc;
    launch_stem_conv(conv_dgrad_args(p.g3, p.c4, p.da3, N, p.H2, p.W2, p.H2, p.W2, 3, 1, 1), st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
    {
      SGnArgs g = gn_args(p.h3, prm->b2_n2_w, prm->b2_n2_b, p.stats[3], N, HW2, F, p.eps);
      g.da = p.da3; g.dh = nullptr; g.dh3 = p.dh3t.p; g.dh_plane = p.dh3t.plane; g.gpart = p.gpart[3];
      launch_stem_gn_bwd(g, st);
      if ((rc = launch_ok("stem_gn_bwd")) != NODE_OK) return rc;
    }
  }
  // block 2, conv1 (3x3 / 2, 64 -> F) + shortcut (1x1 / 2, 64 -> F): both read a2
  launch_stem_wgrad(wgrad_args(p.dh3t, &p.g3, p.a2, p.w3, N, p.H1, p.W1, p.H2, p.W2, 64, F, 3, 2, 1), st);
  if ((rc = launch_ok("stem_wgrad")) != NODE_OK) return rc;
  {   // data gradient of conv1 (3x3 / 2) + the shortcut's (1x1 / 2: one more K segment of the (even, even) pixels)
    SConvArgs c = conv_dgrad_args(p.dh3t, p.c3, p.da2, N, p.H2, p.W2, p.H1, p.W1, 3, 2, 1);
    c.in2 = p.g3.p; c.w2 = p.d2.wd; c.w2_plane = p.d2.plane;
    launch_stem_conv(c, st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  }
  {
    SGnArgs g = gn_args(p.x1, prm->b2_n1_w, prm->b2_n1_b, p.stats[2], N, HW1, 64, p.eps);
    g.da = p.da2; g.dh = nullptr; g.dh3 = p.dx1t.p; g.dh_plane = p.dx1t.plane; g.gpart = p.gpart[2];
    launch_stem_gn_bwd(g, st);
    if ((rc = launch_ok("stem_gn_bwd")) != NODE_OK) return rc;
  }
  // block 1, conv2 (3x3, 64 -> 64): dx1 is the gradient of its output AND of the shortcut s1
  launch_stem_wgrad(wgrad_args(p.dx1t, nullptr, p.a1, p.w2, N, p.H1, p.W1, p.H1, p.W1, 64, 64, 3, 1, 1), st);
  if ((rc = launch_ok("stem_wgrad")) != NODE_OK) return rc;
  launch_stem_conv(conv_dgrad_args(p.dx1t, p.c2, p.da1, N, p.H1, p.W1, p.H1, p.W1, 3, 1, 1), st);
  if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  {
    SGnArgs g = gn_args(p.h1, prm->b1_n2_w, prm->b1_n2_b, p.stats[1], N, HW1, 64, p.eps);
    g.da = p.da1; g.dh = nullptr; g.dh3 = p.dh1t.p; g.dh_plane = p.dh1t.plane; g.gpart = p.gpart[1];
    launch_stem_gn_bwd(g, st);
    if ((rc = launch_ok("stem_gn_bwd")) != NODE_OK) return rc;
  }
  // block 1, conv1 (3x3 / 2) + shortcut (1x1 / 2): both read a0
  launch_stem_wgrad(wgrad_args(p.dh1t, &p.dx1t, p.a0, p.w1, N, p.H0, p.W0, p.H1, p.W1, 64, 64, 3, 2, 1), st);
  if ((rc = launch_ok("stem_wgrad")) != NODE_OK) return rc;
  {
    SConvArgs c = conv_dgrad_args(p.dh1t, p.c1, p.da0, N, p.H1, p.W1, p.H0, p.W0, 3, 2, 1);
    c.in2 = p.dx1t.p; c.w2 = p.d1.wd; c.w2_plane = p.d1.plane;
    launch_stem_conv(c, st);
    if ((rc = launch_ok("stem_conv")) != NODE_OK) return rc;
  }
  {
    SGnArgs g = gn_args(p.h0, prm->b1_n1_w, prm->b1_n1_b, p.stats[0], N, HW0, 64, p.eps);
    g.da = p.da0; g.dh = p.dh0; g.dh3 = nullptr; g.gpart = p.gpart[0];
    launch_stem_gn_bwd(g, st);
    if ((rc = launch_ok("stem_gn_bwd")) != NODE_OK) return rc;
  }
  launch_stem_conv0_wgrad(x, p.dh0, p.w0.slab, N, p.Cin0, p.H, p.W, p.w0.nsplit, p.w0.rps, st);
  if ((rc = launch_ok("stem_conv0_wgrad")) != NODE_OK) return rc;

  SReduceArgs ra;
  memset(&ra, 0, sizeof(ra));
  int nj = 0;
  if (!p.f4) ra.job[nj++] = {p.w4.slab, gr->b2_c2_w, nullptr, 0, p.w4.nsplit, F, F, 9};
  ra.job[nj++] = {p.w3.slab, gr->b2_c1_w, nullptr, 0, p.w3.nsplit, F, 64, 9};
  ra.job[nj++] = {p.w3.slab2, gr->b2_ds_w, nullptr, 0, p.w3.nsplit, F, 64, 1};
  ra.job[nj++] = {p.w2.slab, gr->b1_c2_w, nullptr, 0, p.w2.nsplit, 64, 64, 9};
  ra.job[nj++] = {p.w1.slab, gr->b1_c1_w, nullptr, 0, p.w1.nsplit, 64, 64, 9};
  ra.job[nj++] = {p.w1.slab2, gr->b1_ds_w, nullptr, 0, p.w1.nsplit, 64, 64, 1};
  ra.job[nj++] = {p.w0.slab, gr->conv0_w, gr->conv0_b, 1, p.w0.nsplit, 64, 9 * p.Cin0, 1};
  ra.job[nj++] = {p.gpart[0], gr->b1_n1_w, gr->b1_n1_b, 2, N, 64, 0, 0};
  ra.job[nj++] = {p.gpart[1], gr->b1_n2_w, gr->b1_n2_b, 2, N, 64, 0, 0};
  ra.job[nj++] = {p.gpart[2], gr->b2_n1_w, gr->b2_n1_b, 2, N, 64, 0, 0};
  ra.job[nj++] = {p.gpart[3], gr->b2_n2_w, gr->b2_n2_b, 2, N, F, 0, 0};
  ra.njobs = nj;
  launch_stem_reduce(ra, st);
  if ((rc = launch_ok("stem_reduce")) != NODE_OK) return rc;
  return launch_ok("node_stem_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// diagnostics: one convolution of the family on NCHW tensors
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct OnePlan {
  int YH, YW;
  Filt f;
  Trip xin, dyt;
  float *xn, *dyn, *resn;
  Wg wg;
  size_t bytes;
};
OnePlan make_one(const node_conv_geom* g, void* base) {
  OnePlan p;
  memset(&p, 0, sizeof(p));
  p.YH = (g->x_h + 2 * g->pad - g->k) / g->stride + 1;
  p.YW = (g->x_w + 2 * g->pad - g->k) / g->stride + 1;
  Bump b(base);
  p.f = take_filt(b, g->cout, g->cin, g->k * g->k);
  p.xin = take_trip(b, g->n * g->x_h * g->x_w, g->cin);
  p.dyt = take_trip(b, g->n * p.YH * p.YW, g->cout);
  p.xn = b.take<float>((size_t)g->n * g->x_h * g->x_w * g->cin);
  p.dyn = b.take<float>((size_t)g->n * p.YH * p.YW * g->cout);
  p.resn = b.take<float>((size_t)g->n * (g->x_h * g->x_w * g->cin > p.YH * p.YW * g->cout ? g->x_h * g->x_w * g->cin : p.YH * p.YW * g->cout));
  p.wg = take_wg(b, g->n * p.YH * p.YW, g->cout, g->cin, g->k * g->k, false);
  p.bytes = b.off + 256;
  return p;
}
int check_geom(const node_conv_geom* g) {
  if (!g) return failf(NODE_ERR_NULL, "geometry is NULL");
  if (g->n <= 0 || g->x_h <= 0 || g->x_w <= 0) return failf(NODE_ERR_SHAPE, "bad geometry");
  if (g->cin % 64 != 0 || g->cout % 64 != 0 || g->cin <= 0 || g->cout <= 0) return failf(NODE_ERR_UNSUPPORTED, "cin, cout must be multiples of 64");
  if (!((g->k == 3 && g->pad == 1) || (g->k == 1 && g->pad == 0)) || (g->stride != 1 && g->stride != 2))
    return failf(NODE_ERR_UNSUPPORTED, "3x3 pad 1 or 1x1 pad 0, stride 1 or 2");
  return NODE_OK;
}
}  // namespace

size_t node_stem_conv_workspace_bytes(const node_conv_geom* g) {
  if (check_geom(g) != NODE_OK) return 0;
  return make_one(g, nullptr).bytes;
}

int node_stem_conv(const node_conv_geom* g, int what, const float* x, const float* w, const float* dy, float* result, void* ws,
                   size_t ws_bytes, void* stream) {
  w4_refresh_tuning();     // (NODE_TUNE_W4_*: once per call)
  int rc = check_geom(g);
  if (rc != NODE_OK) return rc;
  if (!result || !ws || (what != 1 && !x) || (what != 2 && !w) || (what != 0 && !dy)) return failf(NODE_ERR_NULL, "a required pointer is NULL");
  if (((uintptr_t)ws) & 255) return failf(NODE_ERR_ARG, "workspace must be 256-byte aligned");
  OnePlan p = make_one(g, ws);
  if (ws_bytes < p.bytes) return failf(NODE_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int XHW = g->x_h * g->x_w, YHW = p.YH * p.YW;
  SPrepArgs pa;
  memset(&pa, 0, sizeof(pa));
  if (what != 2) { pa.job[0] = {w, p.f.wf, p.f.wd, g->cout, g->cin, g->k * g->k}; pa.njobs = 1; }
  pa.zero[0] = p.xin.p + (size_t)p.xin.rows * p.xin.C; pa.zero_plane[0] = p.xin.plane; pa.zero_c[0] = p.xin.C;
  pa.zero[1] = p.dyt.p + (size_t)p.dyt.rows * p.dyt.C; pa.zero_plane[1] = p.dyt.plane; pa.zero_c[1] = p.dyt.C;
  pa.nzero = 2;
  launch_stem_prep(pa, st);
  if (what == 0) {
    launch_stem_from_nchw(x, nullptr, p.xin.p, p.xin.plane, g->n, g->cin, XHW, st);
    launch_stem_conv(conv_fwd_args(p.xin, p.f, p.resn, g->n, g->x_h, g->x_w, p.YH, p.YW, g->k, g->stride, g->pad), st);
    launch_stem_to_nchw(p.resn, result, g->n, g->cout, YHW, st);
  } else if (what == 1) {
    launch_stem_from_nchw(dy, nullptr, p.dyt.p, p.dyt.plane, g->n, g->cout, YHW, st);
    SConvArgs c = conv_dgrad_args(p.dyt, p.f, p.resn, g->n, p.YH, p.YW, g->x_h, g->x_w, g->k, g->stride, g->pad);
    if (g->k == 1 && g->stride == 2) {   // pixels no tap reaches keep a zero gradient
      if (hipMemsetAsync(p.resn, 0, (size_t)g->n * XHW * g->cin * sizeof(float), st) != hipSuccess) return failf(NODE_ERR_HIP, "memset failed");
    }
    launch_stem_conv(c, st);
    launch_stem_to_nchw(p.resn, result, g->n, g->cin, XHW, st);
  } else {
    launch_stem_from_nchw(x, nullptr, p.xin.p, p.xin.plane, g->n, g->cin, XHW, st);
    launch_stem_from_nchw(dy, nullptr, p.dyt.p, p.dyt.plane, g->n, g->cout, YHW, st);
    launch_stem_wgrad(wgrad_args(p.dyt, nullptr, p.xin, p.wg, g->n, g->x_h, g->x_w, p.YH, p.YW, g->cin, g->cout, g->k, g->stride, g->pad), st);
    SReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.job[0] = {p.wg.slab, result, nullptr, 0, p.wg.nsplit, g->cout, g->cin, g->k * g->k};
    ra.njobs = 1;
    launch_stem_reduce(ra, st);
  }
  return launch_ok("node_stem_conv");
}

}  // extern "C"
