// kbench -- kernel-iteration bench for the two GEMM-class kernels of libnode_hip
// (k_conv3x3 implicit GEMM and k_wgrad).  Builds against the library's own
// translation units (tools/build_kbench.sh); with -DNODE_STAMPS the conv kernel
// also records s_memtime / s_memrealtime stamps, from which this tool prints the
// share of prologue / main loop / epilogue and the in-kernel clock.
//
//   kbench conv  N C H W [reps] [tiles...]     forward conv + GN epilogue; tiles = M tile 0 (heuristic) / 64 / 128
//   kbench dgrad N C H W [reps] [tiles...]     data-gradient conv + ReLU/GN-backward epilogue
//   kbench wgrad N C H W [reps] [variants...]  0 generic kernel, 1 geometry-templated kernel
//
// Every variant's output is compared with the first one's; conv and wgrad also against an fp64 host reference.
#include "../neural-ode-features_amd/csrc/node_internal.h"
#include "../include/node_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace node;
namespace node { int dims_for(const node_shape* sh, Dims* out); }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* dev_rand(size_t n, std::mt19937& g, float scale = 1.f, float shift = 0.f, size_t zero_tail = 0) {
  std::vector<float> h(n + zero_tail, 0.f);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (size_t i = 0; i < n; ++i) h[i] = shift + scale * nd(g);
  n += zero_tail;
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
static float* dev_zero(size_t n) {
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0, n * sizeof(float)));
  return d;
}
static std::vector<float> to_host(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
  return h;
}
static double max_abs_diff(const std::vector<float>& a, const std::vector<float>& b, double* ref_max) {
  double m = 0, r = 0;
  for (size_t i = 0; i < a.size(); ++i) { m = std::max(m, (double)std::fabs(a[i] - b[i])); r = std::max(r, (double)std::fabs(b[i])); }
  if (ref_max) *ref_max = r;
  return m;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: kbench conv|dgrad|wgrad N C H W [reps] [variants...]\n"); return 2; }
  std::string what = argv[1];
  node_shape sh;
  sh.n = atoi(argv[2]); sh.c = atoi(argv[3]); sh.h = atoi(argv[4]); sh.w = atoi(argv[5]);
  sh.groups = std::min(32, sh.c); sh.eps = 1e-5f;
  int reps = argc > 6 ? atoi(argv[6]) : 20;
  std::vector<int> variants;
  for (int i = 7; i < argc; ++i) variants.push_back(atoi(argv[i]));
  if (variants.empty()) variants.push_back(0);
  Dims d;
  if (dims_for(&sh, &d) != 0) { fprintf(stderr, "bad shape: %s\n", node_last_error()); return 1; }
  printf("# %s N=%d C=%d H=%d W=%d  BM=%d S=%d mtiles=%d ntile=%d nchunk=%d nsplit=%d\n", what.c_str(), d.N, d.C, d.H, d.W,
         d.BM, d.S, d.mtiles, d.ntile, d.nchunk, d.nsplit);
  std::mt19937 gen(1234);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t numel = d.numel, C = d.C;
  float* in = dev_rand(numel, gen, 1.f, 0.f, C);   // conv inputs carry a tail of C zeros (2-D Winograd halo)
  float* wraw = dev_rand(C * (C + 1) * 9, gen, 0.02f);
  float* bias = dev_rand(C, gen, 0.1f);
  float* gamma = dev_rand(C, gen, 0.25f, 1.f);
  float* beta = dev_rand(C, gen, 0.1f);
  float* act = dev_rand(numel, gen, 1.f, 0.f, C);
  float* xhat = dev_rand(numel, gen);
  float* rstd = dev_rand((size_t)d.N * d.G, gen, 0.1f, 1.f);
  const size_t wsz = (size_t)d.ntile * d.nchunk * 9 * KCH * BN;
  float* wpk = dev_zero(wsz);
  float* tmap = dev_zero((size_t)d.HW * C);
  Ctrl* ctrl;
  CK(hipMalloc(&ctrl, sizeof(Ctrl)));
  launch_set_ctrl(ctrl, 0.3, 0.1, 1, st);
  const bool bwd = what == "dgrad";
  launch_pack_weights(d, wraw, wpk, bwd ? 1 : 0, st);
  launch_tmap(d, wraw, tmap, st);
  CK(hipStreamSynchronize(st));

  const double flops = 2.0 * 9.0 * d.C * d.C * (double)d.N * d.HW;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t nstamp = (size_t)4096 * 8 * 16;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, nstamp * sizeof(unsigned long long)));

  if (what == "conv" || what == "dgrad") {
    // variants = conv M tile (g_conv_bm): 0 heuristic, 64, 128
    std::vector<float*> outs, xh, rs, gp;
    std::vector<Dims> dv;
    std::vector<std::vector<double>> times(variants.size());
    for (size_t v = 0; v < variants.size(); ++v) {
      g_conv_bm = (variants[v] & 0xfff) % 1000;
      g_conv_wino = (variants[v] & 0xfff) >= 2000 ? 2 : (variants[v] & 0xfff) >= 1000 ? 1 : 0;
      Dims dd;
      if (dims_for(&sh, &dd) != 0) { fprintf(stderr, "bad shape: %s\n", node_last_error()); return 1; }
      dv.push_back(dd);
      outs.push_back(dev_zero(numel)); xh.push_back(dev_zero(numel)); rs.push_back(dev_zero((size_t)d.N * d.G));
      gp.push_back(dev_zero((size_t)dd.mtiles * 2 * C + 64));
    }
    g_conv_bm = -1;
    g_conv_wino = -1;
    std::vector<float*> wpks;
    for (size_t v = 0; v < variants.size(); ++v) {
      float* w = dev_zero(conv_packed_elems(dv[v]));
      if (dv[v].wino == 2) launch_pack_weights_w2(dv[v], wraw, w, bwd ? 1 : 0, st);
      else if (dv[v].wino) launch_pack_weights_w(dv[v], wraw, w, bwd ? 1 : 0, st);
      else launch_pack_weights(dv[v], wraw, w, bwd ? 1 : 0, st);
      wpks.push_back(w);
    }
    auto run = [&](size_t v, unsigned long long* stp) {
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.in = in; a.wpacked = wpks[v]; a.mode = bwd ? CM_BWD_RELU_GN : CM_FWD_GN_RELU;
      a.bias = bias; a.tmap = tmap;
      a.et.ctrl = ctrl; a.et.alpha = 0.5f; a.et.tsign = 1.f; a.et.mode = TM_STAGE;
      a.gamma = gamma; a.beta = beta; a.osign = 1.f; a.out = outs[v];
      a.xhat_out = bwd ? nullptr : xh[v]; a.rstd_out = bwd ? nullptr : rs[v];
      a.act = act; a.xhat = xhat; a.rstd = rstd; a.gpart = gp[v];
      a.stamps = stp;
      a.ablate = variants[v] >> 12;
      if (dv[v].csplit) a.raw_out = outs[v];   // split mode: raw conv output (GroupNorm is a separate pass): timing only
      launch_conv(dv[v], a, st);
    };
    for (size_t v = 0; v < variants.size(); ++v) for (int i = 0; i < 3; ++i) run(v, nullptr);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    for (int r = 0; r < reps; ++r)
      for (size_t v = 0; v < variants.size(); ++v) {
        CK(hipEventRecord(e0, st));
        run(v, nullptr);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times[v].push_back(ms * 1e3);
      }
    // fp64 host reference of sample 0 (forward: conv + bias + t*tmap -> GroupNorm -> affine -> ReLU)
    double host_err = -1, host_max = 0;
    if (!bwd) {
      auto hin = to_host(in, (size_t)d.HW * C), hw = to_host(wraw, C * (C + 1) * 9), hb = to_host(bias, C);
      auto hg = to_host(gamma, C), hbt = to_host(beta, C), ho = to_host(outs[0], (size_t)d.HW * C);
      const double tval = 0.3 + 0.5 * 0.1;
      std::vector<double> pre((size_t)d.HW * C);
      for (int p = 0; p < d.HW; ++p)
        for (int co = 0; co < d.C; ++co) {
          const int h = p / d.W, x = p % d.W;
          double acc = hb[co];
          for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
              const int hh = h + kh - 1, xx = x + kw - 1;
              if (hh < 0 || hh >= d.H || xx < 0 || xx >= d.W) continue;
              const float* wrow = &hw[(((size_t)co * (C + 1)) * 3 + kh) * 3 + kw];
              acc += tval * (double)wrow[0];
              const float* irow = &hin[((size_t)hh * d.W + xx) * C];
              for (int ci = 0; ci < d.C; ++ci) acc += (double)irow[ci] * (double)wrow[(size_t)(1 + ci) * 9];
            }
          pre[(size_t)p * C + co] = acc;
        }
      host_err = 0;
      for (int g = 0; g < d.G; ++g) {
        double m = 0, v = 0;
        for (int p = 0; p < d.HW; ++p) for (int cc = 0; cc < d.cpg; ++cc) m += pre[(size_t)p * C + g * d.cpg + cc];
        m /= d.HW * d.cpg;
        for (int p = 0; p < d.HW; ++p) for (int cc = 0; cc < d.cpg; ++cc) { const double e = pre[(size_t)p * C + g * d.cpg + cc] - m; v += e * e; }
        v /= d.HW * d.cpg;
        const double rstdv = 1.0 / std::sqrt(v + 1e-5);
        for (int p = 0; p < d.HW; ++p)
          for (int cc = 0; cc < d.cpg; ++cc) {
            const int c = g * d.cpg + cc;
            const double o = std::max(0.0, (pre[(size_t)p * C + c] - m) * rstdv * hg[c] + hbt[c]);
            host_err = std::max(host_err, std::fabs(o - (double)ho[(size_t)p * C + c]));
            host_max = std::max(host_max, std::fabs(o));
          }
      }
      printf("first variant vs fp64 host reference (sample 0): max err %.3e (ref max %.3e)\n", host_err, host_max);
    }
    auto ref = to_host(outs[0], numel);
    for (size_t v = 0; v < variants.size(); ++v) {
      std::sort(times[v].begin(), times[v].end());
      const double med = times[v][times[v].size() / 2], mn = times[v][0];
      double rmax;
      const double diff = max_abs_diff(to_host(outs[v], numel), ref, &rmax);
      // (dgamma, dbeta) partials are per M tile: compare their sums
      double gdiff = 0;
      if (bwd) {
        auto red = [&](size_t w) {
          auto h = to_host(gp[w], (size_t)dv[w].mtiles * 2 * C);
          std::vector<float> r(2 * C, 0.f);
          for (int m = 0; m < dv[w].mtiles; ++m) for (size_t i = 0; i < 2 * C; ++i) r[i] += h[(size_t)m * 2 * C + i];
          return r;
        };
        double gm;
        gdiff = max_abs_diff(red(v), red(0), &gm);
      }
      printf("tile %4d (BM=%d%s, %d workgroups)  median %8.2f us  min %8.2f us  %6.1f TF (median)  max|out-first| %.3e (ref max %.3e)  gpart diff %.3e\n",
             variants[v] & 0xfff, dv[v].BM, dv[v].wino == 2 ? " winograd-2d" : dv[v].wino ? " winograd" : "", dv[v].mtiles * dv[v].ntile, med, mn, flops / (med * 1e-6) / 1e12, diff, rmax, gdiff);
#ifdef NODE_STAMPS
      CK(hipMemset(stamps, 0, nstamp * sizeof(unsigned long long)));
      run(v, stamps);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> hs(nstamp);
      CK(hipMemcpy(hs.data(), stamps, nstamp * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double pro = 0, mainl = 0, epi = 0, clk = 0, e1 = 0, e2 = 0, e3 = 0, sub[5] = {0, 0, 0, 0, 0}, ph[4] = {0, 0, 0, 0};
      int cnt = 0;
      for (size_t w = 0; w < nstamp / 16; ++w) {
        const unsigned long long* s = &hs[w * 16];
        if (s[1] == 0 || s[4] == 0) continue;
        for (int q = 0; q < 4; ++q) ph[q] += (double)s[12 + q];
        pro += (double)(s[2] - s[1]); mainl += (double)(s[3] - s[2]); epi += (double)(s[4] - s[3]);
        if (s[6] && s[7]) { e1 += (double)(s[6] - s[3]); e2 += (double)(s[7] - s[6]); e3 += (double)(s[4] - s[7]); }
        if (s[8] && s[11]) { for (int q = 0; q < 4; ++q) sub[q] += (double)(s[8 + q] - (q ? s[7 + q] : s[3])); sub[4] += (double)(s[6] - s[11]); }
        if (s[5] > s[0]) clk += (double)(s[4] - s[1]) / (double)(s[5] - s[0]) * 100.0;  // MHz
        ++cnt;
      }
      if (cnt) printf("            stamps over %d waves: prologue %.0f  main %.0f  epilogue %.0f cycles (acc->LDS %.0f, stats %.0f, store %.0f); in-kernel clock %.0f MHz\n",
                      cnt, pro / cnt, mainl / cnt, epi / cnt, e1 / cnt, e2 / cnt, e3 / cnt, clk / cnt);
      {
        double q0 = 0, q1 = 0, q2 = 0, q3 = 0; int c2 = 0;
        for (size_t w = 0; w < nstamp / 16; ++w) {
          const unsigned long long* s = &hs[w * 16];
          if (s[1] == 0 || s[4] == 0 || s[8] == 0 || s[10] == 0 || s[8] < s[1] || s[8] > s[2]) continue;
          q0 += (double)(s[8] - s[1]); q1 += (double)(s[9] - s[8]); q2 += (double)(s[10] - s[9]); q3 += (double)(s[2] - s[10]); ++c2;
        }
        {
          double f1 = 0; int c3 = 0;
          for (size_t w = 0; w < nstamp / 16; ++w) {
            const unsigned long long* s = &hs[w * 16];
            if (s[2] && s[11] > s[2] && s[3] > s[11]) { f1 += (double)(s[11] - s[2]); ++c3; }
          }
          if (c3) printf("            first loop iteration (2 chunks) %.0f cycles\n", f1 / c3);
        }
        if (c2) printf("            prologue: setup %.0f  requests issued %.0f  first chunk landed + transformed %.0f  barrier %.0f\n", q0 / c2, q1 / c2, q2 / c2, q3 / c2);
      }
      if (cnt && ph[0] > 0) printf("            main-loop phases (cycles summed over chunks): first group %.0f  staging %.0f  barrier %.0f  second group %.0f\n",
                                  ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt);
      if (cnt && sub[0] > 0) printf("            output transform: tables+sync %.0f  pass A %.0f  pass B %.0f  pass C %.0f  bias/time %.0f\n",
                                   sub[0] / cnt, sub[1] / cnt, sub[2] / cnt, sub[3] / cnt, sub[4] / cnt);
#endif
    }
  } else if (what == "wgrad") {
    // variant 0: generic kernel (k_wgrad_p); 1: geometry-templated kernel; 2: 1-D Winograd-domain kernel (k_wgrad_w); 3: 2-D (k_wgrad_w2)
    std::vector<float*> wp;
    std::vector<Dims> dv;
    std::vector<std::vector<double>> times(variants.size());
    float* dz = dev_rand(numel, gen, 1.f, 0.f, C);
    for (size_t v = 0; v < variants.size(); ++v) {
      g_wgrad_wino = variants[v] == 3 ? 2 : variants[v] == 2 ? 1 : 0;
      Dims dd;
      if (dims_for(&sh, &dd) != 0) { fprintf(stderr, "bad shape: %s\n", node_last_error()); return 1; }
      dv.push_back(dd);
      wp.push_back(dev_zero((size_t)dd.nsplit * 9 * C * C));
    }
    g_wgrad_wino = -1;
    auto run = [&](size_t v, unsigned long long* stp = nullptr) {
      WgradArgs a;
      memset(&a, 0, sizeof(a));
      a.act = act; a.dz = dz; a.wpart = wp[v];
      g_wgrad_variant = variants[v] == 0 ? 0 : 1;
      a.stamps = stp;
      launch_wgrad(dv[v], a, st);
    };
    for (size_t v = 0; v < variants.size(); ++v) for (int i = 0; i < 3; ++i) run(v);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    for (int r = 0; r < reps; ++r)
      for (size_t v = 0; v < variants.size(); ++v) {
        CK(hipEventRecord(e0, st));
        run(v);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times[v].push_back(ms * 1e3);
      }
    // slab-reduced dW[tap][ci][co] of a variant (Winograd slabs are transformed back on the host)
    auto reduced = [&](size_t v) {
      const int ntap = 9;
      const size_t per = (size_t)ntap * C * C;
      auto h = to_host(wp[v], (size_t)dv[v].nsplit * per);
      std::vector<float> r(per, 0.f);
      for (int s = 0; s < dv[v].nsplit; ++s) for (size_t i = 0; i < per; ++i) r[i] += h[(size_t)s * per + i];
      return r;
    };
    // host reference of dW on a sample of entries (double precision)
    auto hact = to_host(act, numel), hdz = to_host(dz, numel);
    auto refw_v0 = reduced(0);
    double ref_err = 0, ref_max = 0;
    for (int k = 0; k < 64; ++k) {
      const int t = k % 9, ci = (k * 37) % d.C, co = (k * 101 + 3) % d.C;
      double acc = 0;
      for (int n = 0; n < d.N; ++n)
        for (int h = 0; h < d.H; ++h)
          for (int x = 0; x < d.W; ++x) {
            const int hh = h + t / 3 - 1, xx = x + t % 3 - 1;
            if (hh < 0 || hh >= d.H || xx < 0 || xx >= d.W) continue;
            acc += (double)hact[((size_t)n * d.HW + hh * d.W + xx) * C + ci] * (double)hdz[((size_t)n * d.HW + h * d.W + x) * C + co];
          }
      ref_err = std::max(ref_err, std::fabs(acc - (double)refw_v0[((size_t)t * C + ci) * C + co]));
      ref_max = std::max(ref_max, std::fabs(acc));
    }
    printf("first variant vs fp64 host reference on 64 entries: max err %.3e (ref max %.3e)\n", ref_err, ref_max);
    for (size_t v = 0; v < variants.size(); ++v) {
      std::sort(times[v].begin(), times[v].end());
      const double med = times[v][times[v].size() / 2], mn = times[v][0];
      double rmax;
      const double diff = max_abs_diff(reduced(v), refw_v0, &rmax);
      printf("variant %2d%s  median %8.2f us  min %8.2f us  %6.1f TF (median)  max|dW-first| %.3e (ref max %.3e)\n",
             variants[v], dv[v].wgrad_wino == 2 ? " winograd-2d" : dv[v].wgrad_wino ? " winograd" : "", med, mn, flops / (med * 1e-6) / 1e12, diff, rmax);
#ifdef NODE_STAMPS
      CK(hipMemset(stamps, 0, nstamp * sizeof(unsigned long long)));
      run(v, stamps);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> hs(nstamp);
      CK(hipMemcpy(hs.data(), stamps, nstamp * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double pro = 0, mainl = 0, epi = 0, clk = 0, ph[4] = {0, 0, 0, 0};
      int cnt = 0;
      for (size_t w = 0; w < nstamp / 16; ++w) {
        const unsigned long long* s = &hs[w * 16];
        if (s[1] == 0 || s[4] == 0) continue;
        pro += (double)(s[2] - s[1]); mainl += (double)(s[3] - s[2]); epi += (double)(s[4] - s[3]);
        for (int q = 0; q < 4; ++q) ph[q] += (double)s[12 + q];
        if (s[5] > s[0]) clk += (double)(s[4] - s[1]) / (double)(s[5] - s[0]) * 100.0;
        ++cnt;
      }
      if (cnt) printf("            stamps over %d waves: prologue %.0f  main %.0f  store %.0f cycles; in-kernel clock %.0f MHz\n",
                      cnt, pro / cnt, mainl / cnt, epi / cnt, clk / cnt);
      if (cnt && ph[1] > 0) printf("            main-loop phases (cycles summed over units): request %.0f  MFMA loop %.0f  transform+write %.0f  barrier %.0f\n",
                                  ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt);
#endif
    }
    // masked column sums
    float* sp = dev_zero((size_t)d.N * 9 * C);
    for (int i = 0; i < 3; ++i) launch_colsum(d, dz, sp, st);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) launch_colsum(d, dz, sp, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    auto reduce = [&](float* p, size_t per, int nsl) {
      auto h = to_host(p, (size_t)nsl * per);
      std::vector<float> r(per, 0.f);
      for (int s = 0; s < nsl; ++s) for (size_t i = 0; i < per; ++i) r[i] += h[(size_t)s * per + i];
      return r;
    };
    auto hs_ = reduce(sp, 9 * C, d.N);
    double serr = 0, smax = 0;
    for (int t = 0; t < 9; ++t)
      for (int c = 0; c < d.C; ++c) {
        double acc = 0;
        for (int n = 0; n < d.N; ++n)
          for (int h = 0; h < d.H; ++h)
            for (int x = 0; x < d.W; ++x) {
              const int hh = h + t / 3 - 1, xx = x + t % 3 - 1;
              if (hh < 0 || hh >= d.H || xx < 0 || xx >= d.W) continue;
              acc += (double)hdz[((size_t)n * d.HW + h * d.W + x) * C + c];
            }
        serr = std::max(serr, std::fabs(acc - (double)hs_[(size_t)t * C + c]));
        smax = std::max(smax, std::fabs(acc));
      }
    printf("k_colsum %.2f us per launch; vs fp64 host reference: max err %.3e (ref max %.3e)\n", ms * 1e3 / 10, serr, smax);
  } else {
    fprintf(stderr, "unknown bench %s\n", what.c_str());
    return 2;
  }
  return 0;
}
