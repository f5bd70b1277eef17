// Decisions of the adaptive step loop, shared by the kernels that take them: k_step_controller / k_init_controller (one launch per
// decision, kernels_pointwise.hip) and k_tiny_solve (a whole forward solve in one launch, kernels_tiny_solve.hip) -- one source, so
// the two step loops cannot drift apart.  `a.ctrl` may live in global memory or in LDS.
#pragma once
#include "node_internal.h"
#include "../../include/node_hip.h"

namespace node {

// Dormand-Prince / Shampine coefficients (rounded to fp32 exactly as `fp32_tensor * python_float` does on the reference path).
__device__ constexpr float c_CSOL[7] = {
    (float)(35.0 / 384.0), 0.f, (float)(500.0 / 1113.0), (float)(125.0 / 192.0),
    (float)(-2187.0 / 6784.0), (float)(11.0 / 84.0), 0.f};
__device__ constexpr float c_CERR[7] = {
    (float)(35.0 / 384.0 - 1951.0 / 21600.0), 0.f, (float)(500.0 / 1113.0 - 22642.0 / 50085.0),
    (float)(125.0 / 192.0 - 451.0 / 720.0), (float)(-2187.0 / 6784.0 - -12231.0 / 42400.0),
    (float)(11.0 / 84.0 - 649.0 / 6300.0), (float)(-1.0 / 60.0)};

// Step controller, ONE thread.  Mirrors `_adaptive_dopri5_step` / `_optimal_step_size` of the restated solver: accept iff every
// segment's mean squared error ratio <= 1; dt <- dt / clamp(sqrt(max ratio)^(1/5)/0.9, 0.1, 1/dfactor).  t / dt are float64 like
// upstream's adaptive solvers.  ratios[0 .. nseg): the tensor segments' mean squared error ratios (room for the scalar one behind).
__device__ inline void step_controller_decide(const StepCtlArgs& a, float* ratios) {
  Ctrl* c = a.ctrl;
  const double t = c->t, dt = c->dt;
  const float dtf = (float)dt;
  const int step = c->step_idx;
  int nr = a.nseg;
  if (a.has_scalar) {
    float e = (dtf * c_CERR[0]) * c->ts_k[0];
    float s = (dtf * c_CSOL[0]) * c->ts_k[0];
#pragma unroll
    for (int j = 2; j < 7; ++j) { e += (dtf * c_CERR[j]) * c->ts_k[j]; if (j < 6) s += (dtf * c_CSOL[j]) * c->ts_k[j]; }
    const float y1 = c->ts_cur + s;
    const float r = e / (a.atol + a.rtol * fmaxf(fabsf(c->ts_cur), fabsf(y1)));
    ratios[nr++] = a.gbuf != nullptr ? a.gbuf[3] / a.gworld : r * r;     // (global-norm mode: the mean of the ranks' ratios, k_norm_pack)
    c->ts_new = y1;
  }
  bool accept = true, nan = false;
  float maxr = 0.f;
  for (int i = 0; i < nr; ++i) {
    const float r = ratios[i];
    if (!(r <= 1.0f)) accept = false;
    if (r != r) nan = true;
    maxr = fmaxf(maxr, r);
    c->ratio[i] = r;
  }
  for (int i = nr; i < 4; ++i) c->ratio[i] = 0.f;
  double dt_next;
  int done = 0;
  if (a.forced != nullptr) {   // replay: every step accepted, sizes from the list (the last one repeats)
    accept = true;
    const double nd = step + 1 < a.n_forced ? a.forced[step + 1] : -1.0;
    dt_next = nd > 0.0 ? nd : dt;
  } else if (nan) {
    c->status = NODE_ERR_NONFINITE;
    dt_next = dt;
    done = 1;
  } else if (maxr == 0.f) {
    dt_next = dt * 10.0;
  } else {
    const double dfactor = maxr < 1.0f ? 1.0 : 0.2;
    const double er = (double)sqrtf(maxr);
    double factor = pow(er, 0.2) / 0.9;
    factor = fmin(factor, 1.0 / dfactor);
    factor = fmax(0.1, factor);
    dt_next = dt / factor;
  }
  c->t_prev = t;
  c->dt_used = dt;
  c->accept = accept ? 1 : 0;
  if (step == 0) c->first_dt = dt;
  if (a.dt_log != nullptr && step < a.dt_log_cap) a.dt_log[step] = accept ? dt : -dt;
  c->step_idx = step + 1;
  const int j0 = c->j;
  int j1 = j0;
  double t_now = t;
  if (accept) {
    t_now = t + dt;
    c->t = t_now;
    c->n_acc += 1;
    if (a.has_scalar) {  // keep the step's (y0, f0) for dense output, then FSAL
      c->ts_y0_prev = c->ts_cur;
      c->ts_f0_prev = c->ts_k[0];
      c->ts_cur = c->ts_new;
      c->ts_k[0] = c->ts_k[6];
    }
    // targets passed by this step (upstream advances until t >= target, no clamping, then interpolates)
    while (j1 < a.n_targets && !(a.targets[j1] > t_now)) ++j1;
    if (j1 == a.n_targets) {
      done = 1;
      if (a.has_scalar && a.interp_scalar && j1 > j0) {   // scalar segment of the augmented state at the interval's end
        const float t0f = (float)t, t1f = (float)t_now, tjf = (float)a.targets[a.n_targets - 1];
        const float x = (tjf - t0f) / (t1f - t0f);
        float kk[7];
        for (int q = 0; q < 7; ++q) kk[q] = c->ts_k[q];
        kk[0] = c->ts_f0_prev;
        c->ts_cur = interp_one(c->ts_y0_prev, c->ts_new, kk, dtf, x);
      }
    }
  } else {
    c->n_rej += 1;
  }
  c->j0 = j0;
  c->j1 = j1;
  c->j = j1;
  c->dt = dt_next;
  if (!done && !(t_now + dt_next > t_now)) {   // upstream: 'underflow in dt'
    c->status = NODE_ERR_DT_UNDERFLOW;
    done = 1;
  }
  c->done = done;
}

// Hairer initial step (`_select_initial_step`, order argument 4), ONE thread.  sums[seg] = {sum (y0/scale)^2, sum (f0/scale)^2}
// (phase 0) or {sum ((f1-f0)/scale)^2, -} (phase 1).
__device__ inline void init_controller_decide(const InitCtlArgs& a, const float (*sums)[2]) {
  Ctrl* c = a.ctrl;
  if (a.phase == 0) {
    float d0max = 0.f, d1max = 0.f, qmax = -INFINITY;
    for (int sgi = 0; sgi < a.nseg; ++sgi) {
      const float d0 = sqrtf((float)((double)sums[sgi][0] / a.numel[sgi]));
      const float d1 = sqrtf((float)((double)sums[sgi][1] / a.numel[sgi]));
      d0max = fmaxf(d0max, d0);
      d1max = fmaxf(d1max, d1);
      qmax = fmaxf(qmax, d0 / d1);
    }
    if (a.has_scalar) {
      const float sc = a.atol + fabsf(c->ts_cur) * a.rtol;
      float d0 = fabsf(c->ts_cur / sc), d1 = fabsf(c->ts_k[0] / sc);
      if (a.gbuf != nullptr) { d0 = sqrtf(a.gbuf[6] / a.gworld); d1 = sqrtf(a.gbuf[7] / a.gworld); }   // (root mean square over the ranks)
      d0max = fmaxf(d0max, d0);
      d1max = fmaxf(d1max, d1);
      qmax = fmaxf(qmax, d0 / d1);
    }
    float h0;
    if (d0max < 1e-5f || d1max < 1e-5f) h0 = 1e-6f;
    else h0 = 0.01f * qmax;
    c->h0 = h0;
    c->d0 = d0max;
    c->d1 = d1max;
  } else {
    const float h0 = c->h0;
    float d2max = 0.f;
    for (int sgi = 0; sgi < a.nseg; ++sgi) {
      const float d2 = sqrtf((float)((double)sums[sgi][0] / a.numel[sgi])) / h0;
      d2max = fmaxf(d2max, d2);
    }
    if (a.has_scalar) {
      const float sc = a.atol + fabsf(c->ts_cur) * a.rtol;
      d2max = fmaxf(d2max, (a.gbuf != nullptr ? sqrtf(a.gbuf[3] / a.gworld) : fabsf((c->ts_k[1] - c->ts_k[0]) / sc)) / h0);
    }
    float h1;
    if (c->d1 <= 1e-15f && d2max <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
    else h1 = powf(0.01f / fmaxf(c->d1, d2max), 1.0f / 5.0f);
    c->dt = (double)fminf(100.f * h0, h1);
  }
}

}  // namespace node
