// HBM-bound kernels of the integrator: Butcher-tableau stage combine fused with
// GroupNorm+ReLU (and its backward), the wavefront-reduced error norm, the
// device-resident step controller (accept / dt / t never leave the GPU except
// for one 64-byte read-back per step), Hairer's initial step, and the quartic
// dense-output interpolant.
//
// Algorithm follows the restated torchdiffeq spec; the CPU
// statement of the same arithmetic is oracle/torchdiffeq_restated.py.
#include "node_internal.h"
#include "wino4.h"
#include "step_control.h"
#include <cstring>
#include "../../include/node_hip.h"

namespace node {

// (Dormand-Prince / Shampine coefficients c_CSOL / c_CERR, the step controller's and the initial step's decisions: step_control.h)

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// deterministic block sum (256 threads), result valid in every thread
__device__ inline float block_sum_256(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ============================================================================
// Stage combine + GroupNorm + ReLU
//   y_i  = y + scale * sum_j coef_j k_j                  (Butcher row)
//   act  = relu(GN(y_i) * gamma + beta)                   (model.py:341-342)
// One workgroup owns (sample n, a slab of whole groups): the combined values stay
// in LDS between the statistics pass and the normalise pass, so y_i is never
// written to HBM unless the caller asks for it (last stage -> y1).
// ============================================================================
// y + sum_j cf[j] * k[j] at TWO offsets of the same thread, every request issued before the first use.  Written as a
// loop over a run-time term count the compiler waits for each tensor's load before it requests the next: (1 + nk) x 2
// dependent round trips at the head of every combine (12 for the last dopri5 stage).  Same summation order as the
// generic loops below.
template <int NK>
__device__ __forceinline__ void comb_pair(const Comb& c, const float* cf, size_t off0, size_t off1, float4& r0, float4& r1) {
  float4 y0 = ld4(c.y + off0), y1 = ld4(c.y + off1);
  float4 k0[NK > 0 ? NK : 1], k1[NK > 0 ? NK : 1];
#pragma unroll
  for (int j = 0; j < NK; ++j) { k0[j] = ld4(c.k[j] + off0); k1[j] = ld4(c.k[j] + off1); }
  if (NK > 0) {
    float4 s0, s1;
    s0.x = cf[0] * k0[0].x; s0.y = cf[0] * k0[0].y; s0.z = cf[0] * k0[0].z; s0.w = cf[0] * k0[0].w;
    s1.x = cf[0] * k1[0].x; s1.y = cf[0] * k1[0].y; s1.z = cf[0] * k1[0].z; s1.w = cf[0] * k1[0].w;
#pragma unroll
    for (int j = 1; j < NK; ++j) {
      s0.x += cf[j] * k0[j].x; s0.y += cf[j] * k0[j].y; s0.z += cf[j] * k0[j].z; s0.w += cf[j] * k0[j].w;
      s1.x += cf[j] * k1[j].x; s1.y += cf[j] * k1[j].y; s1.z += cf[j] * k1[j].z; s1.w += cf[j] * k1[j].w;
    }
    y0.x += s0.x; y0.y += s0.y; y0.z += s0.z; y0.w += s0.w;
    y1.x += s1.x; y1.y += s1.y; y1.z += s1.z; y1.w += s1.w;
  }
  r0 = y0;
  r1 = y1;
}
__device__ __forceinline__ void comb_pair_any(const Comb& c, const float* cf, size_t off0, size_t off1, float4& r0, float4& r1) {
  switch (c.nk) {
    case 0: comb_pair<0>(c, cf, off0, off1, r0, r1); break;
    case 1: comb_pair<1>(c, cf, off0, off1, r0, r1); break;
    case 2: comb_pair<2>(c, cf, off0, off1, r0, r1); break;
    case 3: comb_pair<3>(c, cf, off0, off1, r0, r1); break;
    case 4: comb_pair<4>(c, cf, off0, off1, r0, r1); break;
    case 5: comb_pair<5>(c, cf, off0, off1, r0, r1); break;
    case 6: comb_pair<6>(c, cf, off0, off1, r0, r1); break;
    default: comb_pair<7>(c, cf, off0, off1, r0, r1); break;
  }
}

__global__ __launch_bounds__(256) void k_combine_gn(CombineGnArgs a, Dims d) {
  if (a.ctrl->done) return;   // a step enqueued past the end of the interval (see Ctrl)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x, c0 = blockIdx.y * d.cs;
  const int csl = min(d.cs, d.C - c0);
  const int cs4 = csl >> 2;
  const int nvec = d.HW * cs4;
  float* tile = smem;                      // [HW][csl]
  float* smean = smem + d.HW * d.cs;       // [cs/cpg]
  float* srstd = smean + d.cs;             // generous

  const float scale = comb_scale(a.comb, a.ctrl);
  float cf[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) cf[j] = scale * a.comb.coef[j];

  if ((nvec & 511) == 0) {   // two units per thread and round, all their requests in flight together
    for (int v = tid; v < nvec; v += 512) {
      const int p0 = v / cs4, q0 = v - p0 * cs4, p1 = (v + 256) / cs4, q1 = (v + 256) - p1 * cs4;
      const size_t off0 = ((size_t)(n * d.HW + p0)) * d.C + c0 + 4 * q0, off1 = ((size_t)(n * d.HW + p1)) * d.C + c0 + 4 * q1;
      float4 y0, y1;
      comb_pair_any(a.comb, cf, off0, off1, y0, y1);
      st4(tile + p0 * csl + 4 * q0, y0);
      st4(tile + p1 * csl + 4 * q1, y1);
      if (a.y_out) { st4(a.y_out + off0, y0); st4(a.y_out + off1, y1); }
    }
  } else
  for (int v = tid; v < nvec; v += 256) {
    const int p = v / cs4, q = v - p * cs4;
    const size_t off = ((size_t)(n * d.HW + p)) * d.C + c0 + 4 * q;
    float4 yv = ld4(a.comb.y + off);
    if (a.comb.nk > 0) {
      float4 s;
      {
        float4 kv = ld4(a.comb.k[0] + off);
        s.x = cf[0] * kv.x; s.y = cf[0] * kv.y; s.z = cf[0] * kv.z; s.w = cf[0] * kv.w;
      }
      for (int j = 1; j < a.comb.nk; ++j) {
        float4 kv = ld4(a.comb.k[j] + off);
        s.x += cf[j] * kv.x; s.y += cf[j] * kv.y; s.z += cf[j] * kv.z; s.w += cf[j] * kv.w;
      }
      yv.x += s.x; yv.y += s.y; yv.z += s.z; yv.w += s.w;
    }
    st4(tile + p * csl + 4 * q, yv);
    if (a.y_out) st4(a.y_out + off, yv);
  }
  __syncthreads();

  const int ngs = csl / d.cpg;
  const int m = d.HW * d.cpg;
  const float inv_m = 1.0f / (float)m;
  for (int gi = wave; gi < ngs; gi += 4) {
    float s = 0.f;
    for (int e = lane; e < m; e += 64) {
      const int p = e / d.cpg, cc = e - p * d.cpg;
      s += tile[p * csl + gi * d.cpg + cc];
    }
    const float mean = wave_sum(s) * inv_m;
    float s2 = 0.f;
    for (int e = lane; e < m; e += 64) {
      const int p = e / d.cpg, cc = e - p * d.cpg;
      const float dv = tile[p * csl + gi * d.cpg + cc] - mean;
      s2 += dv * dv;
    }
    const float var = wave_sum(s2) * inv_m;
    const float rstd = 1.0f / sqrtf(var + d.eps);
    if (lane == 0) {
      smean[gi] = mean;
      srstd[gi] = rstd;
      if (a.rstd_out) a.rstd_out[(size_t)n * d.G + c0 / d.cpg + gi] = rstd;
    }
  }
  __syncthreads();

  for (int v = tid; v < nvec; v += 256) {
    const int p = v / cs4, q = v - p * cs4;
    const size_t off = ((size_t)(n * d.HW + p)) * d.C + c0 + 4 * q;
    const float4 xv = ld4(tile + p * csl + 4 * q);
    const float4 gm = ld4(a.gamma + c0 + 4 * q);
    const float4 bt = ld4(a.beta + c0 + 4 * q);
    float x[4] = {xv.x, xv.y, xv.z, xv.w};
    float g[4] = {gm.x, gm.y, gm.z, gm.w};
    float b[4] = {bt.x, bt.y, bt.z, bt.w};
    float xh[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gl = (4 * q + i) / d.cpg;
      xh[i] = (x[i] - smean[gl]) * srstd[gl];
      float vv = xh[i] * g[i] + b[i];
      if (a.relu) vv = fmaxf(vv, 0.f);
      o[i] = a.osign * vv;
    }
    if (a.act_out) st4(a.act_out + off, make_float4(o[0], o[1], o[2], o[3]));
    if (a.xhat_out) st4(a.xhat_out + off, make_float4(xh[0], xh[1], xh[2], xh[3]));
  }
}

void launch_combine_gn(const Dims& d, const CombineGnArgs& a, hipStream_t s) {
  size_t lds = ((size_t)d.HW * d.cs + 2 * (size_t)d.cs) * sizeof(float);
  hipLaunchKernelGGL(k_combine_gn, dim3(d.N, d.nslab), dim3(256), lds, s, a, d);
}

// ============================================================================
// Top of the backward chain: combine the adjoint state, negate it into the
// cotangent and push it through GroupNorm-3's backward.
//   g   = csign * (a + scale * sum coef_j k^a_j)
//   dz  = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
//   per-sample partials of dgamma = sum g*xhat, dbeta = sum g
// ============================================================================
__global__ __launch_bounds__(256) void k_gn_bwd(GnBwdArgs a, Dims d) {
  if (a.ctrl->done) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x, c0 = blockIdx.y * d.cs;
  const int csl = min(d.cs, d.C - c0);
  const int cs4 = csl >> 2;
  const int nvec = d.HW * cs4;
  float* gt = smem;                           // [HW][csl]   g
  float* xt = gt + d.HW * d.cs;               // [HW][csl]   xhat
  float* sm1 = xt + d.HW * d.cs;              // [cs]
  float* sm2 = sm1 + d.cs;                    // [cs]
  unsigned char* flg = reinterpret_cast<unsigned char*>(sm2 + d.cs);   // [HW] border flags (rounded up to 16 B)
  float* cred = sm2 + d.cs + ((d.HW + 15) / 16) * 4;   // [256][2] channel partials
  float* red9 = cred + 512;                   // [9 * 256] masked column sums across pixel groups
  if (a.spart)
    for (int p = tid; p < d.HW; p += 256) {
      const int h = p / d.W, x = p - h * d.W;
      flg[p] = (unsigned char)((h == 0 ? 1 : 0) | (h == d.H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == d.W - 1 ? 8 : 0));
    }

  const float scale = comb_scale(a.comb, a.ctrl);
  float cf[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) cf[j] = scale * a.comb.coef[j];

  if ((nvec & 511) == 0) {   // two units per thread and round, all their requests in flight together
    for (int v = tid; v < nvec; v += 512) {
      const int p0 = v / cs4, q0 = v - p0 * cs4, p1 = (v + 256) / cs4, q1 = (v + 256) - p1 * cs4;
      const size_t off0 = ((size_t)(n * d.HW + p0)) * d.C + c0 + 4 * q0, off1 = ((size_t)(n * d.HW + p1)) * d.C + c0 + 4 * q1;
      const float4 x0 = ld4(a.xhat + off0), x1 = ld4(a.xhat + off1);
      float4 m0 = make_float4(1.f, 1.f, 1.f, 1.f), m1 = m0;
      if (a.mask_act) { m0 = ld4(a.mask_act + off0); m1 = ld4(a.mask_act + off1); }
      float4 a0, a1;
      comb_pair_any(a.comb, cf, off0, off1, a0, a1);
      if (a.a_out) { st4(a.a_out + off0, a0); st4(a.a_out + off1, a1); }
      float4 g0 = make_float4(a.csign * a0.x, a.csign * a0.y, a.csign * a0.z, a.csign * a0.w);
      float4 g1 = make_float4(a.csign * a1.x, a.csign * a1.y, a.csign * a1.z, a.csign * a1.w);
      g0.x = m0.x > 0.f ? g0.x : 0.f; g0.y = m0.y > 0.f ? g0.y : 0.f; g0.z = m0.z > 0.f ? g0.z : 0.f; g0.w = m0.w > 0.f ? g0.w : 0.f;
      g1.x = m1.x > 0.f ? g1.x : 0.f; g1.y = m1.y > 0.f ? g1.y : 0.f; g1.z = m1.z > 0.f ? g1.z : 0.f; g1.w = m1.w > 0.f ? g1.w : 0.f;
      st4(gt + p0 * csl + 4 * q0, g0);
      st4(gt + p1 * csl + 4 * q1, g1);
      st4(xt + p0 * csl + 4 * q0, x0);
      st4(xt + p1 * csl + 4 * q1, x1);
    }
  } else
  for (int v = tid; v < nvec; v += 256) {
    const int p = v / cs4, q = v - p * cs4;
    const size_t off = ((size_t)(n * d.HW + p)) * d.C + c0 + 4 * q;
    float4 av = ld4(a.comb.y + off);
    if (a.comb.nk > 0) {
      float4 s;
      {
        float4 kv = ld4(a.comb.k[0] + off);
        s.x = cf[0] * kv.x; s.y = cf[0] * kv.y; s.z = cf[0] * kv.z; s.w = cf[0] * kv.w;
      }
      for (int j = 1; j < a.comb.nk; ++j) {
        float4 kv = ld4(a.comb.k[j] + off);
        s.x += cf[j] * kv.x; s.y += cf[j] * kv.y; s.z += cf[j] * kv.z; s.w += cf[j] * kv.w;
      }
      av.x += s.x; av.y += s.y; av.z += s.z; av.w += s.w;
    }
    if (a.a_out) st4(a.a_out + off, av);
    float4 gq = make_float4(a.csign * av.x, a.csign * av.y, a.csign * av.z, a.csign * av.w);
    if (a.mask_act) {
      const float4 mk = ld4(a.mask_act + off);
      gq.x = mk.x > 0.f ? gq.x : 0.f; gq.y = mk.y > 0.f ? gq.y : 0.f;
      gq.z = mk.z > 0.f ? gq.z : 0.f; gq.w = mk.w > 0.f ? gq.w : 0.f;
    }
    st4(gt + p * csl + 4 * q, gq);
    st4(xt + p * csl + 4 * q, ld4(a.xhat + off));
  }
  __syncthreads();

  // per-channel partial sums over the sample's pixels (dgamma, dbeta)
  {
    const int npg = csl <= 256 ? 256 / csl : 1;
    for (int cbase = 0; cbase < csl; cbase += 256) {
      const int cl = cbase + (tid % min(csl, 256));
      const int pg = tid / min(csl, 256);
      float dg = 0.f, db = 0.f;
      if (pg < npg && cl < csl) {
        for (int p = pg; p < d.HW; p += npg) {
          const float g = gt[p * csl + cl];
          dg += g * xt[p * csl + cl];
          db += g;
        }
      }
      cred[tid * 2] = dg;
      cred[tid * 2 + 1] = db;
      __syncthreads();
      if (pg == 0 && cl < csl) {
        const int stride = min(csl, 256);
        for (int r = 1; r < npg; ++r) {
          dg += cred[(r * stride + (tid % stride)) * 2];
          db += cred[(r * stride + (tid % stride)) * 2 + 1];
        }
        a.gpart[((size_t)n * 2 + 0) * d.C + c0 + cl] = dg;
        a.gpart[((size_t)n * 2 + 1) * d.C + c0 + cl] = db;
      }
      __syncthreads();
    }
  }

  const int ngs = csl / d.cpg;
  const int m = d.HW * d.cpg;
  const float inv_m = 1.0f / (float)m;
  for (int gi = wave; gi < ngs; gi += 4) {
    float s1 = 0.f, s2 = 0.f;
    for (int e = lane; e < m; e += 64) {
      const int p = e / d.cpg, cc = e - p * d.cpg;
      const int col = gi * d.cpg + cc;
      const float dxh = gt[p * csl + col] * a.gamma[c0 + col];
      s1 += dxh;
      s2 += dxh * xt[p * csl + col];
    }
    s1 = wave_sum(s1) * inv_m;
    s2 = wave_sum(s2) * inv_m;
    if (lane == 0) { sm1[gi] = s1; sm2[gi] = s2; }
  }
  __syncthreads();

  for (int v = tid; v < nvec; v += 256) {
    const int p = v / cs4, q = v - p * cs4;
    const size_t off = ((size_t)(n * d.HW + p)) * d.C + c0 + 4 * q;
    const float4 gv = ld4(gt + p * csl + 4 * q);
    const float4 xv = ld4(xt + p * csl + 4 * q);
    const float4 gm = ld4(a.gamma + c0 + 4 * q);
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    float x[4] = {xv.x, xv.y, xv.z, xv.w};
    float w[4] = {gm.x, gm.y, gm.z, gm.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gl = (4 * q + i) / d.cpg;
      const float r = a.rstd[(size_t)n * d.G + c0 / d.cpg + gl];
      o[i] = a.osign * (r * (g[i] * w[i] - sm1[gl] - x[i] * sm2[gl]));
    }
    st4(a.dz_out + off, make_float4(o[0], o[1], o[2], o[3]));
    if (a.spart) st4(gt + p * csl + 4 * q, make_float4(o[0], o[1], o[2], o[3]));   // dz tile for the column sums below
  }
  if (a.spart) {   // masked column sums of this sample's dz slab while it is still in LDS (replaces a k_colsum launch)
    __syncthreads();
    for (int cbase = 0; cbase < csl; cbase += 256) {
      const int ncols = min(csl - cbase, 256);
      masked_colsum_tile(gt + cbase, csl, d.HW, flg, ncols, max(1, 256 / ncols), tid, red9,
                         a.spart + (size_t)n * 9 * d.C + c0 + cbase, d.C);
    }
  }
}

void launch_gn_bwd(const Dims& d, const GnBwdArgs& a, hipStream_t s) {
  size_t lds = (2 * (size_t)d.HW * d.cs + 2 * (size_t)d.cs + 512 + 9 * 256) * sizeof(float) + (size_t)d.HW + 16;
  hipLaunchKernelGGL(k_gn_bwd, dim3(d.N, d.nslab), dim3(256), lds, s, a, d);
}

// ============================================================================
// Error norm:  sum_i (err_i / (atol + rtol*max(|y0_i|,|y1_i|)))^2,
//   err = dt * sum_j c_err_j k_j    -- wave-reduced, one partial per workgroup
// For segments whose intermediate stages are never consumed (adj_params) the same
// pass also forms y1 = y0 + dt * sum_j b_j k_j.
// ============================================================================
// One launch for up to three state segments (grid.y = segment): the augmented state's y, a and theta segments used to
// be three launches per step.
struct ErrSegs { ErrSeg seg[3]; float* partial[3]; };
__device__ void error_norm_body(const ErrSeg& seg, const Ctrl* ctrl, float rtol, float atol, float* partial);
__global__ __launch_bounds__(256) void k_error_norm(ErrSegs a, const Ctrl* ctrl, float rtol, float atol) {
  if (ctrl->done) return;
  error_norm_body(a.seg[blockIdx.y], ctrl, rtol, atol, a.partial[blockIdx.y]);
}
__device__ void error_norm_body(const ErrSeg& seg, const Ctrl* ctrl, float rtol, float atol, float* partial) {
  __shared__ float red[4];
  const float dtf = (float)ctrl->dt;
  float ce[7], cb[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) { ce[j] = dtf * c_CERR[j]; cb[j] = dtf * c_CSOL[j]; }
  float acc = 0.f;
  const size_t n4 = seg.n >> 2;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n4; v += stride) {
    const size_t off = v * 4;
    float4 kv[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) kv[j] = (j == 1) ? make_float4(0, 0, 0, 0) : ld4(seg.k[j] + off);
    const float4 y0 = ld4(seg.y0 + off);
    float4 y1;
    if (seg.compute_y1) {
      float4 s;
      s.x = cb[0] * kv[0].x; s.y = cb[0] * kv[0].y; s.z = cb[0] * kv[0].z; s.w = cb[0] * kv[0].w;
#pragma unroll
      for (int j = 2; j < 6; ++j) { s.x += cb[j] * kv[j].x; s.y += cb[j] * kv[j].y; s.z += cb[j] * kv[j].z; s.w += cb[j] * kv[j].w; }
      y1 = make_float4(y0.x + s.x, y0.y + s.y, y0.z + s.z, y0.w + s.w);
      st4(seg.y1 + off, y1);
    } else {
      y1 = ld4(seg.y1 + off);
    }
    float4 e;
    e.x = ce[0] * kv[0].x; e.y = ce[0] * kv[0].y; e.z = ce[0] * kv[0].z; e.w = ce[0] * kv[0].w;
#pragma unroll
    for (int j = 2; j < 7; ++j) { e.x += ce[j] * kv[j].x; e.y += ce[j] * kv[j].y; e.z += ce[j] * kv[j].z; e.w += ce[j] * kv[j].w; }
    float r;
    r = e.x / (atol + rtol * fmaxf(fabsf(y0.x), fabsf(y1.x))); acc += r * r;
    r = e.y / (atol + rtol * fmaxf(fabsf(y0.y), fabsf(y1.y))); acc += r * r;
    r = e.z / (atol + rtol * fmaxf(fabsf(y0.z), fabsf(y1.z))); acc += r * r;
    r = e.w / (atol + rtol * fmaxf(fabsf(y0.w), fabsf(y1.w))); acc += r * r;
    // upstream asserts the state is finite at every step ('non-finite values in state `y`'); fmaxf above drops a
    // NaN operand and ReLU turns a NaN pre-activation into 0, so an infinite / NaN state would otherwise pass
    // unnoticed: 0 * (inf or NaN) = NaN poisons the sum, and the controller reports NODE_ERR_NONFINITE
    acc += 0.f * (((y0.x + y0.y) + (y0.z + y0.w)) + ((y1.x + y1.y) + (y1.z + y1.w)));
  }
  // scalar tail (n % 4), handled by block 0
  if (blockIdx.x == 0) {
    for (size_t i = (n4 << 2) + threadIdx.x; i < seg.n; i += 256) {
      float kk[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) kk[j] = (j == 1) ? 0.f : seg.k[j][i];
      const float y0 = seg.y0[i];
      float y1;
      if (seg.compute_y1) {
        float s = cb[0] * kk[0];
#pragma unroll
        for (int j = 2; j < 6; ++j) s += cb[j] * kk[j];
        y1 = y0 + s;
        seg.y1[i] = y1;
      } else {
        y1 = seg.y1[i];
      }
      float e = ce[0] * kk[0];
#pragma unroll
      for (int j = 2; j < 7; ++j) e += ce[j] * kk[j];
      const float r = e / (atol + rtol * fmaxf(fabsf(y0), fabsf(y1)));
      acc += r * r;
      acc += 0.f * (y0 + y1);
    }
  }
  const float tot = block_sum_256(acc, red);
  // (agent-scope = write-through: the last-arriving workgroup of k_error_norm_ctl, on another XCD, reads it inside the same launch)
  if (threadIdx.x == 0) __hip_atomic_store(&partial[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void launch_error_norm(const ErrSeg* segs, float* const* partial, int nseg, const Ctrl* ctrl, float rtol, float atol, hipStream_t s) {
  ErrSegs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < nseg; ++i) { a.seg[i] = segs[i]; a.partial[i] = partial[i]; }
  hipLaunchKernelGGL(k_error_norm, dim3(ERR_BLOCKS, nseg), dim3(256), 0, s, a, ctrl, rtol, atol);
}

// ============================================================================
// Step controller (one workgroup).  Mirrors `_adaptive_dopri5_step` /
// `_optimal_step_size` of the restated solver: accept iff every segment's mean
// squared error ratio <= 1; dt <- dt / clamp(sqrt(max ratio)^(1/5)/0.9, 0.1, 1/dfactor).
// t / dt are float64 like upstream's adaptive solvers.
// ============================================================================
__device__ inline float reduce_partials_512(const float* p, float* red) {
  float v = p[threadIdx.x] + p[threadIdx.x + 256];
  return block_sum_256(v, red);
}

// COHERENT: the partials were written by other workgroups of the SAME launch (k_error_norm_ctl): agent-scope loads
template <bool COHERENT>
__device__ inline void step_controller_body(const StepCtlArgs& a, float* red, float* ratios) {
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    if (a.gbuf != nullptr) {      // global-norm mode: the sums of all ranks (k_norm_pack + the caller's all-reduce)
      if (threadIdx.x == 0) ratios[sgi] = (float)((double)a.gbuf[sgi] / (a.numel[sgi] * (double)a.gworld));
      continue;
    }
    float tot;
    if (COHERENT) {
      const float* p = a.partial[sgi];
      const float v = __hip_atomic_load(p + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                      __hip_atomic_load(p + threadIdx.x + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tot = block_sum_256(v, red);
    } else {
      tot = reduce_partials_512(a.partial[sgi], red);
    }
    if (threadIdx.x == 0) ratios[sgi] = (float)((double)tot / a.numel[sgi]);
    __syncthreads();
  }
  bool w4_ovf = false;
  if (a.w4sc != nullptr) w4_ovf = w4_gscale_update(a.w4sc, threadIdx.x, red);   // (all 256 threads)
  if (a.gbuf != nullptr && a.gbuf[4] > 0.f) w4_ovf = true;                      // (some rank's cotangent left its scale: all repeat)
  if (threadIdx.x != 0) return;
  if (a.w4sc != nullptr) {
    // fp16-pair operands (wino4.h): the next step's cotangent scale from this step's recorded maximum (above); a step in which a pass
    // met a value beyond its scale's range is REPEATED at the new scale -- nothing accepted, t and dt as they were, not a solver step
    W4Scales* sc = a.w4sc;
    if (w4_ovf && sc->pad[0] < 8u) {
      sc->pad[0] += 1u;                    // (consecutive repeats: bounded)
      Ctrl* c = a.ctrl;
      c->accept = 0;
      c->t_prev = c->t;
      c->dt_used = c->dt;
      c->j0 = c->j1 = c->j;
      return;
    }
    sc->pad[0] = 0u;
  }
  step_controller_decide(a, ratios);
}
__global__ __launch_bounds__(256) void k_step_controller(StepCtlArgs a) {
  __shared__ float red[4];
  __shared__ float ratios[4];
  if (a.ctrl->done) {   // a step enqueued past the end of the interval: nothing was computed, nothing is emitted
    if (threadIdx.x == 0) a.ctrl->j0 = a.ctrl->j1 = a.ctrl->j;
    return;
  }
  step_controller_body<false>(a, red, ratios);
}

void launch_step_controller(const StepCtlArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_step_controller, dim3(1), dim3(256), 0, s, a);
}

// The same two kernels as ONE launch: every workgroup of the error norm leaves its partial sum (write-through), waits for the write's
// acknowledgement and takes a ticket; the last one to arrive is the step controller.  (The hand-off without fences of k_theta_finalize:
// agent-scope stores / loads around an agent-scope counter -- gfx950's memory system, see there.)
__global__ __launch_bounds__(256) void k_error_norm_ctl(ErrSegs a, StepCtlArgs ctl, unsigned* arrive) {
  __shared__ float red[4];
  __shared__ float ratios[4];
  __shared__ int s_last;
  if (ctl.ctrl->done) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ctl.ctrl->j0 = ctl.ctrl->j1 = ctl.ctrl->j;
    return;
  }
  error_norm_body(a.seg[blockIdx.y], ctl.ctrl, ctl.rtol, ctl.atol, a.partial[blockIdx.y]);
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partial's write is acknowledged
    s_last = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream order)
  step_controller_body<true>(ctl, red, ratios);
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_error_norm_ctl's fence-free hand-off is only valid on gfx950 (see k_theta_finalize)"
#endif
void launch_error_norm_ctl(const ErrSeg* segs, float* const* partial, int nseg, const StepCtlArgs& ctl, unsigned* arrive, hipStream_t s) {
  ErrSegs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < nseg; ++i) { a.seg[i] = segs[i]; a.partial[i] = partial[i]; }
  hipLaunchKernelGGL(k_error_norm_ctl, dim3(ERR_BLOCKS, nseg), dim3(256), 0, s, a, ctl, arrive);
}

__global__ __launch_bounds__(256) void k_w4_gscale(W4Scales* sc, int skew) {
  __shared__ float red[4];
  (void)w4_gscale_update(sc, threadIdx.x, red);
  if (threadIdx.x == 0 && skew != 0) sc->e[W4_E_G] += skew;
}
void launch_w4_gscale(W4Scales* sc, hipStream_t s, int skew) { hipLaunchKernelGGL(k_w4_gscale, dim3(1), dim3(256), 0, s, sc, skew); }

// ============================================================================
// Hairer initial step (`_select_initial_step`, order argument 4)
//   phase 0: sum (y0/scale)^2, sum (f0/scale)^2      scale = atol + |y0| rtol
//   phase 1: sum ((f1-f0)/scale)^2
// ============================================================================
struct InitSegs { InitSeg seg[3]; float* partial[3]; };
__device__ inline void init_norms_body(const InitSegs& a, float rtol, float atol, int phase, float* red);
__global__ __launch_bounds__(256) void k_init_norms(InitSegs a, float rtol, float atol, int phase) {
  __shared__ float red[4];
  init_norms_body(a, rtol, atol, phase, red);
}
__device__ inline void init_norms_body(const InitSegs& a, float rtol, float atol, int phase, float* red) {
  const InitSeg& seg = a.seg[blockIdx.y];
  float* partial = a.partial[blockIdx.y];
  float a0 = 0.f, a1 = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < seg.n; i += stride) {
    const float y = seg.y0[i];
    const float sc = atol + fabsf(y) * rtol;
    if (phase == 0) {
      const float u = y / sc, v = seg.f0[i] / sc;
      a0 += u * u;
      a1 += v * v;
    } else {
      const float u = (seg.f1[i] - seg.f0[i]) / sc;
      a0 += u * u;
    }
  }
  const float t0 = block_sum_256(a0, red);
  const float t1 = block_sum_256(a1, red);
  if (threadIdx.x == 0) {      // (agent scope = write-through: k_init_norms_ctl's last workgroup reads them inside the same launch)
    __hip_atomic_store(&partial[blockIdx.x * 2], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partial[blockIdx.x * 2 + 1], t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
void launch_init_norms(const InitSeg* segs, float* const* partial, int nseg, float rtol, float atol, int phase, hipStream_t s) {
  InitSegs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < nseg; ++i) { a.seg[i] = segs[i]; a.partial[i] = partial[i]; }
  hipLaunchKernelGGL(k_init_norms, dim3(ERR_BLOCKS, nseg), dim3(256), 0, s, a, rtol, atol, phase);
}

// global-norm mode: this rank's sums of the coming decision -> gbuf[8] (see NormPackArgs; the caller's hook adds the ranks')
__global__ __launch_bounds__(256) void k_norm_pack(NormPackArgs a) {
  __shared__ float red[4];
  const Ctrl* c = a.ctrl;
  float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const float* p = a.partial[sgi];
    if (a.mode == 0) {
      out[sgi] = reduce_partials_512(p, red);
    } else {
      float v0 = p[threadIdx.x * 2] + p[(threadIdx.x + 256) * 2];
      float v1 = p[threadIdx.x * 2 + 1] + p[(threadIdx.x + 256) * 2 + 1];
      v0 = block_sum_256(v0, red);
      v1 = block_sum_256(v1, red);
      if (a.mode == 1) { out[2 * sgi] = v0; out[2 * sgi + 1] = v1; }
      else out[sgi] = v0;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  if (a.has_scalar) {
    if (a.mode == 0) {        // the scalar segment's squared error ratio, as step_controller_decide forms it
      const float dtf = (float)c->dt;
      float e = (dtf * c_CERR[0]) * c->ts_k[0];
      float s = (dtf * c_CSOL[0]) * c->ts_k[0];
#pragma unroll
      for (int j = 2; j < 7; ++j) { e += (dtf * c_CERR[j]) * c->ts_k[j]; if (j < 6) s += (dtf * c_CSOL[j]) * c->ts_k[j]; }
      const float y1 = c->ts_cur + s;
      const float r = e / (a.atol + a.rtol * fmaxf(fabsf(c->ts_cur), fabsf(y1)));
      out[3] = r * r;
    } else {
      const float sc = a.atol + fabsf(c->ts_cur) * a.rtol;
      if (a.mode == 1) { const float d0 = c->ts_cur / sc, d1 = c->ts_k[0] / sc; out[6] = d0 * d0; out[7] = d1 * d1; }
      else { const float d2 = (c->ts_k[1] - c->ts_k[0]) / sc; out[3] = d2 * d2; }
    }
  }
  if (a.mode == 0 && a.w4sc != nullptr && __hip_atomic_load(&a.w4sc->ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) out[4] = 1.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a.gbuf[i] = out[i];
}
void launch_norm_pack(const NormPackArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_norm_pack, dim3(1), dim3(256), 0, s, a); }

__global__ __launch_bounds__(256) void k_init_controller(InitCtlArgs a) {
  __shared__ float red[4];
  __shared__ float sums[3][2];
  if (a.gbuf != nullptr) {      // global-norm mode: the sums of all ranks; the counts are the ranks' together
    if (threadIdx.x == 0) {
      float gs[3][2];
      InitCtlArgs b = a;
      for (int sgi = 0; sgi < a.nseg; ++sgi) {
        gs[sgi][0] = a.phase == 0 ? a.gbuf[2 * sgi] : a.gbuf[sgi];
        gs[sgi][1] = a.phase == 0 ? a.gbuf[2 * sgi + 1] : 0.f;
        b.numel[sgi] = a.numel[sgi] * (double)a.gworld;
      }
      init_controller_decide(b, gs);
    }
    return;
  }
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const float* p = a.partial[sgi];
    float v0 = p[threadIdx.x * 2] + p[(threadIdx.x + 256) * 2];
    float v1 = p[threadIdx.x * 2 + 1] + p[(threadIdx.x + 256) * 2 + 1];
    v0 = block_sum_256(v0, red);
    v1 = block_sum_256(v1, red);
    if (threadIdx.x == 0) { sums[sgi][0] = v0; sums[sgi][1] = v1; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  init_controller_decide(a, sums);
}
void launch_init_controller(const InitCtlArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_controller, dim3(1), dim3(256), 0, s, a);
}
// norms + decision as ONE launch: the last-arriving workgroup is the controller (the hand-off of k_error_norm_ctl)
__global__ __launch_bounds__(256) void k_init_norms_ctl(InitSegs a, InitCtlArgs ctl, unsigned* arrive) {
  __shared__ float red[4];
  __shared__ float sums[3][2];
  __shared__ int s_last;
  init_norms_body(a, ctl.rtol, ctl.atol, ctl.phase, red);
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int sgi = 0; sgi < ctl.nseg; ++sgi) {
    const float* p = ctl.partial[sgi];
    float v0 = __hip_atomic_load(p + threadIdx.x * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
               __hip_atomic_load(p + (threadIdx.x + 256) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v1 = __hip_atomic_load(p + threadIdx.x * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
               __hip_atomic_load(p + (threadIdx.x + 256) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v0 = block_sum_256(v0, red);
    v1 = block_sum_256(v1, red);
    if (threadIdx.x == 0) { sums[sgi][0] = v0; sums[sgi][1] = v1; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  init_controller_decide(ctl, sums);
}
void launch_init_norms_ctl(const InitSeg* segs, float* const* partial, int nseg, const InitCtlArgs& ctl, unsigned* arrive, hipStream_t s) {
  InitSegs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < nseg; ++i) { a.seg[i] = segs[i]; a.partial[i] = partial[i]; }
  hipLaunchKernelGGL(k_init_norms_ctl, dim3(ERR_BLOCKS, nseg), dim3(256), 0, s, a, ctl, arrive);
}

__global__ void k_set_ctrl(Ctrl* c, double t, double dt, int reset) {
  c->t = t;
  c->dt = dt;
  c->t_prev = t;
  c->dt_used = 0.0;
  c->done = 0; c->step_idx = 0; c->j = 0; c->j0 = 0; c->j1 = 0; c->first_dt = 0.0;
  if (reset) {
    c->accept = 0; c->status = 0; c->n_acc = 0; c->n_rej = 0; c->h0 = 0.f; c->d0 = 0.f; c->d1 = 0.f;
    for (int i = 0; i < 4; ++i) c->ratio[i] = 0.f;
    c->ts_cur = 0.f; c->ts_new = 0.f; c->ts_y0_prev = 0.f; c->ts_f0_prev = 0.f;
    for (int i = 0; i < 7; ++i) c->ts_k[i] = 0.f;
  }
}
void launch_set_ctrl(Ctrl* ctrl, double t, double dt, int reset, hipStream_t s) {
  hipLaunchKernelGGL(k_set_ctrl, dim3(1), dim3(1), 0, s, ctrl, t, dt, reset);
}
// which: 0 ts_cur = v ; 1 rk4 end-of-step: ts_cur += dt*(k0+3k1+3k2+k3)/8 ; 2 ts_k[0] = ts_k[6]
__global__ void k_set_scalar_state(Ctrl* c, float v, int which) {
  if (which == 0) c->ts_cur = v;
  else if (which == 1) {
    const float dtf = (float)c->dt;
    c->ts_cur = c->ts_cur + (c->ts_k[0] + 3.f * c->ts_k[1] + 3.f * c->ts_k[2] + c->ts_k[3]) * (dtf * 0.125f);
  } else if (which == 2) c->ts_k[0] = c->ts_k[6];
}
void launch_set_scalar_state(Ctrl* ctrl, float v, int which, hipStream_t s) {
  hipLaunchKernelGGL(k_set_scalar_state, dim3(1), dim3(1), 0, s, ctrl, v, which);
}

// ============================================================================
// Dense output: quartic through (y0, y1, y_mid, f0, f1) of the last accepted step
//   (`_interp_fit_dopri5` + `_interp_evaluate`, power form like upstream)
// ============================================================================
// ----------------------------------------------------------------------------
// Device-resident stepping: what the host used to do between two steps after reading `Ctrl` back.
// ----------------------------------------------------------------------------
// Dense output of the forward solve for the targets [j0, j1) the finished step passed, written straight into the
// caller's NCHW trajectory: workgroup = (64 channels x 64 pixels of one sample) through an LDS tile, reads
// coalesced along the channels (NHWC state), writes coalesced along the pixels.
__global__ __launch_bounds__(256) void k_emit_outputs(EmitArgs a, Dims d) {
  const Ctrl* c = a.ctrl;
  const int j0 = c->j0, j1 = c->j1;
  if (j1 <= j0) return;
  __shared__ float tile[64][65];
  const int n = blockIdx.z, c0 = blockIdx.x * 64, p0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  const float dt = (float)c->dt_used;
  const float t0f = (float)c->t_prev, t1f = (float)c->t;
  const size_t sample = (size_t)n * d.HW * d.C;
  for (int j = j0; j < j1; ++j) {
    const float x = ((float)a.targets[j] - t0f) / (t1f - t0f);   // upstream rounds t0, t1, t to the state dtype first
    for (int i = ty; i < 64; i += 4) {
      const int p = p0 + i, ch = c0 + tx;
      if (p < d.HW && ch < d.C) {
        const size_t idx = sample + (size_t)p * d.C + ch;
        float kk[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) kk[q] = (q == 1) ? 0.f : a.k[q][idx];
        tile[i][tx] = interp_one(a.y0[idx], a.y1[idx], kk, dt, x);
      }
    }
    __syncthreads();
    float* out = a.y_out + (size_t)j * d.N * d.C * d.HW + sample;
    for (int i = ty; i < 64; i += 4) {
      const int ch = c0 + i, p = p0 + tx;
      if (p < d.HW && ch < d.C) out[(size_t)ch * d.HW + p] = tile[tx][i];
    }
    __syncthreads();
  }
}
void launch_emit_outputs(const Dims& d, const EmitArgs& a, hipStream_t s) {
  dim3 grid((d.C + 63) / 64, (d.HW + 63) / 64, d.N);
  hipLaunchKernelGGL(k_emit_outputs, grid, dim3(256), 0, s, a, d);
}

// The same for a FLAT state (the generic solver, node_flat_*: dynamics evaluated by the caller, no layout): out[j][i]
__global__ __launch_bounds__(256) void k_emit_flat(EmitArgs a, size_t n) {
  const Ctrl* c = a.ctrl;
  const int j0 = c->j0, j1 = c->j1;
  if (j1 <= j0) return;
  const float dt = (float)c->dt_used;
  const float t0f = (float)c->t_prev, t1f = (float)c->t;
  const size_t stride = (size_t)gridDim.x * 256;
  for (int j = j0; j < j1; ++j) {
    const float x = ((float)a.targets[j] - t0f) / (t1f - t0f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      float kk[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) kk[q] = (q == 1) ? 0.f : a.k[q][i];
      a.y_out[(size_t)j * n + i] = interp_one(a.y0[i], a.y1[i], kk, dt, x);
    }
  }
}
void launch_emit_flat(const EmitArgs& a, size_t n, hipStream_t s) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_emit_flat, dim3((unsigned)blocks), dim3(256), 0, s, a, n);
}
// the time an evaluation of the caller's dynamics happens at, as a device float (the caller hands it to its function)
__global__ void k_flat_time(EvalTime et, float* out) { *out = eval_time(et); }
void launch_flat_time(const EvalTime& et, float* out, hipStream_t s) { hipLaunchKernelGGL(k_flat_time, dim3(1), dim3(1), 0, s, et, out); }
// the scalar segment of a flat state lives in the controller: which = -1 its value, 0..6 its stage derivatives
__global__ void k_flat_scalar(Ctrl* c, int which, const float* src, float scale, int accumulate) {
  const float v = scale * src[0];
  if (which < 0) c->ts_cur = (accumulate ? c->ts_cur : 0.f) + v;
  else c->ts_k[which] = (accumulate ? c->ts_k[which] : 0.f) + v;
}
void launch_flat_scalar(Ctrl* ctrl, int which, const float* src, float scale, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(k_flat_scalar, dim3(1), dim3(1), 0, s, ctrl, which, src, scale, accumulate);
}

// Accepted step, interval not finished: y <- y1, k0 <- k6 (FSAL) for every tensor segment.  Augmented solve at the
// end of its interval: every segment <- dense output at the interval's end time, in place (element-wise).
__global__ __launch_bounds__(256) void k_commit(CommitArgs a) {
  const Ctrl* c = a.ctrl;
  if (!c->accept || c->step_idx == 0) return;
  const bool fin = c->done != 0;
  if (fin && !(a.interp_final && c->j1 > c->j0 && c->status == 0)) return;
  // (a step enqueued past the end: the controller left j0 == j1, so nothing happens here either)
  const size_t stride = (size_t)gridDim.x * 256, start = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (!fin) {
    for (int sg = 0; sg < a.nseg; ++sg) {
      const size_t n4 = a.n[sg] >> 2;
      float4* y = reinterpret_cast<float4*>(a.y[sg]);
      const float4* y1 = reinterpret_cast<const float4*>(a.y1[sg]);
      float4* k0 = reinterpret_cast<float4*>(a.k0[sg]);
      const float4* k6 = reinterpret_cast<const float4*>(a.k6[sg]);
      for (size_t i = start; i < n4; i += stride) { y[i] = y1[i]; k0[i] = k6[i]; }
      for (size_t i = (n4 << 2) + start; i < a.n[sg]; i += stride) { a.y[sg][i] = a.y1[sg][i]; a.k0[sg][i] = a.k6[sg][i]; }
    }
    return;
  }
  const float dt = (float)c->dt_used;
  const float t0f = (float)c->t_prev, t1f = (float)c->t;
  const float x = ((float)a.targets[0] - t0f) / (t1f - t0f);
  for (int sg = 0; sg < a.nseg; ++sg)
    for (size_t i = start; i < a.n[sg]; i += stride) {
      float kk[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) kk[q] = (q == 1) ? 0.f : a.k[sg][q][i];
      a.y[sg][i] = interp_one(a.y[sg][i], a.y1[sg][i], kk, dt, x);
    }
}
void launch_commit(const CommitArgs& a, hipStream_t s) {
  size_t nmax = 0;
  for (int i = 0; i < a.nseg; ++i) nmax = a.n[i] > nmax ? a.n[i] : nmax;
  size_t blocks = (nmax / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_commit, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

__global__ void k_set_target(double* targets, double t_end) { targets[0] = t_end; }
void launch_set_target(double* targets, double t_end, hipStream_t s) {
  hipLaunchKernelGGL(k_set_target, dim3(1), dim3(1), 0, s, targets, t_end);
}
// Deferred completion: what the host would have read back, left in the caller's device record; a miss (the steps
// enqueued did not finish the interval, or it stopped with a status) also bumps the caller's flag, on which the
// optimizer step is predicated.
__global__ void k_export_record(const Ctrl* c, node_step_record* r, float* miss_flag, int expect) {
  const int miss = !(c->done && c->status == 0 && c->step_idx <= expect);   // (steps past the end did nothing)
  r->done = c->done; r->status = c->status; r->steps = c->step_idx; r->accepted = c->n_acc; r->rejected = c->n_rej;
  r->miss = miss; r->t = c->t; r->dt = c->dt; r->first_dt = c->first_dt; r->t_prev = c->t_prev; r->dt_used = c->dt_used;
  if (miss && miss_flag != nullptr) *miss_flag += 1.f;
}
void launch_export_record(const Ctrl* ctrl, node_step_record* rec, float* miss_flag, int expect_steps, hipStream_t s) {
  hipLaunchKernelGGL(k_export_record, dim3(1), dim3(1), 0, s, ctrl, rec, miss_flag, expect_steps);
}
__global__ void k_set_interval(Ctrl* c, double t, double dt) {
  c->t = t; c->dt = dt; c->t_prev = t; c->dt_used = 0.0;
  c->done = 0; c->step_idx = 0; c->j = 0; c->j0 = 0; c->j1 = 0; c->first_dt = 0.0; c->accept = 0;
}
void launch_set_interval(Ctrl* ctrl, double t, double dt, hipStream_t s) {
  hipLaunchKernelGGL(k_set_interval, dim3(1), dim3(1), 0, s, ctrl, t, dt);
}

__global__ __launch_bounds__(256) void k_axpy(float* y, const float* x, float alpha, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] += alpha * x[i];
}
void launch_axpy(float* y, const float* x, float alpha, size_t n, hipStream_t s) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_axpy, dim3((unsigned)blocks), dim3(256), 0, s, y, x, alpha, n);
}
__global__ __launch_bounds__(256) void k_scatter_axpy(ScatterArgs a) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t n4 = a.n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(a.src)[i];
    for (int q = 0; q < a.nt; ++q) {
      float4* d = reinterpret_cast<float4*>(a.dst[q]) + i;
      float4 o = *d;
      const float c = a.coef[q];
      o.x += c * v.x; o.y += c * v.y; o.z += c * v.z; o.w += c * v.w;
      *d = o;
    }
  }
  if (blockIdx.x == 0)
    for (size_t i = (n4 << 2) + threadIdx.x; i < a.n; i += 256)
      for (int q = 0; q < a.nt; ++q) a.dst[q][i] += a.coef[q] * a.src[i];
}
void launch_scatter_axpy(const ScatterArgs& a, hipStream_t s) {
  if (a.nt <= 0 || a.n == 0) return;
  size_t blocks = (a.n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_scatter_axpy, dim3((unsigned)blocks), dim3(256), 0, s, a);
}
__global__ __launch_bounds__(256) void k_fill(float* p, float v, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
void launch_fill(float* p, float v, size_t n, hipStream_t s) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) return;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, p, v, n);
}

// adj_t <- adj_t - <f_i, g_i>     (adjoint: "effect of moving the current time measurement point")
__global__ __launch_bounds__(256) void k_dot_partial(const float* a, const float* b, size_t n, float* partial) {
  __shared__ float red[4];
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += a[i] * b[i];
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_dot_final(Ctrl* c, const float* partial, float sign, float* out_dot) {
  __shared__ float red[4];
  const float tot = sign * reduce_partials_512(partial, red);
  if (threadIdx.x == 0) {
    c->ts_cur = c->ts_cur - tot;
    if (out_dot) *out_dot = tot;
  }
}
void launch_dot_sub_scalar(Ctrl* ctrl, const float* a, const float* b, size_t n, float sign, float* partial, float* out_dot, hipStream_t s) {
  hipLaunchKernelGGL(k_dot_partial, dim3(ERR_BLOCKS), dim3(256), 0, s, a, b, n, partial);
  hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(256), 0, s, ctrl, partial, sign, out_dot);
}
__global__ void k_copy_scalar_out(const Ctrl* c, float* dst) { *dst = c->ts_cur; }
void launch_copy_scalar_out(const Ctrl* ctrl, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_scalar_out, dim3(1), dim3(1), 0, s, ctrl, dst);
}

// ============================================================================
// theta-segment stage derivative: reduce the split-K / per-tile partials written
// by the wgrad GEMM and the GroupNorm-backward epilogues into one vector in the
// internal theta layout, times osign (= tsign, the reverse-time negation).
// ============================================================================
// ONE launch (a kernel on this box costs >= 5 us however little it does, and this used to be three):
//   bulk of the vector (the last 2 x wb workgroups): the two conv-weight blocks, out[r] = osign * sum_sp wpart[sp][r], as
//     float4 with four slabs in flight per thread (the split-K slabs are 16 x 2.36 MB at C = 256: HBM-bound);
//   the small pieces (the first 5 x nsm workgroups; 26 C values): GroupNorm affine gradients (jobs 0..2), time-channel taps
//     and conv biases (jobs 3, 4) -- column sums of short matrices ([rows][2C] per-tile GroupNorm partials,
//     [N][9C] per-sample masked dz sums); one 64-column chunk, 64 columns x 4 row groups, per workgroup;
//   vjp_t = sum_layers sum_{tap,co} W[co][0][tap] * S[tap][co]  (d conv / d t = time-channel border map): each
//     conv-job workgroup leaves the dot product of its 64 columns in a.sred's tail, the last one to arrive
//     (agent-scope fences around a device counter) adds them in a fixed order -- deterministic.
__global__ __launch_bounds__(256) void k_theta_finalize(ThetaFinalizeArgs a, Dims d) {
  if (a.ctrl->done) return;
  __shared__ float red[256];
  __shared__ int s_last;
  const ThetaLayout L = theta_layout(d.C);
  const int C = d.C;
  // linear grid: [5 jobs x nsm column chunks of the small pieces][2 layers x wb blocks of the bulk sums] -- the small
  // jobs first, so that their arrival chain overlaps the bulk; no workgroup is launched just to exit
  const int nsm = (9 * C + 63) / 64;
  const int bx_all = blockIdx.x;
  const int wb = ((int)gridDim.x - 5 * nsm) / 2;
  if (bx_all >= 5 * nsm) {
    const int layer = (bx_all - 5 * nsm) / wb, bx = (bx_all - 5 * nsm) - layer * wb;
    const size_t CC = (size_t)C * C;
    const size_t n4 = 9 * CC / 4;   // C % 4 == 0
    const float4* wp = reinterpret_cast<const float4*>(a.wpart[layer]);
    float4* out = reinterpret_cast<float4*>(a.theta_out + L.wc[layer]);   // 16-B aligned: every block size is a multiple of C
    const size_t stride = (size_t)wb * 256;
    if (a.dU != nullptr) {   // F(4x4,3x3)-domain gradients, every element written once by k_w4_wgrad: dW = G^T dU G
      // (measured, round 6: one (ci, co) pair per thread instead of four -- 2 x 256 instead of 2 x 64 workgroups with work -- 15.8 against
      //  13.0 us: the launch is one round of 36 loads per thread either way, and 4-byte loads are four times the requests)
      const size_t cc4 = CC / 4;
      const float4* du = reinterpret_cast<const float4*>(a.dU + (size_t)layer * W4_COMPS * CC);
      for (size_t i = (size_t)bx * 256 + threadIdx.x; i < cc4; i += stride) {
        float4 tq[6][3];   // t[xi][kw] = sum_nu dU[xi][nu] G[nu][kw]
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          float4 u[6];
#pragma unroll
          for (int nu = 0; nu < 6; ++nu) u[nu] = du[(size_t)(xi * 6 + nu) * cc4 + i];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
              const float gq = (float)W4_G[nu][kw];
              if (gq != 0.f) { s.x += gq * u[nu].x; s.y += gq * u[nu].y; s.z += gq * u[nu].z; s.w += gq * u[nu].w; }
            }
            tq[xi][kw] = s;
          }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
              const float gq = (float)W4_G[xi][kh];
              if (gq != 0.f) { s.x += gq * tq[xi][kw].x; s.y += gq * tq[xi][kw].y; s.z += gq * tq[xi][kw].z; s.w += gq * tq[xi][kw].w; }
            }
            out[(size_t)(kh * 3 + kw) * cc4 + i] = make_float4(a.osign * s.x, a.osign * s.y, a.osign * s.z, a.osign * s.w);
          }
      }
      return;
    }
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n4; i += stride) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int sp = 0;
      for (; sp + 4 <= d.nsplit; sp += 4) {
        const float4 v0 = wp[(size_t)sp * n4 + i], v1 = wp[(size_t)(sp + 1) * n4 + i];
        const float4 v2 = wp[(size_t)(sp + 2) * n4 + i], v3 = wp[(size_t)(sp + 3) * n4 + i];
        acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
        acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; sp < d.nsplit; ++sp) {
        const float4 v = wp[(size_t)sp * n4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      out[i] = make_float4(a.osign * acc.x, a.osign * acc.y, a.osign * acc.z, a.osign * acc.w);
    }
    return;
  }
  const int job = bx_all / nsm, bxs = bx_all - job * nsm;
  const bool gn = job < 3;
  const int layer = gn ? job : job - 3;
  const int ncol = gn ? 2 * C : 9 * C;
  if (bxs * 64 >= ncol) return;
  const int rows = gn ? a.gpart_rows[layer] : (a.spart_rows > 0 ? a.spart_rows : d.N);
  const float* src = gn ? a.gpart[layer] : a.spart[layer];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = bxs * 64 + cl;
  float v = 0.f;
  if (col < ncol) {
    int r = rg;
    for (; r + 12 < rows; r += 16) {
      const float v0 = src[(size_t)r * ncol + col], v1 = src[(size_t)(r + 4) * ncol + col];
      const float v2 = src[(size_t)(r + 8) * ncol + col], v3 = src[(size_t)(r + 12) * ncol + col];
      v += (v0 + v1) + (v2 + v3);
    }
    for (; r < rows; r += 4) v += src[(size_t)r * ncol + col];
  }
  red[threadIdx.x] = v;
  __syncthreads();
  if (rg != 0) return;          // the first wave finishes the job
  const bool on = col < ncol;
  v = (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
  if (gn) {   // columns [0, C) = dgamma, [C, 2C) = dbeta
    if (on) {
      const int which = col >= C ? 1 : 0;
      a.theta_out[(which ? L.b[layer] : L.g[layer]) + (col - which * C)] = a.osign * v;
    }
    return;
  }
  // column = tap * C + co
  if (on) {
    a.theta_out[L.wt[layer] + col] = a.osign * (v * eval_time(a.et));  // time-channel taps: t * masked sums
    if (col >= 4 * C && col < 5 * C) a.theta_out[L.cb[layer] + (col - 4 * C)] = a.osign * v;  // conv bias = centre tap
  }
  const int nb = (9 * C + 63) / 64;                    // conv-job workgroups per layer
  float* dotpart = a.sred + (size_t)2 * 9 * C;         // [2][nb] partial dot products, then the arrival counter
  unsigned* counter = reinterpret_cast<unsigned*>(dotpart + 2 * nb);
  const float part = wave_sum(on ? v * a.wtime[layer][col] : 0.f);   // wtime: [tap][co], gathered once per solve (k_wtime)
  // Hand-off without fences: an agent-scope release here would write back this XCD's whole L2, which is full of
  // the wgrad slabs' dirty lines (measured: +12 us on the launch).  The partial is an agent-scope (write-through)
  // store, acknowledged (vmcnt(0)) before the agent-scope count; the last arriver reads with agent-scope loads.
  // This leans on gfx950's memory system (agent-scope stores write through the XCD's L2; s_waitcnt vmcnt(0) waits
  // for the write acknowledgement), not on the HIP memory model -- so it is tied to the architecture at compile
  // time, and tests/test_gpu_parity.py::test_vjp_t_is_deterministic_over_repeated_launches watches it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_theta_finalize's fence-free hand-off is only valid on gfx950: use release/acquire on the counter elsewhere"
#endif
  if (cl == 0) {
    __hip_atomic_store(dotpart + layer * nb + bxs, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(2 * nb - 1);
  }
  __builtin_amdgcn_wave_barrier();
  if (!s_last) return;                                  // (one wave: LDS write above is visible after the wave barrier)
  float tot = 0.f;
  for (int i = cl; i < 2 * nb; i += 64) tot += __hip_atomic_load(dotpart + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tot = wave_sum(tot);
  if (cl == 0) {
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream order)
    if (a.write_scalar) a.ctrl->ts_k[a.kidx] = a.osign * tot;
    if (a.vjp_t_out) *a.vjp_t_out = a.osign * tot;
  }
}

void launch_theta_finalize(const Dims& d, const ThetaFinalizeArgs& a, hipStream_t s) {
  size_t wblocks = (9 * (size_t)d.C * d.C / 4 + 255) / 256;
  if (a.dU != nullptr) wblocks = ((size_t)d.C * d.C / 4 + 255) / 256;      // F(4x4,3x3)-domain gradients: one float4 of (ci, co) pairs per thread
  if (wblocks > 1024) wblocks = 1024;
  const size_t nsmall = (9 * (size_t)d.C + 63) / 64;
  hipLaunchKernelGGL(k_theta_finalize, dim3((unsigned)(5 * nsmall + 2 * wblocks)), dim3(256), 0, s, a, d);
}

// out = y + scale * sum_j coef_j k_j   (flat; fixed-grid solver's end-of-step update)
__global__ __launch_bounds__(256) void k_lincomb(Comb c, const Ctrl* ctrl, float* out, size_t n) {
  const float scale = comb_scale(c, ctrl);
  float cf[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) cf[j] = scale * c.coef[j];
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int j = 0; j < c.nk; ++j) s += cf[j] * c.k[j][i];
    out[i] = c.y[i] + s;
  }
}
void launch_lincomb(const Comb& c, const Ctrl* ctrl, float* out, size_t n, hipStream_t s) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_lincomb, dim3((unsigned)blocks), dim3(256), 0, s, c, ctrl, out, n);
}

}  // namespace node
