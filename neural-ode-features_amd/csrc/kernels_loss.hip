// The last two operations of the training step's forward pass and the first two of its backward -- the classifier's
// Linear layer (reference model.py:244-250, `nn.Linear(in_ch, out)` behind Flatten) and the loss of the training loop
// (train.py:43 `F.cross_entropy(p, y)`, train.py:44,46 the running loss / accuracy sums) -- as ONE launch each way
// (SURVEY.md 8f rank 2).  On PyTorch-ROCm they are 3 hipBLASLt GEMMs + ~10 ATen launches (log_softmax, nll_loss, their
// backwards, a bias reduction, fills) per step: the only foreign kernels that were left in the timed window, and the
// places where the host fell behind the GPU (profiles/r04_cfg2_steps.txt, "idle by pair").
//
//   k_head_loss_fwd:  logits[n][o] = sum_c pooled[n][c] W[o][c] + b[o]                (skipped when logits are given)
//                     loss = reduce_n (logsumexp_o logits[n] - logits[n][target[n]])   (skipped when no target is given)
//                     stat = {loss, number of samples whose arg-max class is the target}
//   k_head_loss_bwd:  dlogits[n][o] = g (softmax(logits[n])[o] - [o == target[n]]) (/ N for the mean)   (or given)
//                     dW[o][c] = sum_n dlogits[n][o] pooled[n][c];  db[o] = sum_n dlogits[n][o];
//                     dpooled[n][c] = sum_o dlogits[n][o] W[o][c]
//
// All sums in a fixed order (no float atomics): bit-reproducible from launch to launch.  The work is tiny (128 x 10 x 256
// multiply-adds at cfg 2): one wave per sample forward, one workgroup per block of 16 input channels backward.
#include "node_internal.h"
#include "../../include/node_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdio>

namespace node {
int set_error(int code, const char* msg);

namespace {

struct LossArgs {
  const float* pooled;      // [N][C] or nullptr (logits given)
  const float* weight;      // [O][C]
  const float* bias;        // [O] or nullptr
  const long long* target;  // [N] or nullptr
  float* logits;            // [N][O]
  float* loss;              // [1]
  float* stat;              // [2] or nullptr
  float* scratch;           // arrival counter (word 0; a FIXED place: launches of different batch sizes share the scratch), then [2 * nwg] partials from word 4
  int N, C, O;
  int sum_reduction;
  int OL, G, NCH;           // class lanes per group (power of two <= 64), channel groups (64 / OL, or 1), 64-class chunks
  int w_lds;                // W^T staged in LDS
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
  return v;
}

constexpr int LOSS_SPW = 4;   // samples per workgroup: one per wave

// LDS: [wt: G * ((C / G) + 1) * OL * NCH floats (when w_lds)] [prow: 4 waves x (C + G)] [lrow: 4 waves x NCH * 64] [red: 16]
__global__ __launch_bounds__(256) void k_head_loss_fwd(const LossArgs a) {
  extern __shared__ __align__(16) float lsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, O = a.O, OL = a.OL, G = a.G, NCH = a.NCH;
  const int CG = C / G, OW = OL * NCH;            // channels per group, class slots per channel row
  float* wt = lsm;
  float* prow = wt + (a.w_lds ? (size_t)G * (CG + 1) * OW : 0) + (size_t)wave * (C + G);
  float* lrow = lsm + (a.w_lds ? (size_t)G * (CG + 1) * OW : 0) + (size_t)LOSS_SPW * (C + G) + (size_t)wave * (NCH * 64);
  float* red = lsm + (a.w_lds ? (size_t)G * (CG + 1) * OW : 0) + (size_t)LOSS_SPW * (C + G) + (size_t)LOSS_SPW * (NCH * 64);   // [9]
  const int n = blockIdx.x * LOSS_SPW + wave;
  const bool live = n < a.N;
  const int g = lane / OL, ol = lane - g * OL;

  if (a.pooled != nullptr) {
    if (a.w_lds) {      // W^T, group-blocked: wt[(g (CG + 1) + cl) OW + slot], slot = chunk * OL + ol
      for (int idx = tid; idx < O * C; idx += 256) {
        const int o = idx / C, c = idx - o * C;
        const int gg = c / CG, cl = c - gg * CG;
        wt[((size_t)gg * (CG + 1) + cl) * OW + o] = a.weight[idx];
      }
    }
    if (live)
      for (int c = lane; c < C; c += 64) prow[c + c / CG] = a.pooled[(size_t)n * C + c];
    __syncthreads();
    if (live) {
      for (int ch = 0; ch < NCH; ++ch) {
        const int o = ch * OL + ol;          // (NCH > 1 only with OL = 64, G = 1)
        float acc = 0.f;
        if (o < O) {
          if (a.w_lds) {
            const float* wr = wt + ((size_t)g * (CG + 1)) * OW + o;
            const float* pr = prow + g * (CG + 1);
#pragma unroll 8
            for (int cl = 0; cl < CG; ++cl) acc = fmaf(pr[cl], wr[(size_t)cl * OW], acc);
          } else {
            const float* wr = a.weight + (size_t)o * C + g * CG;
            const float* pr = prow + g * (CG + 1);
#pragma unroll 8
            for (int cl = 0; cl < CG; ++cl) acc = fmaf(pr[cl], wr[cl], acc);
          }
        }
        for (int s = OL; s < 64; s <<= 1) acc += __shfl_xor(acc, s);     // over the channel groups (fixed order)
        if (o < O) acc += a.bias != nullptr ? a.bias[o] : 0.f;
        lrow[ch * 64 + lane] = (g == 0 && o < O) ? acc : -INFINITY;   // (lanes of the groups g > 0 are the slots >= OL: masked)
        if (g == 0 && o < O) a.logits[(size_t)n * O + o] = acc;
      }
    }
  } else if (live) {
    for (int s = lane; s < NCH * 64; s += 64) {
      const int ch = s >> 6, o = ch * OL + (s & 63);
      lrow[s] = ((s & 63) < OL && o < O) ? a.logits[(size_t)n * O + o] : -INFINITY;
    }
  }
  if (a.target == nullptr) return;
  __syncthreads();

  float loss_n = 0.f, hit = 0.f;
  if (live) {
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int ch = 0; ch < NCH; ++ch) m = fmaxf(m, lrow[ch * 64 + lane]);
    m = wave_max(m);
    float se = 0.f;
    for (int ch = 0; ch < NCH; ++ch) {
      const float v = lrow[ch * 64 + lane];
      se += v == -INFINITY ? 0.f : expf(v - m);
      if (v == m && lane < OL) am = min(am, ch * OL + lane);
    }
    se = wave_add(se);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) am = min(am, __shfl_xor(am, s));     // first class that attains the maximum
    const long long tg = a.target[n];
    if (tg >= 0 && tg < O) {
      const int t = (int)tg;
      const float lt = lrow[(t / OL) * 64 + (t % OL)];
      loss_n = logf(se) + m - lt;
      hit = am == t ? 1.f : 0.f;
    } else {
      loss_n = NAN;       // (no ignore_index: the reference's loaders never produce one)
    }
  }
  if (lane == 0) { red[wave * 2] = loss_n; red[wave * 2 + 1] = hit; }
  __syncthreads();
  // workgroup partial -> scratch; the last workgroup to arrive sums the partials in index order.  Fence-free hand-off
  // as in k_theta_finalize (kernels_pointwise.hip): agent-scope store, acknowledged, then the agent-scope count; the
  // reader uses agent-scope loads.  Tied to gfx950 (agent-scope stores write through the XCD's L2).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_head_loss_fwd's fence-free hand-off is only valid on gfx950"
#endif
  const unsigned nwg = gridDim.x;
  unsigned* counter = reinterpret_cast<unsigned*>(a.scratch);
  float* parts = a.scratch + 4;
  if (tid == 0) {
    const float pl = (red[0] + red[2]) + (red[4] + red[6]), ph = (red[1] + red[3]) + (red[5] + red[7]);
    __hip_atomic_store(parts + 2 * (size_t)blockIdx.x, pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(parts + 2 * (size_t)blockIdx.x + 1, ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    red[8] = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1 ? 1.f : 0.f;
  }
  __syncthreads();
  if (red[8] == 0.f) return;
  float sl = 0.f, sh = 0.f;
  for (unsigned i = tid; i < nwg; i += 256) {
    sl += __hip_atomic_load(parts + 2 * (size_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh += __hip_atomic_load(parts + 2 * (size_t)i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sl = wave_add(sl);
  sh = wave_add(sh);
  __syncthreads();
  if (lane == 0) { red[wave * 2] = sl; red[wave * 2 + 1] = sh; }
  __syncthreads();
  if (tid == 0) {
    const float tl = (red[0] + red[2]) + (red[4] + red[6]), th = (red[1] + red[3]) + (red[5] + red[7]);
    const float out = a.sum_reduction ? tl : tl / (float)a.N;
    *a.loss = out;
    if (a.stat != nullptr) { a.stat[0] = out; a.stat[1] = th; }
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream order)
  }
}

struct LossBwdArgs {
  const float* pooled;       // [N][C] or nullptr (CE alone: only d_logits is produced)
  const float* weight;       // [O][C]
  const float* logits;       // [N][O]
  const long long* target;   // [N]
  const float* grad_loss;    // [1] or nullptr (= 1)
  const float* grad_logits;  // [N][O] or nullptr (computed from logits + target)
  float* d_logits;           // [N][O] or nullptr
  float* d_pooled;           // [N][C]
  float* d_weight;           // [O][C]
  float* d_bias;             // [O] or nullptr
  int N, C, O;
  int sum_reduction;
  int CB;                    // input channels per workgroup (Linear present)
  int NS;                    // samples per LDS chunk
};

// dl[n][o] of samples [n0, n0 + ns) into LDS (row stride O).  The chunk's logits (or the given gradient) come in as ONE coalesced
// copy -- a wave that fetched its samples' rows one after the other paid an L2 round trip per sample (35 us per launch at 128
// samples: the profile of the first version) -- then every wave turns its rows into gradients in place.
__device__ __forceinline__ void loss_dl_chunk(const LossBwdArgs& a, float* dl, int n0, int ns, float scale, int write_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, O = a.O;
  const float* src = (a.grad_logits != nullptr ? a.grad_logits : a.logits) + (size_t)n0 * O;
  for (int idx = threadIdx.x; idx < ns * O; idx += 256) dl[idx] = src[idx];
  __syncthreads();
  if (a.grad_logits != nullptr) return;
  if (O <= 64) {
    // few classes: ONE LANE per sample walks its row in LDS (no cross-lane reduction at all; the wave-per-sample form below spends
    // its time in twelve dependent ds_bpermute hops per sample: 40 us per launch at 128 samples x 10 classes)
    for (int i = threadIdx.x; i < ns; i += 256) {
      float* row = dl + (size_t)i * O;
      float m = -INFINITY;
      for (int o = 0; o < O; ++o) m = fmaxf(m, row[o]);
      float se = 0.f;
      for (int o = 0; o < O; ++o) se += expf(row[o] - m);
      const float inv = 1.f / se;
      const int t = (int)a.target[n0 + i];
      for (int o = 0; o < O; ++o) {
        const float v = scale * (expf(row[o] - m) * inv - (o == t ? 1.f : 0.f));
        row[o] = v;
        if (write_out && a.d_logits != nullptr) a.d_logits[(size_t)(n0 + i) * O + o] = v;
      }
    }
    return;
  }
  // many classes: a wave per sample, the classes over its lanes; the wave's targets (samples wave, wave + 4, ...: <= 128 of
  // them, NS <= 512), one or two per lane, fetched once
  int tg[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = wave + 4 * (lane + 64 * h);
    tg[h] = i < ns ? (int)a.target[n0 + i] : -1;
  }
  for (int i = wave, k = 0; i < ns; i += 4, ++k) {
    float* row = dl + (size_t)i * O;
    float m = -INFINITY;
    for (int o = lane; o < O; o += 64) m = fmaxf(m, row[o]);
    m = wave_max(m);
    float se = 0.f;
    for (int o = lane; o < O; o += 64) se += expf(row[o] - m);
    se = wave_add(se);
    const float inv = 1.f / se;
    const int t = __shfl(k < 64 ? tg[0] : tg[1], k & 63);
    for (int o = lane; o < O; o += 64) {
      const float v = scale * (expf(row[o] - m) * inv - (o == t ? 1.f : 0.f));
      row[o] = v;
      if (write_out && a.d_logits != nullptr) a.d_logits[(size_t)(n0 + i) * O + o] = v;
    }
  }
}

// LDS: dl [NS][O] | wblk [O][CB] | pl [NS][CB]
__global__ __launch_bounds__(256) void k_head_loss_bwd(const LossBwdArgs a) {
  extern __shared__ __align__(16) float lsm[];
  const int tid = threadIdx.x;
  const int N = a.N, C = a.C, O = a.O, NS = a.NS;
  const float g = a.grad_loss != nullptr ? *a.grad_loss : 1.f;
  const float scale = a.sum_reduction ? g : g / (float)N;
  float* dl = lsm;
  if (a.pooled == nullptr) {      // cross-entropy alone: d_logits, NS samples per workgroup
    const int n0 = blockIdx.x * NS, ns = min(NS, N - n0);
    loss_dl_chunk(a, dl, n0, ns, scale, 1);
    return;
  }
  const int CB = a.CB, c0 = blockIdx.x * CB;
  float* wblk = dl + (size_t)NS * O;            // [O][CB]
  float* pl = wblk + (size_t)O * CB;            // [NS][CB]
  for (int idx = tid; idx < O * CB; idx += 256) wblk[idx] = a.weight[(size_t)(idx / CB) * C + c0 + (idx % CB)];
  // dW: every thread owns the outputs e = o * CB + cc = tid, tid + 256, ... (at most MAXE of them: E <= 4096, checked by
  // the launcher) and walks all samples; db: thread o of workgroup 0 (o = tid + 256 k, O <= 1024)
  const int E = O * CB;
  constexpr int MAXE = 16;
  float accw[MAXE];
#pragma unroll
  for (int k = 0; k < MAXE; ++k) accw[k] = 0.f;
  float accb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = 0; n0 < N; n0 += NS) {
    const int ns = min(NS, N - n0);
    __syncthreads();                             // (the previous chunk's readers are done with dl / pl)
    loss_dl_chunk(a, dl, n0, ns, scale, blockIdx.x == 0);
    for (int idx = tid; idx < ns * CB; idx += 256) pl[idx] = a.pooled[(size_t)(n0 + idx / CB) * C + c0 + (idx % CB)];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      const int e = tid + k * 256;
      if (e < E) {
        const int o = e / CB, cc = e - o * CB;
        float s = accw[k];
#pragma unroll 4
        for (int i = 0; i < ns; ++i) s = fmaf(dl[(size_t)i * O + o], pl[i * CB + cc], s);
        accw[k] = s;
      }
    }
    if (blockIdx.x == 0 && a.d_bias != nullptr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = tid + k * 256;
        if (o < O) {
          float s = accb[k];
          for (int i = 0; i < ns; ++i) s += dl[(size_t)i * O + o];
          accb[k] = s;
        }
      }
    }
    // dpooled[n][c0 + cc] = sum_o dl[n][o] W[o][c0 + cc]
    for (int idx = tid; idx < ns * CB; idx += 256) {
      const int i = idx / CB, cc = idx - i * CB;
      float s = 0.f;
      for (int o = 0; o < O; ++o) s = fmaf(dl[(size_t)i * O + o], wblk[o * CB + cc], s);
      a.d_pooled[(size_t)(n0 + i) * C + c0 + cc] = s;
    }
  }
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    const int e = tid + k * 256;
    if (e < E) a.d_weight[(size_t)(e / CB) * C + c0 + (e % CB)] = accw[k];
  }
  if (blockIdx.x == 0 && a.d_bias != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (tid + k * 256 < O) a.d_bias[tid + k * 256] = accb[k];
  }
}

int bad(int code, const char* msg) { return set_error(code, msg); }

}  // namespace
}  // namespace node

using namespace node;

extern "C" {

size_t node_head_loss_scratch_bytes(int n) {
  if (n <= 0) return 0;
  return ((size_t)2 * ((n + LOSS_SPW - 1) / LOSS_SPW) + 4) * sizeof(float);
}

int node_head_loss_fwd(const node_head_loss* h, void* stream) {
  if (!h) return bad(NODE_ERR_NULL, "node_head_loss is NULL");
  if (h->n <= 0 || h->classes <= 0) return bad(NODE_ERR_SHAPE, "n and classes must be positive");
  if (h->classes > 1024) return bad(NODE_ERR_UNSUPPORTED, "at most 1024 classes");
  if (!h->logits) return bad(NODE_ERR_NULL, "logits is NULL");
  if (h->pooled && (!h->weight || h->c <= 0)) return bad(NODE_ERR_NULL, "pooled given without weight / c");
  if (!h->pooled && !h->target) return bad(NODE_ERR_ARG, "nothing to do: neither pooled (Linear) nor target (loss) given");
  if (h->target && (!h->loss || !h->scratch)) return bad(NODE_ERR_NULL, "target given without loss / scratch");
  LossArgs a;
  a.pooled = h->pooled; a.weight = h->weight; a.bias = h->bias; a.target = reinterpret_cast<const long long*>(h->target);
  a.logits = h->logits; a.loss = h->loss; a.stat = h->stat; a.scratch = h->scratch;
  a.N = h->n; a.C = h->pooled ? h->c : 0; a.O = h->classes; a.sum_reduction = h->reduction == NODE_REDUCE_SUM ? 1 : 0;
  // lane map of a wave: OL class lanes x G channel groups (few classes: the C-long dot product is cut into 64 / OL pieces
  // that meet by shuffles), or all 64 lanes as class slots (NCH chunks of 64 classes, one channel group)
  int OL = 1;
  while (OL < a.O) OL *= 2;
  if (a.pooled && OL <= 32 && a.C % (64 / OL) == 0) { a.OL = OL; a.G = 64 / OL; a.NCH = 1; }
  else { a.OL = 64; a.G = 1; a.NCH = (a.O + 63) / 64; }
  const size_t CG = a.pooled ? (size_t)a.C / a.G : 0, OW = (size_t)a.OL * a.NCH;
  size_t wt_floats = a.pooled ? (size_t)a.G * (CG + 1) * OW : 0;
  a.w_lds = a.pooled && wt_floats * sizeof(float) <= 96 * 1024;
  if (!a.w_lds) wt_floats = 0;
  const size_t lds = (wt_floats + (size_t)LOSS_SPW * (a.C + a.G) + (size_t)LOSS_SPW * a.NCH * 64 + 16) * sizeof(float);
  if (lds > 150 * 1024) return bad(NODE_ERR_UNSUPPORTED, "head loss: in_features too large for one wave's LDS row");
  static bool attr[MAX_DEVICES] = {};
  allow_full_lds(reinterpret_cast<const void*>(k_head_loss_fwd), attr);
  const int grid = (a.N + LOSS_SPW - 1) / LOSS_SPW;
  hipLaunchKernelGGL(k_head_loss_fwd, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return bad(NODE_ERR_HIP, hipGetErrorString(e));
  return NODE_OK;
}

int node_head_loss_bwd(const node_head_loss* h, const node_head_loss_grad* gr, void* stream) {
  if (!h || !gr) return bad(NODE_ERR_NULL, "node_head_loss / node_head_loss_grad is NULL");
  if (h->n <= 0 || h->classes <= 0) return bad(NODE_ERR_SHAPE, "n and classes must be positive");
  if (!gr->grad_logits && (!h->logits || !h->target)) return bad(NODE_ERR_NULL, "logits + target (or grad_logits) required");
  LossBwdArgs a;
  a.pooled = h->pooled; a.weight = h->weight; a.logits = h->logits; a.target = reinterpret_cast<const long long*>(h->target);
  a.grad_loss = gr->grad_loss; a.grad_logits = gr->grad_logits; a.d_logits = gr->d_logits;
  a.d_pooled = gr->d_pooled; a.d_weight = gr->d_weight; a.d_bias = gr->d_bias;
  a.N = h->n; a.C = h->c; a.O = h->classes; a.sum_reduction = h->reduction == NODE_REDUCE_SUM ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  static bool attr[MAX_DEVICES] = {};
  allow_full_lds(reinterpret_cast<const void*>(k_head_loss_bwd), attr);
  if (!h->pooled) {
    if (!gr->d_logits) return bad(NODE_ERR_NULL, "cross-entropy backward without d_logits");
    if (gr->grad_logits) return bad(NODE_ERR_ARG, "grad_logits given and no Linear layer: nothing to compute");
    a.CB = 0;
    a.NS = 8192 / a.O < 1 ? 1 : 8192 / a.O;
    if (a.NS > 64) a.NS = 64;
    const int grid = (a.N + a.NS - 1) / a.NS;
    hipLaunchKernelGGL(k_head_loss_bwd, dim3(grid), dim3(256), (size_t)a.NS * a.O * sizeof(float), st, a);
  } else {
    if (!h->weight || !gr->d_pooled || !gr->d_weight) return bad(NODE_ERR_NULL, "Linear backward needs weight, d_pooled, d_weight");
    a.CB = a.C % 16 == 0 ? 16 : a.C % 4 == 0 ? 4 : 1;
    while (a.CB > 1 && (size_t)a.O * a.CB > 4096) a.CB /= 4;
    if ((size_t)a.O * a.CB > 4096) return bad(NODE_ERR_UNSUPPORTED, "head loss backward: too many classes");
    a.NS = 8192 / a.O < 1 ? 1 : 8192 / a.O;
    if (a.NS > 512) a.NS = 512;
    if (a.NS > a.N) a.NS = a.N;
    const size_t lds = ((size_t)a.NS * a.O + (size_t)a.O * a.CB + (size_t)a.NS * a.CB) * sizeof(float);
    hipLaunchKernelGGL(k_head_loss_bwd, dim3(a.C / a.CB), dim3(256), lds, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return bad(NODE_ERR_HIP, hipGetErrorString(e));
  return NODE_OK;
}

}  // extern "C"
