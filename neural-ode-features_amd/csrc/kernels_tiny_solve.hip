// Latency path, second form: a WHOLE forward dopri5 solve of a tiny state in ONE launch (review item "one launch per evaluation at
// bs = 1", evaluate.py:97-142: every test image solved on its own, ~26 evaluations of model.py:339-348 per image).
//
// kernels_tiny.hip runs an evaluation as two launches and a step as 12 + 5: at [1,256,8,8] a launch is ~11 us of which the
// work is ~3 -- the rest is the kernel boundary, the filter slice fetched again from L2 (6.9 MB per evaluation, the same bytes
// every time) and a chain of dependent round trips.  Here the grid stays resident for the solve:
//   * workgroup (sample n, channel block gp of 16 output channels, slice ks of 32 input channels) keeps ITS slice of BOTH
//     convolutions' filters -- 2 x 27 KB of exact bf16 triples -- in LDS from the first instruction to the last: after the
//     prologue no filter byte moves;
//   * a convolution is: the 32-channel slice of the input block arrives (8 KB), nine K steps of v_mfma_f32_16x16x32_bf16 (six
//     part products each: fp32 accuracy, the split of kernels_w4.hip), the 64 x 16 partial sums go to the block's REDUCER
//     (the workgroup with ks == gp mod KS), which adds the slices in slice order, applies bias + t x time map + GroupNorm
//     (+ ReLU) and publishes the block -- workers of the next convolution wait for exactly the two blocks they read
//     (point-to-point: no grid barrier anywhere in an evaluation);
//   * the reducers hold the solver state of their block -- y, y1, k1..k7: 36 registers per lane -- so the Butcher combines,
//     the error norm, dense output and FSAL touch no memory; the step's decision needs ONE all-reducer exchange of a partial
//     sum per step (two more for the initial step), and every reducer takes the decision redundantly, through the same code
//     as k_step_controller (step_control.h);
//   * hand-off idiom: every word that crosses workgroups travels as an 8-byte {value, tag} pair (one agent-scope 64-bit store /
//     load: single-copy atomic), tag = {28-bit nonce of the solve, 4-bit version of the evaluation}: the consumer polls the
//     DATA and knows each word by its tag -- no flag behind the data, no store acknowledgement before a flag, no counter
//     (the low-latency protocol of the collective libraries; a hand-off is one store and one load that sees it).  Nothing is
//     zeroed between solves (stale pairs carry an older nonce); every wait is a bounded spin -- a deadline on the constant
//     100 MHz clock -- that raises a grid-wide abort word, so the grid drains whatever happens to a neighbour.
//     A slot is rewritten every evaluation with NO consumer acknowledgement, and no version is ever lost (advisor, round 5), because a
//     block's KS slices together read EVERY block of the sample: reducer gp republishes act[1] (version e + 1) only behind the conv-1
//     partial sums e + 1 of all its slices, whose workers read act[0] e + 1 of ALL blocks, which every reducer publishes only behind
//     the conv-2 partial sums e of all ITS slices -- so every worker of the sample has read its act[1] words of version e by then.  The
//     same chain, one convolution later, covers act[0]; a worker rewrites its part[cv] slot only behind activations whose publication
//     needed that slot's previous version consumed.  The one exchange without such a chain -- the step decision's partial sums,
//     which all reducers read -- is double-buffered by exchange parity (errpart).
//   * one launch and one stream synchronisation per solve: the launch splits the model's fp32 filters itself, forms the time
//     channel's border maps, copies y0 into the trajectory and writes the record into the caller's pinned host copy.
// The grid must be co-resident (one workgroup per CU: N x (C/16) x (C/32) <= CUs; 128 workgroups at [1,256,8,8]); a grid that is
// not runs into its deadline, drains, and node_solve_fwd repeats the solve on kernels_tiny.hip.  Measured: DESIGN.md 4.7,
// profiles/r05_latency_bs1.txt (17.0 / 15.2 us per evaluation at [1,256,8,8], tol 1e-3 / 1e-5, host included; kernels_tiny.hip 34.3 / 29.7).
#include "node_internal.h"
#include "step_control.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace node {

typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s_bf16x2 __attribute__((ext_vector_type(2)));
typedef float s_f32x2 __attribute__((ext_vector_type(2)));
typedef float s_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s_u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned TS_VERS = 15;             // versions cycle 0 .. 14 (a slot is rewritten every evaluation: consecutive versions differ); 15 = the solve is over
constexpr int TS_PITCH = 40;                 // bf16 elements per padded-pixel row of an LDS part plane (32 channels + 8)
constexpr long long TS_DEADLINE = 25000000;  // 0.25 s of the 100 MHz constant clock (a solve takes ~1 ms): only a grid that is not whole gets there

// Butcher rows as the host path builds them: (float) of the double coefficient (Solver::make_comb), scaled by (float) dt at use
__device__ constexpr float TS_BETA[6][6] = {
    {(float)(1.0 / 5), 0.f, 0.f, 0.f, 0.f, 0.f},
    {(float)(3.0 / 40), (float)(9.0 / 40), 0.f, 0.f, 0.f, 0.f},
    {(float)(44.0 / 45), (float)(-56.0 / 15), (float)(32.0 / 9), 0.f, 0.f, 0.f},
    {(float)(19372.0 / 6561), (float)(-25360.0 / 2187), (float)(64448.0 / 6561), (float)(-212.0 / 729), 0.f, 0.f},
    {(float)(9017.0 / 3168), (float)(-355.0 / 33), (float)(46732.0 / 5247), (float)(49.0 / 176), (float)(-5103.0 / 18656), 0.f},
    {(float)(35.0 / 384), 0.f, (float)(500.0 / 1113), (float)(125.0 / 192), (float)(-2187.0 / 6784), (float)(11.0 / 84)},
};
__device__ constexpr float TS_ALPHA[6] = {(float)(1.0 / 5), (float)(3.0 / 10), (float)(4.0 / 5), (float)(8.0 / 9), 1.f, 1.f};
// row `r` of the tableau and its node as immediates (indexing the arrays with a run-time row is a load from constant memory: two
// dependent scalar-cache round trips on the critical path of every stage)
__device__ __forceinline__ void ts_row(int r, float (&b)[6], float& alpha) {
#define TS_ROW(R)                                                                  \
  case R:                                                                          \
    b[0] = TS_BETA[R][0]; b[1] = TS_BETA[R][1]; b[2] = TS_BETA[R][2];              \
    b[3] = TS_BETA[R][3]; b[4] = TS_BETA[R][4]; b[5] = TS_BETA[R][5];              \
    alpha = TS_ALPHA[R];                                                           \
    break;
  switch (r) {
    TS_ROW(0) TS_ROW(1) TS_ROW(2) TS_ROW(3) TS_ROW(4)
    default:
    TS_ROW(5)
  }
#undef TS_ROW
}

__device__ __forceinline__ void ts_split8(const float* v, s_u32x4& hh, s_u32x4& mm, s_u32x4& ll) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const s_f32x2 x = {v[2 * i], v[2 * i + 1]};
    const s_bf16x2 h = __builtin_convertvector(x, s_bf16x2);
    const s_f32x2 r = x + (-__builtin_convertvector(h, s_f32x2));
    const s_bf16x2 m = __builtin_convertvector(r, s_bf16x2);
    const s_f32x2 t = r + (-__builtin_convertvector(m, s_f32x2));
    const s_bf16x2 l = __builtin_convertvector(t, s_bf16x2);
    hh[i] = __builtin_bit_cast(unsigned, h);
    mm[i] = __builtin_bit_cast(unsigned, m);
    ll[i] = __builtin_bit_cast(unsigned, l);
  }
}

typedef unsigned long long ts_pair;
__device__ __forceinline__ void ts_put(ts_pair* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((ts_pair)tag << 32) | (ts_pair)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ ts_pair ts_get(const ts_pair* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A wave collects NV tagged words (those with use[i]); returns 0 when every one carries `tag`, 2 when one carries `tag_done`
// (the solve is over), 1 when the grid gave up (abort word == nonce, or this wait ran into its deadline).  Wave-uniform.
template <int NV, bool CHECK_DONE>
__device__ __forceinline__ int ts_gather(const ts_pair* const (&ptr)[NV], const bool (&use)[NV], unsigned tag, unsigned tag_done,
                                         float (&out)[NV], unsigned* abort_word, unsigned nonce, int* diag_rounds = nullptr) {
  long long deadline = 0;
  unsigned it = 0;
  for (;;) {
    bool ok = true, dn = false;
    if (diag_rounds != nullptr && (threadIdx.x & 63) == 0) atomicAdd(diag_rounds, 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      ts_pair w = 0;
      if (use[i]) w = ts_get(ptr[i]);
      const unsigned t = (unsigned)(w >> 32);
      if (use[i]) {
        if (CHECK_DONE && t == tag_done) dn = true;
        else if (t != tag) ok = false;
      }
      out[i] = __uint_as_float((unsigned)w);
    }
    if (CHECK_DONE && __any(dn)) return 2;
    if (__all(ok)) return 0;
    if ((++it & 15u) == 0) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nonce) return 1;
      const long long now = wall_clock64();
      if (deadline == 0) deadline = now + TS_DEADLINE;
      else if (now > deadline) { __hip_atomic_store(abort_word, nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 1; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Gathers read TWO tagged words per request (16 bytes): each 8-byte half validates itself, so a request torn between its halves
// is harmless, and the number of requests -- what an uncached read costs by (measured: ~100 ns per wave-level request whatever its
// width) -- halves.  The compiler has no 16-byte atomic load: inline assembly, every request and the wait inside ONE statement (a
// later use of the results cannot move above the wait).
#define TS_Q "global_load_dwordx4 "
__device__ __forceinline__ void ts_load_quads(const ts_pair* const (&p)[4], s_u32x4 (&o)[4]) {
  asm volatile(TS_Q "%0, %4, off sc1\n\t" TS_Q "%1, %5, off sc1\n\t" TS_Q "%2, %6, off sc1\n\t" TS_Q "%3, %7, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
               : "memory");
}
__device__ __forceinline__ void ts_load_quads(const ts_pair* const (&p)[14], s_u32x4 (&o)[14]) {
  asm volatile(TS_Q "%0, %14, off sc1\n\t" TS_Q "%1, %15, off sc1\n\t" TS_Q "%2, %16, off sc1\n\t" TS_Q "%3, %17, off sc1\n\t"
               TS_Q "%4, %18, off sc1\n\t" TS_Q "%5, %19, off sc1\n\t" TS_Q "%6, %20, off sc1\n\t" TS_Q "%7, %21, off sc1\n\t"
               TS_Q "%8, %22, off sc1\n\t" TS_Q "%9, %23, off sc1\n\t" TS_Q "%10, %24, off sc1\n\t" TS_Q "%11, %25, off sc1\n\t"
               TS_Q "%12, %26, off sc1\n\t" TS_Q "%13, %27, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]),
                 "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]),
                 "v"(p[11]), "v"(p[12]), "v"(p[13])
               : "memory");
}
#undef TS_Q
// two tagged words in one 16-byte store (the consumer reads them the same way); the wait state keeps a later VALU write off the
// data registers while the store still reads them (the compiler's hazard recogniser does not see inline assembly)
__device__ __forceinline__ void ts_put2(ts_pair* p, float v0, float v1, unsigned tag) {
  const s_u32x4 w = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");
}

// A wave collects NQ pairs of tagged words (those with use[i]): ts_gather's contract, two words per request.  out[2 i], out[2 i + 1].
template <int NQ, bool CHECK_DONE>
__device__ __forceinline__ int ts_gather2(const ts_pair* const (&ptr)[NQ], const bool (&use)[NQ], unsigned tag, unsigned tag_done,
                                          float (&out)[2 * NQ], unsigned* abort_word, unsigned nonce, int* diag_rounds = nullptr) {
  long long deadline = 0;
  unsigned it = 0;
  for (;;) {
    bool ok = true, dn = false;
    if (diag_rounds != nullptr && (threadIdx.x & 63) == 0) atomicAdd(diag_rounds, 1);
    s_u32x4 w[NQ];
    ts_load_quads(ptr, w);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if (use[i]) {
        if (CHECK_DONE && (w[i][1] == tag_done || w[i][3] == tag_done)) dn = true;
        else if (w[i][1] != tag || w[i][3] != tag) ok = false;
      }
      out[2 * i] = __uint_as_float(w[i][0]);
      out[2 * i + 1] = __uint_as_float(w[i][2]);
    }
    if (CHECK_DONE && __any(dn)) return 2;
    if (__all(ok)) return 0;
    if ((++it & 15u) == 0) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nonce) return 1;
      const long long now = wall_clock64();
      if (deadline == 0) deadline = now + TS_DEADLINE;
      else if (now > deadline) { __hip_atomic_store(abort_word, nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 1; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Before a workgroup asks for a whole block it watches ONE word of it per producer (lane l of wave 0 <-> producer l): a waiting grid
// that polls whole blocks moves terabytes per second of nothing and delays the very stores it waits for (measured: a hand-off took
// 3 - 5 us that way).  Same return codes as ts_gather; wave 0 only.
template <bool CHECK_DONE>
__device__ __forceinline__ int ts_watch(const ts_pair* word, bool active, unsigned tag, unsigned tag_done, unsigned* abort_word, unsigned nonce) {
  long long deadline = 0;
  unsigned it = 0;
  for (;;) {
    bool ok = true, dn = false;
    if (active) {
      const unsigned t = (unsigned)(ts_get(word) >> 32);
      if (CHECK_DONE && t == tag_done) dn = true;
      else if (t != tag) ok = false;
    }
    if (CHECK_DONE && __any(dn)) return 2;
    if (__all(ok)) return 0;
    if ((++it & 31u) == 0) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nonce) return 1;
      const long long now = wall_clock64();
      if (deadline == 0) deadline = now + TS_DEADLINE;
      else if (now > deadline) { __hip_atomic_store(abort_word, nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 1; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// sum over the lanes that share a GroupNorm group inside a 16-lane row (cpg = 1, 2, 4, 8, 16 consecutive columns), on the DPP path:
// quad permutes, then the half-row / row mirrors (values are uniform inside the smaller group by then, so a mirror is the butterfly)
__device__ __forceinline__ float ts_dpp_add(float v, int ctrl_sel) {
  const int x = __builtin_bit_cast(int, v);
  int y;
  switch (ctrl_sel) {
    case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); break;     // quad_perm [1,0,3,2]
    case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false); break;     // quad_perm [2,3,0,1]
    case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); break;    // row_half_mirror
    default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false); break;   // row_mirror
  }
  return v + __builtin_bit_cast(float, y);
}
// sum over the four 16-lane rows of a wave, result in every lane.  Inline assembly: handed the same value twice, the builtins'
// two results are folded into one by the compiler (hipcc 7.2: `v_add_f32 v1, v1, v1` after the swap)
__device__ __forceinline__ float ts_rows_sum(float x) {
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
  float r = p + q, t = r;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r), "+v"(t));
  return r + t;
}
__device__ __forceinline__ float ts_group_sum(float v, int cpg) {
  if (cpg >= 2) v = ts_dpp_add(v, 0);
  if (cpg >= 4) v = ts_dpp_add(v, 1);
  if (cpg >= 8) v = ts_dpp_add(v, 2);
  if (cpg >= 16) v = ts_dpp_add(v, 3);
  return v;
}

struct TinySolveArgs {
  const float* y0;              // NCHW [N][C][HW]
  float* y_out;                 // NCHW [n_targets][N][C][HW]: slot j <-> target j
  float* y_first;               // NCHW [N][C][HW]: the trajectory's slot of t0 (a copy of y0), nullable
  const float* w[2];            // the convolutions' filters [C][C + 1][3][3] (input channel 0 = time, model.py:321-322), as the model holds them
  const float* bias[2];
  const float* gamma[3];
  const float* beta[3];
  ts_pair* act[2];              // NHWC [N][HW][C] tagged words: the convolutions' inputs, block by block
  ts_pair* part[2];             // [N GP][KS][4 elements][4 waves][64 lanes] tagged partial sums
  ts_pair* errpart;             // [2][R][2]: the reducers' partial sums of a step decision (double-buffered by exchange parity)
  unsigned* abort_word;         // == nonce: some wait of this solve ran into its deadline
  Ctrl* ctrl;                   // out: the final record, in the workspace (what k_export_record reads) ...
  Ctrl* ctrl_host;              // ... and straight into the caller's pinned host record (visible when the launch has completed)
  const double* targets; int n_targets;      // device array, or nullptr: the times ride in the arguments (<= 8 of them)
  double targets_inline[8];
  const double* forced; int n_forced;
  double* dt_log; int dt_log_cap;
  double t0;
  long long max_steps;
  float rtol, atol, tsign, eps;
  unsigned nonce;               // 28 bits, unique per solve of this process (stale words of earlier solves never match)
  int stamps;                   // diagnostics library: print the in-kernel timeline
  int N, C, H, W, cpg, KS, GP;
};

// LDS: Wf [2 convs][27 fragments][64 lanes] 16 B | A [3 parts][PP][TS_PITCH] bf16 | floats: red_a[256] red_b[256] bsum[16] | ints [12] | Ctrl | target times [8]
__global__ __launch_bounds__(256) void k_tiny_solve(const TinySolveArgs a) {
  extern __shared__ __align__(16) unsigned char lsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int C = a.C, H = a.H, W = a.W, HW = H * W, KS = a.KS, GP = a.GP, cpg = a.cpg;
  const int Wp = W + 2, PP = (H + 2) * Wp;
  s_u32x4* Wf = reinterpret_cast<s_u32x4*>(lsm);
  unsigned short* A = reinterpret_cast<unsigned short*>(lsm + 2 * 27 * 64 * 16);
  const size_t plane = (size_t)PP * TS_PITCH;
  float* fl = reinterpret_cast<float*>(lsm + 2 * 27 * 64 * 16 + ((3 * plane * 2 + 15) & ~(size_t)15));
  float* red_a = fl;
  float* red_b = fl + 256;
  float* bsum = fl + 512;                          // [0..3], [8..11]: wave partials of two sums
  int* li = reinterpret_cast<int*>(fl + 528);      // outcome of a wait, per wave: [0..3] activations, [4..7] partial sums, [8..11] agree()
  Ctrl* lc = reinterpret_cast<Ctrl*>(fl + 540);    // (16-byte aligned: fl is, 540 floats = 2160 B)
  double* ltg = reinterpret_cast<double*>(fl + 540 + (sizeof(Ctrl) + 15) / 16 * 4);      // the target times, when they came in the arguments

#ifdef NODE_DIAG
  // in-kernel timeline (diagnostics library only, NODE_TUNE_TINY_STAMPS=1): workgroups 0 (a reducer) and 1 (a worker) stamp the
  // constant 100 MHz clock at the phases of their first 48 convolutions and print the differences when the solve is over
  __shared__ long long stamp[48][10];
  __shared__ int rounds[4];      // gather rounds (summed over the four waves): activations, partial sums; calls of each
  if (tid < 4) rounds[tid] = 0;
#define TS_ROUNDS(k) (a.stamps ? &rounds[k] : nullptr)
  int stamp_row = 0;
#define TS_STAMP(k) do { if (a.stamps && blockIdx.x < 2 && tid == 0 && stamp_row < 48) stamp[stamp_row][k] = wall_clock64(); } while (0)
#define TS_STAMP_NEXT() do { ++stamp_row; } while (0)
#else
#define TS_ROUNDS(k) nullptr
#define TS_STAMP(k) do {} while (0)
#define TS_STAMP_NEXT() do {} while (0)
#endif
  int r = blockIdx.x;
  const int ks = r % KS; r /= KS;
  const int gp = r % GP;
  const int n = r / GP;
  const int R = a.N * GP, rid = n * GP + gp;
  const bool is_red = ks == gp % KS;
  const unsigned nonce = a.nonce, tag_done = (nonce << 4) | TS_VERS;
  auto tag_of = [&](unsigned version) { return (nonce << 4) | (version % TS_VERS); };

  // ---- prologue: this workgroup's filter slices of both convolutions, straight from the model's fp32 tensors, split into exact bf16
  // triples in B-fragment order (lane = 16 kq + col holds W[16 gp + col][32 ks + 8 kq .. + 7][tap]) -> LDS, for the whole solve.
  // Wave w takes the taps w, w + 4, w + 8 (no packing launch, no packed copy in memory)
#pragma unroll
  for (int cv = 0; cv < 2; ++cv) {
    const float* wsrc = a.w[cv] + ((size_t)(gp * 16 + col) * (C + 1) + ks * 32 + kq * 8 + 1) * 9;
    for (int tap = wave; tap < 9; tap += 4) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = wsrc[e * 9 + tap];
      s_u32x4 hh, mm, ll;
      ts_split8(v, hh, mm, ll);
      s_u32x4* dst = Wf + cv * 27 * 64 + tap * 3 * 64 + lane;
      dst[0] = hh; dst[64] = mm; dst[128] = ll;
    }
  }
  // zero the activation planes once: the halo stays zero, the interior is rewritten by every staging
  for (int i = tid; i < (int)(3 * plane / 2); i += 256) reinterpret_cast<unsigned*>(A)[i] = 0u;

  // ---- reducer state (meaningful in reducers only)
  const int c = gp * 16 + col;
  int pix[4];
  bool on[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { pix[i] = wave * 16 + 4 * kq + i; on[i] = pix[i] < HW; }
  float y[4] = {0.f, 0.f, 0.f, 0.f}, y1[4] = {0.f, 0.f, 0.f, 0.f}, k[7][4];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) k[j][i] = 0.f;
  float tm1[4] = {0.f, 0.f, 0.f, 0.f}, tm2[4] = {0.f, 0.f, 0.f, 0.f};
  float b1 = 0.f, b2 = 0.f, g1 = 0.f, e1 = 0.f, g2 = 0.f, e2 = 0.f, g3 = 0.f, e3 = 0.f;
  if (tid < 8) ltg[tid] = a.targets_inline[tid];
  const double* targets = a.targets != nullptr ? a.targets : ltg;
  if (is_red) {
    // the time channel is constant over the image: its convolution is t x (sum of its taps that fall inside the image) -- the border
    // map of k_time_prep, here for this lane's channel and four pixels
    float wt1[9], wt2[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      wt1[tp] = a.w[0][(size_t)c * (C + 1) * 9 + tp];
      wt2[tp] = a.w[1][(size_t)c * (C + 1) * 9 + tp];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (on[i]) {
        y[i] = a.y0[((size_t)n * C + c) * HW + pix[i]];
        if (a.y_first != nullptr) a.y_first[((size_t)n * C + c) * HW + pix[i]] = y[i];
        const int ph = pix[i] / W, px = pix[i] % W;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int hh2 = ph + kh - 1, ww2 = px + kw - 1;
            if (hh2 >= 0 && hh2 < H && ww2 >= 0 && ww2 < W) { s1 += wt1[kh * 3 + kw]; s2 += wt2[kh * 3 + kw]; }
          }
        tm1[i] = s1;
        tm2[i] = s2;
      }
    b1 = a.bias[0][c]; b2 = a.bias[1][c];
    g1 = a.gamma[0][c]; e1 = a.beta[0][c];
    g2 = a.gamma[1][c]; e2 = a.beta[1][c];
    g3 = a.gamma[2][c]; e3 = a.beta[2][c];
    if (tid == 0) {      // k_set_ctrl(reset = 1)
      memset(lc, 0, sizeof(Ctrl));
      lc->t = a.t0; lc->t_prev = a.t0;
      lc->dt = a.forced != nullptr ? a.forced[0] : 0.0;
    }
  }
  __syncthreads();

  const float inv_m = 1.f / (float)(cpg * HW);
  const double numel = (double)a.N * C * HW;

  // GroupNorm over (cpg channels x HW pixels) of this 16-channel block: v <- [relu](((v - mean) rstd) gamma + beta); masked entries stay 0
  // a wave's group total stays in registers (DPP inside the 16-lane row, v_permlane16/32_swap across the rows); the four waves meet
  // in LDS: red_a / red_b [group][wave]
  auto group_norm = [&](float (&v)[4], float gm, float bt, bool relu) {
    const int grp = col / cpg;
    float s = ts_rows_sum(ts_group_sum((v[0] + v[1]) + (v[2] + v[3]), cpg));
    // (no barrier in front: a wave gets here through the second barrier of the previous call, behind which nobody reads red_a; red_b is
    //  written behind the first barrier of this call, behind which nobody reads the previous red_b)
    if (kq == 0 && (col & (cpg - 1)) == 0) red_a[grp * 4 + wave] = s;
    __syncthreads();
    const float4 ta = *reinterpret_cast<const float4*>(red_a + grp * 4);
    const float mean = ((ta.x + ta.y) + (ta.z + ta.w)) * inv_m;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (on[i]) { const float dv = v[i] - mean; q += dv * dv; }
    q = ts_rows_sum(ts_group_sum(q, cpg));
    if (kq == 0 && (col & (cpg - 1)) == 0) red_b[grp * 4 + wave] = q;
    __syncthreads();
    const float4 tb = *reinterpret_cast<const float4*>(red_b + grp * 4);
    const float rstd = 1.0f / sqrtf(((tb.x + tb.y) + (tb.z + tb.w)) * inv_m + a.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = ((v[i] - mean) * rstd) * gm + bt;
      if (relu) o = fmaxf(o, 0.f);
      v[i] = on[i] ? o : 0.f;
    }
  };
  // a block of a convolution's input (64 pixels x 16 channels) -> tagged words; nobody waits for the stores
  auto publish = [&](ts_pair* act, const float (&v)[4], unsigned tag) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (on[i]) ts_put(act + ((size_t)n * HW + pix[i]) * C + c, v[i], tag);
  };
  // the outcome of a wave-level wait, agreed on by the workgroup: 0 go on, 1 the grid gave up, 2 the solve is over
  auto agree = [&](int code) -> int {
    if (lane == 0) li[8 + wave] = code;
    __syncthreads();
    const int c0 = li[8], c1 = li[9], c2 = li[10], c3 = li[11];
    const int any1 = (c0 == 1) | (c1 == 1) | (c2 == 1) | (c3 == 1);
    const int any2 = (c0 == 2) | (c1 == 2) | (c2 == 2) | (c3 == 2);
    __syncthreads();                       // (li[8..11] are free again)
    return any1 ? 1 : (any2 ? 2 : 0);
  };
  // deterministic sum over the workgroup of two values, result in every thread
  auto block_sum2 = [&](float v0, float v1, float& o0, float& o1) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); }
    __syncthreads();
    if (lane == 0) { bsum[wave] = v0; bsum[8 + wave] = v1; }
    __syncthreads();
    o0 = (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]);
    o1 = (bsum[8] + bsum[9]) + (bsum[10] + bsum[11]);
  };
  // all-reducer exchange number `xno` (1, 2, ...): everybody's (p0, p1) summed in reducer order; false = the grid gave up
  unsigned xno = 0;
  auto exchange = [&](float p0, float p1, float& tot0, float& tot1) -> bool {
    ++xno;
    const unsigned xtag = tag_of(xno);
    ts_pair* all = a.errpart + (size_t)(xno & 1u) * R * 2;
    float w0, w1;
    block_sum2(p0, p1, w0, w1);
    if (tid == 0) {
      ts_put(all + 2 * rid, w0, xtag);
      ts_put(all + 2 * rid + 1, w1, xtag);
    }
    const int src = tid < R ? tid : 0;
    const ts_pair* const ptr[2] = {all + 2 * src, all + 2 * src + 1};
    const bool use[2] = {tid < R, tid < R};
    float q[2];
    const int code = agree(ts_gather<2, false>(ptr, use, xtag, 0u, q, a.abort_word, nonce));
    if (code != 0) return false;
    block_sum2(q[0], q[1], tot0, tot1);
    return true;
  };

  // ---- reducer: the first evaluation's input, y0 -> GroupNorm-1 -> ReLU
  enum { PH_F0 = 0, PH_PROBE = 1, PH_STAGE = 2 };
  int phase = PH_F0, stage = 0, slot = 0;
  float tnow = a.tsign * (float)a.t0;
  unsigned ver = 0;
  if (is_red) {
    float v[4] = {y[0], y[1], y[2], y[3]};
    group_norm(v, g1, e1, true);
    publish(a.act[0], v, tag_of(1u));
  }
  bool failed = false;

  for (;;) {
    ++ver;
    const unsigned tag = tag_of(ver);
#pragma unroll 1
    for (int cv = 0; cv < 2; ++cv) {
      // ---- worker: the two 16-channel blocks of this slice arrive as tagged words; stage them as bf16 triples
      TS_STAMP(0);
      {
        // wave w takes the pixels 16 w .. 16 w + 15 of the slice, four per request: lane <-> (pixel, channel pair), so a request is
        // four contiguous 256-byte runs (a first version gave each lane eight consecutive channels: 64 scattered 8-byte reads per
        // request, 16 x the bytes through the fabric, 2 - 4 us per hand-off)
        const ts_pair* src = a.act[cv] + (size_t)n * HW * C + ks * 32;
        const int cp = lane & 15;
        const ts_pair* ptr[4];
        bool use[4];
        int lds_at[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = wave * 16 + 4 * j + (lane >> 4);
          use[j] = p < HW;
          ptr[j] = src + (size_t)(use[j] ? p : 0) * C + 2 * cp;
          lds_at[j] = ((p / W + 1) * Wp + (p % W) + 1) * TS_PITCH + 2 * cp;
        }
        // each wave waits for ITS pixel tile on its own: it watches the tile's last word in each of the two blocks (two 8-byte polls per
        // round), then asks for its 4 KB; the outcome meets the other waves' at the barrier the staging needs anyway
        const int plast = (wave * 16 + 15 < HW ? wave * 16 + 15 : HW - 1);
        int code = ts_watch<true>(src + (size_t)plast * C + 16 * (lane & 1) + 15, lane < 2 && wave * 16 < HW, tag, tag_done, a.abort_word, nonce);
        TS_STAMP(1);
        float f[8];
        if (code == 0) code = ts_gather2<4, true>(ptr, use, tag, tag_done, f, a.abort_word, nonce, TS_ROUNDS(0));
        TS_STAMP(2);
        if (code == 0) {
          s_u32x4 hh, mm, ll;
          ts_split8(f, hh, mm, ll);       // (pair j = the two channels of pixel j: one 32-bit LDS word per part)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (use[j]) {
              *reinterpret_cast<unsigned*>(A + lds_at[j]) = hh[j];
              *reinterpret_cast<unsigned*>(A + plane + lds_at[j]) = mm[j];
              *reinterpret_cast<unsigned*>(A + 2 * plane + lds_at[j]) = ll[j];
            }
        }
        if (lane == 0) li[wave] = code;
      }
      __syncthreads();
      {
        const int c0 = li[0], c1 = li[1], c2 = li[2], c3 = li[3];
        if ((c0 | c1 | c2 | c3) != 0) { failed = c0 == 1 || c1 == 1 || c2 == 1 || c3 == 1; goto finished; }
      }
      TS_STAMP(3);
      // ---- products: wave = pixel tile, nine K steps (tap x 32 channels), six part products each
      s_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      {
        const int p = wave * 16 + col;
        const bool inside = p < HW;
        const unsigned short* abase = A + (size_t)(inside ? (p / W) * Wp + (p % W) : 0) * TS_PITCH + kq * 8;
        const s_u32x4* wf = Wf + cv * 27 * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const unsigned short* ap = inside ? abase + ((tap / 3) * Wp + (tap % 3)) * TS_PITCH : A + kq * 8;     // (outside: the zero halo)
          const s_bf16x8 Ah = __builtin_bit_cast(s_bf16x8, *reinterpret_cast<const s_u32x4*>(ap));
          const s_bf16x8 Am = __builtin_bit_cast(s_bf16x8, *reinterpret_cast<const s_u32x4*>(ap + plane));
          const s_bf16x8 Al = __builtin_bit_cast(s_bf16x8, *reinterpret_cast<const s_u32x4*>(ap + 2 * plane));
          const s_bf16x8 Bh = __builtin_bit_cast(s_bf16x8, wf[(tap * 3 + 0) * 64]);
          const s_bf16x8 Bm = __builtin_bit_cast(s_bf16x8, wf[(tap * 3 + 1) * 64]);
          const s_bf16x8 Bl = __builtin_bit_cast(s_bf16x8, wf[(tap * 3 + 2) * 64]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc, 0, 0, 0);     // smallest products first
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc, 0, 0, 0);
        }
      }
      // accumulator: lane holds output channel c = 16 gp + col, pixels 16 wave + 4 kq + i

      TS_STAMP(4);
      ts_pair* part = a.part[cv] + (size_t)rid * KS * 1024;
      if (!is_red) {
        // ---- worker: tagged partial sums to the block's reducer
        ts_pair* mine = part + (size_t)ks * 1024 + (size_t)(wave * 64 + lane) * 2;       // [slice][half][wave][lane][2]: a request is 1 KB contiguous
        ts_put2(mine, acc[0], acc[1], tag);
        ts_put2(mine + 512, acc[2], acc[3], tag);
        __syncthreads();                   // (the activation planes are free for the next staging)
        TS_STAMP(5);
        TS_STAMP_NEXT();
        continue;
      }
      // ---- reducer: the other slices' partial sums, added in slice order
      if (KS > 1) {
        const ts_pair* ptr[14];
        bool use[14];
#pragma unroll
        for (int q = 0; q < 7; ++q) {          // slot q of the seven OTHER slices <-> slice q (q < ks) or q + 1
          const int sl = q < ks ? q : q + 1;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            use[2 * q + h] = sl < KS;
            ptr[2 * q + h] = part + (size_t)(sl < KS ? sl : 0) * 1024 + h * 512 + (size_t)(wave * 64 + lane) * 2;
          }
        }
        // each wave waits for ITS quarter of the seven other slices: lanes 0 - 7 watch the last word wave `wave` of a slice writes
        int code;
        {
          const int q = lane & 7;
          code = ts_watch<false>(part + (size_t)q * 1024 + 512 + (size_t)(wave * 64 + 63) * 2 + 1, lane < 8 && q < KS && q != ks, tag, 0u, a.abort_word, nonce);
        }
        TS_STAMP(5);
        float got[28];
        if (code == 0) code = ts_gather2<14, false>(ptr, use, tag, 0u, got, a.abort_word, nonce, TS_ROUNDS(1));
        TS_STAMP(6);
        if (lane == 0) li[4 + wave] = code;
        __syncthreads();
        if ((li[4] | li[5] | li[6] | li[7]) != 0) { failed = true; goto finished; }
        // got[4 q + i]: slot q; the sum runs in slice order, this workgroup's own slice in its place (slots past KS: zeros)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = 0.f;
#pragma unroll
          for (int q = 0; q < 7; ++q) {
            if (q == ks) s += acc[i];
            s += use[2 * q] ? got[4 * q + i] : 0.f;
          }
          if (ks == 7) s += acc[i];
          acc[i] = s;
        }
      }
      if (cv == 0) {
        // conv1 epilogue: + bias + t x time map, GroupNorm-2, ReLU -> the second convolution's input
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = on[i] ? acc[i] + b1 + tnow * tm1[i] : 0.f;
        TS_STAMP(8);
        group_norm(v, g2, e2, true);
        TS_STAMP(9);
        publish(a.act[1], v, tag);
        TS_STAMP(7);
        TS_STAMP_NEXT();
        continue;
      }
      // conv2 epilogue: + bias + t x time map, GroupNorm-3, orientation -> k[slot]
      {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = on[i] ? acc[i] + b2 + tnow * tm2[i] : 0.f;
        TS_STAMP(8);
        group_norm(v, g3, e3, false);
        TS_STAMP(9);
#pragma unroll
        for (int j = 0; j < 7; ++j)
          if (j == slot)
#pragma unroll
            for (int i = 0; i < 4; ++i) k[j][i] = v[i] * a.tsign;
      }

      // ---- reducer: what follows this evaluation
      const unsigned next_tag = tag_of(ver + 1);
      bool start_step = false;
      if (phase == PH_F0) {
        if (a.forced != nullptr) {
          start_step = true;
        } else {
          // Hairer initial step, phase 0 (k_init_norms / k_init_controller)
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (on[i]) {
              const float sc = a.atol + fabsf(y[i]) * a.rtol;
              const float u = y[i] / sc, w = k[0][i] / sc;
              s0 += u * u;
              s1 += w * w;
            }
          float t0s, t1s;
          if (!exchange(s0, s1, t0s, t1s)) { failed = true; goto finished; }
          if (tid == 0) {
            InitCtlArgs ic;
            memset(&ic, 0, sizeof(ic));
            ic.ctrl = lc; ic.numel[0] = numel; ic.nseg = 1; ic.phase = 0; ic.rtol = a.rtol; ic.atol = a.atol;
            const float sums[1][2] = {{t0s, t1s}};
            init_controller_decide(ic, sums);
          }
          __syncthreads();
          const float h0 = lc->h0;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = on[i] ? y[i] + (0.f + (h0 * 1.0f) * k[0][i]) : 0.f;
          tnow = a.tsign * ((float)lc->t + h0);
          phase = PH_PROBE; slot = 1;
          group_norm(v, g1, e1, true);
          publish(a.act[0], v, next_tag);
        }
      } else if (phase == PH_PROBE) {
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (on[i]) {
            const float sc = a.atol + fabsf(y[i]) * a.rtol;
            const float u = (k[1][i] - k[0][i]) / sc;
            s0 += u * u;
          }
        float t0s, t1s;
        if (!exchange(s0, 0.f, t0s, t1s)) { failed = true; goto finished; }
        if (tid == 0) {
          InitCtlArgs ic;
          memset(&ic, 0, sizeof(ic));
          ic.ctrl = lc; ic.numel[0] = numel; ic.nseg = 1; ic.phase = 1; ic.rtol = a.rtol; ic.atol = a.atol;
          const float sums[1][2] = {{t0s, t1s}};
          init_controller_decide(ic, sums);
        }
        __syncthreads();
        start_step = true;
      } else if (stage < 5) {
        // the next stage's Butcher combine (the sixth one is y1)
        const float dtf = (float)lc->dt;
        const int nk = stage + 2;
        float brow[6], alpha_next;
        ts_row(stage + 1, brow, alpha_next);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sk = 0.f;
#pragma unroll
          for (int j = 0; j < 6; ++j)
            if (j < nk) sk += (dtf * brow[j]) * k[j][i];
          v[i] = on[i] ? y[i] + sk : 0.f;
        }
        ++stage;
        if (stage == 5) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y1[i] = v[i];
        }
        slot = stage + 1;
        tnow = a.tsign * ((float)lc->t + alpha_next * dtf);
        group_norm(v, g1, e1, true);
        publish(a.act[0], v, next_tag);
      } else {
        // ---- the step's seventh evaluation is in: error norm, decision, dense output, FSAL
        const float dtf = (float)lc->dt;
        float accn = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (on[i]) {
            float e = (dtf * c_CERR[0]) * k[0][i];
#pragma unroll
            for (int j = 2; j < 7; ++j) e += (dtf * c_CERR[j]) * k[j][i];
            const float rr = e / (a.atol + a.rtol * fmaxf(fabsf(y[i]), fabsf(y1[i])));
            accn += rr * rr;
            accn += 0.f * (y[i] + y1[i]);      // (a non-finite state poisons the sum: NODE_ERR_NONFINITE, see k_error_norm)
          }
        float tot, unused;
        if (!exchange(accn, 0.f, tot, unused)) { failed = true; goto finished; }
        if (tid == 0) {
          StepCtlArgs sc;
          memset(&sc, 0, sizeof(sc));
          sc.ctrl = lc; sc.numel[0] = numel; sc.nseg = 1; sc.has_scalar = 0; sc.rtol = a.rtol; sc.atol = a.atol;
          sc.targets = targets; sc.n_targets = a.n_targets;
          sc.forced = a.forced; sc.n_forced = a.n_forced;
          sc.dt_log = rid == 0 ? a.dt_log : nullptr; sc.dt_log_cap = a.dt_log_cap;
          float ratios[4] = {(float)((double)tot / numel), 0.f, 0.f, 0.f};
          step_controller_decide(sc, ratios);
          if (!lc->done && (long long)lc->step_idx >= a.max_steps) { lc->status = NODE_ERR_MAX_STEPS; lc->done = 1; }
        }
        __syncthreads();
        const int j0 = lc->j0, j1 = lc->j1;
        if (j1 > j0) {
          const float dtu = (float)lc->dt_used, t0f = (float)lc->t_prev, t1f = (float)lc->t;
          for (int j = j0; j < j1; ++j) {
            const float x = ((float)targets[j] - t0f) / (t1f - t0f);
            float* out = a.y_out + (((size_t)j * a.N + n) * C + c) * HW;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (on[i]) {
                const float kk[7] = {k[0][i], 0.f, k[2][i], k[3][i], k[4][i], k[5][i], k[6][i]};
                out[pix[i]] = interp_one(y[i], y1[i], kk, dtu, x);
              }
          }
        }
        if (lc->done) goto finished;
        if (lc->accept) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { y[i] = y1[i]; k[0][i] = k[6][i]; }
        }
        start_step = true;
      }
      if (start_step) {
        const float dtf = (float)lc->dt;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = on[i] ? y[i] + (0.f + (dtf * TS_BETA[0][0]) * k[0][i]) : 0.f;
        phase = PH_STAGE; stage = 0; slot = 1;
        tnow = a.tsign * ((float)lc->t + TS_ALPHA[0] * dtf);
        group_norm(v, g1, e1, true);
        publish(a.act[0], v, next_tag);
      }
      TS_STAMP(7);
      TS_STAMP_NEXT();
    }
  }

finished:
#ifdef NODE_DIAG
  if (a.stamps && blockIdx.x < 2 && tid == 0) {
    printf("wg %d: gather rounds (4 waves): activations %d, partial sums %d over %d convolutions\n", (int)blockIdx.x, rounds[0], rounds[1], stamp_row);
    for (int i = 0; i < stamp_row && i < 48; ++i) {
      const long long b = stamp[i][0];
      if (blockIdx.x == 0)
        printf("wg 0 conv %2d: top %7lld | +watch %4lld +gather %4lld +stage %4lld +mfma %4lld +watch-parts %4lld +gather-parts %4lld +sum %4lld +norm %4lld +epilogue %4lld (x10 ns)\n", i,
               b % 10000000, stamp[i][1] - b, stamp[i][2] - b, stamp[i][3] - b, stamp[i][4] - b, stamp[i][5] - b, stamp[i][6] - b, stamp[i][8] - b, stamp[i][9] - b, stamp[i][7] - b);
      else
        printf("wg 1 conv %2d: top %7lld | +watch %4lld +gather %4lld +stage %4lld +mfma %4lld +put %4lld (x10 ns)\n", i,
               b % 10000000, stamp[i][1] - b, stamp[i][2] - b, stamp[i][3] - b, stamp[i][4] - b, stamp[i][5] - b);
    }
  }
#endif
  if (failed) {      // some wait ran into its deadline: the record says so, whoever notices first
    if (tid == 0) {
      a.ctrl->status = NODE_ERR_HIP; a.ctrl->done = 1;
      if (a.ctrl_host != nullptr) { a.ctrl_host->status = NODE_ERR_HIP; a.ctrl_host->done = 1; }
    }
    return;          // (the abort word releases everybody else)
  }
  if (!is_red) return;
  __syncthreads();
  if (rid == 0 && tid == 0) {
    *a.ctrl = *lc;
    if (a.ctrl_host != nullptr) *a.ctrl_host = *lc;
  }
  // the workers of the next evaluation are polling this block: tell them the solve is over
  const float zero[4] = {0.f, 0.f, 0.f, 0.f};
  publish(a.act[0], zero, tag_done);
}

}  // namespace

// Geometry the resident solve takes: 32 | C <= 256 (<= 8 slices per block), GroupNorm groups that tile a 16-channel block
// (a power of two <= 16 channels per group), an image of <= 64 pixels (one pixel tile per wave), a grid that fits the chip
// one workgroup per CU.  NODE_TUNE_TINY_RESIDENT=0 keeps the two-launches-per-evaluation path.
bool tiny_resident_ok(const Dims& d) {
  const char* e = getenv("NODE_TUNE_TINY_RESIDENT");      // (read per solve, not per launch: tests switch it inside one process)
  if (e != nullptr && atoi(e) == 0) return false;
  if (d.C % 32 != 0 || d.C > 256 || d.HW > 64 || d.W > 62) return false;
  if (d.cpg < 1 || d.cpg > 16 || (d.cpg & (d.cpg - 1)) != 0) return false;
  // compute units of the CURRENT device (cached per device: a process may drive several), one workgroup each -- the kernel's LDS
  // leaves room for exactly one per CU, which hipOccupancyMaxActiveBlocksPerMultiprocessor confirms once per device.  (Other streams
  // and processes can still hold CUs when the launch arrives: that is what the bounded waits and the fallback are for.)
  static std::atomic<int> cus_of[MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) { (void)hipGetLastError(); return false; }
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (cus == 0) {
    int n = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = -1; }
    if (n > 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_tiny_solve), 256, 64 * 1024) != hipSuccess ||
                  per_cu < 1)) { (void)hipGetLastError(); n = -1; }
    cus = n > 0 ? n : -1;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  const long grid = (long)d.N * (d.C / 16) * (d.C / 32);
  if (grid > cus) return false;
  if ((size_t)(d.H + 2) * (d.W + 2) > 128) return false;           // (LDS planes of the padded image)
  return true;
}
// tagged words (8 bytes each) of the hand-off buffers: two activation buffers | two partial-sum buffers | the decision exchange
size_t tiny_resident_handoff_words(const Dims& d) {
  return 2 * d.numel + 2 * (size_t)d.N * (d.C / 16) * (d.C / 32) * 1024 + (size_t)2 * d.N * (d.C / 16) * 2 + 16;
}
void launch_tiny_solve(const Dims& d, const TinyResidentArgs& b, hipStream_t s) {
  TinySolveArgs a;
  memset(&a, 0, sizeof(a));
  a.y0 = b.y0; a.y_out = b.y_out;
  a.y_first = b.y_first;
  for (int i = 0; i < 2; ++i) { a.w[i] = b.w[i]; a.bias[i] = b.bias[i]; }
  {
    ts_pair* w = reinterpret_cast<ts_pair*>(b.handoff);
    const size_t parts = (size_t)d.N * (d.C / 16) * (d.C / 32) * 1024;
    a.act[0] = w; a.act[1] = w + d.numel;
    a.part[0] = w + 2 * d.numel; a.part[1] = a.part[0] + parts;
    a.errpart = a.part[1] + parts;
    a.abort_word = reinterpret_cast<unsigned*>(a.errpart + (size_t)2 * d.N * (d.C / 16) * 2);
  }
  a.ctrl = b.ctrl; a.ctrl_host = b.ctrl_host; a.nonce = b.nonce;
  for (int i = 0; i < 8; ++i) a.targets_inline[i] = b.targets_inline[i];
  { const char* e = getenv("NODE_TUNE_TINY_STAMPS"); a.stamps = e ? atoi(e) : 0; }
  for (int i = 0; i < 3; ++i) { a.gamma[i] = b.gamma[i]; a.beta[i] = b.beta[i]; }
  a.targets = b.targets; a.n_targets = b.n_targets; a.forced = b.forced; a.n_forced = b.n_forced;
  a.dt_log = b.dt_log; a.dt_log_cap = b.dt_log_cap; a.t0 = b.t0; a.max_steps = b.max_steps;
  a.rtol = b.rtol; a.atol = b.atol; a.tsign = b.tsign; a.eps = d.eps;
  a.N = d.N; a.C = d.C; a.H = d.H; a.W = d.W; a.cpg = d.cpg; a.KS = d.C / 32; a.GP = d.C / 16;
  const size_t plane = (size_t)(d.H + 2) * (d.W + 2) * TS_PITCH;
  const size_t lds = (size_t)2 * 27 * 64 * 16 + ((3 * plane * 2 + 15) & ~(size_t)15) + 540 * sizeof(float) + (sizeof(Ctrl) + 15) / 16 * 16 + 8 * sizeof(double) + 64;
  static bool attr[MAX_DEVICES] = {};
  allow_full_lds(reinterpret_cast<const void*>(k_tiny_solve), attr);
  int grid = d.N * a.GP * a.KS;
  { const char* e = getenv("NODE_TUNE_TINY_RESIDENT"); if (e != nullptr && atoi(e) == 2 && grid > 1) --grid; }    // test hook: a grid that is NOT whole -- every wait must run into its deadline, the grid must drain
  hipLaunchKernelGGL(k_tiny_solve, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace node
