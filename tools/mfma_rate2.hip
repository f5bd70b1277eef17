// mfma_rate2 -- one wave per SIMD, nine independent 32x32 accumulators (k_wgrad_p's shape): cycles per
// MFMA with nothing else, with interleaved ds_read_b32 operand reads, and with per-step address VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return; } } while (0)

template <int FLAGS>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, int iters, int W, int W2, int npx, int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int tid = threadIdx.x;
  for (int i = tid; i < 24576; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    lds[i] = (FLAGS & 16) ? ((float)(h & 0xffff) / 32768.f - 1.f) : 0.001f * (i & 1023);
  }
  int* atab = reinterpret_cast<int*>(lds + 24576);
  for (int i = tid; i < 80; i += blockDim.x) atab[i] = (11 + (i / 8 + 1) * 10 + (i % 8) + 1) * 64;
  __syncthreads();
  float a0[9], a1[9], b0, b1;
  for (int t = 0; t < 9; ++t) { a0[t] = lds[tid + t * 64]; a1[t] = a0[t]; }
  b0 = lds[tid + 4096]; b1 = b0;
  int px = tid >> 5 & 1, prow = 0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define SB __builtin_amdgcn_sched_barrier(0)
  const int hi = (tid >> 5) & 1, abase = (tid & 31) + ((tid >> 7) & 1) * 32, bbase = (tid & 31) + ((tid >> 6) & 1) * 32;
  int toff[9];
  for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * W2 + (t % 3 - 1)) * 64;
  int slot_nxt = atab[2 + hi];
  const float* As = lds; const float* Zs = lds + 8192 + 4096;
#define TSTEP(CUR, BCUR, NXT, BNXT, KP)                                                \
  {                                                                                    \
    const int p1 = 2 * ((KP) + 1) + hi;                                                \
    const bool ok = p1 < npx;                                                          \
    const int aoff = abase + ((FLAGS & 32) ? (((KP) + 1) & 31) * 128 + hi * 64 + 21 * 64 : (ok ? slot_nxt : 11 * 64)); \
    const int zoff = (ok ? p1 : 64) * 64 + bbase;                                      \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                     \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(CUR[t], BCUR, acc[t], 0, 0, 0);    \
      if (t == 0) { if (!(FLAGS & 32)) slot_nxt = atab[min(p1 + 2, 67)]; BNXT = Zs[zoff]; } \
      NXT[t] = (FLAGS & 64) ? As[aoff + (t / 3 - 1) * 640 + (t % 3 - 1) * 64] : As[aoff + toff[t]]; \
      SB;                                                                              \
    }                                                                                  \
  }
#define STEP(CUR, BCUR, NXT, BNXT)                                                     \
  {                                                                                    \
    int aoff = tid & 63;                                                               \
    if (FLAGS & 2) { px += 2; while (px >= W) { px -= W; ++prow; } aoff += ((prow & 7) * 10 + px) * 64; } \
    SB;                                                                                \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                     \
      if (FLAGS & 4) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(CUR[t]), "v"(BCUR)); \
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(CUR[t], BCUR, acc[t], 0, 0, 0); \
      if (FLAGS & 1) { NXT[t] = lds[aoff + t * 64]; if (t == 8) BNXT = lds[aoff + 8192]; } \
      SB;                                                                              \
    }                                                                                  \
  }
  if (FLAGS & 8) {
    for (int it = 0; it < iters / 16; ++it) {
      int kp = 0;
      for (; kp + 1 < npairs; kp += 2) {
        TSTEP(a0, b0, a1, b1, kp)
        TSTEP(a1, b1, a0, b0, kp + 1)
      }
    }
  } else {
  for (int it = 0; it < iters; ++it) {
    STEP(a0, b0, a1, b1)
    STEP(a1, b1, a0, b0)
  }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int FLAGS>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  const int blocks = 256, iters = 500, threads = 256;
  CK(hipMalloc(&out, blocks * threads * sizeof(float)));
  CK(hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long)));
  CK(hipFuncSetAttribute((const void*)k_rate<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<FLAGS>), dim3(blocks), dim3(threads), 100 << 10, 0, out, cyc, iters, 8, 10, 64, 32);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_rate<FLAGS>), dim3(blocks), dim3(threads), 100 << 10, 0, out, cyc, iters, 8, 10, 64, 32);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * 4);
  CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
  const double tf = (double)blocks * 4 * iters * 18.0 * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-40s cycles per MFMA %.1f   wall %.1f TF\n", name, avg / (iters * 18.0), tf);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<0>("1w x 9 acc: mfma only");
  run<1>("1w x 9 acc: + interleaved ds_read_b32");
  run<3>("1w x 9 acc: + reads + address VALU");
  run<4>("1w x 9 acc in AGPRs: mfma only");
  run<5>("1w x 9 acc in AGPRs: + reads");
  run<8>("1w x 9 acc: wgrad-like table-driven step");
  run<24>("1w x 9 acc: wgrad-like step, random data");
  run<8 + 32>("wgrad-like: no slot table (VALU-only address)");
  run<8 + 64>("wgrad-like: immediate tap offsets");
  run<8 + 32 + 64>("wgrad-like: no table + immediate taps");
  run<16>("1w x 9 acc: mfma only, random data");
  run<17>("1w x 9 acc: + reads, random data");
  return 0;
}
