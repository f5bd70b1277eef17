// mfma_rate -- what does one SIMD retire per cycle on v_mfma_f32_32x32x2_f32 when the loop around
// the MFMAs looks more and more like k_conv3x3_p's?  (design input)
//   F_OPS   operands rotate over 8 VGPRs fed by ds_read_b128 (2 reads per 4 MFMAs)
//   F_BAR   one s_barrier per 16 MFMAs
//   F_LDSW  one ds_write_b128 + global_load_dwordx4 per 16 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return; } } while (0)

template <int FLAGS>
__global__ __launch_bounds__(512) void k_rate(float* out, unsigned long long* cyc, const float* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 0.001f * i;
  __syncthreads();
  float4 a0 = *reinterpret_cast<const float4*>(lds + tid * 4), b0 = *reinterpret_cast<const float4*>(lds + 2048 + tid * 4);
  float4 a1 = a0, b1 = b0;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* la = lds + (tid & 63) * 36;
  const float* lb = lds + 4096 + (tid & 63) * 36;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define SB __builtin_amdgcn_sched_barrier(0)
#define M4(A, B)                                                          \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, B.x, acc, 0, 0, 0);     \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, B.y, acc, 0, 0, 0);     \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, B.z, acc, 0, 0, 0);     \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, B.w, acc, 0, 0, 0);
#define LD(A, B, O)                                                       \
  if (FLAGS & 1) { A = *reinterpret_cast<const float4*>(la + (O)); B = *reinterpret_cast<const float4*>(lb + (O)); }
  for (int it = 0; it < iters; ++it) {
    if (FLAGS & 4) g = *reinterpret_cast<const float4*>(src + (size_t)(it & 63) * 2048 + tid * 4);
    LD(a1, b1, 4); SB; M4(a0, b0); SB;
    LD(a0, b0, 8); SB; M4(a1, b1); SB;
    LD(a1, b1, 12); SB; M4(a0, b0); SB;
    LD(a0, b0, 0); SB; M4(a1, b1); SB;
    if (FLAGS & 4) *reinterpret_cast<float4*>(lds + 6144 + (tid & 127) * 4) = g;
    if (FLAGS & 2) __syncthreads();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int FLAGS>
void run(int threads, size_t ldsb, const char* name) {
  float *out, *src; unsigned long long* cyc;
  const int blocks = 256, iters = 1000;
  CK(hipMalloc(&out, blocks * threads * sizeof(float)));
  CK(hipMalloc(&src, 64 * 2048 * sizeof(float)));
  CK(hipMemset(src, 0, 64 * 2048 * sizeof(float)));
  CK(hipMalloc(&cyc, blocks * 16 * sizeof(unsigned long long)));
  CK(hipFuncSetAttribute((const void*)k_rate<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<FLAGS>), dim3(blocks), dim3(threads), ldsb, 0, out, cyc, src, iters);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_rate<FLAGS>), dim3(blocks), dim3(threads), ldsb, 0, out, cyc, src, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = (double)blocks * (threads / 64) * iters * 16.0 * 4096.0 / (ms * 1e-3) / 1e12;
  std::vector<unsigned long long> h(blocks * threads / 64);
  CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
  const double wps = threads / 256.0;
  printf("%-44s waves/SIMD %.0f  LDS %3zu KB  cycles per MFMA per SIMD %.1f   wall %.1f TF (%.3f ms)\n", name, wps, ldsb >> 10, avg / (iters * 16.0 * wps), tf, ms);
  CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(src));
}

int main() {
  run<0>(512, 40 << 10, "2w: mfma only");
  run<0>(512, 150 << 10, "2w: mfma only, 150 KB LDS (1 WG/CU)");
  run<1>(512, 150 << 10, "2w: + ds_read_b128 operands");
  run<3>(512, 150 << 10, "2w: + operands + barrier/16");
  run<7>(512, 150 << 10, "2w: + operands + barrier + stage");
  run<2>(512, 150 << 10, "2w: barrier/16 only");
  run<0>(256, 150 << 10, "1w: mfma only");
  run<1>(256, 150 << 10, "1w: + ds_read_b128 operands");
  run<3>(256, 150 << 10, "1w: + operands + barrier/16");
  run<7>(256, 150 << 10, "1w: + operands + barrier + stage");
  return 0;
}
