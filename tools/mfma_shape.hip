// Which bf16 MFMA shape should the component GEMM's 64 x 64 wave tile use?  Same tile, same six part products per K = 32 step:
//   32x32x16: 4 accumulator tiles x 2 K halves x 6 = 48 MFMAs (16 passes each)     16x16x32: 16 tiles x 6 = 96 MFMAs (8 passes each)
// operands in registers (no memory), optionally NV VALU instructions per step standing in for the in-register split of V.  Whole chip
// (256 workgroups x 4 waves, one wave per SIMD), short (~20 us, a launch of k_w4_gemm64b) and long runs: sustained bf16 TFLOP/s and the
// shader clock the chip holds (s_memtime / s_memrealtime).     hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o tools/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHAPE, int NV>
__global__ __launch_bounds__(256) void k_shape(float* out, unsigned long long* clk, int steps) {
  const int tid = threadIdx.x;
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int e = 0; e < 8; ++e) { a[p][e] = (__bf16)(0.001f * (tid + e + p)); b[p][e] = (__bf16)(0.002f * (tid - e + p)); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.01f * (tid + i);
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc32[t][r] = 0.f;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) acc16[t][r] = 0.f;
  unsigned long long c0, c1, r0, r1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int s = 0; s < steps; ++s) {
    if (SHAPE == 32) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc32[t], 0, 0, 0);
          acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc32[t], 0, 0, 0);
          acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc32[t], 0, 0, 0);
          acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc32[t], 0, 0, 0);
          acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc32[t], 0, 0, 0);
          acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc32[t], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc16[t], 0, 0, 0);
        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc16[t], 0, 0, 0);
        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc16[t], 0, 0, 0);
        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc16[t], 0, 0, 0);
        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc16[t], 0, 0, 0);
        acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc16[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0001f + v[(i + 3) & 7];
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float sum = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc32[t][r];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) sum += acc16[t][r];
  for (int i = 0; i < 8; ++i) sum += v[i];
  out[blockIdx.x * 256 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int NV>
int run(float* out, unsigned long long* clk, int steps, const char* name) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_shape<SHAPE, NV>), dim3(256), dim3(256), 0, 0, out, clk, steps);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((k_shape<SHAPE, NV>), dim3(256), dim3(256), 0, 0, out, clk, steps);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[512];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
  const double us_in = real / 256 / 100.0, ghz = cyc / real / 10.0;      // s_memrealtime: 100 MHz
  const double flops = 256.0 * 4 * steps * 64.0 * 64 * 32 * 2 * 6;       // per launch: six part products of a 64 x 64 x 32 step per wave
  printf("%-10s VALU/step %2d  steps %5d: %7.2f us in the loop, %7.2f us per launch by events, %6.0f bf16 TFLOP/s in the loop, clock held %.2f GHz\n", name, NV, steps,
         us_in, ms * 1e3 / reps, flops / (us_in * 1e-6) / 1e12, ghz);
  return 0;
}

int main() {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 512 * 8));
  for (int steps : {8, 64, 1024}) {       // k_w4_gemm64b at C = 256 is 8 steps of K = 32 per tile (+ the shared component)
    run<32, 0>(out, clk, steps, "32x32x16");
    run<16, 0>(out, clk, steps, "16x16x32");
    run<32, 32>(out, clk, steps, "32x32x16");
    run<16, 32>(out, clk, steps, "16x16x32");
  }
  return 0;
}
