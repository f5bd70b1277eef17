// What MFMA rate does the chip SUSTAIN on the fp16-pair GEMM's arithmetic -- v_mfma_f32_32x32x16_f16, one wave per SIMD, sixteen 32 x 32
// accumulators per wave (k_w4_gemm256h's 128 x 128 quarter), operands in registers -- when the operands are (a) small structured numbers
// (what tools/mfma_shape.hip multiplies) and (b) random fp16 pairs (h random in [-4, 4), l of random sign and 2^-11 of it: what V and U
// pairs look like)?  Whole chip, 256 workgroups x 4 waves, ~100 us.  Prints time in the loop, fp16 TFLOP/s, the shader clock held
// (s_memtime / s_memrealtime) and the cycles per MFMA; both for 32x32x16 and for the same quarter as 64 accumulators of 16x16x32.     hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int RANDOM>
__global__ __launch_bounds__(256, 1) void k_power(float* out, unsigned long long* clk, int steps) {
  const int tid = threadIdx.x;
  unsigned seed = 12345u + 977u * (blockIdx.x * 256 + tid);
  f16x8 a[4][2], b[4][2];      // four row / column blocks, parts h | l
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      if (RANDOM) {
        const float ha = ((int)(rnd(seed) & 0xffff) - 32768) * (4.f / 32768), hb = ((int)(rnd(seed) & 0xffff) - 32768) * (4.f / 32768);
        a[i][0][e] = (_Float16)ha; a[i][1][e] = (_Float16)(ha * (((int)(rnd(seed) & 0xfff) - 2048) * (1.f / 2048 / 2048)));
        b[i][0][e] = (_Float16)hb; b[i][1][e] = (_Float16)(hb * (((int)(rnd(seed) & 0xfff) - 2048) * (1.f / 2048 / 2048)));
      } else {
        a[i][0][e] = (_Float16)(0.001f * (tid + e + i)); a[i][1][e] = (_Float16)(0.0001f * (tid + e));
        b[i][0][e] = (_Float16)(0.002f * (tid - e + i)); b[i][1][e] = (_Float16)(0.0002f * (tid - e));
      }
    }
  f32x16 acc[4][4];
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.f;
  unsigned long long c0, c1, r0, r1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int s = 0; s < steps; ++s) {
    if (RANDOM < 2) {
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)      // product-major: (l,h) of all tiles, (h,l), (h,h)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r][pr == 0 ? 1 : 0], b[c][pr == 1 ? 1 : 0], acc[r][c], 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)         // RANDOM == 2: random pairs, TILE-major (the three products of a tile back to back)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r][pr == 0 ? 1 : 0], b[c][pr == 1 ? 1 : 0], acc[r][c], 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float sum = 0.f;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) sum += acc[r][c][q];
  out[blockIdx.x * 256 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}


// the same 128 x 128 quarter and the same products as 64 accumulators of v_mfma_f32_16x16x32_f16 (a K = 32 step per instruction: half as
// many steps, 192 MFMAs of 16 cycles per step)
template <int RANDOM>
__global__ __launch_bounds__(256, 1) void k_power16(float* out, unsigned long long* clk, int steps) {
  const int tid = threadIdx.x;
  unsigned seed = 54321u + 977u * (blockIdx.x * 256 + tid);
  f16x8 a[8][2], b[8][2];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) {
      if (RANDOM) {
        const float ha = ((int)(rnd(seed) & 0xffff) - 32768) * (4.f / 32768), hb = ((int)(rnd(seed) & 0xffff) - 32768) * (4.f / 32768);
        a[i][0][e] = (_Float16)ha; a[i][1][e] = (_Float16)(ha * (((int)(rnd(seed) & 0xfff) - 2048) * (1.f / 2048 / 2048)));
        b[i][0][e] = (_Float16)hb; b[i][1][e] = (_Float16)(hb * (((int)(rnd(seed) & 0xfff) - 2048) * (1.f / 2048 / 2048)));
      } else {
        a[i][0][e] = (_Float16)(0.001f * (tid + e + i)); a[i][1][e] = (_Float16)(0.0001f * (tid + e));
        b[i][0][e] = (_Float16)(0.002f * (tid - e + i)); b[i][1][e] = (_Float16)(0.0002f * (tid - e));
      }
    }
  f32x4 acc[8][8];
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) for (int q = 0; q < 4; ++q) acc[r][c][q] = 0.f;
  unsigned long long c0, c1, r0, r1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int s = 0; s < steps; s += 2) {      // one K = 32 step = two of the other kernel's
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r][pr == 0 ? 1 : 0], b[c][pr == 1 ? 1 : 0], acc[r][c], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float sum = 0.f;
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) for (int q = 0; q < 4; ++q) sum += acc[r][c][q];
  out[blockIdx.x * 256 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int RANDOM, int SHAPE>
int run(float* out, unsigned long long* clk, int steps, const char* name) {
  auto launch = [&]() {
    if (SHAPE == 32) hipLaunchKernelGGL((k_power<RANDOM>), dim3(256), dim3(256), 0, 0, out, clk, steps);
    else hipLaunchKernelGGL((k_power16<RANDOM>), dim3(256), dim3(256), 0, 0, out, clk, steps);
  };
  for (int rep = 0; rep < 3; ++rep) launch();
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 20; ++rep) launch();
  CK(hipDeviceSynchronize());
  unsigned long long h[512];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
  const double us_in = real / 256 / 100.0, ghz = cyc / real / 10.0;      // s_memrealtime: 100 MHz
  const double nm = SHAPE == 32 ? 48.0 * steps : 96.0 * steps;           // MFMAs per wave (16x16x32: 192 per two steps)
  const double flops = 256.0 * 4 * nm * (SHAPE == 32 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2);
  printf("%-22s steps %5d: %8.2f us in the loop, %6.0f fp16 TFLOP/s, clock held %.2f GHz, %.1f cycles per MFMA\n", name, steps, us_in,
         flops / (us_in * 1e-6) / 1e12, ghz, cyc / 256 / nm);
  return 0;
}

int main() {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 512 * 8));
  for (int steps : {64, 128, 1024}) {      // k_w4_gemm256h at C = 1024: 64 steps per tile, two tiles per CU and launch
    run<0, 32>(out, clk, steps, "32x32x16 structured");
    run<1, 32>(out, clk, steps, "32x32x16 random pairs");
    run<2, 32>(out, clk, steps, "32x32x16 rnd, tile-major");
    run<0, 16>(out, clk, steps, "16x16x32 structured");
    run<1, 16>(out, clk, steps, "16x16x32 random pairs");
  }
  return 0;
}
