// mfma_rate3 -- two waves per SIMD (512 threads): does VALU work issued by one wave steal matrix-pipe time from
// its SIMD partner?  Per chunk every wave issues 32 MFMAs (four independent accumulators); waves 0-3 ("stagers")
// additionally issue NV VALU instructions between the two halves; optional barrier per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return; } } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)

// MODE bit0: barrier per chunk; bit1: stagers skip their MFMAs; bit2: partners skip their MFMAs;
// bit3: VALU in all waves (symmetric staging); bit4: dependent VALU chain (else 8 independent chains)
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k_rate(float* out, unsigned long long* cyc, int chunks) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int tid = threadIdx.x;
  const bool stager = tid < 256;
  float a = 0.001f * tid, b = 0.002f * tid;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.01f * (tid + i);
  const bool do_mfma = stager ? !(MODE & 2) : !(MODE & 4);
  const bool do_valu = stager || (MODE & 8);
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int c = 0; c < chunks; ++c) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
    }
    SB;
    if (do_valu) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (MODE & 32) {          // DPP quad permutes (the input transform's cross-row exchange)
          v[i & 7] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i & 7]), 0x5A, 0xf, 0xf, true) ^ (int)(i * 77));
        } else if (MODE & 64) {   // 16-byte LDS writes
          *reinterpret_cast<float4*>(lds + ((tid * 4 + (i & 7) * 2048) & 16383)) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (MODE & 128) {  // packed fp32 adds
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 x = {v[(2 * i) & 6], v[((2 * i) & 6) + 1]}, y = {0.5f, 0.25f};
          x = x + y;
          v[(2 * i) & 6] = x.x; v[((2 * i) & 6) + 1] = x.y;
        } else if (MODE & 16) v[0] = __builtin_fmaf(v[0], 1.0001f, 0.5f);
        else v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, 0.5f);
      }
    }
    SB;
    if (MODE & 1) __builtin_amdgcn_s_barrier();
    SB;
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i & 3], 0, 0, 0);
    }
    SB;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int MODE, int NV>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  const int blocks = 256, chunks = 200, threads = 512;
  CK(hipMalloc(&out, blocks * threads * sizeof(float)));
  CK(hipMalloc(&cyc, blocks * 8 * sizeof(unsigned long long)));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_rate<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, chunks);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * 8);
  CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double st = 0, pa = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? st : pa) += (double)h[b * 8 + w];
  st /= blocks * 4.0 * chunks; pa /= blocks * 4.0 * chunks;
  printf("%-62s NV=%3d  cycles/chunk: stagers %7.0f  partners %7.0f\n", name, NV, st, pa);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<0, 0>("both MFMA, no VALU, no barrier (floor 4096)");
  run<1, 0>("both MFMA, no VALU, barrier");
  run<4, 0>("stagers MFMA only (partners idle; floor 2048)");
  run<2, 200>("stagers VALU only, partners MFMA only (floor 2048)");
  run<2 + 16, 200>("stagers dependent VALU only, partners MFMA only");
  run<0, 200>("stagers MFMA+VALU, partners MFMA, no barrier");
  run<1, 200>("stagers MFMA+VALU, partners MFMA, barrier");
  run<1 + 16, 200>("stagers MFMA+dependent VALU, partners MFMA, barrier");
  run<1 + 8, 100>("all waves MFMA+VALU (symmetric), barrier");
  run<1, 400>("stagers MFMA+VALU, partners MFMA, barrier");
  run<1, 100>("stagers MFMA+VALU, partners MFMA, barrier");
  run<1 + 8, 50>("all waves MFMA + 50 fma, barrier");
  run<1 + 8 + 32, 50>("all waves MFMA + 50 DPP-xor, barrier");
  run<1 + 8 + 128, 50>("all waves MFMA + 50 packed adds, barrier");
  run<1 + 8 + 64, 8>("all waves MFMA + 8 ds_write_b128, barrier");
  run<1 + 32, 100>("stagers MFMA + 100 DPP-xor, partners MFMA, barrier");
  run<2 + 32, 200>("stagers DPP-xor only, partners MFMA only");
  run<2 + 64, 32>("stagers ds_write_b128 only, partners MFMA only");
  return 0;
}
