// What does the memory system of one MI355X sustain for the component GEMM's traffic SHAPE -- 28.3 MB read + 18.9 MB written per launch
// (cfg 2: V pairs + filter pairs in, M out), nothing computed?  One launch = one "GEMM" of pure traffic; buffers either rotate through
// more than the 256 MB of MALL (cold) or stay the same (warm: what follows a producer in the step).  Reported per launch, HIP events around
// a run of launches and the average kernel duration from the events of single launches.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_stream.hip -o tools/hbm_stream && tools/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_stream(const u4v* __restrict__ in, size_t nin, u4v* __restrict__ out, size_t nout) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  u4v acc = {0, 0, 0, 0};
  size_t i = tid;
  for (; i + (INFLIGHT - 1) * nth < nin; i += INFLIGHT * nth) {
    u4v v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) v[k] = __builtin_nontemporal_load(in + i + k * nth);
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) acc ^= v[k];
  }
  for (; i < nin; i += nth) acc ^= in[i];
  for (size_t o = tid; o < nout; o += nth) __builtin_nontemporal_store(acc, out + o);
}

int main() {
  const size_t rbytes = (size_t)36 * 512 * 256 * 4 + (size_t)36 * 256 * 256 * 4, wbytes = (size_t)36 * 512 * 256 * 4;
  const int NSET = 8;                                     // 8 x 47 MB = 378 MB > MALL
  std::vector<u4v*> in(NSET), out(NSET);
  for (int s = 0; s < NSET; ++s) {
    CK(hipMalloc(&in[s], rbytes)); CK(hipMalloc(&out[s], wbytes));
    CK(hipMemset(in[s], 1, rbytes)); CK(hipMemset(out[s], 0, wbytes));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {256, 512, 1024, 2048, 4096};
  printf("traffic per launch: %.1f MB read + %.1f MB written\n", rbytes / 1e6, wbytes / 1e6);
  for (int warm = 0; warm < 2; ++warm)
    for (int g : grids) {
      const int reps = 200;
      for (int r = 0; r < 16; ++r) { const int s = warm ? 0 : r % NSET; hipLaunchKernelGGL(k_stream<8>, dim3(g), dim3(256), 0, st, in[s], rbytes / 16, out[s], wbytes / 16); }
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) { const int s = warm ? 0 : r % NSET; hipLaunchKernelGGL(k_stream<8>, dim3(g), dim3(256), 0, st, in[s], rbytes / 16, out[s], wbytes / 16); }
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("%s  %4d workgroups   %.2f us per launch   %.2f TB/s\n", warm ? "warm (same buffers)   " : "cold (rotating 378 MB)", g, us, (rbytes + wbytes) / us / 1e6);
    }
  return 0;
}
