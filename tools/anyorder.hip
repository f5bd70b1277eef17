// Does hipExtAnyOrderLaunch let a kernel start beside its predecessor in the SAME stream on this stack (gfx950, ROCm 7.2)?  hip_ext.h says the
// flag "is not supported on AMD GFX9xx boards".  Pairs of a 64-workgroup kernel that spins ~20 us: 200 pairs in order, then 200 pairs whose
// second launch carries the flag.  In order: ~2 x 20 us per pair; overlapped: ~20.
//   hipcc --offload-arch=gfx950 -O3 tools/anyorder.hip -o tools/anyorder && tools/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int* sink; hipMalloc(&sink, 4); hipMemset(sink, 0, 4);
  const long long ticks = 2000;      // 100 MHz constant clock: 20 us
  for (int mode = 0; mode < 2; ++mode) {
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, st, ticks, sink);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) {
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, st, ticks, sink);
      if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, st, ticks, sink);
      else hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
    }
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f us per pair (%s)\n", mode ? "second launch with hipExtAnyOrderLaunch" : "in order", ms * 1e3 / 200, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
