// One-way latency of a tagged word between two workgroups of one launch, by memory scope and by placement (same XCD / another XCD):
// what a hand-off inside a resident kernel costs on gfx950.   hipcc --offload-arch=gfx950 -O3 tools/pingpong.hip -o tools/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SCOPE>
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// workgroup `a` and workgroup `b` bounce a counter `iters` times; everybody else leaves.  flags[0]: a -> b, flags[64]: b -> a
template <int SCOPE>
__global__ void k_pingpong(unsigned* flags, int a, int b, int iters, long long* out, unsigned* xcc) {
  if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  unsigned* mine = flags + ((int)blockIdx.x == a ? 0 : 64);
  const unsigned* theirs = flags + ((int)blockIdx.x == a ? 64 : 0);
  const long long t0 = wall_clock64();
  long long guard = 0;
  for (int i = 1; i <= iters; ++i) {
    if ((int)blockIdx.x == a) {
      st<SCOPE>(mine, (unsigned)i);
      while (ld<SCOPE>(theirs) < (unsigned)i) { if (++guard > 400000000LL) return; }
    } else {
      while (ld<SCOPE>(theirs) < (unsigned)i) { if (++guard > 400000000LL) return; }
      st<SCOPE>(mine, (unsigned)i);
    }
  }
  if ((int)blockIdx.x == a) out[0] = wall_clock64() - t0;
}

// fan-in: `nprod` producers each store `words` tagged 8-byte words; one consumer polls ONE word per producer, then reads everything
// once.  Reports the time from "go" (consumer's own store) to "all data read", per round.
__global__ void k_fanin(unsigned long long* data, unsigned* go, int nprod, int words_per_lane, int iters, long long* out) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  if (wg > nprod) return;
  if (wg == 0) {   // consumer
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
      if (tid == 0) __hip_atomic_store(go, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid < nprod) {
        const unsigned long long* w = data + ((size_t)tid * 256 + 255) * words_per_lane + (words_per_lane - 1);
        long long guard = 0;
        while ((unsigned)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != (unsigned)it) { if (++guard > 100000000LL) break; }
      }
      __syncthreads();
      unsigned long long acc = 0;
      for (int p = 0; p < nprod; ++p)
        for (int i = 0; i < words_per_lane; ++i) {
          unsigned long long v;
          long long guard = 0;
          do { v = __hip_atomic_load(data + ((size_t)p * 256 + tid) * words_per_lane + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(v >> 32) != (unsigned)it && ++guard < 100000000LL);
          acc += v;
        }
      if (acc == 0x1234567ull) out[1] = 1;
      __syncthreads();
    }
    if (tid == 0) out[0] = wall_clock64() - t0;
    return;
  }
  const int p = wg - 1;
  for (int it = 1; it <= iters; ++it) {
    if (tid == 0) { long long guard = 0; while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) { if (++guard > 100000000LL) break; } }
    __syncthreads();
    for (int i = 0; i < words_per_lane; ++i)
      __hip_atomic_store(data + ((size_t)p * 256 + tid) * words_per_lane + i, ((unsigned long long)it << 32) | (unsigned)(tid + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  unsigned *flags, *xcc, *go;
  long long* out;
  unsigned long long* data;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&xcc, 4096)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&go, 256));
  CK(hipMalloc(&data, (size_t)16 * 256 * 8 * 8));
  const int iters = 2000;
  unsigned hx[64];
  // (workgroup scope, sc0: the polling load keeps hitting its own L1 -- the loop never ends; not run)
  for (int scope = 0; scope < 3; scope += 2) {
    for (int partner : {1, 8, 4}) {
      CK(hipMemset(flags, 0, 4096)); CK(hipMemset(out, 0, 64));
      if (scope == 0) hipLaunchKernelGGL(k_pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(16), dim3(64), 0, 0, flags, 0, partner, iters, out, xcc);
      if (scope == 1) hipLaunchKernelGGL(k_pingpong<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(16), dim3(64), 0, 0, flags, 0, partner, iters, out, xcc);
      if (scope == 2) hipLaunchKernelGGL(k_pingpong<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(16), dim3(64), 0, 0, flags, 0, partner, iters, out, xcc);
      CK(hipDeviceSynchronize());
      long long t = 0;
      CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost));
      printf("scope %s  wg 0 (xcc %u) <-> wg %d (xcc %u): one way %.0f ns%s\n", scope == 0 ? "agent    " : scope == 1 ? "workgroup" : "system   ", hx[0], partner,
             hx[partner], t * 10.0 / (2.0 * iters), t == 0 ? "  (did not finish)" : "");
    }
  }
  for (int nprod : {1, 7})
    for (int wpl : {1, 4}) {
      CK(hipMemset(data, 0, (size_t)16 * 256 * 8 * 8)); CK(hipMemset(go, 0, 256)); CK(hipMemset(out, 0, 64));
      hipLaunchKernelGGL(k_fanin, dim3(nprod + 1), dim3(256), 0, 0, data, go, nprod, wpl, 500, out);
      CK(hipDeviceSynchronize());
      long long t = 0;
      CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
      printf("fan-in: %2d producers x 256 lanes x %d words: go -> all read %.0f ns per round\n", nprod, wpl, t * 10.0 / 500);
    }
  return 0;
}
