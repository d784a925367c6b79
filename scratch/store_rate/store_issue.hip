// Microbenchmark: what one 1-KiB store costs the wave that issues it (cycles per store, per wave), for NW waves of a
// 512-thread workgroup storing back to back, with a 64-bit per-lane address (global_store_dwordx4 v[a:a+1], v[d:d+3], off)
// or a scalar base + 32-bit lane offset (global_store_dwordx4 vOff, v[d:d+3], s[b:b+1]).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool SADDR>
__global__ __launch_bounds__(512, 2) void k(char* dst, long long pitch, int iters, int nw, long long wg_stride, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* base = dst + (long long)blockIdx.x * wg_stride;
  const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  __syncthreads();
  if (wave >= nw) return;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    char* tb = base + (long long)(it & 7) * 1024 * pitch;   // a wave writes 128 rows x 128 B per pass, 8 waves: 1024 rows
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 128 * wave + 8 * s + (lane >> 3);           // 8 rows x 128 B per store
      const unsigned off = (unsigned)(row * pitch + (lane & 7) * 16);
      if (SADDR) {
        asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(tb) : "memory");
      } else {
        u32x4* p = (u32x4*)(tb + off);
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  const long long pitch = 6144;
  const long long per_wg = 8ll * 1024 * pitch;   // 8 passes of 1024 rows, then reused
  char* dst; (void)hipMalloc(&dst, per_wg * 8 + (1 << 20)); (void)hipMemset(dst, 0, per_wg * 8 + (1 << 20));
  unsigned long long* out; (void)hipMalloc(&out, 8 * 8 * 8);
  const int iters = 200;
  for (int saddr = 0; saddr < 2; ++saddr)
    for (int grid : {1, 8})
      for (int nw : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
          if (saddr) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(512), 0, 0, dst, pitch, iters, nw, per_wg, out);
          else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(512), 0, 0, dst, pitch, iters, nw, per_wg, out);
        }
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(64);
        (void)hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int w = 0; w < nw; ++w) m = std::max(m, (double)h[w]);
        printf("%s, %d workgroup(s), %d waves storing: %6.1f cycles per store and wave, %5.1f cycles per store CU-wide\n",
               saddr ? "scalar base + 32-bit offset" : "64-bit lane address      ", grid, nw, m / (iters * 16.0), m / (iters * 16.0 * nw));
      }
  return 0;
}
