// Microbenchmark: what rate can one CU sustain for LDS-DMA (global_load_lds_dwordx4, 1-KiB pieces of 8 rows x 128 B)?
// One 512-thread workgroup per CU; NW of its 8 waves issue pieces back to back, DEPTH pieces in flight per wave.
// Sources: a 64-KiB window per workgroup that stays in L2 ("l2"), or a stream through a buffer far larger than the caches ("hbm").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define GPTR(p) ((const void __attribute__((address_space(1)))*)(p))
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void k(const char* src, long long window, long long row_stride, int iters, int nw,
                                            unsigned long long* out, int resident) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (resident ? (long long)(blockIdx.x % 64) * window : (long long)blockIdx.x * window);
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (wave < nw) {
    t0 = __builtin_amdgcn_s_memtime();
    long long off = (long long)wave * 8 * row_stride;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* g = base + (off % window) + (long long)(lane >> 3) * row_stride + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
        off += (long long)nw * 8 * row_stride;
      }
      if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (DEPTH >= 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  }
}

int main(int argc, char** argv) {
  const long long big = 4ll << 30;
  char* src; hipMalloc(&src, big); hipMemset(src, 1, big);
  unsigned long long* out; hipMalloc(&out, 256 * 8 * 8);
  hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  for (int resident = 1; resident >= 0; --resident)
    for (long long rs : {128ll, 1536ll})
      for (int nw : {1, 2, 4, 8})
        for (int depth : {2, 4, 8}) {
          const long long window = resident ? 65536 : (big / 256 / (8 * rs)) * (8 * rs);
          const int it = resident ? iters : 400;
          for (int rep = 0; rep < 2; ++rep) {
            if (depth == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 65536, 0, src, window, rs, it, nw, out, resident);
            else if (depth == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 65536, 0, src, window, rs, it, nw, out, resident);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 65536, 0, src, window, rs, it, nw, out, resident);
          }
          hipDeviceSynchronize();
          std::vector<unsigned long long> h(256 * 8);
          hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
          std::vector<double> c;
          for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < nw; ++w) m = std::max(m, h[b * 8 + w]); c.push_back((double)m); }
          std::sort(c.begin(), c.end());
          const double bytes = (double)it * depth * nw * 1024.0;
          printf("%s rowstride %5lld waves %d depth %d: %6.1f B/clk/CU (median CU), %5.1f clk per piece\n", resident ? "L2 " : "HBM", rs, nw, depth,
                 bytes / c[128], c[128] / ((double)it * depth * nw));
        }
  return 0;
}
