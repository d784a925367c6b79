// Microbenchmark: the epilogue pattern (8 blocks of 50 VALU + two 1-KiB stores per wave) on 32 CUs while the other 224
// CUs stream operands by LDS-DMA the way the K loop does (8 waves, 1-KiB pieces), from an L2-resident window or from a
// buffer far larger than the caches.  Question: does a CU that stores out of phase with the others get its stores out
// as fast as it does on an idle chip?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define GPTR(p) ((const void __attribute__((address_space(1)))*)(p))
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))

__global__ __launch_bounds__(512, 2) void k(char* dst, const char* src, long long window, int resident, int stream_on, int tiles,
                                            int stream_iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool storer = ((blockIdx.x >> 3) & 7) == 0;      // 4 workgroups per XCD
  const long long pitch = 6144;
  if (storer) {
    const int g = wave >> 2, c = wave & 3;
    char* base = dst + (long long)blockIdx.x * 8 * 256 * pitch;
    f32x2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, d0 = {3.f, 4.f}, d1 = {5.f, 6.f};
    unsigned long long acc = 0;
    for (int t = 0; t < tiles; ++t) {
      char* p0 = base + (long long)(t & 7) * 256 * pitch + (long long)(128 * g + (lane >> 3)) * pitch + c * 128 + (lane & 7) * 16;
      __syncthreads();
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int v = 0; v < 25; ++v) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d0) : "v"(a), "v"(b));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d1) : "v"(a), "v"(b));
        }
        const u32x4 w0 = {__builtin_bit_cast(unsigned, d0.x), __builtin_bit_cast(unsigned, d0.y), __builtin_bit_cast(unsigned, d1.x), 7u};
        const u32x4 w1 = {__builtin_bit_cast(unsigned, d1.y), __builtin_bit_cast(unsigned, d0.x), 9u, 11u};
        __builtin_nontemporal_store(w0, (u32x4*)p0);
        __builtin_nontemporal_store(w1, (u32x4*)(p0 + 8 * pitch));
        p0 += 16 * pitch;
      }
      acc += __builtin_amdgcn_s_memtime() - t0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // a K loop's worth of quiet between two epilogues (the storers are out of phase with nobody, they just idle)
      for (int q = 0; q < 20; ++q) __builtin_amdgcn_s_sleep(127);
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = acc;
  } else if (stream_on) {
    const char* base = src + (resident ? (long long)(blockIdx.x % 64) * window : (long long)blockIdx.x * window);
    long long off = (long long)wave * 8 * 128;
    for (int i = 0; i < stream_iters; ++i) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const char* gp = base + (off % window) + (long long)(lane >> 3) * 128 + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds(GPTR(gp), LPTR(smem + (wave * 4 + d) * 1024), 16, 0, 0);
        off += 8ll * 8 * 128;
      }
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

int main() {
  const long long big = 2ll << 30;
  char* src; (void)hipMalloc(&src, big); (void)hipMemset(src, 1, big);
  char* dst; (void)hipMalloc(&dst, 256ll * 8 * 256 * 6144 + (1 << 20)); (void)hipMemset(dst, 0, 256ll * 8 * 256 * 6144 + (1 << 20));
  unsigned long long* out; (void)hipMalloc(&out, 256 * 8 * 8);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int tiles = 30;
  for (int mode = 0; mode < 3; ++mode) {     // 0: the rest of the chip idle, 1: streaming out of L2, 2: streaming from HBM
    const int resident = mode == 1;
    const long long window = resident ? 65536 : (big / 256 / 1024) * 1024;
    // long enough that the streamers outlast the storers (30 tiles x (~8 k + 20 x 8 k cycles))
    const int stream_iters = mode == 1 ? 60000 : 25000;
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, dst, src, window, resident, mode != 0, tiles, stream_iters, out);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> g0, g1;
    for (int b = 0; b < 256; ++b)
      if (((b >> 3) & 7) == 0) { g0.push_back((double)h[b * 8 + 0] / tiles / 8); g1.push_back((double)h[b * 8 + 4] / tiles / 8); }
    std::sort(g0.begin(), g0.end()); std::sort(g1.begin(), g1.end());
    printf("%s: %6.0f cycles per block (older wave), %6.0f (younger wave)\n",
           mode == 0 ? "32 CUs storing, 224 idle                 " : mode == 1 ? "32 CUs storing, 224 streaming out of L2  " : "32 CUs storing, 224 streaming from HBM   ",
           g0[g0.size() / 2], g1[g1.size() / 2]);
  }
  return 0;
}
