// Microbenchmark: the epilogue's shape without the GEMM: every wave of a 512-thread workgroup runs 8 blocks of
// { NV packed-fp32 VALU instructions on its data registers; two 1-KiB non-temporal stores }.  Cycles per block and wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV>
__global__ __launch_bounds__(512, 2) void k(char* dst, long long pitch, int tiles, long long wg_stride, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave >> 2, c = wave & 3;
  char* base = dst + (long long)blockIdx.x * wg_stride;
  f32x2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, d0 = {3.f, 4.f}, d1 = {5.f, 6.f};
  unsigned long long acc = 0;
  __syncthreads();
  for (int t = 0; t < tiles; ++t) {
    char* tb = base + (long long)(t & 7) * 256 * pitch;
    char* p0 = tb + (long long)(128 * g + (lane >> 3)) * pitch + c * 128 + (lane & 7) * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int v = 0; v < NV / 2; ++v) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d0) : "v"(a), "v"(b));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d1) : "v"(a), "v"(b));
      }
      const u32x4 w0 = {__builtin_bit_cast(unsigned, d0.x), __builtin_bit_cast(unsigned, d0.y), __builtin_bit_cast(unsigned, d1.x), 7u};
      const u32x4 w1 = {__builtin_bit_cast(unsigned, d1.y), __builtin_bit_cast(unsigned, d0.x), 9u, 11u};
      __builtin_nontemporal_store(w0, (u32x4*)p0);
      __builtin_nontemporal_store(w1, (u32x4*)(p0 + 8 * pitch));
      p0 += 16 * pitch;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    acc += t1 - t0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane == 0) out[blockIdx.x * 8 + wave] = acc;
}

int main() {
  const long long pitch = 6144;
  const int tiles = 40;
  const long long per_wg = 8ll * 256 * pitch;
  char* dst; (void)hipMalloc(&dst, per_wg * 256 + (1 << 20)); (void)hipMemset(dst, 0, per_wg * 256 + (1 << 20));
  unsigned long long* out; (void)hipMalloc(&out, 256 * 8 * 8);
  auto run = [&](int nv, int grid) {
    for (int rep = 0; rep < 3; ++rep) {
      if (nv == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
      else if (nv == 50) hipLaunchKernelGGL(k<50>, dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
      else hipLaunchKernelGGL(k<130>, dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
    }
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> g0, g1;
    for (int b = 0; b < grid; ++b) { g0.push_back((double)h[b * 8 + 0] / tiles / 8); g1.push_back((double)h[b * 8 + 4] / tiles / 8); }
    std::sort(g0.begin(), g0.end()); std::sort(g1.begin(), g1.end());
    printf("%3d VALU per block, %3d workgroups: %6.0f cycles per block (wave 0), %6.0f (wave 4)\n", nv, grid, g0[grid / 2], g1[grid / 2]);
  };
  for (int nv : {0, 50, 130})
    for (int grid : {1, 32, 256}) run(nv, grid);
  return 0;
}
