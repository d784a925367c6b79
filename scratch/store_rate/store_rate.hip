// Microbenchmark: how fast does one CU get a 256 x 256 bf16 output tile (128 KiB) out, as the GEMM epilogue writes it:
// 8 waves, 16 stores of 1 KiB each (8 rows x 128 B, row pitch 6144 B) per wave and tile, non-temporal or plain.
// Grid sizes 256 / 64 / 8 / 1 workgroups tell a per-CU limit from a shared one (XCD write link, memory).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int SEG>
__global__ __launch_bounds__(512, 2) void k(char* dst, long long pitch, int tiles, long long tile_stride_wg, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // wave (g, c): rows 128 g .., columns 64 c ..; store s: rows 16 (s / 2) + 8 (s & 1) + (lane & 7), 8 lanes x 16 B per row
  const int g = wave >> 2, c = wave & 3;
  char* base = dst + (long long)blockIdx.x * tile_stride_wg;
  unsigned long long t0, t1, acc = 0;
  const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  __syncthreads();
  for (int t = 0; t < tiles; ++t) {
    char* tb = base + (long long)t * 256 * pitch;      // next 256 rows
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    constexpr int LPR = SEG / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;          // rows per instruction
    // the workgroup's 8 waves tile the 256 x 512-byte output: wave w owns 512 / SEG column groups? keep it simple:
    // wave w writes rows [w * 16 * RPI / 8 ...): 16 instructions x RPI rows, SEG bytes at column offset (w % (512 / SEG)) * SEG
    constexpr int CG = 512 / SEG;          // column groups of SEG bytes in the tile's 512-byte rows
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int cgp = wave % CG;
      const int rband = wave / CG;                         // 8 / CG row bands
      const int row = (rband * 16 + s) * RPI + lane / LPR; // 16 * RPI rows per band
      u32x4* p = (u32x4*)(tb + (long long)(row % 256) * pitch + cgp * SEG + (lane % LPR) * 16);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    acc += t1 - t0;
  }
  if (lane == 0) out[blockIdx.x * 8 + wave] = acc;
}

int main() {
  const long long pitch = 6144;                          // fc1's output row: 3072 bf16
  const int tiles = 24;
  const long long per_wg = (long long)tiles * 256 * pitch;   // 37.7 MB per workgroup, written once
  char* dst; hipMalloc(&dst, per_wg * 256);
  hipMemset(dst, 0, per_wg * 256);
  unsigned long long* out; hipMalloc(&out, 256 * 8 * 8);
  auto run = [&](int seg, int grid) {
    for (int rep = 0; rep < 3; ++rep) {
      if (seg == 64) hipLaunchKernelGGL((k<true, 64>), dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
      else if (seg == 128) hipLaunchKernelGGL((k<true, 128>), dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
      else if (seg == 256) hipLaunchKernelGGL((k<true, 256>), dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
      else hipLaunchKernelGGL((k<true, 512>), dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
    }
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (int b = 0; b < grid; ++b) { unsigned long long m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, h[b * 8 + w]); cyc.push_back((double)m / tiles); }
    std::sort(cyc.begin(), cyc.end());
    printf("non-temporal, %4d contiguous bytes per row and instruction (%2d rows), grid %3d: %7.0f cycles per 128-KiB tile = %5.1f B/clk/CU\n",
           seg, 1024 / seg, grid, cyc[grid / 2], 131072.0 / cyc[grid / 2]);
  };
  for (int seg : {64, 128, 256, 512})
    for (int grid : {256, 32, 1}) run(seg, grid);
  return 0;
}
