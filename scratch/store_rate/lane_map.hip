// Microbenchmark: does it matter WHICH lanes of a store instruction write a row's 128 contiguous bytes?
// Every instruction writes 8 rows x 128 B.  "adjacent": lanes 8 r .. 8 r + 7 write row r (consecutive lanes = consecutive
// 16-byte chunks).  "strided": lanes r, r + 8, ..., r + 56 write row r (lane & 7 = row, lane >> 3 = chunk) - the layout
// the GEMM epilogue has after its row-pair exchange.  "gemm": strided, chunks in the epilogue's order (0, 2, 1, 3, 4, 6, 5, 7).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MAP>
__global__ __launch_bounds__(512, 2) void k(char* dst, long long pitch, int tiles, long long wg_stride, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave >> 2, c = wave & 3;
  char* base = dst + (long long)blockIdx.x * wg_stride;
  int row, chunk;
  if (MAP == 0) { row = lane >> 3; chunk = lane & 7; }
  else if (MAP == 1) { row = lane & 7; chunk = lane >> 3; }
  else { row = lane & 7; const int q = (lane >> 4) & 3, h = (lane >> 3) & 1; chunk = ((q & 1) ? 2 + (q >> 1) : (q >> 1)) + 4 * h; }
  const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  unsigned long long acc = 0;
  __syncthreads();
  for (int t = 0; t < tiles; ++t) {
    char* tb = base + (long long)(t & 7) * 256 * pitch;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 128 * g + 8 * s + row;
      __builtin_nontemporal_store(v, (u32x4*)(tb + (long long)r * pitch + c * 128 + chunk * 16));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += __builtin_amdgcn_s_memtime() - t0;
  }
  if (lane == 0) out[blockIdx.x * 8 + wave] = acc;
}

int main() {
  const long long pitch = 6144;
  const int tiles = 24;
  const long long per_wg = 8ll * 256 * pitch;
  char* dst; (void)hipMalloc(&dst, per_wg * 256 + (1 << 20)); (void)hipMemset(dst, 0, per_wg * 256 + (1 << 20));
  unsigned long long* out; (void)hipMalloc(&out, 256 * 8 * 8);
  const char* names[3] = {"adjacent lanes per row", "strided lanes per row ", "the epilogue's layout "};
  for (int map = 0; map < 3; ++map)
    for (int grid : {1, 32, 256}) {
      for (int rep = 0; rep < 3; ++rep) {
        if (map == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
        else if (map == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
        else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, dst, pitch, tiles, per_wg, out);
      }
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * 8);
      (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> cyc;
      for (int b = 0; b < grid; ++b) { unsigned long long m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, h[b * 8 + w]); cyc.push_back((double)m / tiles); }
      std::sort(cyc.begin(), cyc.end());
      printf("%s, %3d workgroups: %7.0f cycles per 128-KiB tile = %5.1f B/clk/CU\n", names[map], grid, cyc[grid / 2], 131072.0 / cyc[grid / 2]);
    }
  return 0;
}
