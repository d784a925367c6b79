// Microbenchmark: the load twin of lane_map.hip - 16 loads of 8 rows x 128 B per wave and tile (the residual read of the
// GEMM's residual epilogue), with a row's 8 lanes adjacent or 8 lanes apart.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MAP>
__global__ __launch_bounds__(512, 2) void k(const char* src, long long pitch, int tiles, long long wg_stride, unsigned long long* out, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = wave >> 2, c = wave & 3;
  const char* base = src + (long long)blockIdx.x * wg_stride;
  int row, chunk;
  if (MAP == 0) { row = lane >> 3; chunk = lane & 7; } else { row = lane & 7; chunk = lane >> 3; }
  unsigned long long acc = 0;
  unsigned x = 0;
  __syncthreads();
  for (int t = 0; t < tiles; ++t) {
    const char* tb = base + (long long)t * 256 * pitch;     // fresh rows every tile (streamed, as the residual is)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u32x4 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 128 * g + 8 * s + row;
      v[s] = *(const u32x4*)(tb + (long long)r * pitch + c * 128 + chunk * 16);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) x ^= v[s].x ^ v[s].w;
    asm volatile("" ::"v"(x));
    acc += __builtin_amdgcn_s_memtime() - t0;
  }
  if (lane == 0) out[blockIdx.x * 8 + wave] = acc;
  if (x == 0x12345678u) sink[0] = x;
}

int main() {
  const long long pitch = 1536;         // proj / fc2: the residual stream's row (768 bf16)
  const int tiles = 24;
  const long long per_wg = (long long)tiles * 256 * pitch;
  char* src; (void)hipMalloc(&src, per_wg * 256 + (1 << 20)); (void)hipMemset(src, 1, per_wg * 256 + (1 << 20));
  unsigned long long* out; (void)hipMalloc(&out, 256 * 8 * 8);
  unsigned* sink; (void)hipMalloc(&sink, 64);
  const char* names[2] = {"adjacent lanes per row", "strided lanes per row "};
  for (int map = 0; map < 2; ++map)
    for (int grid : {1, 32, 256}) {
      for (int rep = 0; rep < 3; ++rep) {
        if (map == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, src, pitch, tiles, per_wg, out, sink);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, src, pitch, tiles, per_wg, out, sink);
      }
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * 8);
      (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> cyc;
      for (int b = 0; b < grid; ++b) { unsigned long long m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, h[b * 8 + w]); cyc.push_back((double)m / tiles); }
      std::sort(cyc.begin(), cyc.end());
      printf("loads, %s, %3d workgroups: %7.0f cycles per 128-KiB tile = %5.1f B/clk/CU\n", names[map], grid, cyc[grid / 2], 131072.0 / cyc[grid / 2]);
    }
  return 0;
}
