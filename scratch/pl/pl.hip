#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 31, 32, 47, 48, 63}) printf("lane %2d: r0=%3u r1=%3u\n", l, h[l], h[l + 64]);
  return 0;
}
