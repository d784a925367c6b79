// Shared device helpers and host-side error plumbing for libvitm_gfx950 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/vitm_gfx950.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define GPTR(p) ((const void __attribute__((address_space(1)))*)(p))
#define LPTR(p) ((void __attribute__((address_space(3)))*)(p))

__device__ __forceinline__ float bf2f(bf16_t u) {
  return __builtin_bit_cast(float, (unsigned)u << 16);
}
// round-to-nearest-even, NaN-preserving (hipcc lowers the cast to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> one dword of two bf16 (lo in bits 0-15), a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7): the fp32-accurate form.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float r = fmaf(-p, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_accurate(float v) {
  return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752f));
}

// GELU(x) = x * Phi(x) for bf16 outputs, two values at a time in packed fp32 math (no transcendentals):
// Phi(x) - 1/2 = xc * Q(xc^2) with xc = clamp(x, -4, 4) and Q a degree-8 minimax polynomial: |error of Phi| <= 7e-6 on
// [-4, 4] with Phi > 0 throughout (2.6e-5 at -4), so nothing is floored; beyond +-4 Phi keeps its value there, an error
// of at most 3.2e-5 * |x|: far below one bf16 ulp of any output that is not itself below 1e-3.
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  f32x2_t xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f);
  const f32x2_t t = xc * xc;
  f32x2_t q = t * 8.065587196e-11f + -7.004855253e-09f;
  q = q * t + 2.716520232e-07f;
  q = q * t + -6.295502535e-06f;
  q = q * t + 9.891200638e-05f;
  q = q * t + -1.133939351e-03f;
  q = q * t + 9.877516551e-03f;
  q = q * t + -6.641063552e-02f;
  q = q * t + 3.989227214e-01f;
  return x * (xc * q + 0.5f);
}

// d/dx GELU(x) = Phi(x) + x * phi(x), two values at a time: Phi from the same polynomial as gelu_erf2, here shifted by
// its value at -4 and floored at 0 (|error| <= 2e-5),
// phi(x) = exp(-x^2 / 2) / sqrt(2 pi) through v_exp_f32.  For bf16 gradients.
__device__ __forceinline__ f32x2_t dgelu_erf2(f32x2_t x) {
  f32x2_t xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f);
  const f32x2_t t = xc * xc;
  f32x2_t q = t * 8.065587196e-11f + -7.004855253e-09f;
  q = q * t + 2.716520232e-07f;
  q = q * t + -6.295502535e-06f;
  q = q * t + 9.891200638e-05f;
  q = q * t + -1.133939351e-03f;
  q = q * t + 9.877516551e-03f;
  q = q * t + -6.641063552e-02f;
  q = q * t + 3.989227214e-01f;
  f32x2_t phi = xc * q + (0.5f - 2.5e-5f);
  phi.x = fmaxf(phi.x, 0.0f);
  phi.y = fmaxf(phi.y, 0.0f);
  const f32x2_t x2 = x * x;
  f32x2_t pdf;
  pdf.x = 0.3989422804f * __builtin_amdgcn_exp2f(-0.72134752f * x2.x);
  pdf.y = 0.3989422804f * __builtin_amdgcn_exp2f(-0.72134752f * x2.y);
  return phi + x * pdf;
}
// the same function on NP independent pairs with the Horner steps interleaved across pairs (one dependent chain
// per pair would leave the packed-math pipe waiting on itself)
template <int NP>
__device__ __forceinline__ void gelu_erf2_n(f32x2_t* x) {
  f32x2_t xc[NP], t[NP], q[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    xc[k].x = __builtin_amdgcn_fmed3f(x[k].x, -4.0f, 4.0f);
    xc[k].y = __builtin_amdgcn_fmed3f(x[k].y, -4.0f, 4.0f);
    t[k] = xc[k] * xc[k];
    q[k] = t[k] * 8.065587196e-11f + -7.004855253e-09f;
  }
  const float cf[7] = {2.716520232e-07f, -6.295502535e-06f, 9.891200638e-05f, -1.133939351e-03f,
                       9.877516551e-03f, -6.641063552e-02f, 3.989227214e-01f};
#pragma unroll
  for (int c = 0; c < 7; ++c)
#pragma unroll
    for (int k = 0; k < NP; ++k) q[k] = q[k] * t[k] + cf[c];
#pragma unroll
  for (int k = 0; k < NP; ++k) x[k] = x[k] * (xc[k] * q[k] + 0.5f);
}
__device__ __forceinline__ float gelu_erf(float v) {
  f32x2_t t = {v, v};
  return gelu_erf2(t).x;
}

// ---- host side ------------------------------------------------------------------------------
void vitm_set_error(const char* fmt, ...);

#define VITM_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      vitm_set_error(__VA_ARGS__);         \
      return VITM_EINVAL;                  \
    }                                      \
  } while (0)

// One-time (per kernel instantiation and device) opt-in to more than 64 KiB of dynamic LDS; cheap afterwards.
#define VITM_LDS_OPTIN(kfn, bytes, what)                                                            \
  do {                                                                                              \
    static std::atomic<unsigned long long> done__{0};                                               \
    int dev__ = 0;                                                                                  \
    (void)hipGetDevice(&dev__);                                                                     \
    const unsigned long long bit__ = 1ull << (dev__ & 63);                                          \
    if (!(done__.load(std::memory_order_relaxed) & bit__)) {                                        \
      if (hipFuncSetAttribute((const void*)(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)) != hipSuccess) { \
        vitm_set_error("%s: cannot reserve %d bytes of LDS", what, (int)(bytes));                   \
        (void)hipGetLastError();                                                                    \
        return VITM_ELAUNCH;                                                                        \
      }                                                                                             \
      done__.fetch_or(bit__, std::memory_order_relaxed);                                            \
    }                                                                                               \
  } while (0)

// compute units of the calling thread's current device (cached per device)
int vitm_cu_count();

#define VITM_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      vitm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return VITM_ELAUNCH;                                                        \
    }                                                                             \
  } while (0)
