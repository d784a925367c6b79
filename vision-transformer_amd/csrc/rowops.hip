// Row-wise (HBM-bound) kernels of the ViT forward for gfx950, one 64-lane wave per row, 16-byte loads:
//   ln_stats        (mean, rstd) per row -- the statistics half of a LayerNorm folded into the next GEMM
//   layernorm       full LayerNorm (final norm / stand-alone)
//   embed_finalize  class-token row, optional patch norm, norm_pre, + positional embedding, in place
//   scale_channels  LayerScale stand-alone
//   prepack_linear  one-time weight packing (LayerNorm / LayerScale folding, bf16, K padding)
//   casts           f32 <-> bf16
// A row is held in registers (NC chunks of 8 values per lane: 4 for D <= 2048, 16 up to 8192) so mean and variance
// are the exact two-pass form over one read of the row.
#include "common.h"

namespace {

constexpr int NC_SMALL = 4, NC_WIDE = 16;   // 8-element chunks per lane
constexpr int MAXD_SMALL = 64 * 8 * NC_SMALL;   // 2048
constexpr int MAXD = 64 * 8 * NC_WIDE;          // 8192
constexpr int ROWS_PER_WG = 4;        // 256 threads

template <int NC>
struct Row {
  float v[NC][8];
};

template <typename T>
__device__ __forceinline__ void load8(const T* p, float* v);
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) {
  const u16x8_t u = *(const u16x8_t*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(u[e]);
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float* v) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
  uint4 o;
  o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
  o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
  *(uint4*)p = o;
}
__device__ __forceinline__ void store8(float* p, const float* v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <typename T, int NC>
__device__ __forceinline__ void load_row(Row<NC>& r, const T* x, int D, int lane) {
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = (lane + 64 * t) * 8;
    if (c < D) {
      load8<T>(x + c, r.v[t]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[t][e] = 0.f;
    }
  }
}

// two-pass mean / rstd of the row held in r (biased variance, eps inside the sqrt)
template <int NC>
__device__ __forceinline__ void row_stats(const Row<NC>& r, int D, int lane, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NC; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += r.v[t][e];
  mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    if ((lane + 64 * t) * 8 < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = r.v[t][e] - mean;
        q = fmaf(d, d, q);
      }
    }
  }
  const float var = wave_sum(q) / (float)D;
  rstd = 1.0f / sqrtf(var + eps);
}

template <int NC>
__device__ __forceinline__ void normalize_row(Row<NC>& r, int D, int lane, float mean, float rstd,
                                              const float* w, const float* b) {
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = (lane + 64 * t) * 8;
    if (c < D) {
      float wv[8], bv[8];
      load8<float>(w + c, wv);
      load8<float>(b + c, bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[t][e] = fmaf((r.v[t][e] - mean) * rstd, wv[e], bv[e]);
    }
  }
}

template <typename T, int NC = NC_SMALL>
__global__ __launch_bounds__(256) void ln_stats_kernel(const T* __restrict__ x, int64_t ldx, int M, int D,
                                                       float eps, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
  if (m >= M) return;
  Row<NC> r;
  load_row<T>(r, x + (int64_t)m * ldx, D, lane);
  float mean, rstd;
  row_stats(r, D, lane, eps, mean, rstd);
  if (lane == 0) *(float2*)(stats + 2 * (int64_t)m) = make_float2(mean, rstd);
}

template <typename TI, typename TO, int NC = NC_SMALL>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int64_t ldx,
                                                        TO* __restrict__ out, int64_t ldo, int M, int D,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
  if (m >= M) return;
  Row<NC> r;
  load_row<TI>(r, x + (int64_t)m * ldx, D, lane);
  float mean, rstd;
  row_stats(r, D, lane, eps, mean, rstd);
  normalize_row(r, D, lane, mean, rstd, w, b);
  TO* o = out + (int64_t)m * ldo;
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = (lane + 64 * t) * 8;
    if (c < D) store8(o + c, r.v[t]);
  }
}

template <typename T, int NC = NC_SMALL>
__global__ __launch_bounds__(256) void embed_finalize_kernel(T* __restrict__ x, int64_t ldx, int rows,
                                                             int ntok, int D, const float* __restrict__ cls,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ en_w,
                                                             const float* __restrict__ en_b, float en_eps,
                                                             const float* __restrict__ np_w,
                                                             const float* __restrict__ np_b, float np_eps,
                                                             float* __restrict__ stats, float stats_eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int n = m % ntok;
  T* xr = x + (int64_t)m * ldx;
  Row<NC> r;
  float mean, rstd;
  if (n == 0 && cls != nullptr) {
    load_row<float>(r, cls, D, lane);
  } else {
    load_row<T>(r, xr, D, lane);
    if (en_w != nullptr) {  // PatchEmbed.norm_layer acts on patch tokens only
      row_stats(r, D, lane, en_eps, mean, rstd);
      normalize_row(r, D, lane, mean, rstd, en_w, en_b);
    }
  }
  if (np_w != nullptr) {
    row_stats(r, D, lane, np_eps, mean, rstd);
    normalize_row(r, D, lane, mean, rstd, np_w, np_b);
  }
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = (lane + 64 * t) * 8;
    if (c < D) {
      if (pos != nullptr) {
        float pv[8];
        load8<float>(pos + (int64_t)n * D + c, pv);
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[t][e] += pv[e];
      }
      store8(xr + c, r.v[t]);
    }
  }
  if (stats != nullptr) {   // (mean, rstd) of the row as stored, for the LayerNorm folded into the first GEMM that reads x
    if (sizeof(T) == 2) {   // bf16 storage: the statistics are those of the rounded values, exactly what vitm_ln_stats sees
#pragma unroll
      for (int t = 0; t < NC; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[t][e] = bf2f(f2bf(r.v[t][e]));
    }
    row_stats(r, D, lane, stats_eps, mean, rstd);
    if (lane == 0) *(float2*)(stats + 2 * (int64_t)m) = make_float2(mean, rstd);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* __restrict__ x, int64_t ldx,
                                                             T* __restrict__ out, int64_t ldo, int M,
                                                             int D, const float* __restrict__ gamma) {
  const int cpr = D / 8;
  const int64_t total = (int64_t)M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;
    float v[8], g[8];
    load8<T>(x + m * ldx + c, v);
    load8<float>(gamma + c, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= g[e];
    store8(out + m * ldo + c, v);
  }
}

// Per-head LayerNorm of q and k (Attention(qk_norm=True)), in place on the packed qkv activation.  A slot of SL
// lanes (the next power of two >= head_dim / 8) owns one head vector; reductions stay inside the slot.
template <typename T, int HD>
__global__ __launch_bounds__(256) void qk_norm_kernel(T* __restrict__ qkv, int64_t ld, int64_t nvec, int H,
                                                      const float* __restrict__ qw, const float* __restrict__ qb,
                                                      float qeps, const float* __restrict__ kw,
                                                      const float* __restrict__ kb, float keps) {
  constexpr int LPV = HD / 8;
  constexpr int SL = LPV <= 4 ? 4 : (LPV <= 8 ? 8 : 16);
  const int64_t vec = ((int64_t)blockIdx.x * 256 + threadIdx.x) / SL;
  const int sub = threadIdx.x % SL;
  const bool live = vec < nvec && sub < LPV;
  const int64_t m = live ? vec / (2 * H) : 0;
  const int j = live ? (int)(vec - m * 2 * H) : 0;
  const bool isk = j >= H;
  const int h = isk ? j - H : j;
  T* ptr = qkv + m * ld + (int64_t)(isk ? H * HD : 0) + h * HD + sub * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (live) load8<T>(ptr, v);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
#pragma unroll
  for (int o = 1; o < SL; o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)HD;
  float q = 0.f;
  if (live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) q = fmaf(v[e] - mean, v[e] - mean, q);
  }
#pragma unroll
  for (int o = 1; o < SL; o <<= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)HD + (isk ? keps : qeps));
  if (live) {
    float wv[8], bv[8];
    load8<float>((isk ? kw : qw) + sub * 8, wv);
    load8<float>((isk ? kb : qb) + sub * 8, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf((v[e] - mean) * rstd, wv[e], bv[e]);
    store8(ptr, v);
  }
}

template <typename T>
int launch_qk_norm(void* qkv, int64_t ld, int M, int H, int hd, const float* qw, const float* qb, float qeps,
                   const float* kw, const float* kb, float keps, hipStream_t st) {
  const int64_t nvec = (int64_t)M * 2 * H;
  const int lpv = hd / 8, sl = lpv <= 4 ? 4 : (lpv <= 8 ? 8 : 16);
  const dim3 grid((unsigned)((nvec * sl + 255) / 256)), blk(256);
  switch (hd) {
    case 32: hipLaunchKernelGGL((qk_norm_kernel<T, 32>), grid, blk, 0, st, (T*)qkv, ld, nvec, H, qw, qb, qeps, kw, kb, keps); break;
    case 64: hipLaunchKernelGGL((qk_norm_kernel<T, 64>), grid, blk, 0, st, (T*)qkv, ld, nvec, H, qw, qb, qeps, kw, kb, keps); break;
    case 96: hipLaunchKernelGGL((qk_norm_kernel<T, 96>), grid, blk, 0, st, (T*)qkv, ld, nvec, H, qw, qb, qeps, kw, kb, keps); break;
    case 128: hipLaunchKernelGGL((qk_norm_kernel<T, 128>), grid, blk, 0, st, (T*)qkv, ld, nvec, H, qw, qb, qeps, kw, kb, keps); break;
    default:
      vitm_set_error("vitm_qk_norm: head_dim=%d unsupported (32, 64, 96, 128)", hd);
      return VITM_EUNSUPPORTED;
  }
  VITM_CHECK_LAUNCH("vitm_qk_norm");
  return VITM_OK;
}

// Patch matrix of an NCHW image batch (bf16 out): row (b, cls + patch) holds the patch's (c, ky, kx) values in the
// order of Conv2d.weight.reshape(D, -1); the class-token row (when cls = 1) and the K padding are zero.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col_kernel(const TI* __restrict__ img, TO* __restrict__ out, int64_t ldo,
                                                     int64_t rows, int C, int H, int W, int ph, int pw, int gw,
                                                     int npatch, int kreal, int kpad, int cls, int vec) {
  const int cpr = kpad / 8;
  const int64_t total = rows * cpr;
  const int pp = ph * pw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cpr;
    const int k0 = (int)(i - row * cpr) * 8;
    const int64_t b = row / (npatch + cls);
    const int n = (int)(row - b * (npatch + cls)) - cls;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (n >= 0 && k0 < kreal) {
      const int py = n / gw, px = n - py * gw;
      const int64_t base = ((int64_t)b * C * H + (int64_t)py * ph) * W + (int64_t)px * pw;
      if (vec && k0 + 8 <= kreal) {
        const int c = k0 / pp, kr = k0 - c * pp;
        const int ky = kr / pw, kx = kr - ky * pw;
        load8<TI>(img + base + ((int64_t)c * H + ky) * W + kx, v);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + e;
          if (k < kreal) {
            const int c = k / pp, kr = k - c * pp;
            const int ky = kr / pw, kx = kr - ky * pw;
            const TI t = img[base + ((int64_t)c * H + ky) * W + kx];
            if (sizeof(TI) == 2) v[e] = bf2f(*(const bf16_t*)&t); else v[e] = *(const float*)&t;
          }
        }
      }
    }
    store8(out + row * ldo + k0, v);
  }
}

// out = gelu_erf(x) elementwise (stand-alone activation): erf to 1.5e-7 in both dtypes
template <typename T>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ out, int64_t ldo,
                                                   int M, int D) {
  const int cpr = D / 8;
  const int64_t total = (int64_t)M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;
    float v[8];
    load8<T>(x + m * ldx + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_accurate(v[e]);
    store8(out + m * ldo + c, v);
  }
}

// out = act(x) elementwise for the activations an MLP may be built with besides the fused erf GELU
template <typename T, int KIND>
__global__ __launch_bounds__(256) void activation_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ out,
                                                         int64_t ldo, int M, int D) {
  const int cpr = D / 8;
  const int64_t total = (int64_t)M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;
    float v[8];
    load8<T>(x + m * ldx + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e];
      if (KIND == VITM_ACT_RELU) v[e] = fmaxf(t, 0.f);
      else if (KIND == VITM_ACT_GELU_TANH) v[e] = 0.5f * t * (1.f + tanhf(0.7978845608028654f * (t + 0.044715f * t * t * t)));
      else if (KIND == VITM_ACT_SILU) v[e] = t / (1.f + __expf(-t));
      else if (KIND == VITM_ACT_SIGMOID) v[e] = 1.f / (1.f + __expf(-t));
      else if (KIND == VITM_ACT_TANH) v[e] = tanhf(t);
      else v[e] = gelu_erf_accurate(t);
    }
    store8(out + m * ldo + c, v);
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// one workgroup per output row n
template <typename TW, typename TP>
__global__ __launch_bounds__(256) void prepack_linear_kernel(const TW* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ ln_g,
                                                             const float* __restrict__ ln_b,
                                                             const float* __restrict__ oscale, int K, int kpad,
                                                             TP* __restrict__ wp, float* __restrict__ s,
                                                             float* __restrict__ c, float* __restrict__ g) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  const TW* wr = w + (int64_t)n * K;
  TP* wo = wp + (int64_t)n * kpad;
  float ssum = 0.f, csum = 0.f;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    float o = 0.f;   // the value as the GEMM will see it (rounded to the packed type)
    if (k < K) {
      float wv;
      if (sizeof(TW) == 2) wv = bf2f(*(const bf16_t*)(wr + k)); else wv = *(const float*)(wr + k);
      const float gk = ln_g ? ln_g[k] : 1.0f;
      o = wv * gk;
      if (sizeof(TP) == 2) o = bf2f(f2bf(o));
      ssum += o;
      if (ln_b) csum = fmaf(ln_b[k], wv, csum);
    }
    if (sizeof(TP) == 2) *(bf16_t*)(wo + k) = f2bf(o); else *(float*)(wo + k) = o;
  }
  ssum = block_sum_256(ssum, sh);
  csum = block_sum_256(csum, sh);
  if (threadIdx.x == 0) {
    const float os = oscale ? oscale[n] : 1.0f;
    if (s) s[n] = ssum;
    c[n] = (csum + (bias ? bias[n] : 0.f)) * os;
    if (g) g[n] = os;
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8<float>(src + i * 8, v);
    store8(dst + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = f2bf(src[n8 * 8 + threadIdx.x]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8<bf16_t>(src + i * 8, v);
    store8(dst + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = bf2f(src[n8 * 8 + threadIdx.x]);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline unsigned row_grid(int M) { return (unsigned)((M + ROWS_PER_WG - 1) / ROWS_PER_WG); }
inline unsigned stream_grid(int64_t items) {
  int64_t g = (items + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int vitm_ln_stats(const void* x, int x_dtype, int64_t ldx, int M, int D, float eps, float* stats,
                             vitm_stream_t stream) {
  VITM_REQUIRE(x && stats, "vitm_ln_stats: null pointer");
  VITM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= MAXD, "vitm_ln_stats: D=%d must be a multiple of 8 and <= %d", D, MAXD);
  VITM_REQUIRE(ldx >= D && ldx % 8 == 0 && al16(x) && ((uintptr_t)stats & 7) == 0, "vitm_ln_stats: ldx/alignment");
  hipStream_t st = (hipStream_t)stream;
  const bool wide = D > MAXD_SMALL;   // rows beyond 2048 values: 16 chunks per lane
  if (x_dtype == VITM_BF16) {
    if (wide) hipLaunchKernelGGL((ln_stats_kernel<bf16_t, NC_WIDE>), dim3(row_grid(M)), dim3(256), 0, st, (const bf16_t*)x, ldx, M, D, eps, stats);
    else hipLaunchKernelGGL(ln_stats_kernel<bf16_t>, dim3(row_grid(M)), dim3(256), 0, st, (const bf16_t*)x, ldx, M, D, eps, stats);
  } else if (x_dtype == VITM_F32) {
    if (wide) hipLaunchKernelGGL((ln_stats_kernel<float, NC_WIDE>), dim3(row_grid(M)), dim3(256), 0, st, (const float*)x, ldx, M, D, eps, stats);
    else hipLaunchKernelGGL(ln_stats_kernel<float>, dim3(row_grid(M)), dim3(256), 0, st, (const float*)x, ldx, M, D, eps, stats);
  } else {
    vitm_set_error("vitm_ln_stats: bad x_dtype %d", x_dtype);
    return VITM_EINVAL;
  }
  VITM_CHECK_LAUNCH("vitm_ln_stats");
  return VITM_OK;
}

extern "C" int vitm_layernorm(const void* x, int x_dtype, int64_t ldx, void* out, int out_dtype, int64_t ldo, int M,
                              int D, const float* w, const float* b, float eps, vitm_stream_t stream) {
  VITM_REQUIRE(x && out && w && b, "vitm_layernorm: null pointer");
  VITM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= MAXD, "vitm_layernorm: D=%d must be a multiple of 8 and <= %d", D, MAXD);
  VITM_REQUIRE(ldx >= D && ldo >= D && ldx % 8 == 0 && ldo % 8 == 0 && al16(x) && al16(out) && al16(w) && al16(b),
               "vitm_layernorm: leading dimensions must be multiples of 8 and pointers 16-byte aligned");
  VITM_REQUIRE((x_dtype == VITM_BF16 || x_dtype == VITM_F32) && (out_dtype == VITM_BF16 || out_dtype == VITM_F32),
               "vitm_layernorm: bad dtype (x %d, out %d)", x_dtype, out_dtype);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(row_grid(M)), blk(256);
#define VITM_LN_LAUNCH(TI, TO)                                                                                       \
  do {                                                                                                               \
    if (D > MAXD_SMALL)                                                                                              \
      hipLaunchKernelGGL((layernorm_kernel<TI, TO, NC_WIDE>), grid, blk, 0, st, (const TI*)x, ldx, (TO*)out, ldo, M, D, w, b, eps); \
    else                                                                                                             \
      hipLaunchKernelGGL((layernorm_kernel<TI, TO>), grid, blk, 0, st, (const TI*)x, ldx, (TO*)out, ldo, M, D, w, b, eps);          \
  } while (0)
  if (x_dtype == VITM_BF16 && out_dtype == VITM_BF16) VITM_LN_LAUNCH(bf16_t, bf16_t);
  else if (x_dtype == VITM_BF16) VITM_LN_LAUNCH(bf16_t, float);
  else if (out_dtype == VITM_BF16) VITM_LN_LAUNCH(float, bf16_t);
  else VITM_LN_LAUNCH(float, float);
#undef VITM_LN_LAUNCH
  VITM_CHECK_LAUNCH("vitm_layernorm");
  return VITM_OK;
}

extern "C" int vitm_embed_finalize_stats(void* x, int x_dtype, int64_t ldx, int B, int ntok, int D, const float* cls,
                                         const float* pos, const float* en_w, const float* en_b, float en_eps,
                                         const float* np_w, const float* np_b, float np_eps, float* stats,
                                         float stats_eps, vitm_stream_t stream) {
  VITM_REQUIRE(x, "vitm_embed_finalize: null x");
  VITM_REQUIRE(stats == nullptr || (((uintptr_t)stats & 7) == 0 && stats_eps > 0.f), "vitm_embed_finalize_stats: stats alignment / eps");
  VITM_REQUIRE(x_dtype == VITM_BF16 || x_dtype == VITM_F32, "vitm_embed_finalize: bad x_dtype %d", x_dtype);
  VITM_REQUIRE(B > 0 && ntok > 0 && D > 0 && D % 8 == 0 && D <= MAXD, "vitm_embed_finalize: D=%d must be a multiple of 8 and <= %d", D, MAXD);
  VITM_REQUIRE(ldx >= D && ldx % 8 == 0 && al16(x), "vitm_embed_finalize: ldx/alignment");
  VITM_REQUIRE((en_w == nullptr) == (en_b == nullptr) && (np_w == nullptr) == (np_b == nullptr),
               "vitm_embed_finalize: LayerNorm weight and bias must be given together");
  VITM_REQUIRE(al16(cls) && al16(pos) && al16(en_w) && al16(en_b) && al16(np_w) && al16(np_b), "vitm_embed_finalize: vectors must be 16-byte aligned");
  const int rows = B * ntok;
#define VITM_EF_LAUNCH(T, NC)                                                                                          \
  hipLaunchKernelGGL((embed_finalize_kernel<T, NC>), dim3(row_grid(rows)), dim3(256), 0, (hipStream_t)stream, (T*)x, ldx, \
                     rows, ntok, D, cls, pos, en_w, en_b, en_eps, np_w, np_b, np_eps, stats, stats_eps)
  if (x_dtype == VITM_BF16) {
    if (D > MAXD_SMALL) VITM_EF_LAUNCH(bf16_t, NC_WIDE);
    else VITM_EF_LAUNCH(bf16_t, NC_SMALL);
  } else {
    if (D > MAXD_SMALL) VITM_EF_LAUNCH(float, NC_WIDE);
    else VITM_EF_LAUNCH(float, NC_SMALL);
  }
#undef VITM_EF_LAUNCH
  VITM_CHECK_LAUNCH("vitm_embed_finalize");
  return VITM_OK;
}

extern "C" int vitm_embed_finalize(void* x, int x_dtype, int64_t ldx, int B, int ntok, int D, const float* cls,
                                   const float* pos, const float* en_w, const float* en_b, float en_eps,
                                   const float* np_w, const float* np_b, float np_eps, vitm_stream_t stream) {
  return vitm_embed_finalize_stats(x, x_dtype, ldx, B, ntok, D, cls, pos, en_w, en_b, en_eps, np_w, np_b, np_eps, nullptr,
                                   0.f, stream);
}

extern "C" int vitm_qk_norm(void* qkv, int dtype, int64_t ldqkv, int M, int H, int head_dim, const float* q_w,
                            const float* q_b, float q_eps, const float* k_w, const float* k_b, float k_eps,
                            vitm_stream_t stream) {
  VITM_REQUIRE(qkv && q_w && q_b && k_w && k_b, "vitm_qk_norm: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_qk_norm: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && H > 0 && head_dim > 0, "vitm_qk_norm: M, H, head_dim must be positive");
  VITM_REQUIRE(ldqkv >= 3 * (int64_t)H * head_dim && ldqkv % 8 == 0 && al16(qkv), "vitm_qk_norm: ldqkv/alignment");
  VITM_REQUIRE(al16(q_w) && al16(q_b) && al16(k_w) && al16(k_b), "vitm_qk_norm: vectors must be 16-byte aligned");
  VITM_REQUIRE((int64_t)M * 2 * H * 16 / 256 < (1ll << 31), "vitm_qk_norm: too many rows for one launch");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VITM_BF16) return launch_qk_norm<bf16_t>(qkv, ldqkv, M, H, head_dim, q_w, q_b, q_eps, k_w, k_b, k_eps, st);
  return launch_qk_norm<float>(qkv, ldqkv, M, H, head_dim, q_w, q_b, q_eps, k_w, k_b, k_eps, st);
}

extern "C" int vitm_im2col(const void* img, int img_dtype, int B, int C, int H, int W, int ph, int pw, void* out,
                           int out_dtype, int64_t ldo, int kpad, int cls_slot, vitm_stream_t stream) {
  VITM_REQUIRE(img && out, "vitm_im2col: null pointer");
  VITM_REQUIRE(img_dtype == VITM_BF16 || img_dtype == VITM_F32, "vitm_im2col: bad img_dtype %d", img_dtype);
  VITM_REQUIRE(out_dtype == VITM_BF16 || out_dtype == VITM_F32, "vitm_im2col: bad out_dtype %d", out_dtype);
  VITM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0, "vitm_im2col: bad shape");
  const int gh = H / ph, gw = W / pw;
  VITM_REQUIRE(gh > 0 && gw > 0, "vitm_im2col: patch larger than the image");
  const int kreal = C * ph * pw;
  VITM_REQUIRE(kpad >= kreal && kpad % 8 == 0 && ldo >= kpad && ldo % 8 == 0 && al16(out), "vitm_im2col: kpad/ldo/alignment");
  const int vec = (pw % 8 == 0) && (W % 8 == 0) && (((uintptr_t)img & 31) == 0);
  const int64_t rows = (int64_t)B * (gh * gw + (cls_slot ? 1 : 0));
  const dim3 grid(stream_grid(rows * (kpad / 8)));
  hipStream_t st = (hipStream_t)stream;
#define VITM_IM2COL(TI, TO)                                                                                         \
  hipLaunchKernelGGL((im2col_kernel<TI, TO>), grid, dim3(256), 0, st, (const TI*)img, (TO*)out, ldo, rows, C, H, W, ph, pw, \
                     gw, gh * gw, kreal, kpad, cls_slot ? 1 : 0, vec)
  if (img_dtype == VITM_F32 && out_dtype == VITM_BF16) VITM_IM2COL(float, bf16_t);
  else if (img_dtype == VITM_F32) VITM_IM2COL(float, float);
  else if (out_dtype == VITM_BF16) VITM_IM2COL(bf16_t, bf16_t);
  else VITM_IM2COL(bf16_t, float);
#undef VITM_IM2COL
  VITM_CHECK_LAUNCH("vitm_im2col");
  return VITM_OK;
}

extern "C" int vitm_gelu(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D, vitm_stream_t stream) {
  VITM_REQUIRE(x && out, "vitm_gelu: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_gelu: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && ldx >= D && ldo >= D && ldx % 8 == 0 && ldo % 8 == 0, "vitm_gelu: D/ld must be multiples of 8");
  VITM_REQUIRE(al16(x) && al16(out), "vitm_gelu: pointers must be 16-byte aligned");
  const dim3 grid(stream_grid((int64_t)M * (D / 8)));
  if (dtype == VITM_BF16) {
    hipLaunchKernelGGL(gelu_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, D);
  } else {
    hipLaunchKernelGGL(gelu_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)out, ldo, M, D);
  }
  VITM_CHECK_LAUNCH("vitm_gelu");
  return VITM_OK;
}

template <int KIND>
static void launch_activation(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D, hipStream_t st) {
  const dim3 grid(stream_grid((int64_t)M * (D / 8)));
  if (dtype == VITM_BF16) {
    hipLaunchKernelGGL((activation_kernel<bf16_t, KIND>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, D);
  } else {
    hipLaunchKernelGGL((activation_kernel<float, KIND>), grid, dim3(256), 0, st, (const float*)x, ldx, (float*)out, ldo, M, D);
  }
}

extern "C" int vitm_activation(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D, int kind,
                               vitm_stream_t stream) {
  VITM_REQUIRE(x && out, "vitm_activation: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_activation: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && ldx >= D && ldo >= D && ldx % 8 == 0 && ldo % 8 == 0, "vitm_activation: D/ld must be multiples of 8");
  VITM_REQUIRE(al16(x) && al16(out), "vitm_activation: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case VITM_ACT_GELU_ERF: launch_activation<VITM_ACT_GELU_ERF>(x, ldx, out, ldo, dtype, M, D, st); break;
    case VITM_ACT_RELU: launch_activation<VITM_ACT_RELU>(x, ldx, out, ldo, dtype, M, D, st); break;
    case VITM_ACT_GELU_TANH: launch_activation<VITM_ACT_GELU_TANH>(x, ldx, out, ldo, dtype, M, D, st); break;
    case VITM_ACT_SILU: launch_activation<VITM_ACT_SILU>(x, ldx, out, ldo, dtype, M, D, st); break;
    case VITM_ACT_SIGMOID: launch_activation<VITM_ACT_SIGMOID>(x, ldx, out, ldo, dtype, M, D, st); break;
    case VITM_ACT_TANH: launch_activation<VITM_ACT_TANH>(x, ldx, out, ldo, dtype, M, D, st); break;
    default:
      vitm_set_error("vitm_activation: unknown activation kind %d", kind);
      return VITM_EINVAL;
  }
  VITM_CHECK_LAUNCH("vitm_activation");
  return VITM_OK;
}

extern "C" int vitm_scale_channels(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D,
                                   const float* gamma, vitm_stream_t stream) {
  VITM_REQUIRE(x && out && gamma, "vitm_scale_channels: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_scale_channels: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && ldx >= D && ldo >= D && ldx % 8 == 0 && ldo % 8 == 0, "vitm_scale_channels: D/ld must be multiples of 8");
  VITM_REQUIRE(al16(x) && al16(out) && al16(gamma), "vitm_scale_channels: pointers must be 16-byte aligned");
  const dim3 grid(stream_grid((int64_t)M * (D / 8)));
  if (dtype == VITM_BF16) {
    hipLaunchKernelGGL(scale_channels_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, D, gamma);
  } else {
    hipLaunchKernelGGL(scale_channels_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)out, ldo, M, D, gamma);
  }
  VITM_CHECK_LAUNCH("vitm_scale_channels");
  return VITM_OK;
}

extern "C" int vitm_prepack_linear(const void* w, int w_dtype, const float* bias, const float* ln_gamma,
                                   const float* ln_beta, const float* out_scale, int N, int K, int kpad,
                                   void* wp, int wp_dtype, float* s, float* c, float* g, vitm_stream_t stream) {
  VITM_REQUIRE(w && wp && c, "vitm_prepack_linear: null w/wp/c");
  VITM_REQUIRE(N > 0 && K > 0 && kpad >= K, "vitm_prepack_linear: bad N/K/kpad");
  VITM_REQUIRE((w_dtype == VITM_BF16 || w_dtype == VITM_F32) && (wp_dtype == VITM_BF16 || wp_dtype == VITM_F32),
               "vitm_prepack_linear: bad dtype (w %d, wp %d)", w_dtype, wp_dtype);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)N), blk(256);
  if (w_dtype == VITM_F32 && wp_dtype == VITM_BF16) {
    hipLaunchKernelGGL((prepack_linear_kernel<float, bf16_t>), grid, blk, 0, st, (const float*)w, bias, ln_gamma, ln_beta, out_scale, K, kpad, (bf16_t*)wp, s, c, g);
  } else if (w_dtype == VITM_BF16 && wp_dtype == VITM_BF16) {
    hipLaunchKernelGGL((prepack_linear_kernel<bf16_t, bf16_t>), grid, blk, 0, st, (const bf16_t*)w, bias, ln_gamma, ln_beta, out_scale, K, kpad, (bf16_t*)wp, s, c, g);
  } else if (w_dtype == VITM_F32) {
    hipLaunchKernelGGL((prepack_linear_kernel<float, float>), grid, blk, 0, st, (const float*)w, bias, ln_gamma, ln_beta, out_scale, K, kpad, (float*)wp, s, c, g);
  } else {
    hipLaunchKernelGGL((prepack_linear_kernel<bf16_t, float>), grid, blk, 0, st, (const bf16_t*)w, bias, ln_gamma, ln_beta, out_scale, K, kpad, (float*)wp, s, c, g);
  }
  VITM_CHECK_LAUNCH("vitm_prepack_linear");
  return VITM_OK;
}

extern "C" int vitm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, vitm_stream_t stream) {
  VITM_REQUIRE(src && dst && n > 0, "vitm_cast_f32_to_bf16: bad arguments");
  VITM_REQUIRE(al16(src) && al16(dst), "vitm_cast_f32_to_bf16: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  VITM_CHECK_LAUNCH("vitm_cast_f32_to_bf16");
  return VITM_OK;
}

extern "C" int vitm_cast_bf16_to_f32(const void* src, float* dst, int64_t n, vitm_stream_t stream) {
  VITM_REQUIRE(src && dst && n > 0, "vitm_cast_bf16_to_f32: bad arguments");
  VITM_REQUIRE(al16(src) && al16(dst), "vitm_cast_bf16_to_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
  VITM_CHECK_LAUNCH("vitm_cast_bf16_to_f32");
  return VITM_OK;
}
