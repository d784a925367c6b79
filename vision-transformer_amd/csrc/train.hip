// Kernels of the training step for gfx950 (vitm/training/train.py:208-241): the pieces of the backward pass that are not
// GEMMs, the loss, gradient clipping and the optimizer update.  Activation-sized kernels exist for fp32 (the reference's
// training precision) and bf16 (mixed precision: bf16 activations, fp32 statistics / reductions / parameters).
//
//   transpose          dst[n, m] = src[m, n], K padding of the destination zeroed (operands of the weight-gradient GEMM)
//   colsum             out[c, d] = sum over the rows of chunk c of a[m, d] (* b[m, d]): bias / LayerScale / LayerNorm
//                      parameter gradients, two deterministic stages (chunks, then one chunk over the partials)
//   layernorm_bwd      dx (+ dx_add) and per-workgroup partial sums of dw, db
//   gelu_bwd           dpre = dh * d/dx gelu_erf(pre)
//   cross_entropy      mean softmax cross-entropy over the batch and its gradient (nn.CrossEntropyLoss, train.py:187)
//   sumsq / clip_coef / scale   torch.nn.utils.clip_grad_norm_(parameters, 1.0) on one flat gradient buffer (train.py:222)
//   adamw              torch.optim.AdamW update (train.py:68-72), one launch per parameter tensor
//
// Everything that reduces does so in a fixed order: results are bit-identical from run to run.
#include "common.h"

namespace {

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
__device__ __forceinline__ float ldv(const float* p) { return *p; }
__device__ __forceinline__ float ldv(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv(bf16_t* p, float v) { *p = f2bf(v); }
// eight consecutive elements as fp32 (one 16-byte access for bf16, two for fp32); p is 16-byte aligned
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  const u16x8_t t = *(const u16x8_t*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  uint4 w;
  w.x = pack_bf2(v[0], v[1]); w.y = pack_bf2(v[2], v[3]); w.z = pack_bf2(v[4], v[5]); w.w = pack_bf2(v[6], v[7]);
  *(uint4*)p = w;
}
// the same eight elements as they come from memory: requested early, converted when they are needed
struct Raw8f { float4 a, b; };
__device__ __forceinline__ Raw8f ldraw8(const float* p) { return Raw8f{*(const float4*)p, *(const float4*)(p + 4)}; }
__device__ __forceinline__ u16x8_t ldraw8(const bf16_t* p) { return *(const u16x8_t*)p; }
__device__ __forceinline__ void cvt8(const Raw8f& r, float (&v)[8]) {
  v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void cvt8(const u16x8_t& r, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bf2f(r[e]);
}
template <typename T>
inline bool vec_ok(const void* p, int64_t ld) {   // rows start 16-byte aligned
  return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (ld * (int64_t)sizeof(T)) % 16 == 0);
}
inline unsigned stream_grid(int64_t items, int per = 256, int cap = 4096) {
  int64_t g = (items + per - 1) / per;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- transpose: 32x32 tiles through LDS ------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ src, int64_t lds_, TD* __restrict__ dst,
                                                        int64_t ldd, int M, int N, int Mpad,
                                                        const float* __restrict__ row_scale) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 8 * i, n = n0 + tx;
    tile[ty + 8 * i][tx] = (m < M && n < N) ? ldv(src + (int64_t)m * lds_ + n) * (row_scale ? row_scale[m] : 1.0f) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, m = m0 + tx;
    if (n < N && m < Mpad) stv(dst + (int64_t)n * ldd + m, tile[tx][ty + 8 * i]);   // rows m >= M were loaded as zero
  }
}

// ---- column sums ---------------------------------------------------------------------------------------------
// workgroup = 64 columns x 4 row lanes; blockIdx.y = row chunk
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b,
                                                     int64_t ldb, float* __restrict__ out, int64_t ldo, int M, int D,
                                                     int rows_per_chunk) {
  __shared__ float sh[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  r1 = r1 < M ? r1 : M;
  float s = 0.f;
  if (col < D) {
    for (int m = r0 + rl; m < r1; m += 4) {
      const float v = ldv(a + (int64_t)m * lda + col);
      s += b ? v * ldv(b + (int64_t)m * ldb + col) : v;
    }
  }
  sh[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < D) out[(int64_t)blockIdx.y * ldo + col] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// vector form: a thread owns 8 adjacent columns (D % 8 == 0, rows 16-byte aligned); workgroup = 32 column groups x 8
// row lanes, so a workgroup covers 256 columns; blockIdx.y = row chunk
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b,
                                                      int64_t ldb, float* __restrict__ out, int64_t ldo, int M, int D,
                                                      int rows_per_chunk) {
  __shared__ float sh[8][32][9];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cg) * 8;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  r1 = r1 < M ? r1 : M;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (col < D) {
    for (int m = r0 + rl; m < r1; m += 8) {
      float av[8];
      ld8(a + (int64_t)m * lda + col, av);
      if (b) {
        float bv[8];
        ld8(b + (int64_t)m * ldb + col, bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = fmaf(av[e], bv[e], s[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += av[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[rl][cg][e] = s[e];
  __syncthreads();
  // 256 threads = 256 columns of this workgroup
  const int c = threadIdx.x, g = c >> 3, e = c & 7;
  const int oc = blockIdx.x * 256 + c;
  if (oc < D) {
    float t = ((sh[0][g][e] + sh[1][g][e]) + (sh[2][g][e] + sh[3][g][e])) + ((sh[4][g][e] + sh[5][g][e]) + (sh[6][g][e] + sh[7][g][e]));
    out[(int64_t)blockIdx.y * ldo + oc] = t;
  }
}

// column sums of TWO fp32 [rows, D] arrays in one launch (the per-workgroup partial sums of a LayerNorm's dw and db):
// workgroup = 16 columns x 64 row lanes, blockIdx.y picks the array; a thread sums rows rl, rl + 64, ... in that order,
// the 64 row lanes are added pairwise in a fixed tree.
__global__ __launch_bounds__(1024) void colsum_pair_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out_a, float* __restrict__ out_b, int rows,
                                                           int D) {
  __shared__ float sh[64][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cl;
  const float* src = blockIdx.y ? b : a;   // b == nullptr: launched with one array only
  float s = 0.f;
  if (col < D)
    for (int m = rl; m < rows; m += 64) s += src[(int64_t)m * D + col];
  sh[rl][cl] = s;
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (rl < h) sh[rl][cl] += sh[rl + h][cl];
    __syncthreads();
  }
  if (rl == 0 && col < D) (blockIdx.y ? out_b : out_a)[col] = sh[0][cl];
}

// ---- LayerScale gradient out of the weight gradient of the Linear in front of it -----------------------------------
// y = x + gamma * z, z = h W^T + b.  With dy the gradient at y:  dgamma[n] = sum_m dy[m, n] z[m, n]
//   = sum_k W[n, k] (sum_m dy[m, n] h[m, k]) + b[n] sum_m dy[m, n] = sum_k W[n, k] dWraw[n, k] + b[n] dbraw[n],
// where dWraw / dbraw are the Linear's gradients taken WITHOUT gamma; its true gradients are gamma[n] times those.  So z
// never has to be stored or read again: one workgroup per output row n forms the dot product (W rounded to bf16 first
// when the forward multiplied by the bf16 weight) and scales the row in place.
__global__ __launch_bounds__(256) void layerscale_grads_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ dw, float* __restrict__ db,
                                                               const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                               int K, int round_bf16) {
  __shared__ float sh[256];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* wr = w + (int64_t)n * K;
  float* dr = dw + (int64_t)n * K;
  const float g = gamma[n];
  float s = 0.f;
  for (int k = t; k < K; k += 256) {
    float wv = wr[k];
    if (round_bf16) wv = bf2f(f2bf(wv));
    const float d = dr[k];
    s = fmaf(wv, d, s);
    dr[k] = d * g;
  }
  sh[t] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (t < h) sh[t] += sh[t + h];
    __syncthreads();
  }
  if (t == 0) {
    float r = sh[0];
    if (bias) {
      r = fmaf(bias[n], db[n], r);   // the bias is added in fp32 in the GEMM epilogue: no rounding
      db[n] *= g;
    }
    dgamma[n] = r;
  }
}

// ---- LayerNorm backward: one wave per row at a time, a workgroup of 4 waves strides over the rows ------------------
constexpr int LN_MAXC = 4;   // 8-element chunks per lane: D <= 2048
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                            const T* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ w, float eps,
                                                            const T* __restrict__ dx_add, int64_t ldadd,
                                                            T* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ dwp, float* __restrict__ dbp, int M, int D) {
  __shared__ float red[2][4][64 * 8];   // per wave, one 8-chunk at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[LN_MAXC][8], gw[LN_MAXC][8], gb[LN_MAXC][8];
#pragma unroll
  for (int t = 0; t < LN_MAXC; ++t) {
    const int c = (lane + 64 * t) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wv[t][e] = c < D ? w[c + e] : 0.f;
      gw[t][e] = 0.f;
      gb[t][e] = 0.f;
    }
  }
  const float inv_d = 1.0f / (float)D;
  for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
    float xv[LN_MAXC][8], gv[LN_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXC; ++t) {
      const int c = (lane + 64 * t) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[t][e] = c < D ? ldv(x + (int64_t)m * ldx + c + e) : 0.f;
        gv[t][e] = c < D ? ldv(dy + (int64_t)m * lddy + c + e) : 0.f;
        s += xv[t][e];
      }
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXC; ++t)
      if ((lane + 64 * t) * 8 < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) q = fmaf(xv[t][e] - mean, xv[t][e] - mean, q);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXC; ++t)
      if ((lane + 64 * t) * 8 < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[t][e] - mean) * rstd;
          const float g = gv[t][e] * wv[t][e];
          gw[t][e] = fmaf(gv[t][e], xh, gw[t][e]);
          gb[t][e] += gv[t][e];
          xv[t][e] = xh;       // keep xhat
          gv[t][e] = g;        // keep dy * w
          c1 += g;
          c2 = fmaf(g, xh, c2);
        }
      }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int t = 0; t < LN_MAXC; ++t) {
      const int c = (lane + 64 * t) * 8;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float r = rstd * (gv[t][e] - c1 - xv[t][e] * c2);
          if (dx_add) r += ldv(dx_add + (int64_t)m * ldadd + c + e);
          stv(dx + (int64_t)m * lddx + c + e, r);
        }
      }
    }
  }
  // the four waves' partial parameter gradients -> one row of the partial arrays per workgroup
#pragma unroll
  for (int t = 0; t < LN_MAXC; ++t) {
    const int c = (lane + 64 * t) * 8;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[0][wave][lane * 8 + e] = gw[t][e];
      red[1][wave][lane * 8 + e] = gb[t][e];
    }
    __syncthreads();
    if (wave == 0 && c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = lane * 8 + e;
        dwp[(int64_t)blockIdx.x * D + c + e] = (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
        dbp[(int64_t)blockIdx.x * D + c + e] = (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
      }
    }
  }
}

// Fast form: rows 16-byte aligned, NC = ceil(D / 512) chunks of 8 columns per lane (registers sized to the row), 16-byte
// accesses.  The wave's next row (x, dy and dx_add) is requested before the current row's three reductions, so a row
// costs its arithmetic, not two trips to memory.
template <typename T, int NC>
__global__ __launch_bounds__(256) void layernorm_bwd8_kernel(const T* __restrict__ x, int64_t ldx,
                                                             const T* __restrict__ dy, int64_t lddy,
                                                             const float* __restrict__ w, float eps,
                                                             const T* __restrict__ dx_add, int64_t ldadd,
                                                             T* __restrict__ dx, int64_t lddx,
                                                             float* __restrict__ dwp, float* __restrict__ dbp, int M, int D) {
  __shared__ float red[2][4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[NC][8], gw[NC][8], gb[NC][8];
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = (lane + 64 * t) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wv[t][e] = c < D ? w[c + e] : 0.f;
      gw[t][e] = 0.f;
      gb[t][e] = 0.f;
    }
  }
  const float inv_d = 1.0f / (float)D;
  decltype(ldraw8(x)) xr[NC], gr[NC], ar[NC];
  auto request = [&](int m) {
#pragma unroll
    for (int t = 0; t < NC; ++t) {
      const int c = (lane + 64 * t) * 8;
      if (c < D) {
        xr[t] = ldraw8(x + (int64_t)m * ldx + c);
        gr[t] = ldraw8(dy + (int64_t)m * lddy + c);
        if (dx_add) ar[t] = ldraw8(dx_add + (int64_t)m * ldadd + c);
      }
    }
  };
  const int step = gridDim.x * 4;
  int m = blockIdx.x * 4 + wave;
  if (m < M) request(m);
  for (; m < M; m += step) {
    float xv[NC][8], gv[NC][8];
    decltype(ldraw8(x)) ac[NC];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NC; ++t) {
      const int c = (lane + 64 * t) * 8;
      if (c < D) {
        cvt8(xr[t], xv[t]);
        cvt8(gr[t], gv[t]);
        ac[t] = ar[t];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[t][e] = gv[t][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[t][e];
    }
    if (m + step < M) request(m + step);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NC; ++t)
      if ((lane + 64 * t) * 8 < D) {
#pragma unroll
        for (int e = 0; e < 8; ++e) q = fmaf(xv[t][e] - mean, xv[t][e] - mean, q);
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int t = 0; t < NC; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {          // columns past D hold zeros: they add nothing
        const float xh = (xv[t][e] - mean) * rstd;
        const float g = gv[t][e] * wv[t][e];
        gw[t][e] = fmaf(gv[t][e], xh, gw[t][e]);
        gb[t][e] += gv[t][e];
        xv[t][e] = xh;
        gv[t][e] = g;
        c1 += g;
        c2 = fmaf(g, xh, c2);
      }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int t = 0; t < NC; ++t) {
      const int c = (lane + 64 * t) * 8;
      if (c < D) {
        float rr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rr[e] = rstd * (gv[t][e] - c1 - xv[t][e] * c2);
        if (dx_add) {
          float av[8];
          cvt8(ac[t], av);
#pragma unroll
          for (int e = 0; e < 8; ++e) rr[e] += av[e];
        }
        st8(dx + (int64_t)m * lddx + c, rr);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = (lane + 64 * t) * 8;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[0][wave][lane * 8 + e] = gw[t][e];
      red[1][wave][lane * 8 + e] = gb[t][e];
    }
    __syncthreads();
    if (wave == 0 && c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = lane * 8 + e;
        dwp[(int64_t)blockIdx.x * D + c + e] = (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
        dbp[(int64_t)blockIdx.x * D + c + e] = (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
      }
    }
  }
}

template <typename T>
void launch_ln_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* w, float eps, const void* dx_add,
                   int64_t ldadd, void* dx, int64_t lddx, float* dwp, float* dbp, int nblocks, int M, int D, hipStream_t st) {
  const T *xp = (const T*)x, *gp = (const T*)dy, *ap = (const T*)dx_add;
  T* op = (T*)dx;
  const dim3 grid((unsigned)nblocks), blk(256);
  if (vec_ok<T>(x, ldx) && vec_ok<T>(dy, lddy) && vec_ok<T>(dx_add, ldadd) && vec_ok<T>(dx, lddx)) {
    const int nc = (D + 511) / 512;
#define VITM_LNB(NCV) hipLaunchKernelGGL((layernorm_bwd8_kernel<T, NCV>), grid, blk, 0, st, xp, ldx, gp, lddy, w, eps, ap, ldadd, op, lddx, dwp, dbp, M, D)
    if (nc == 1) VITM_LNB(1);
    else if (nc == 2) VITM_LNB(2);
    else if (nc == 3) VITM_LNB(3);
    else VITM_LNB(4);
#undef VITM_LNB
    return;
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel<T>, grid, blk, 0, st, xp, ldx, gp, lddy, w, eps, ap, ldadd, op, lddx, dwp, dbp, M, D);
}

// ---- backward of the per-head LayerNorm of q and k (Attention(qk_norm=True), parallel block) ------------------------
// A "vector" is one head of one token: HD elements, 8 per lane on HD / 8 lanes of an SL-lane group.  blockIdx.y picks
// q (0) or k (1); a thread walks vectors with a fixed lane-in-group, so it sums the parameter gradients of its own 8
// elements in registers.  dqkv holds d(loss)/d(normalised q, k) on entry and d(loss)/d(q, k) on return (v untouched).
template <typename T, int HD>
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const T* __restrict__ qkv, int64_t ld, T* __restrict__ dqkv,
                                                          int64_t ldd, int64_t nvec, int H, const float* __restrict__ qw,
                                                          float qeps, const float* __restrict__ kw, float keps,
                                                          float* __restrict__ partial) {
  constexpr int LPV = HD / 8;
  constexpr int SL = LPV <= 4 ? 4 : (LPV <= 8 ? 8 : 16);
  constexpr int GPB = 256 / SL;                 // lane groups per workgroup
  __shared__ float red[2][GPB][HD];
  const bool isk = blockIdx.y != 0;
  const int grp = threadIdx.x / SL, sub = threadIdx.x % SL;
  const bool lane_live = sub < LPV;
  const float eps = isk ? keps : qeps;
  float wv[8], gw[8], gb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    wv[e] = lane_live ? (isk ? kw : qw)[sub * 8 + e] : 0.f;
    gw[e] = 0.f;
    gb[e] = 0.f;
  }
  const int64_t coloff = isk ? (int64_t)H * HD : 0;
  const float inv = 1.0f / (float)HD;
  // every lane of a group runs the same number of iterations (the shuffles need them all)
  for (int64_t vec = (int64_t)blockIdx.x * GPB + grp; vec < nvec; vec += (int64_t)gridDim.x * GPB) {
    const int64_t m = vec / H;
    const int h = (int)(vec - m * H);
    const int64_t off = coloff + (int64_t)h * HD + sub * 8;
    float x[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = g[e] = 0.f;
    if (lane_live) {
      ld8(qkv + m * ld + off, x);
      ld8(dqkv + m * ldd + off, g);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e];
#pragma unroll
    for (int o = 1; o < SL; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv;
    float q = 0.f;
    if (lane_live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) q = fmaf(x[e] - mean, x[e] - mean, q);
    }
#pragma unroll
    for (int o = 1; o < SL; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q * inv + eps);
    float c1 = 0.f, c2 = 0.f;
    if (lane_live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (x[e] - mean) * rstd;
        gw[e] = fmaf(g[e], xh, gw[e]);
        gb[e] += g[e];
        const float gg = g[e] * wv[e];
        x[e] = xh;
        g[e] = gg;
        c1 += gg;
        c2 = fmaf(gg, xh, c2);
      }
    }
#pragma unroll
    for (int o = 1; o < SL; o <<= 1) {
      c1 += __shfl_xor(c1, o, 64);
      c2 += __shfl_xor(c2, o, 64);
    }
    c1 *= inv;
    c2 *= inv;
    if (lane_live) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = rstd * (g[e] - c1 - x[e] * c2);
      st8(dqkv + m * ldd + off, r);
    }
  }
  // per-workgroup partial sums: partial[(2 isk + which) * gridDim.x + blockIdx.x][HD]
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (lane_live) {
      red[0][grp][sub * 8 + e] = gw[e];
      red[1][grp][sub * 8 + e] = gb[e];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * HD; i += 256) {
    const int which = i / HD, c = i - which * HD;
    float t = 0.f;
    for (int gi = 0; gi < GPB; ++gi) t += red[which][gi][c];
    partial[((int64_t)(2 * (isk ? 1 : 0) + which) * gridDim.x + blockIdx.x) * HD + c] = t;
  }
}

template <typename T>
int launch_qk_norm_bwd(const void* qkv, int64_t ld, void* dqkv, int64_t ldd, int M, int H, int hd, const float* qw,
                       float qeps, const float* kw, float keps, float* partial, int nblocks, hipStream_t st) {
  const int64_t nvec = (int64_t)M * H;
  const dim3 grid((unsigned)nblocks, 2), blk(256);
#define VITM_QKB(HDV) hipLaunchKernelGGL((qk_norm_bwd_kernel<T, HDV>), grid, blk, 0, st, (const T*)qkv, ld, (T*)dqkv, ldd, nvec, H, qw, qeps, kw, keps, partial)
  switch (hd) {
    case 32: VITM_QKB(32); break;
    case 64: VITM_QKB(64); break;
    case 96: VITM_QKB(96); break;
    case 128: VITM_QKB(128); break;
    default:
      vitm_set_error("vitm_qk_norm_bwd: head_dim=%d unsupported (32, 64, 96, 128)", hd);
      return VITM_EUNSUPPORTED;
  }
#undef VITM_QKB
  VITM_CHECK_LAUNCH("vitm_qk_norm_bwd");
  return VITM_OK;
}

// ---- GELU backward: eight elements per thread when the rows allow 16-byte accesses ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd8_kernel(const T* __restrict__ pre, int64_t ldp, const T* __restrict__ dh,
                                                        int64_t ldh, T* __restrict__ dpre, int64_t ldd, int M, int D8) {
  const int64_t total = (int64_t)M * D8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D8;
    const int c = (int)(i - m * D8) * 8;
    float v[8], g[8];
    ld8(pre + m * ldp + c, v);
    ld8(dh + m * ldh + c, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cdf = 0.5f * (1.0f + erf_as(v[e] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * v[e] * v[e]);
      g[e] *= fmaf(v[e], pdf, cdf);
    }
    st8(dpre + m * ldd + c, g);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void residual_scale8_kernel(const T* __restrict__ r, int64_t ldr, const T* __restrict__ z,
                                                              int64_t ldz, const float* __restrict__ g, T* __restrict__ out,
                                                              int64_t ldo, int M, int D8) {
  const int64_t total = (int64_t)M * D8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D8;
    const int c = (int)(i - m * D8) * 8;
    float rv[8], zv[8];
    ld8(r + m * ldr + c, rv);
    ld8(z + m * ldz + c, zv);
#pragma unroll
    for (int e = 0; e < 8; ++e) rv[e] = fmaf(g[c + e], zv[e], rv[e]);
    st8(out + m * ldo + c, rv);
  }
}

// ---- GELU backward ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ pre, int64_t ldp, const T* __restrict__ dh,
                                                       int64_t ldh, T* __restrict__ dpre, int64_t ldd, int M, int D) {
  const int64_t total = (int64_t)M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D;
    const int c = (int)(i - m * D);
    const float v = ldv(pre + m * ldp + c);
    const float cdf = 0.5f * (1.0f + erf_as(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    stv(dpre + m * ldd + c, ldv(dh + m * ldh + c) * fmaf(v, pdf, cdf));
  }
}

// ---- LayerScale + residual add that keeps the branch output -------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void residual_scale_kernel(const T* __restrict__ r, int64_t ldr, const T* __restrict__ z,
                                                             int64_t ldz, const float* __restrict__ g, T* __restrict__ out,
                                                             int64_t ldo, int M, int D) {
  const int64_t total = (int64_t)M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D;
    const int c = (int)(i - m * D);
    stv(out + m * ldo + c, fmaf(g[c], ldv(z + m * ldz + c), ldv(r + m * ldr + c)));
  }
}

// ---- softmax cross-entropy, mean over the batch: one wave per row (rows spread over the chip), then one wave adds
// the per-row losses in a fixed order ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_entropy_rows_kernel(const float* __restrict__ logits, int64_t ld,
                                                                 const int64_t* __restrict__ labels, int B, int C,
                                                                 float* __restrict__ row_loss,
                                                                 float* __restrict__ dlogits, int64_t lddl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= B) return;   // wave-uniform
  const float inv_b = 1.0f / (float)B;
  const float* row = logits + (int64_t)r * ld;
  float mx = -3.0e38f;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(row[c] - mx);
  s = wave_sum(s);
  // a label outside [0, C) -- nn.CrossEntropyLoss's ignore_index (-100) included, which this kernel does not implement --
  // must not become an out-of-bounds read: the row's loss is NaN (so the mean is) and its gradient row is zero
  const int64_t y64 = labels[r];
  const bool ok = y64 >= 0 && y64 < (int64_t)C;
  const int y = ok ? (int)y64 : 0;
  if (lane == 0) row_loss[r] = ok ? mx + __logf(s) - row[y] : __builtin_nanf("");
  if (dlogits) {
    const float is = 1.0f / s;
    for (int c = lane; c < C; c += 64)
      dlogits[(int64_t)r * lddl + c] = ok ? (__expf(row[c] - mx) * is - (c == y ? 1.f : 0.f)) * inv_b : 0.f;
  }
}
__global__ __launch_bounds__(64) void cross_entropy_mean_kernel(const float* __restrict__ row_loss, int B, float* __restrict__ loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 64) s += row_loss[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// ---- gradient norm, clip coefficient, in-place scale ---------------------------------------------------------------
constexpr int SUMSQ_PARTS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ parts) {
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x;
  int64_t hi = lo + per;
  hi = hi < n ? hi : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s = fmaf(x[i], x[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// out[0] = total norm, out[1] = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(64) void clip_coef_kernel(const float* __restrict__ parts, int nparts, float max_norm,
                                                       float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 64) s += parts[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    out[0] = norm;
    out[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
  }
}
__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ coef) {
  const float c = coef[0];
  if (c == 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= c;
}

// ---- AdamW (decoupled weight decay, bias-corrected; torch.optim.AdamW with amsgrad=False, maximize=False) ------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = fmaf(beta1, m[i], (1.0f - beta1) * gi);      // lerp(m, g, 1 - beta1)
    const float vi = fmaf(beta2, v[i], (1.0f - beta2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

// the same update with the seven scalars (lr, beta1, beta2, eps, weight decay, 1 - beta1^step, sqrt(1 - beta2^step)) read
// from device memory: a captured launch then follows a learning-rate schedule without being captured again
__global__ __launch_bounds__(256) void adamw_hyper_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n, const float* __restrict__ hyper) {
  const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2_sqrt = hyper[6];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = fmaf(beta1, m[i], (1.0f - beta1) * gi);
    const float vi = fmaf(beta2, v[i], (1.0f - beta2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace

extern "C" int vitm_transpose(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd, int M, int N,
                              int m_pad, const float* row_scale, vitm_stream_t stream) {
  VITM_REQUIRE(src && dst, "vitm_transpose: null pointer");
  VITM_REQUIRE((src_dtype == VITM_BF16 || src_dtype == VITM_F32) && (dst_dtype == VITM_BF16 || dst_dtype == VITM_F32),
               "vitm_transpose: bad dtype %d -> %d", src_dtype, dst_dtype);
  VITM_REQUIRE(M > 0 && N > 0 && lds >= N && m_pad >= M && ldd >= m_pad, "vitm_transpose: bad shape");
  dim3 grid((unsigned)((m_pad + 31) / 32), (unsigned)((N + 31) / 32));
  VITM_REQUIRE(grid.y < 65536, "vitm_transpose: N too large");
  hipStream_t st = (hipStream_t)stream;
  const bool sf = src_dtype == VITM_F32, df = dst_dtype == VITM_F32;
  if (sf && df) hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, lds, (float*)dst, ldd, M, N, m_pad, row_scale);
  else if (sf) hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, lds, (bf16_t*)dst, ldd, M, N, m_pad, row_scale);
  else if (df) hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, lds, (float*)dst, ldd, M, N, m_pad, row_scale);
  else hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, M, N, m_pad, row_scale);
  VITM_CHECK_LAUNCH("vitm_transpose");
  return VITM_OK;
}

extern "C" int vitm_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, int dtype, float* out, int64_t ldo,
                           int M, int D, int nchunks, vitm_stream_t stream) {
  VITM_REQUIRE(a && out, "vitm_colsum: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_colsum: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && nchunks > 0 && nchunks < 65536 && lda >= D && ldo >= D && (!b || ldb >= D),
               "vitm_colsum: bad shape");
  const int rpc = (M + nchunks - 1) / nchunks;
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = dtype == VITM_F32;
  const bool vec = D % 8 == 0 && (f32 ? vec_ok<float>(a, lda) && vec_ok<float>(b, ldb) : vec_ok<bf16_t>(a, lda) && vec_ok<bf16_t>(b, ldb));
  if (vec) {
    const dim3 grid((unsigned)((D + 255) / 256), (unsigned)nchunks);
    if (f32) hipLaunchKernelGGL(colsum8_kernel<float>, grid, dim3(256), 0, st, (const float*)a, lda, (const float*)b, ldb, out, ldo, M, D, rpc);
    else hipLaunchKernelGGL(colsum8_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, out, ldo, M, D, rpc);
  } else {
    const dim3 grid((unsigned)((D + 63) / 64), (unsigned)nchunks);
    if (f32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)a, lda, (const float*)b, ldb, out, ldo, M, D, rpc);
    else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, out, ldo, M, D, rpc);
  }
  VITM_CHECK_LAUNCH("vitm_colsum");
  return VITM_OK;
}

extern "C" int vitm_colsum_pair(const float* a, const float* b, int rows, int D, float* out_a, float* out_b,
                                vitm_stream_t stream) {
  VITM_REQUIRE(a && out_a && (!b == !out_b), "vitm_colsum_pair: null pointer (b and out_b go together)");
  VITM_REQUIRE(rows > 0 && D > 0 && (int64_t)((D + 15) / 16) < 65536, "vitm_colsum_pair: bad shape");
  hipLaunchKernelGGL(colsum_pair_kernel, dim3((unsigned)((D + 15) / 16), b ? 2 : 1), dim3(1024), 0, (hipStream_t)stream, a, b, out_a,
                     out_b, rows, D);
  VITM_CHECK_LAUNCH("vitm_colsum_pair");
  return VITM_OK;
}

extern "C" int vitm_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* w, float eps,
                                  const void* dx_add, int64_t ldadd, void* dx, int64_t lddx, int dtype, float* dw_partial,
                                  float* db_partial, int nblocks, int M, int D, vitm_stream_t stream) {
  VITM_REQUIRE(x && dy && w && dx && dw_partial && db_partial, "vitm_layernorm_bwd: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_layernorm_bwd: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXC, "vitm_layernorm_bwd: D=%d must be a multiple of 8 and <= %d", D, 64 * 8 * LN_MAXC);
  VITM_REQUIRE(ldx >= D && lddy >= D && lddx >= D && (!dx_add || ldadd >= D) && nblocks > 0, "vitm_layernorm_bwd: bad leading dimension");
  if (dtype == VITM_F32) launch_ln_bwd<float>(x, ldx, dy, lddy, w, eps, dx_add, ldadd, dx, lddx, dw_partial, db_partial, nblocks, M, D, (hipStream_t)stream);
  else launch_ln_bwd<bf16_t>(x, ldx, dy, lddy, w, eps, dx_add, ldadd, dx, lddx, dw_partial, db_partial, nblocks, M, D, (hipStream_t)stream);
  VITM_CHECK_LAUNCH("vitm_layernorm_bwd");
  return VITM_OK;
}

extern "C" int vitm_qk_norm_bwd(const void* qkv, int64_t ldqkv, void* dqkv, int64_t lddqkv, int dtype, int M, int H,
                                int head_dim, const float* q_w, float q_eps, const float* k_w, float k_eps, float* partial,
                                int nblocks, vitm_stream_t stream) {
  VITM_REQUIRE(qkv && dqkv && q_w && k_w && partial, "vitm_qk_norm_bwd: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_qk_norm_bwd: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && H > 0 && nblocks > 0 && nblocks < 65536, "vitm_qk_norm_bwd: bad shape");
  const int64_t D = (int64_t)H * head_dim;
  VITM_REQUIRE(ldqkv >= 3 * D && lddqkv >= 3 * D, "vitm_qk_norm_bwd: leading dimensions must cover 3*H*head_dim");
  const bool ok = dtype == VITM_F32 ? vec_ok<float>(qkv, ldqkv) && vec_ok<float>(dqkv, lddqkv)
                                    : vec_ok<bf16_t>(qkv, ldqkv) && vec_ok<bf16_t>(dqkv, lddqkv);
  VITM_REQUIRE(ok, "vitm_qk_norm_bwd: rows must start 16-byte aligned");
  if (dtype == VITM_F32)
    return launch_qk_norm_bwd<float>(qkv, ldqkv, dqkv, lddqkv, M, H, head_dim, q_w, q_eps, k_w, k_eps, partial, nblocks, (hipStream_t)stream);
  return launch_qk_norm_bwd<bf16_t>(qkv, ldqkv, dqkv, lddqkv, M, H, head_dim, q_w, q_eps, k_w, k_eps, partial, nblocks, (hipStream_t)stream);
}

extern "C" int vitm_gelu_bwd(const void* pre, int64_t ldp, const void* dh, int64_t ldh, void* dpre, int64_t ldd, int dtype, int M,
                             int D, vitm_stream_t stream) {
  VITM_REQUIRE(pre && dh && dpre, "vitm_gelu_bwd: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_gelu_bwd: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && ldp >= D && ldh >= D && ldd >= D, "vitm_gelu_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = dtype == VITM_F32;
  const bool vec = D % 8 == 0 && (f32 ? vec_ok<float>(pre, ldp) && vec_ok<float>(dh, ldh) && vec_ok<float>(dpre, ldd)
                                      : vec_ok<bf16_t>(pre, ldp) && vec_ok<bf16_t>(dh, ldh) && vec_ok<bf16_t>(dpre, ldd));
  if (vec) {
    const dim3 grid(stream_grid((int64_t)M * (D / 8), 256, 8192));
    if (f32) hipLaunchKernelGGL(gelu_bwd8_kernel<float>, grid, dim3(256), 0, st, (const float*)pre, ldp, (const float*)dh, ldh, (float*)dpre, ldd, M, D / 8);
    else hipLaunchKernelGGL(gelu_bwd8_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)pre, ldp, (const bf16_t*)dh, ldh, (bf16_t*)dpre, ldd, M, D / 8);
  } else {
    const dim3 grid(stream_grid((int64_t)M * D));
    if (f32) hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)pre, ldp, (const float*)dh, ldh, (float*)dpre, ldd, M, D);
    else hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)pre, ldp, (const bf16_t*)dh, ldh, (bf16_t*)dpre, ldd, M, D);
  }
  VITM_CHECK_LAUNCH("vitm_gelu_bwd");
  return VITM_OK;
}

extern "C" int vitm_residual_scale(const void* resid, int64_t ldr, const void* z, int64_t ldz, const float* gamma, void* out,
                                   int64_t ldo, int dtype, int M, int D, vitm_stream_t stream) {
  VITM_REQUIRE(resid && z && gamma && out, "vitm_residual_scale: null pointer");
  VITM_REQUIRE(dtype == VITM_BF16 || dtype == VITM_F32, "vitm_residual_scale: bad dtype %d", dtype);
  VITM_REQUIRE(M > 0 && D > 0 && ldr >= D && ldz >= D && ldo >= D, "vitm_residual_scale: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = dtype == VITM_F32;
  const bool vec = D % 8 == 0 && (f32 ? vec_ok<float>(resid, ldr) && vec_ok<float>(z, ldz) && vec_ok<float>(out, ldo)
                                      : vec_ok<bf16_t>(resid, ldr) && vec_ok<bf16_t>(z, ldz) && vec_ok<bf16_t>(out, ldo));
  if (vec) {
    const dim3 grid(stream_grid((int64_t)M * (D / 8), 256, 8192));
    if (f32) hipLaunchKernelGGL(residual_scale8_kernel<float>, grid, dim3(256), 0, st, (const float*)resid, ldr, (const float*)z, ldz, gamma, (float*)out, ldo, M, D / 8);
    else hipLaunchKernelGGL(residual_scale8_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)resid, ldr, (const bf16_t*)z, ldz, gamma, (bf16_t*)out, ldo, M, D / 8);
  } else {
    const dim3 grid(stream_grid((int64_t)M * D));
    if (f32) hipLaunchKernelGGL(residual_scale_kernel<float>, grid, dim3(256), 0, st, (const float*)resid, ldr, (const float*)z, ldz, gamma, (float*)out, ldo, M, D);
    else hipLaunchKernelGGL(residual_scale_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)resid, ldr, (const bf16_t*)z, ldz, gamma, (bf16_t*)out, ldo, M, D);
  }
  VITM_CHECK_LAUNCH("vitm_residual_scale");
  return VITM_OK;
}

extern "C" int vitm_layerscale_grads(const float* w, const float* bias, float* dw, float* db, const float* gamma,
                                     float* dgamma, int N, int K, int round_bf16, vitm_stream_t stream) {
  VITM_REQUIRE(w && dw && gamma && dgamma && (!bias == !db), "vitm_layerscale_grads: null pointer (bias and db go together)");
  VITM_REQUIRE(N > 0 && K > 0, "vitm_layerscale_grads: bad shape");
  hipLaunchKernelGGL(layerscale_grads_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, w, bias, dw, db, gamma,
                     dgamma, K, round_bf16);
  VITM_CHECK_LAUNCH("vitm_layerscale_grads");
  return VITM_OK;
}

extern "C" int vitm_cross_entropy_f32(const float* logits, int64_t ld, const int64_t* labels, int B, int C, float* loss,
                                      float* dlogits, int64_t lddl, float* row_loss, vitm_stream_t stream) {
  VITM_REQUIRE(logits && labels && loss && row_loss, "vitm_cross_entropy_f32: null pointer");
  VITM_REQUIRE(B > 0 && C > 0 && ld >= C && (!dlogits || lddl >= C), "vitm_cross_entropy_f32: bad shape");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, logits, ld, labels, B, C, row_loss, dlogits, lddl);
  hipLaunchKernelGGL(cross_entropy_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)row_loss, B, loss);
  VITM_CHECK_LAUNCH("vitm_cross_entropy_f32");
  return VITM_OK;
}

// ---- the same loss for callers that cannot allocate: the per-row losses go to a file-scope device array instead of a
// caller-supplied buffer.  Each GPU holds its own copy of the array; its address on a device is looked up once, with
// that device current, and kept.
constexpr int CE_NOALLOC_MAX_ROWS = 16384;
static __device__ __align__(256) float scratch[CE_NOALLOC_MAX_ROWS];

static int scratch_on_current_device(hipStream_t st, float** out) {
  constexpr int MAXDEV = 64;
  static std::atomic<float*> ptrs[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
    vitm_set_error("vitm_cross_entropy_f32_noalloc: cannot identify the current device");
    return VITM_ELAUNCH;
  }
  // the kernels run on the stream's GPU and must get THAT GPU's copy of the array: a stream of another device than
  // the calling thread's current one is refused (the null stream belongs to the current device)
  hipDevice_t sdev = dev;
  if (st != nullptr && hipStreamGetDevice(st, &sdev) != hipSuccess) {
    (void)hipGetLastError();
    vitm_set_error("vitm_cross_entropy_f32_noalloc: cannot identify the stream's device");
    return VITM_ELAUNCH;
  }
  if ((int)sdev != dev) {
    vitm_set_error("vitm_cross_entropy_f32_noalloc: the stream belongs to device %d but the current device is %d "
                   "(make the stream's device current before the call)", (int)sdev, dev);
    return VITM_EINVAL;
  }
  float* p = ptrs[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
    if (e != hipSuccess || sym == nullptr) {
      vitm_set_error("vitm_cross_entropy_f32_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
      (void)hipGetLastError();
      return VITM_ELAUNCH;
    }
    p = (float*)sym;
    ptrs[dev].store(p, std::memory_order_release);   // every thread that races here stores the same address
  }
  *out = p;
  return VITM_OK;
}

extern "C" int vitm_cross_entropy_f32_noalloc(const float* logits, int64_t ld, const int64_t* labels, int B, int C,
                                              float* loss, float* dlogits, int64_t lddl, vitm_stream_t stream) {
  VITM_REQUIRE(logits && labels && loss, "vitm_cross_entropy_f32_noalloc: null pointer");
  VITM_REQUIRE(B > 0 && C > 0 && ld >= C && (!dlogits || lddl >= C), "vitm_cross_entropy_f32_noalloc: bad shape");
  VITM_REQUIRE(B <= CE_NOALLOC_MAX_ROWS, "vitm_cross_entropy_f32_noalloc: B=%d exceeds the built-in scratch of %d rows "
               "(use vitm_cross_entropy_f32 with a caller-supplied row_loss buffer)", B, CE_NOALLOC_MAX_ROWS);
  float* row_loss = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = scratch_on_current_device(st, &row_loss);
  if (rc != VITM_OK) return rc;
  // rows 0 .. B-1 of the array are rewritten in full by every call: nothing has to start at zero
  hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, logits, ld, labels, B, C, row_loss, dlogits, lddl);
  hipLaunchKernelGGL(cross_entropy_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)row_loss, B, loss);
  VITM_CHECK_LAUNCH("vitm_cross_entropy_f32_noalloc");
  return VITM_OK;
}
extern "C" int vitm_cross_entropy_noalloc_max_rows(void) { return CE_NOALLOC_MAX_ROWS; }

extern "C" int vitm_grad_clip_f32(float* grads, int64_t n, float max_norm, float* scratch, float* norm_and_coef,
                                  vitm_stream_t stream) {
  VITM_REQUIRE(grads && scratch && norm_and_coef, "vitm_grad_clip_f32: null pointer");
  VITM_REQUIRE(n > 0 && max_norm > 0.f, "vitm_grad_clip_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_PARTS), dim3(256), 0, st, grads, n, scratch);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, scratch, SUMSQ_PARTS, max_norm, norm_and_coef);
  hipLaunchKernelGGL(scale_by_kernel, dim3(stream_grid(n)), dim3(256), 0, st, grads, n, norm_and_coef + 1);
  VITM_CHECK_LAUNCH("vitm_grad_clip_f32");
  return VITM_OK;
}
extern "C" int vitm_grad_clip_scratch_floats(void) { return SUMSQ_PARTS; }

extern "C" int vitm_adamw_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, vitm_stream_t stream) {
  VITM_REQUIRE(p && g && exp_avg && exp_avg_sq, "vitm_adamw_f32: null pointer");
  VITM_REQUIRE(n > 0 && step > 0, "vitm_adamw_f32: n and step must be positive");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  VITM_CHECK_LAUNCH("vitm_adamw_f32");
  return VITM_OK;
}

extern "C" int vitm_adamw_f32_hyper(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper,
                                    vitm_stream_t stream) {
  VITM_REQUIRE(p && g && exp_avg && exp_avg_sq && hyper, "vitm_adamw_f32_hyper: null pointer");
  VITM_REQUIRE(n > 0, "vitm_adamw_f32_hyper: n must be positive");
  hipLaunchKernelGGL(adamw_hyper_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n, hyper);
  VITM_CHECK_LAUNCH("vitm_adamw_f32_hyper");
  return VITM_OK;
}
