// fp32 backward of the multi-head self-attention for gfx950 (training step; the reference trains in fp32).
//
// Given qkv, the forward output o, its log-sum-exp (log2 domain, from vitm_attention_f32_lse) and d_o, with
// s = scale * q k^T, p = softmax(s), o = p v:
//     delta_i = sum_d d_o[i,d] o[i,d]
//     d_v[j]  = sum_i p[i,j] d_o[i]
//     d_s[i,j] = p[i,j] (d_o[i] . v[j] - delta_i)
//     d_q[i]  = scale * sum_j d_s[i,j] k[j]          d_k[j] = scale * sum_i d_s[i,j] q[i]
// p is recomputed from q, k and the saved log-sum-exp: nothing of size N x N is stored.  Two kernels with the thread
// layout of the forward (four adjacent lanes share a row, each a quarter of head_dim):
//   attn_bwd_dq   a workgroup owns 64 queries and sweeps the keys (K, V chunks in LDS); also writes delta;
//   attn_bwd_dkv  a workgroup owns 64 keys and sweeps the queries (Q, d_o chunks + their lse and delta in LDS).
// Each output element is accumulated by one thread in a fixed order: deterministic, no atomics.
#include "common.h"

namespace {

constexpr int RPB = 64;      // rows (queries or keys) per workgroup
constexpr int SUB = 4;       // swept rows per step

__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, int64_t ldq, const float* __restrict__ o,
                                                          int64_t ldo, const float* __restrict__ d_o, int64_t lddo,
                                                          const float* __restrict__ lse, float* __restrict__ dqkv,
                                                          int64_t lddq, float* __restrict__ delta, int N, int H, float scale,
                                                          int kchunk, int nrb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = HD / 4, V4 = DP / 4, CPR = HD / 4;
  float* Ks = (float*)smem;
  float* Vs = Ks + (size_t)kchunk * HD;
  const int tid = threadIdx.x;
  const int ql = tid >> 2, part = tid & 3;
  const int bh = blockIdx.x / nrb, rb = blockIdx.x - bh * nrb;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const float sl2e = scale * 1.4426950408889634f;
  const float* base = qkv + (int64_t)b * N * ldq + h * HD;
  int qrow = rb * RPB + ql;
  const bool live = qrow < N;
  qrow = live ? qrow : N - 1;

  float q[DP], go[DP], dq[DP];
  float dl = 0.f;
  {
    const float* qp = base + (int64_t)qrow * ldq + part * DP;
    const float* op = o + ((int64_t)b * N + qrow) * ldo + h * HD + part * DP;
    const float* gp = d_o + ((int64_t)b * N + qrow) * lddo + h * HD + part * DP;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      q[d] = qp[d] * sl2e;
      go[d] = gp[d];
      dl = fmaf(go[d], op[d], dl);
      dq[d] = 0.f;
    }
  }
  dl = quad_sum(dl);
  const float L = lse[((int64_t)b * H + h) * N + qrow];

  for (int c0 = 0; c0 < N; c0 += kchunk) {
    if (c0 > 0) __syncthreads();
    const int rows = (N - c0) < kchunk ? (N - c0) : kchunk;
    const int rows_pad = (rows + SUB - 1) / SUB * SUB;
    for (int idx = tid; idx < rows_pad * CPR; idx += 256) {
      const int row = idx / CPR, ch = idx - row * CPR;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (row < rows) {
        const float* src = base + (int64_t)(c0 + row) * ldq + ch * 4;
        kv = *(const float4*)(src + D);
        vv = *(const float4*)(src + 2 * D);
      }
      *(float4*)(Ks + row * HD + ch * 4) = kv;
      *(float4*)(Vs + row * HD + ch * 4) = vv;
    }
    __syncthreads();
    for (int j0 = 0; j0 < rows_pad; j0 += SUB) {
#pragma unroll
      for (int jj = 0; jj < SUB; ++jj) {
        const float* kp = Ks + (j0 + jj) * HD + part * DP;
        const float* vp = Vs + (j0 + jj) * HD + part * DP;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const float4 kk = *(const float4*)(kp + 4 * v);
          const float4 vv = *(const float4*)(vp + 4 * v);
          s = fmaf(q[4 * v + 0], kk.x, s); s = fmaf(q[4 * v + 1], kk.y, s);
          s = fmaf(q[4 * v + 2], kk.z, s); s = fmaf(q[4 * v + 3], kk.w, s);
          dp = fmaf(go[4 * v + 0], vv.x, dp); dp = fmaf(go[4 * v + 1], vv.y, dp);
          dp = fmaf(go[4 * v + 2], vv.z, dp); dp = fmaf(go[4 * v + 3], vv.w, dp);
        }
        s = quad_sum(s);
        dp = quad_sum(dp);
        const float p = (j0 + jj < rows) ? __builtin_amdgcn_exp2f(s - L) : 0.f;   // keys past the end: nothing
        const float ds = p * (dp - dl);
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const float4 kk = *(const float4*)(kp + 4 * v);
          dq[4 * v + 0] = fmaf(ds, kk.x, dq[4 * v + 0]); dq[4 * v + 1] = fmaf(ds, kk.y, dq[4 * v + 1]);
          dq[4 * v + 2] = fmaf(ds, kk.z, dq[4 * v + 2]); dq[4 * v + 3] = fmaf(ds, kk.w, dq[4 * v + 3]);
        }
      }
    }
  }
  if (live) {
    float* dst = dqkv + ((int64_t)b * N + qrow) * lddq + h * HD + part * DP;
#pragma unroll
    for (int v = 0; v < V4; ++v)
      *(float4*)(dst + 4 * v) = make_float4(dq[4 * v] * scale, dq[4 * v + 1] * scale, dq[4 * v + 2] * scale, dq[4 * v + 3] * scale);
    if (part == 0) delta[((int64_t)b * H + h) * N + qrow] = dl;
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                           const float* __restrict__ d_o, int64_t lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int64_t lddq, int N, int H, float scale,
                                                           int qchunk, int nrb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = HD / 4, V4 = DP / 4, CPR = HD / 4;
  float* Qs = (float*)smem;
  float* Gs = Qs + (size_t)qchunk * HD;
  float* Ls = Gs + (size_t)qchunk * HD;   // lse per query of the chunk
  float* Ds = Ls + qchunk;                // delta per query
  const int tid = threadIdx.x;
  const int kl = tid >> 2, part = tid & 3;
  const int bh = blockIdx.x / nrb, rb = blockIdx.x - bh * nrb;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const float sl2e = scale * 1.4426950408889634f;
  const float* base = qkv + (int64_t)b * N * ldq + h * HD;
  int krow = rb * RPB + kl;
  const bool live = krow < N;
  krow = live ? krow : N - 1;

  float kk[DP], vv[DP], dk[DP], dv[DP];
  {
    const float* kp = base + (int64_t)krow * ldq + D + part * DP;
    const float* vp = base + (int64_t)krow * ldq + 2 * D + part * DP;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      kk[d] = kp[d] * sl2e;
      vv[d] = vp[d];
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
  }

  for (int c0 = 0; c0 < N; c0 += qchunk) {
    if (c0 > 0) __syncthreads();
    const int rows = (N - c0) < qchunk ? (N - c0) : qchunk;
    const int rows_pad = (rows + SUB - 1) / SUB * SUB;
    for (int idx = tid; idx < rows_pad * CPR; idx += 256) {
      const int row = idx / CPR, ch = idx - row * CPR;
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
      if (row < rows) {
        qv = *(const float4*)(base + (int64_t)(c0 + row) * ldq + ch * 4);
        gv = *(const float4*)(d_o + ((int64_t)b * N + c0 + row) * lddo + h * HD + ch * 4);
      }
      *(float4*)(Qs + row * HD + ch * 4) = qv;
      *(float4*)(Gs + row * HD + ch * 4) = gv;
    }
    for (int row = tid; row < rows_pad; row += 256) {
      const bool ok = row < rows;
      Ls[row] = ok ? lse[((int64_t)b * H + h) * N + c0 + row] : 1e30f;     // padded queries: p = exp2(-inf) = 0
      Ds[row] = ok ? delta[((int64_t)b * H + h) * N + c0 + row] : 0.f;
    }
    __syncthreads();
    for (int i0 = 0; i0 < rows_pad; i0 += SUB) {
#pragma unroll
      for (int ii = 0; ii < SUB; ++ii) {
        const float* qp = Qs + (i0 + ii) * HD + part * DP;
        const float* gp = Gs + (i0 + ii) * HD + part * DP;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const float4 qq = *(const float4*)(qp + 4 * v);
          const float4 gg = *(const float4*)(gp + 4 * v);
          s = fmaf(kk[4 * v + 0], qq.x, s); s = fmaf(kk[4 * v + 1], qq.y, s);
          s = fmaf(kk[4 * v + 2], qq.z, s); s = fmaf(kk[4 * v + 3], qq.w, s);
          dp = fmaf(vv[4 * v + 0], gg.x, dp); dp = fmaf(vv[4 * v + 1], gg.y, dp);
          dp = fmaf(vv[4 * v + 2], gg.z, dp); dp = fmaf(vv[4 * v + 3], gg.w, dp);
        }
        s = quad_sum(s);
        dp = quad_sum(dp);
        const float p = __builtin_amdgcn_exp2f(s - Ls[i0 + ii]);
        const float ds = p * (dp - Ds[i0 + ii]);
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const float4 qq = *(const float4*)(qp + 4 * v);
          const float4 gg = *(const float4*)(gp + 4 * v);
          dv[4 * v + 0] = fmaf(p, gg.x, dv[4 * v + 0]); dv[4 * v + 1] = fmaf(p, gg.y, dv[4 * v + 1]);
          dv[4 * v + 2] = fmaf(p, gg.z, dv[4 * v + 2]); dv[4 * v + 3] = fmaf(p, gg.w, dv[4 * v + 3]);
          dk[4 * v + 0] = fmaf(ds, qq.x, dk[4 * v + 0]); dk[4 * v + 1] = fmaf(ds, qq.y, dk[4 * v + 1]);
          dk[4 * v + 2] = fmaf(ds, qq.z, dk[4 * v + 2]); dk[4 * v + 3] = fmaf(ds, qq.w, dk[4 * v + 3]);
        }
      }
    }
  }
  if (live) {
    float* dkp = dqkv + ((int64_t)b * N + krow) * lddq + D + h * HD + part * DP;
    float* dvp = dqkv + ((int64_t)b * N + krow) * lddq + 2 * D + h * HD + part * DP;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      *(float4*)(dkp + 4 * v) = make_float4(dk[4 * v] * scale, dk[4 * v + 1] * scale, dk[4 * v + 2] * scale, dk[4 * v + 3] * scale);
      *(float4*)(dvp + 4 * v) = make_float4(dv[4 * v], dv[4 * v + 1], dv[4 * v + 2], dv[4 * v + 3]);
    }
  }
}

template <int HD>
int launch_bwd(const float* qkv, int64_t ldq, const float* o, int64_t ldo, const float* d_o, int64_t lddo, const float* lse,
               float* dqkv, int64_t lddq, float* delta, int B, int N, int H, float scale, hipStream_t st) {
  constexpr int CH_MAX = (6144 / HD) / SUB * SUB;       // K + V (or Q + d_o) chunk: 48 KiB, + lse/delta
  const int npad = (N + SUB - 1) / SUB * SUB;
  const int chunk = npad < CH_MAX ? npad : CH_MAX;
  const int nrb = (N + RPB - 1) / RPB;
  const dim3 grid((unsigned)((int64_t)nrb * B * H)), blk(256);
  {
    auto kfn = attn_bwd_dq_kernel<HD>;
    const int smem = 2 * chunk * HD * 4;
    VITM_LDS_OPTIN(kfn, 2 * CH_MAX * HD * 4, "vitm_attention_bwd_f32");
    hipLaunchKernelGGL(kfn, grid, blk, smem, st, qkv, ldq, o, ldo, d_o, lddo, lse, dqkv, lddq, delta, N, H, scale, chunk, nrb);
  }
  {
    auto kfn = attn_bwd_dkv_kernel<HD>;
    const int smem = 2 * chunk * HD * 4 + 2 * chunk * 4;
    VITM_LDS_OPTIN(kfn, 2 * CH_MAX * HD * 4 + 2 * CH_MAX * 4, "vitm_attention_bwd_f32");
    hipLaunchKernelGGL(kfn, grid, blk, smem, st, qkv, ldq, d_o, lddo, lse, delta, dqkv, lddq, N, H, scale, chunk, nrb);
  }
  VITM_CHECK_LAUNCH("vitm_attention_bwd_f32");
  return VITM_OK;
}

}  // namespace

extern "C" int vitm_attention_bwd_f32(const float* qkv, int64_t ldqkv, const float* o, int64_t ldo, const float* d_o,
                                      int64_t lddo, const float* lse, float* dqkv, int64_t lddqkv, float* delta, int B, int N,
                                      int H, int head_dim, float scale, vitm_stream_t stream) {
  VITM_REQUIRE(qkv && o && d_o && lse && dqkv && delta, "vitm_attention_bwd_f32: null pointer");
  VITM_REQUIRE(B > 0 && N > 0 && H > 0 && scale > 0.f, "vitm_attention_bwd_f32: bad arguments");
  const int D = H * head_dim;
  VITM_REQUIRE(ldqkv >= 3 * D && lddqkv >= 3 * D && ldo >= D && lddo >= D && ldqkv % 4 == 0 && lddqkv % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0,
               "vitm_attention_bwd_f32: leading dimensions too small or not multiples of 4");
  VITM_REQUIRE((((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15) == 0, "vitm_attention_bwd_f32: pointers must be 16-byte aligned");
  VITM_REQUIRE((int64_t)B * H * ((N + RPB - 1) / RPB) < (1ll << 31), "vitm_attention_bwd_f32: grid too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 32: return launch_bwd<32>(qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv, lddqkv, delta, B, N, H, scale, st);
    case 64: return launch_bwd<64>(qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv, lddqkv, delta, B, N, H, scale, st);
    case 96: return launch_bwd<96>(qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv, lddqkv, delta, B, N, H, scale, st);
    case 128: return launch_bwd<128>(qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv, lddqkv, delta, B, N, H, scale, st);
    default: break;
  }
  vitm_set_error("vitm_attention_bwd_f32: head_dim=%d unsupported (32, 64, 96, 128)", head_dim);
  return VITM_EUNSUPPORTED;
}
