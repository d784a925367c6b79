// fp32 multi-head self-attention forward for gfx950 (the full-precision twin of attention.hip):
// out = softmax(scale * q k^T) v per (image, head), every operand and every product in fp32.
//
// This is the accuracy path (fp32 callers expect 1e-4 agreement with an fp32 reference), so it uses
// plain fp32 FMAs rather than the bf16 matrix cores.  One workgroup of 256 threads owns 64 queries of
// one (image, head); four adjacent lanes share a query, each holding a quarter of the head dimension of
// q and of the output accumulator.  K and V rows are staged through LDS in chunks and swept eight keys
// at a time with an online softmax: partial dot products are summed over the four lanes with two
// cross-lane adds, the running maximum / sum are rescaled once per eight keys.
#include "common.h"

namespace {

constexpr int QPB = 64;      // queries per workgroup
constexpr int SUB = 8;       // keys per online-softmax step

// HD is the head dimension the arithmetic runs at (a multiple of 16); hd <= HD is the real one (a multiple of 4).
// EXACT (hd == HD) compiles the padding away; otherwise columns hd .. HD-1 of q, K and V are zeros and the matching
// output columns are not stored.
template <int HD, bool EXACT>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                       float* __restrict__ out, int64_t ldo, int N, int H,
                                                       float sl2e, int kchunk, int nqb, float* __restrict__ lse, int hd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = HD / 4;      // head-dim slice per lane
  constexpr int V4 = DP / 4;      // float4 per slice
  constexpr int CPR = HD / 4;     // float4 per K/V row
  float* Ks = (float*)smem;
  float* Vs = Ks + (size_t)kchunk * HD;

  const int tid = threadIdx.x;
  const int ql = tid >> 2, part = tid & 3;
  const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
  const int b = bh / H, h = bh - b * H;
  if (EXACT) hd = HD;
  const int D = H * hd;
  const float* base = qkv + (int64_t)b * N * ldq + h * hd;

  int qrow = qb * QPB + ql;
  const bool store = qrow < N;
  qrow = store ? qrow : N - 1;   // rows past N work on a copy of the last row and are never stored

  float q[DP], o[DP];
  {
    const float* qp = base + (int64_t)qrow * ldq + part * DP;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const float4 t = (EXACT || part * DP + 4 * v < hd) ? *(const float4*)(qp + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
      q[4 * v + 0] = t.x * sl2e; q[4 * v + 1] = t.y * sl2e; q[4 * v + 2] = t.z * sl2e; q[4 * v + 3] = t.w * sl2e;
    }
  }
#pragma unroll
  for (int d = 0; d < DP; ++d) o[d] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  for (int c0 = 0; c0 < N; c0 += kchunk) {
    if (c0 > 0) __syncthreads();
    const int rows = (N - c0) < kchunk ? (N - c0) : kchunk;
    const int rows_pad = (rows + SUB - 1) / SUB * SUB;
    for (int idx = tid; idx < rows_pad * CPR; idx += 256) {
      const int row = idx / CPR, ch = idx - row * CPR;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (row < rows && (EXACT || ch * 4 < hd)) {
        const float* src = base + (int64_t)(c0 + row) * ldq + ch * 4;
        kv = *(const float4*)(src + D);
        vv = *(const float4*)(src + 2 * D);
      }
      *(float4*)(Ks + row * HD + ch * 4) = kv;
      *(float4*)(Vs + row * HD + ch * 4) = vv;
    }
    __syncthreads();

    for (int j0 = 0; j0 < rows_pad; j0 += SUB) {
      float s[SUB];
#pragma unroll
      for (int jj = 0; jj < SUB; ++jj) {
        const float* kp = Ks + (j0 + jj) * HD + part * DP;
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const float4 t = *(const float4*)(kp + 4 * v);
          acc = fmaf(q[4 * v + 0], t.x, acc);
          acc = fmaf(q[4 * v + 1], t.y, acc);
          acc = fmaf(q[4 * v + 2], t.z, acc);
          acc = fmaf(q[4 * v + 3], t.w, acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        s[jj] = (j0 + jj < rows) ? acc : -1e30f;   // block-uniform: keys past the end contribute nothing
      }
      float mt = s[0];
#pragma unroll
      for (int jj = 1; jj < SUB; ++jj) mt = fmaxf(mt, s[jj]);
      const float mn = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - mn);
      m_run = mn;
      float ps = 0.f;
#pragma unroll
      for (int jj = 0; jj < SUB; ++jj) {
        s[jj] = __builtin_amdgcn_exp2f(s[jj] - mn);
        ps += s[jj];
      }
      l_run = fmaf(l_run, alpha, ps);
#pragma unroll
      for (int d = 0; d < DP; ++d) o[d] *= alpha;
#pragma unroll
      for (int jj = 0; jj < SUB; ++jj) {
        const float* vp = Vs + (j0 + jj) * HD + part * DP;
#pragma unroll
        for (int v = 0; v < V4; ++v) {
          const float4 t = *(const float4*)(vp + 4 * v);
          o[4 * v + 0] = fmaf(s[jj], t.x, o[4 * v + 0]);
          o[4 * v + 1] = fmaf(s[jj], t.y, o[4 * v + 1]);
          o[4 * v + 2] = fmaf(s[jj], t.z, o[4 * v + 2]);
          o[4 * v + 3] = fmaf(s[jj], t.w, o[4 * v + 3]);
        }
      }
    }
  }

  if (store) {
    const float inv = 1.0f / l_run;
    float* op = out + ((int64_t)b * N + qrow) * ldo + h * hd + part * DP;
#pragma unroll
    for (int v = 0; v < V4; ++v)
      if (EXACT || part * DP + 4 * v < hd) *(float4*)(op + 4 * v) = make_float4(o[4 * v] * inv, o[4 * v + 1] * inv, o[4 * v + 2] * inv, o[4 * v + 3] * inv);
    // log2-domain log-sum-exp of the scaled scores, kept for the backward pass: p = exp2(s * scale * log2e - lse)
    if (lse != nullptr && part == 0) lse[((int64_t)b * H + h) * N + qrow] = m_run + __log2f(l_run);
  }
}

template <int HD, bool EXACT>
int launch_attn_f32(const float* qkv, int64_t ldq, float* out, int64_t ldo, int B, int N, int H, float scale,
                    float* lse, int hd, hipStream_t st) {
  int kchunk = (8192 / HD) / SUB * SUB;          // 64 KiB of LDS for K and V together
  const int npad = (N + SUB - 1) / SUB * SUB;
  if (npad < kchunk) kchunk = npad;
  const int smem = 2 * kchunk * HD * 4;
  auto kfn = attn_f32_kernel<HD, EXACT>;
  VITM_LDS_OPTIN(kfn, 2 * ((8192 / HD) / SUB * SUB) * HD * 4, "vitm_attention_f32");   // once, for the largest chunk
  const int nqb = (N + QPB - 1) / QPB;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((int64_t)nqb * B * H)), dim3(256), smem, st, qkv, ldq, out, ldo, N, H,
                     scale * 1.4426950408889634f, kchunk, nqb, lse, hd);
  VITM_CHECK_LAUNCH("vitm_attention_f32");
  return VITM_OK;
}

}  // namespace

static int attention_f32_impl(const float* qkv, int64_t ldqkv, float* out, int64_t ldo, int B, int N, int H,
                              int head_dim, float scale, float* lse, vitm_stream_t stream) {
  VITM_REQUIRE(qkv && out, "vitm_attention_f32: null pointer");
  VITM_REQUIRE(B > 0 && N > 0 && H > 0, "vitm_attention_f32: B, N, H must be positive");
  VITM_REQUIRE((int64_t)B * H * ((N + QPB - 1) / QPB) < (1ll << 31), "vitm_attention_f32: grid too large for one launch");
  const int D = H * head_dim;
  VITM_REQUIRE(ldqkv >= 3 * D && ldo >= D && ldqkv % 4 == 0 && ldo % 4 == 0,
               "vitm_attention_f32: ldqkv >= 3*H*head_dim, ldo >= H*head_dim, both multiples of 4");
  VITM_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "vitm_attention_f32: pointers must be 16-byte aligned");
  VITM_REQUIRE(scale > 0.f, "vitm_attention_f32: scale must be positive");
  hipStream_t st = (hipStream_t)stream;
  if (head_dim <= 0 || head_dim % 4 != 0 || head_dim > 128) {
    vitm_set_error("vitm_attention_f32: head_dim=%d unsupported (a multiple of 4, at most 128)", head_dim);
    return VITM_EUNSUPPORTED;
  }
  const int hd = head_dim;
  switch ((hd + 31) / 32) {   // the arithmetic runs at the next multiple of 32
    case 1: return hd == 32 ? launch_attn_f32<32, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                            : launch_attn_f32<32, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
    case 2: return hd == 64 ? launch_attn_f32<64, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                            : launch_attn_f32<64, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
    case 3: return hd == 96 ? launch_attn_f32<96, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                            : launch_attn_f32<96, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
    default: return hd == 128 ? launch_attn_f32<128, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                              : launch_attn_f32<128, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
  }
}

extern "C" int vitm_attention_f32(const float* qkv, int64_t ldqkv, float* out, int64_t ldo, int B, int N, int H,
                                  int head_dim, float scale, vitm_stream_t stream) {
  return attention_f32_impl(qkv, ldqkv, out, ldo, B, N, H, head_dim, scale, nullptr, stream);
}

extern "C" int vitm_attention_f32_lse(const float* qkv, int64_t ldqkv, float* out, int64_t ldo, float* lse, int B, int N,
                                      int H, int head_dim, float scale, vitm_stream_t stream) {
  VITM_REQUIRE(lse, "vitm_attention_f32_lse: null lse");
  return attention_f32_impl(qkv, ldqkv, out, ldo, B, N, H, head_dim, scale, lse, stream);
}
