// Fused multi-head self-attention forward for gfx950: out = softmax(scale * q k^T) v per (image, head).
//
// One workgroup (8 waves) per (image, head) and group of 8 query blocks; one wave owns a 32-query
// block.  K and V rows of the head are staged once per workgroup into LDS (chunks of up to 256 keys)
// and every wave sweeps them in 32-key tiles with an online softmax:
//
//   S^T = K_tile . Q^T      v_mfma_f32_32x32x16_bf16, keys on the accumulator rows, queries on the lanes,
//                           so each lane owns one query column: row max / row sum need no cross-lane
//                           work beyond one exchange between the two 32-lane halves;
//   P^T = exp2(S^T * scale*log2e - m)   in registers; the f32 accumulator tile, converted pairwise to
//                           bf16, already IS the B operand of the next product (its k order is the
//                           accumulator's row order);
//   O^T += V^T_tile . P^T   V^T fragments come from row-major V rows in LDS through
//                           ds_read_b64_tr_b16 (hardware transposing read), fetched in that same k order.
//
// LDS rows are padded by 16 bytes (row = head_dim*2 + 16 bytes) which makes the 16-byte K fragment
// reads bank-conflict free.  The score matrix never leaves registers.
#include "common.h"

namespace {

constexpr int QB = 32;          // queries per wave
constexpr int KT = 32;          // keys per tile
constexpr int WAVES = 8;
constexpr int KCHUNK_MAX = 256;  // keys resident in LDS at a time

__device__ __forceinline__ u16x4_t tr_read(const char* lds_addr) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lds_addr));
  return __builtin_bit_cast(u16x4_t, v);
}

template <int HD>
__global__ __launch_bounds__(WAVES * 64) void attn_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                          bf16_t* __restrict__ out, int64_t ldo, int N,
                                                          int H, float sl2e, int kchunk, int nqg,
                                                          float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16;     // LDS row stride in bytes
  constexpr int CPR = HD / 8;         // 16-byte chunks per row
  constexpr int KS = HD / 16;         // k-steps of S^T = K.Q^T
  constexpr int DB = HD / 32;         // 32-wide output blocks of O^T
  char* Ks = smem;
  char* Vs = smem + kchunk * RS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.x / nqg, qg = blockIdx.x - bh * nqg;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const int q0 = (qg * WAVES + wave) * QB;
  const bool active = q0 < N;  // wave-uniform

  const bf16_t* base = qkv + (int64_t)b * N * ldq + h * HD;

  // Q fragments (B operand of S^T = K.Q^T): lane (r, hh) holds Q[q0 + r][16 s + 8 hh + 0..7]
  bf16x8_t qf[KS];
  {
    int qrow = q0 + r;
    qrow = qrow < N ? qrow : N - 1;  // rows past N are computed on a copy of the last row and never stored
    const bf16_t* qp = base + (int64_t)qrow * ldq + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *(const bf16x8_t*)(qp + 16 * s);
  }

  f32x16_t o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // transposing-read lane addressing: 16-lane group g16 reads keys kb..kb+3 x 16 columns
  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (4 * hh + (li >> 2)) * RS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;

  for (int c0 = 0; c0 < N; c0 += kchunk) {
    if (c0 > 0) __syncthreads();  // everyone is done with the previous chunk
    const int rows = (N - c0) < kchunk ? (N - c0) : kchunk;
    const int rows_pad = (rows + KT - 1) / KT * KT;
    for (int idx = tid; idx < rows_pad * CPR; idx += WAVES * 64) {
      const int row = idx / CPR, ch = idx - row * CPR;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (row < rows) {
        const bf16_t* src = base + (int64_t)(c0 + row) * ldq + ch * 8;
        kv = *(const uint4*)(src + D);
        vv = *(const uint4*)(src + 2 * D);
      }
      *(uint4*)(Ks + row * RS + ch * 16) = kv;
      *(uint4*)(Vs + row * RS + ch * 16) = vv;
    }
    __syncthreads();
    if (!active) continue;

    const int ntiles = rows_pad / KT;
    for (int kt = 0; kt < ntiles; ++kt) {
      // ---- S^T tile: keys on rows, queries on lanes
      f32x16_t sacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
      const char* kp = Ks + (kt * KT + r) * RS + hh * 16;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8_t a = *(const bf16x8_t*)(kp + s * 32);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], sacc, 0, 0, 0);
      }
      if (kt * KT + KT > rows) {  // ragged last tile: keys past N contribute nothing
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * KT + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= rows) sacc[e] = -1e30f;
        }
      }
      // ---- online softmax (one query per lane; the two lane halves hold different keys)
      float mt = sacc[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) mt = fmaxf(mt, sacc[e]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * sl2e);
      const float mb = mn * sl2e;
      m_run = mn;
      float ps = 0.f;
      float pr[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        pr[e] = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -mb));
        ps += pr[e];
      }
      l_run = fmaf(l_run, alpha, ps);
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
      // P^T as the B operand: k-step s2 takes accumulator registers 8 s2 .. 8 s2 + 7
      bf16x8_t pb[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u16x8_t t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = f2bf(pr[8 * s2 + j]);
        pb[s2] = __builtin_bit_cast(bf16x8_t, t);
      }
      // ---- O^T += V^T . P^T ; element j of the A fragment must be key 16 s2 + 8 (j>>2) + 4 hh + (j&3)
      const char* vp = Vs + kt * KT * RS + tr_off;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const char* a0 = vp + (16 * s2) * RS + d * 64;
          const u16x4_t lo = tr_read(a0);
          const u16x4_t hi = tr_read(a0 + 8 * RS);
          u16x8_t t;
          t[0] = lo[0]; t[1] = lo[1]; t[2] = lo[2]; t[3] = lo[3];
          t[4] = hi[0]; t[5] = hi[1]; t[6] = hi[2]; t[7] = hi[3];
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, t), pb[s2], o[d], 0, 0, 0);
        }
      }
    }
  }

  if (!active) return;
  const float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l;
  // log2-domain log-sum-exp of the scaled scores (training): p = exp2(s * scale * log2e - lse)
  if (lse != nullptr && hh == 0 && q0 + r < N) lse[((int64_t)b * H + h) * N + q0 + r] = m_run * sl2e + __log2f(l);
  if (q0 + r < N) {
    bf16_t* op = out + ((int64_t)b * N + q0 + r) * ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack_bf2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
        w.y = pack_bf2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        *(uint2*)(op + 32 * d + 8 * g + 4 * hh) = w;
      }
  }
}

template <int HD>
int launch_attn(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale,
                float* lse, hipStream_t st) {
  const int npad = (N + KT - 1) / KT * KT;
  const int kchunk = npad < KCHUNK_MAX ? npad : KCHUNK_MAX;
  const int smem = 2 * kchunk * (HD * 2 + 16);
  auto kfn = attn_kernel<HD>;
  VITM_LDS_OPTIN(kfn, 2 * KCHUNK_MAX * (HD * 2 + 16), "vitm_attention");   // once, for the largest chunk
  const int nqb = (N + QB - 1) / QB;
  const int nqg = (nqb + WAVES - 1) / WAVES;
  dim3 grid((unsigned)((int64_t)nqg * B * H));
  hipLaunchKernelGGL(kfn, grid, dim3(WAVES * 64), smem, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                     scale * 1.4426950408889634f, kchunk, nqg, lse);
  VITM_CHECK_LAUNCH("vitm_attention");
  return VITM_OK;
}

}  // namespace

static int attention_impl(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int N, int H,
                          int head_dim, float scale, float* lse, vitm_stream_t stream) {
  VITM_REQUIRE(qkv && out, "vitm_attention: null pointer");
  VITM_REQUIRE(B > 0 && N > 0 && H > 0, "vitm_attention: B, N, H must be positive");
  VITM_REQUIRE((int64_t)B * H * ((N + QB * WAVES - 1) / (QB * WAVES)) < (1ll << 31), "vitm_attention: grid too large for one launch");
  const int D = H * head_dim;
  VITM_REQUIRE(ldqkv >= 3 * D && ldo >= D && ldqkv % 8 == 0 && ldo % 8 == 0,
               "vitm_attention: ldqkv >= 3*H*head_dim, ldo >= H*head_dim, both multiples of 8");
  VITM_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "vitm_attention: pointers must be 16-byte aligned");
  VITM_REQUIRE(scale > 0.f, "vitm_attention: scale must be positive");
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 32: return launch_attn<32>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, st);
    case 64: return launch_attn<64>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, st);
    case 96: return launch_attn<96>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, st);
    case 128: return launch_attn<128>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, st);
    default: break;
  }
  vitm_set_error("vitm_attention: head_dim=%d unsupported (32, 64, 96, 128)", head_dim);
  return VITM_EUNSUPPORTED;
}

extern "C" int vitm_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int N, int H,
                              int head_dim, float scale, vitm_stream_t stream) {
  return attention_impl(qkv, ldqkv, out, ldo, B, N, H, head_dim, scale, nullptr, stream);
}

extern "C" int vitm_attention_lse(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, float* lse, int B, int N, int H,
                                  int head_dim, float scale, vitm_stream_t stream) {
  VITM_REQUIRE(lse, "vitm_attention_lse: null lse");
  return attention_impl(qkv, ldqkv, out, ldo, B, N, H, head_dim, scale, lse, stream);
}
