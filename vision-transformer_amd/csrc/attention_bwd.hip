// Backward of the fused bf16 attention for gfx950 (mixed-precision training; backward of vitm/models/models.py:88-93).
//
// With S = q k^T, P = exp2(S * scale*log2e - lse) (lse from the forward), delta = rowsum(dO * O):
//   dV = P^T dO          dP = dO V^T          dS = P * (dP - delta)
//   dQ = scale * dS K    dK = scale * dS^T Q
// Two kernels, each a copy of the forward kernel's structure (a wave owns 32 rows whose fragments stay in registers,
// the other side streams through LDS in 32-row tiles, v_mfma_f32_32x32x16_bf16 everywhere, nothing atomic, the score
// matrix never leaves registers):
//   attn_bwd_dq_kernel   a wave owns 32 queries: S^T = K Q^T and dP^T = V dO^T put queries on the lanes, so lse and
//                        delta are one value per lane; dS^T, rounded to bf16, is the B operand of dQ^T += K^T dS^T with
//                        K^T fetched from the row-major K tile by ds_read_b64_tr_b16.  It also computes delta.
//   attn_bwd_dkv_kernel  a wave owns 32 keys: S = Q K^T and dP = dO V^T put keys on the lanes and queries on the
//                        accumulator rows (lse / delta come from LDS per row); P and dS, rounded to bf16, are the
//                        B operands of dV^T += dO^T P and dK^T += Q^T dS with dO^T / Q^T from transposing reads.
// Products are bf16 x bf16 with fp32 accumulation; softmax statistics, delta and the accumulators are fp32.
#include "common.h"

namespace {

constexpr int RB = 32;           // rows owned by a wave
constexpr int TT = 32;           // streamed rows per tile
constexpr int WAVES = 8;
constexpr int CHUNK_MAX = 256;   // streamed rows resident in LDS at a time

__device__ __forceinline__ u16x4_t tr_read(const char* lds_addr) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lds_addr));
  return __builtin_bit_cast(u16x4_t, v);
}

// 32 x 32 tile = rows(LDS, row-major, 32 of them from `rows`) . frag^T; accumulator register e is row
// (e & 3) + 8 (e >> 2) + 4 hh, the lane's column is r
template <int KS>
__device__ __forceinline__ f32x16_t tile_mm(const char* rows, const bf16x8_t (&frag)[KS]) {
  f32x16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const bf16x8_t a = *(const bf16x8_t*)(rows + s * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag[s], acc, 0, 0, 0);
  }
  return acc;
}

// the accumulator tile as the B operand of the next product: k-step s2 takes registers 8 s2 .. 8 s2 + 7
__device__ __forceinline__ void to_b_operand(const float (&v)[16], bf16x8_t (&pb)[2]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    u16x8_t t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = f2bf(v[8 * s2 + j]);
    pb[s2] = __builtin_bit_cast(bf16x8_t, t);
  }
}

// acc[d] += X^T . B for the 32-row tile at `tile` (row-major rows, stride RS): X^T fragments by transposing reads whose
// element order is the accumulator row order of to_b_operand
template <int DB, int RS>
__device__ __forceinline__ void tile_mm_t(const char* tile, const bf16x8_t (&pb)[2], f32x16_t (&acc)[DB]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const char* a0 = tile + (16 * s2) * RS + d * 64;
      const u16x4_t lo = tr_read(a0);
      const u16x4_t hi = tr_read(a0 + 8 * RS);
      u16x8_t t;
      t[0] = lo[0]; t[1] = lo[1]; t[2] = lo[2]; t[3] = lo[3];
      t[4] = hi[0]; t[5] = hi[1]; t[6] = hi[2]; t[7] = hi[3];
      acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, t), pb[s2], acc[d], 0, 0, 0);
    }
  }
}

// a wave's transposed accumulator (rows = head-dim, lanes = owned rows) to a row-major bf16 destination row
template <int DB>
__device__ __forceinline__ void store_rows(bf16_t* op, const f32x16_t (&acc)[DB], float mul, int hh) {
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = pack_bf2(acc[d][4 * g + 0] * mul, acc[d][4 * g + 1] * mul);
      w.y = pack_bf2(acc[d][4 * g + 2] * mul, acc[d][4 * g + 3] * mul);
      *(uint2*)(op + 32 * d + 8 * g + 4 * hh) = w;
    }
}

// stage rows [c0, c0 + rows) of two row-major sources into the two LDS tiles, zero rows up to rows_pad
template <int HD>
__device__ __forceinline__ void stage_two(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, char* As, char* Bs,
                                          int c0, int rows, int rows_pad, int tid) {
  constexpr int RS = HD * 2 + 16, CPR = HD / 8;
  for (int idx = tid; idx < rows_pad * CPR; idx += WAVES * 64) {
    const int row = idx / CPR, ch = idx - row * CPR;
    uint4 av = make_uint4(0, 0, 0, 0), bv = make_uint4(0, 0, 0, 0);
    if (row < rows) {
      av = *(const uint4*)(a + (int64_t)(c0 + row) * lda + ch * 8);
      bv = *(const uint4*)(b + (int64_t)(c0 + row) * ldb + ch * 8);
    }
    *(uint4*)(As + row * RS + ch * 16) = av;
    *(uint4*)(Bs + row * RS + ch * 16) = bv;
  }
}

// The same for sequences that fit one workgroup (rows_pad <= 256): every request goes out before the first one is
// waited for.  In the loop above a thread's iterations are one trip to memory each, and with one workgroup on the CU
// nothing else covers them (measured: loading and storing alone were 163 of a 270 us launch).
template <int HD>
__device__ __forceinline__ void stage_two_once(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, char* As, char* Bs,
                                               int rows, int rows_pad, int tid) {
  constexpr int RS = HD * 2 + 16, CPR = HD / 8, NIT = CPR / 2;   // 256 rows x CPR chunks over 512 threads
  uint4 av[NIT], bv[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int idx = tid + j * (WAVES * 64), row = idx / CPR, ch = idx - row * CPR;
    const int rr = row < rows ? row : rows - 1;   // rows past the end: any valid row, zeros are written instead
    av[j] = *(const uint4*)(a + (int64_t)rr * lda + ch * 8);
    bv[j] = *(const uint4*)(b + (int64_t)rr * ldb + ch * 8);
  }
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int idx = tid + j * (WAVES * 64), row = idx / CPR, ch = idx - row * CPR;
    if (row < rows_pad) {
      *(uint4*)(As + row * RS + ch * 16) = row < rows ? av[j] : make_uint4(0, 0, 0, 0);
      *(uint4*)(Bs + row * RS + ch * 16) = row < rows ? bv[j] : make_uint4(0, 0, 0, 0);
    }
  }
}

template <int HD>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                                 const bf16_t* __restrict__ o, int64_t ldo,
                                                                 const bf16_t* __restrict__ d_o, int64_t lddo,
                                                                 const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                                 int64_t lddq, float* __restrict__ delta, int N, int H,
                                                                 float scale, float sl2e, int chunk, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16, KS = HD / 16, DB = HD / 32;
  char* Ks = smem;
  char* Vs = smem + chunk * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.x / ngrp, grp = blockIdx.x - bh * ngrp;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const int q0 = (grp * WAVES + wave) * RB;
  const bool active = q0 < N;   // wave-uniform
  int qrow = q0 + r;
  qrow = qrow < N ? qrow : N - 1;   // rows past N work on a copy of the last row and are never stored
  const int64_t trow = (int64_t)b * N + qrow;

  const bf16_t* base = qkv + (int64_t)b * N * ldq + h * HD;
  bf16x8_t qf[KS], dof[KS];
  float dl = 0.f;
  {
    const bf16_t* qp = qkv + trow * ldq + h * HD + 8 * hh;
    const bf16_t* dp = d_o + trow * lddo + h * HD + 8 * hh;
    const bf16_t* op = o + trow * ldo + h * HD + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[s] = *(const bf16x8_t*)(qp + 16 * s);
      dof[s] = *(const bf16x8_t*)(dp + 16 * s);
      const u16x8_t dv = __builtin_bit_cast(u16x8_t, dof[s]);
      const u16x8_t ov = *(const u16x8_t*)(op + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = fmaf(bf2f(dv[j]), bf2f(ov[j]), dl);
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const int64_t sidx = ((int64_t)b * H + h) * N + qrow;
  if (active && hh == 0 && q0 + r < N) delta[sidx] = dl;
  const float Lq = lse[sidx];

  f32x16_t acc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (4 * hh + (li >> 2)) * RS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;

  for (int c0 = 0; c0 < N; c0 += chunk) {
    if (c0 > 0) __syncthreads();
    const int rows = (N - c0) < chunk ? (N - c0) : chunk;
    const int rows_pad = (rows + TT - 1) / TT * TT;
    stage_two<HD>(base + D, ldq, base + 2 * D, ldq, Ks, Vs, c0, rows, rows_pad, tid);
    __syncthreads();
    if (!active) continue;
    for (int kt = 0; kt < rows_pad / TT; ++kt) {
      const f32x16_t sacc = tile_mm<KS>(Ks + (kt * TT + r) * RS + hh * 16, qf);   // S^T: keys on rows
      const f32x16_t dpac = tile_mm<KS>(Vs + (kt * TT + r) * RS + hh * 16, dof);  // dP^T
      float ds[16];
      const bool ragged = kt * TT + TT > rows;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -Lq));
        if (ragged && kt * TT + (e & 3) + 8 * (e >> 2) + 4 * hh >= rows) p = 0.f;   // keys past N
        ds[e] = p * (dpac[e] - dl);
      }
      bf16x8_t sb[2];
      to_b_operand(ds, sb);
      tile_mm_t<DB, RS>(Ks + kt * TT * RS + tr_off, sb, acc);                    // dQ^T += K^T dS^T
    }
  }
  if (!active || q0 + r >= N) return;
  store_rows<DB>(dqkv + trow * lddq + h * HD, acc, scale, hh);
}

template <int HD>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                                  const bf16_t* __restrict__ d_o, int64_t lddo,
                                                                  const float* __restrict__ lse,
                                                                  const float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, int64_t lddq, int N, int H,
                                                                  float scale, float sl2e, int chunk, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16, KS = HD / 16, DB = HD / 32;
  char* Qs = smem;
  char* Os = smem + chunk * RS;
  float* Ls = (float*)(smem + 2 * chunk * RS);
  float* Ds = Ls + chunk;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.x / ngrp, grp = blockIdx.x - bh * ngrp;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const int k0 = (grp * WAVES + wave) * RB;
  const bool active = k0 < N;
  int krow = k0 + r;
  krow = krow < N ? krow : N - 1;
  const int64_t trow = (int64_t)b * N + krow;

  bf16x8_t kf[KS], vf[KS];
  {
    const bf16_t* kp = qkv + trow * ldq + D + h * HD + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kf[s] = *(const bf16x8_t*)(kp + 16 * s);
      vf[s] = *(const bf16x8_t*)(kp + D + 16 * s);
    }
  }
  f32x16_t dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dk[d][e] = 0.f;
      dv[d][e] = 0.f;
    }

  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (4 * hh + (li >> 2)) * RS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;
  const bf16_t* qbase = qkv + (int64_t)b * N * ldq + h * HD;
  const bf16_t* obase = d_o + (int64_t)b * N * lddo + h * HD;
  const float* lbase = lse + ((int64_t)b * H + h) * N;
  const float* dbase = delta + ((int64_t)b * H + h) * N;

  for (int c0 = 0; c0 < N; c0 += chunk) {
    if (c0 > 0) __syncthreads();
    const int rows = (N - c0) < chunk ? (N - c0) : chunk;
    const int rows_pad = (rows + TT - 1) / TT * TT;
    stage_two<HD>(qbase, ldq, obase, lddo, Qs, Os, c0, rows, rows_pad, tid);
    for (int i = tid; i < rows_pad; i += WAVES * 64) {
      Ls[i] = i < rows ? lbase[c0 + i] : 1e30f;    // queries past N: P = exp2(-inf) = 0
      Ds[i] = i < rows ? dbase[c0 + i] : 0.f;
    }
    __syncthreads();
    if (!active) continue;
    for (int qt = 0; qt < rows_pad / TT; ++qt) {
      const f32x16_t sacc = tile_mm<KS>(Qs + (qt * TT + r) * RS + hh * 16, kf);   // S: queries on rows, keys on lanes
      const f32x16_t dpac = tile_mm<KS>(Os + (qt * TT + r) * RS + hh * 16, vf);   // dP
      float p[16], ds[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 L4 = *(const float4*)(Ls + qt * TT + 8 * g + 4 * hh);
        const float4 D4 = *(const float4*)(Ds + qt * TT + 8 * g + 4 * hh);
        const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * g + i;
          p[e] = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -Lr[i]));
          ds[e] = p[e] * (dpac[e] - Dr[i]);
        }
      }
      bf16x8_t pb[2], sb[2];
      to_b_operand(p, pb);
      to_b_operand(ds, sb);
      tile_mm_t<DB, RS>(Os + qt * TT * RS + tr_off, pb, dv);    // dV^T += dO^T P
      tile_mm_t<DB, RS>(Qs + qt * TT * RS + tr_off, sb, dk);    // dK^T += Q^T dS
    }
  }
  if (!active || k0 + r >= N) return;
  bf16_t* op = dqkv + trow * lddq + D + h * HD;
  store_rows<DB>(op, dk, scale, hh);
  store_rows<DB>(op + D, dv, 1.0f, hh);
}

// Both of the above in one workgroup per (image, head), for sequences that fit one workgroup's 256 rows: every operand
// element comes from memory once and there is one launch.  Phase 1 is the dQ kernel (K, V staged in LDS, the wave's q /
// dO rows in registers); then each wave takes its own 32 K / V rows out of LDS as the fragments of phase 2, the waves put
// their q / dO rows (still in registers) and lse / delta where K and V were, and phase 2 is the dK / dV kernel on those.
// Rows past N inside the last tile are copies of row N - 1 with lse = 1e30: their probabilities are exactly 0.
// (Measured at 197 tokens, head dim 64, 3072 pairs: 260-270 us against 133 + 171 us for the two kernels.  Where the time
// goes: with the tile loops compiled out -- loads, delta, barriers and stores only -- the launch still takes 163 us: at one
// workgroup per CU (phase 2 needs ~170 registers) a pair's 100 KB in and 75 KB out, the trips to memory in front of them
// and the workgroup's start-up are not covered by anything, 13.6 us of a pair's 22.  What was tried on the compute
// side changes little for that reason: a one-pass form (score tiles computed once, dS handed through LDS to the idle
// eighth wave for dQ: 20 instead of 28 MFMAs and half the exp2 per tile pair) ran at 290 us, 270 with the dQ wave
// idle; unrolling the tile loops by two: no change; all staging requests issued before the first wait
// (stage_two_once): -1.3 %.  Two persistent forms that brought the next pair's K / V into a second LDS buffer under
// the compute -- through every thread's registers, or by the eighth wave with L2 touches for the row operands -- ran
// 1.5 % SLOWER than this kernel as written (register spills on the first, unexplained on the second); overlapping
// that memory time is what a further version has to get right.)
template <int HD>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                                    const bf16_t* __restrict__ o, int64_t ldo,
                                                                    const bf16_t* __restrict__ d_o, int64_t lddo,
                                                                    const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                                    int64_t lddq, float* __restrict__ delta, int N, int H,
                                                                    float scale, float sl2e, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16, KS = HD / 16, DB = HD / 32;
  char* As = smem;                   // K, then Q
  char* Bs = smem + npad * RS;       // V, then dO
  float* Ls = (float*)(smem + 2 * npad * RS);
  float* Ds = Ls + npad;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const int q0 = wave * RB;
  const bool active = q0 < N;   // wave-uniform
  int qrow = q0 + r;
  qrow = qrow < N ? qrow : N - 1;
  const int64_t trow = (int64_t)b * N + qrow;
  const bf16_t* base = qkv + (int64_t)b * N * ldq + h * HD;
  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (4 * hh + (li >> 2)) * RS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;
  const int ntile = npad / TT;

  bf16x8_t qf[KS], dof[KS];
  float dl = 0.f;
  {
    const bf16_t* qp = qkv + trow * ldq + h * HD + 8 * hh;
    const bf16_t* dp = d_o + trow * lddo + h * HD + 8 * hh;
    const bf16_t* op = o + trow * ldo + h * HD + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[s] = *(const bf16x8_t*)(qp + 16 * s);
      dof[s] = *(const bf16x8_t*)(dp + 16 * s);
      const u16x8_t dv = __builtin_bit_cast(u16x8_t, dof[s]);
      const u16x8_t ov = *(const u16x8_t*)(op + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = fmaf(bf2f(dv[j]), bf2f(ov[j]), dl);
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const int64_t sidx = ((int64_t)b * H + h) * N + qrow;
  if (active && hh == 0 && q0 + r < N) delta[sidx] = dl;
  const float Lq = lse[sidx];

  stage_two_once<HD>(base + D, ldq, base + 2 * D, ldq, As, Bs, N, npad, tid);
  __syncthreads();

  bf16x8_t kf[KS], vf[KS];
  if (active) {
    // ---- phase 1: dQ of the wave's 32 queries
    f32x16_t acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    for (int kt = 0; kt < ntile; ++kt) {
      const f32x16_t sacc = tile_mm<KS>(As + (kt * TT + r) * RS + hh * 16, qf);   // S^T: keys on rows
      const f32x16_t dpac = tile_mm<KS>(Bs + (kt * TT + r) * RS + hh * 16, dof);  // dP^T
      float ds[16];
      const bool ragged = kt * TT + TT > N;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -Lq));
        if (ragged && kt * TT + (e & 3) + 8 * (e >> 2) + 4 * hh >= N) p = 0.f;   // keys past N
        ds[e] = p * (dpac[e] - dl);
      }
      bf16x8_t sb[2];
      to_b_operand(ds, sb);
      tile_mm_t<DB, RS>(As + kt * TT * RS + tr_off, sb, acc);                    // dQ^T += K^T dS^T
    }
    if (q0 + r < N) store_rows<DB>(dqkv + trow * lddq + h * HD, acc, scale, hh);
    // the wave's 32 keys and values for phase 2 (rows past N are zero rows in LDS; their results are not stored)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kf[s] = *(const bf16x8_t*)(As + (q0 + r) * RS + (16 * s + 8 * hh) * 2);
      vf[s] = *(const bf16x8_t*)(Bs + (q0 + r) * RS + (16 * s + 8 * hh) * 2);
    }
  }
  __syncthreads();   // everyone is done with K and V in LDS
  if (active) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      *(bf16x8_t*)(As + (q0 + r) * RS + (16 * s + 8 * hh) * 2) = qf[s];
      *(bf16x8_t*)(Bs + (q0 + r) * RS + (16 * s + 8 * hh) * 2) = dof[s];
    }
    if (hh == 0) {
      Ls[q0 + r] = q0 + r < N ? Lq : 1e30f;    // queries past N: P = exp2(-inf) = 0
      Ds[q0 + r] = q0 + r < N ? dl : 0.f;
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- phase 2: dK and dV of the wave's 32 keys
  f32x16_t dk[DB], dv[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dk[d][e] = 0.f;
      dv[d][e] = 0.f;
    }
  for (int qt = 0; qt < ntile; ++qt) {
    const f32x16_t sacc = tile_mm<KS>(As + (qt * TT + r) * RS + hh * 16, kf);   // S: queries on rows, keys on lanes
    const f32x16_t dpac = tile_mm<KS>(Bs + (qt * TT + r) * RS + hh * 16, vf);   // dP
    float p[16], ds[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 L4 = *(const float4*)(Ls + qt * TT + 8 * g + 4 * hh);
      const float4 D4 = *(const float4*)(Ds + qt * TT + 8 * g + 4 * hh);
      const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = 4 * g + i;
        p[e] = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -Lr[i]));
        ds[e] = p[e] * (dpac[e] - Dr[i]);
      }
    }
    bf16x8_t pb[2], sb[2];
    to_b_operand(p, pb);
    to_b_operand(ds, sb);
    tile_mm_t<DB, RS>(Bs + qt * TT * RS + tr_off, pb, dv);    // dV^T += dO^T P
    tile_mm_t<DB, RS>(As + qt * TT * RS + tr_off, sb, dk);    // dK^T += Q^T dS
  }
  if (q0 + r >= N) return;
  bf16_t* op = dqkv + trow * lddq + D + h * HD;
  store_rows<DB>(op, dk, scale, hh);
  store_rows<DB>(op + D, dv, 1.0f, hh);
}

// The same two phases with FOUR waves per workgroup, each owning two 32-row blocks (w and w + 4): half the threads per
// workgroup, so two workgroups -- two (image, head) pairs -- share a CU (2 x 66 KiB of LDS, two waves per SIMD as before)
// and one pair's loads, trips to memory and stores run under the other pair's tile loops.
constexpr int FW = 4;
template <int HD>
__global__ __launch_bounds__(FW * 64, 2) void attn_bwd_fused4_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                                     const bf16_t* __restrict__ o, int64_t ldo,
                                                                     const bf16_t* __restrict__ d_o, int64_t lddo,
                                                                     const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                                     int64_t lddq, float* __restrict__ delta, int N, int H,
                                                                     float scale, float sl2e, int npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = HD * 2 + 16, KS = HD / 16, DB = HD / 32, CPR = HD / 8, NIT = CPR;   // 256 rows x CPR chunks over 256 threads
  char* As = smem;                   // K, then Q
  char* Bs = smem + npad * RS;       // V, then dO
  float* Ls = (float*)(smem + 2 * npad * RS);
  float* Ds = Ls + npad;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const bf16_t* base = qkv + (int64_t)b * N * ldq + h * HD;
  const int g16 = lane >> 4, li = lane & 15;
  const int tr_off = (4 * hh + (li >> 2)) * RS + (16 * (g16 & 1) + 4 * (li & 3)) * 2;
  const int ntile = npad / TT;

  // ---- requests: the two row blocks' q / dO / o rows and lse, then K and V for LDS -- all before the first wait
  bf16x8_t qf[2][KS], dof[2][KS];
  u16x8_t of[2][KS];
  float Lq[2];
  int64_t trow[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    int qrow = (wave + FW * sub) * RB + r;
    qrow = qrow < N ? qrow : N - 1;
    trow[sub] = (int64_t)b * N + qrow;
    const bf16_t* qp = qkv + trow[sub] * ldq + h * HD + 8 * hh;
    const bf16_t* dp = d_o + trow[sub] * lddo + h * HD + 8 * hh;
    const bf16_t* op = o + trow[sub] * ldo + h * HD + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[sub][s] = *(const bf16x8_t*)(qp + 16 * s);
      dof[sub][s] = *(const bf16x8_t*)(dp + 16 * s);
      of[sub][s] = *(const u16x8_t*)(op + 16 * s);
    }
    Lq[sub] = lse[((int64_t)b * H + h) * N + qrow];
  }
  {
    uint4 av[NIT], bv[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int idx = tid + j * (FW * 64), row = idx / CPR, ch = idx - row * CPR;
      const int rr = row < N ? row : N - 1;
      av[j] = *(const uint4*)(base + D + (int64_t)rr * ldq + ch * 8);
      bv[j] = *(const uint4*)(base + 2 * D + (int64_t)rr * ldq + ch * 8);
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int idx = tid + j * (FW * 64), row = idx / CPR, ch = idx - row * CPR;
      if (row < npad) {
        *(uint4*)(As + row * RS + ch * 16) = row < N ? av[j] : make_uint4(0, 0, 0, 0);
        *(uint4*)(Bs + row * RS + ch * 16) = row < N ? bv[j] : make_uint4(0, 0, 0, 0);
      }
    }
  }
  float dl[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u16x8_t dv = __builtin_bit_cast(u16x8_t, dof[sub][s]);
#pragma unroll
      for (int j = 0; j < 8; ++j) t = fmaf(bf2f(dv[j]), bf2f(of[sub][s][j]), t);
    }
    dl[sub] = t + __shfl_xor(t, 32, 64);
    const int q0 = (wave + FW * sub) * RB;
    if (hh == 0 && q0 + r < N) delta[((int64_t)b * H + h) * N + q0 + r] = dl[sub];
  }
  __syncthreads();

  // ---- phase 1: dQ of the wave's two query blocks
  bf16x8_t kf[2][KS], vf[2][KS];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int q0 = (wave + FW * sub) * RB;
    if (q0 >= N) continue;   // wave-uniform
    f32x16_t acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    for (int kt = 0; kt < ntile; ++kt) {
      const f32x16_t sacc = tile_mm<KS>(As + (kt * TT + r) * RS + hh * 16, qf[sub]);   // S^T: keys on rows
      const f32x16_t dpac = tile_mm<KS>(Bs + (kt * TT + r) * RS + hh * 16, dof[sub]);  // dP^T
      float ds[16];
      const bool ragged = kt * TT + TT > N;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -Lq[sub]));
        if (ragged && kt * TT + (e & 3) + 8 * (e >> 2) + 4 * hh >= N) p = 0.f;   // keys past N
        ds[e] = p * (dpac[e] - dl[sub]);
      }
      bf16x8_t sb[2];
      to_b_operand(ds, sb);
      tile_mm_t<DB, RS>(As + kt * TT * RS + tr_off, sb, acc);                    // dQ^T += K^T dS^T
    }
    if (q0 + r < N) store_rows<DB>(dqkv + trow[sub] * lddq + h * HD, acc, scale, hh);
    // the block's 32 keys and values for phase 2 (rows past N are zero rows in LDS; their results are not stored)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kf[sub][s] = *(const bf16x8_t*)(As + (q0 + r) * RS + (16 * s + 8 * hh) * 2);
      vf[sub][s] = *(const bf16x8_t*)(Bs + (q0 + r) * RS + (16 * s + 8 * hh) * 2);
    }
  }
  __syncthreads();   // everyone is done with K and V in LDS
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int q0 = (wave + FW * sub) * RB;
    if (q0 >= N) continue;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      *(bf16x8_t*)(As + (q0 + r) * RS + (16 * s + 8 * hh) * 2) = qf[sub][s];
      *(bf16x8_t*)(Bs + (q0 + r) * RS + (16 * s + 8 * hh) * 2) = dof[sub][s];
    }
    if (hh == 0) {
      Ls[q0 + r] = q0 + r < N ? Lq[sub] : 1e30f;    // queries past N: P = exp2(-inf) = 0
      Ds[q0 + r] = q0 + r < N ? dl[sub] : 0.f;
    }
  }
  __syncthreads();

  // ---- phase 2: dK and dV of the wave's two key blocks
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int k0 = (wave + FW * sub) * RB;
    if (k0 >= N) continue;
    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        dk[d][e] = 0.f;
        dv[d][e] = 0.f;
      }
    for (int qt = 0; qt < ntile; ++qt) {
      const f32x16_t sacc = tile_mm<KS>(As + (qt * TT + r) * RS + hh * 16, kf[sub]);   // S: queries on rows, keys on lanes
      const f32x16_t dpac = tile_mm<KS>(Bs + (qt * TT + r) * RS + hh * 16, vf[sub]);   // dP
      float p[16], ds[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 L4 = *(const float4*)(Ls + qt * TT + 8 * g + 4 * hh);
        const float4 D4 = *(const float4*)(Ds + qt * TT + 8 * g + 4 * hh);
        const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * g + i;
          p[e] = __builtin_amdgcn_exp2f(fmaf(sacc[e], sl2e, -Lr[i]));
          ds[e] = p[e] * (dpac[e] - Dr[i]);
        }
      }
      bf16x8_t pb[2], sb[2];
      to_b_operand(p, pb);
      to_b_operand(ds, sb);
      tile_mm_t<DB, RS>(Bs + qt * TT * RS + tr_off, pb, dv);    // dV^T += dO^T P
      tile_mm_t<DB, RS>(As + qt * TT * RS + tr_off, sb, dk);    // dK^T += Q^T dS
    }
    if (k0 + r < N) {
      bf16_t* op = dqkv + trow[sub] * lddq + D + h * HD;
      store_rows<DB>(op, dk, scale, hh);
      store_rows<DB>(op + D, dv, 1.0f, hh);
    }
  }
}

template <int HD>
int launch_bwd(const bf16_t* qkv, int64_t ldq, const bf16_t* o, int64_t ldo, const bf16_t* d_o, int64_t lddo,
               const float* lse, bf16_t* dqkv, int64_t lddq, float* delta, int B, int N, int H, float scale,
               hipStream_t st) {
  constexpr int RS = HD * 2 + 16;
  const int npad = (N + TT - 1) / TT * TT;
  const int chunk = npad < CHUNK_MAX ? npad : CHUNK_MAX;
  const int ngrp = (N + RB * WAVES - 1) / (RB * WAVES);
  const dim3 grid((unsigned)((int64_t)ngrp * B * H)), blk(WAVES * 64);
  const float sl2e = scale * 1.4426950408889634f;
#ifndef ATTN_BWD_NO_FUSED4
  if (ngrp == 1 && HD <= 64) {   // ... by four waves, two workgroups to a CU
    auto kfn = attn_bwd_fused4_kernel<(HD <= 64 ? HD : 64)>;
    const int lds = 2 * npad * RS + 2 * npad * 4;
    VITM_LDS_OPTIN(kfn, 2 * CHUNK_MAX * RS + 2 * CHUNK_MAX * 4, "vitm_attention_bwd");
    hipLaunchKernelGGL(kfn, dim3((unsigned)(B * H)), dim3(FW * 64), lds, st, qkv, ldq, o, ldo, d_o, lddo, lse, dqkv, lddq, delta,
                       N, H, scale, sl2e, npad);
    VITM_CHECK_LAUNCH("vitm_attention_bwd");
    return VITM_OK;
  }
#endif
  if (ngrp == 1) {   // the whole sequence in one workgroup: one kernel, every operand read once
    auto kfn = attn_bwd_fused_kernel<HD>;
    const int lds = 2 * npad * RS + 2 * npad * 4;
    VITM_LDS_OPTIN(kfn, 2 * CHUNK_MAX * RS + 2 * CHUNK_MAX * 4, "vitm_attention_bwd");
    hipLaunchKernelGGL(kfn, dim3((unsigned)(B * H)), blk, lds, st, qkv, ldq, o, ldo, d_o, lddo, lse, dqkv, lddq, delta, N, H,
                       scale, sl2e, npad);
    VITM_CHECK_LAUNCH("vitm_attention_bwd");
    return VITM_OK;
  }
  {
    auto kfn = attn_bwd_dq_kernel<HD>;
    VITM_LDS_OPTIN(kfn, 2 * CHUNK_MAX * RS, "vitm_attention_bwd");
    hipLaunchKernelGGL(kfn, grid, blk, 2 * chunk * RS, st, qkv, ldq, o, ldo, d_o, lddo, lse, dqkv, lddq, delta, N, H, scale,
                       sl2e, chunk, ngrp);
    VITM_CHECK_LAUNCH("vitm_attention_bwd");
  }
  {
    auto kfn = attn_bwd_dkv_kernel<HD>;
    VITM_LDS_OPTIN(kfn, 2 * CHUNK_MAX * RS + 2 * CHUNK_MAX * 4, "vitm_attention_bwd");
    hipLaunchKernelGGL(kfn, grid, blk, 2 * chunk * RS + 2 * chunk * 4, st, qkv, ldq, d_o, lddo, lse, delta, dqkv, lddq, N, H,
                       scale, sl2e, chunk, ngrp);
    VITM_CHECK_LAUNCH("vitm_attention_bwd");
  }
  return VITM_OK;
}

}  // namespace

extern "C" int vitm_attention_bwd(const void* qkv, int64_t ldqkv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                                  const float* lse, void* dqkv, int64_t lddqkv, float* delta, int B, int N, int H,
                                  int head_dim, float scale, vitm_stream_t stream) {
  VITM_REQUIRE(qkv && o && d_o && lse && dqkv && delta, "vitm_attention_bwd: null pointer");
  VITM_REQUIRE(B > 0 && N > 0 && H > 0 && scale > 0.f, "vitm_attention_bwd: bad arguments");
  const int D = H * head_dim;
  VITM_REQUIRE(ldqkv >= 3 * D && lddqkv >= 3 * D && ldo >= D && lddo >= D && ldqkv % 8 == 0 && lddqkv % 8 == 0 &&
                   ldo % 8 == 0 && lddo % 8 == 0,
               "vitm_attention_bwd: leading dimensions must cover the rows and be multiples of 8");
  VITM_REQUIRE((((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15) == 0,
               "vitm_attention_bwd: pointers must be 16-byte aligned");
  VITM_REQUIRE((int64_t)B * H * ((N + RB * WAVES - 1) / (RB * WAVES)) < (1ll << 31), "vitm_attention_bwd: grid too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *q = (const bf16_t*)qkv, *op = (const bf16_t*)o, *dop = (const bf16_t*)d_o;
  bf16_t* dq = (bf16_t*)dqkv;
  switch (head_dim) {
    case 32: return launch_bwd<32>(q, ldqkv, op, ldo, dop, lddo, lse, dq, lddqkv, delta, B, N, H, scale, st);
    case 64: return launch_bwd<64>(q, ldqkv, op, ldo, dop, lddo, lse, dq, lddqkv, delta, B, N, H, scale, st);
    case 96: return launch_bwd<96>(q, ldqkv, op, ldo, dop, lddo, lse, dq, lddqkv, delta, B, N, H, scale, st);
    case 128: return launch_bwd<128>(q, ldqkv, op, ldo, dop, lddo, lse, dq, lddqkv, delta, B, N, H, scale, st);
    default: break;
  }
  vitm_set_error("vitm_attention_bwd: head_dim=%d unsupported (32, 64, 96, 128)", head_dim);
  return VITM_EUNSUPPORTED;
}
