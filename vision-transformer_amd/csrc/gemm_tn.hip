// Weight-gradient GEMM for gfx950: out[n, k] = sum_m a[m, n] * b[m, k]  (dW = dy^T x of nn.Linear's backward), bf16
// operands, fp32 accumulation and fp32 output.
//
// Both operands are row-major activations whose ROWS are the reduction index, the opposite of what an MFMA fragment
// wants (eight consecutive reduction elements per lane).  Tiles of 64 reduction rows x 128 columns of each operand are
// staged row-major in LDS (coalesced 16-byte chunks, rows padded by 64 bytes) and the fragments are fetched with
// ds_read_b64_tr_b16, the transposing LDS read: no transposed copy of an activation is ever written to HBM.
//
// The output is small (a weight matrix) and the reduction is long (every token of the batch), so the reduction is split
// over workgroup slices: each slice writes its own fp32 partial tile and a second kernel adds the slices in a
// fixed order (deterministic; no atomics).  Workgroup = 8 waves on a 256 x 256 output tile (128 x 64 per wave as 4 x 2
// v_mfma_f32_32x32x16_bf16 tiles) or 4 waves on 128 x 128; the global loads of the next 64-row step are in flight while the current one is
// multiplied (register staging, two LDS buffers, one barrier per step).
#include "common.h"

namespace {

constexpr int TN_BM = 64;             // reduction rows per step
// LDS row strides are (tile row bytes) + 64, i.e. 64 mod 256: the four rows a transposing read touches per half-wave
// (4 x 64 bytes) then fall in four different quarters of the 64 banks (SQ_LDS_BANK_CONFLICT = 0)
constexpr int TN_MIN_ROWS = 512;      // reduction rows a slice should at least own
#ifndef TN_DEPTH
#define TN_DEPTH 2                    // 64-row steps of operand loads in flight per thread
#endif

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ u16x4_t tn_tr_read(const char* lds_addr) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lds_addr));
  return __builtin_bit_cast(u16x4_t, v);
}

template <int RS>
__device__ __forceinline__ bf16x8_t tn_frag(const char* a0) {
  const u16x4_t lo = tn_tr_read(a0);
  const u16x4_t hi = tn_tr_read(a0 + 8 * RS);
  u16x8_t t;
  t[0] = lo[0]; t[1] = lo[1]; t[2] = lo[2]; t[3] = lo[3];
  t[4] = hi[0]; t[5] = hi[1]; t[6] = hi[2]; t[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, t);
}

// BN_ x BK_ = output tile of a workgroup, FI = 32-row fragments per wave (a wave owns 32 FI x 64 of the tile):
//   128 x 128, FI 2: 4 waves as 2 x 2, two workgroups per CU (small outputs)
//   256 x 256, FI 4: 8 waves as 2 x 4, one workgroup per CU -- half the operand bytes per MAC.  Operands reach a CU at
//                    15-20 B/clk whatever the tile, prefetch depth or number of resident workgroups (measured), so the
//                    MACs per operand byte of the tile set the speed: 64 -> ~650 TFLOP/s, 128 -> ~1 PFLOP/s.
template <int BN_, int BK_, int FI, int DEPTH>
__global__ __launch_bounds__((BN_ / (32 * FI)) * (BK_ / 64) * 64, BN_ == 128 ? 2 : 1) void gemm_tn_kernel(
    const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b, int64_t ldb, float* __restrict__ dst,
    int64_t ldd, int64_t slice_stride, float* __restrict__ csum, const float* __restrict__ row_scale, int M, int N,
    int K, int rows_per_slice, int tiles_k, int nslices, int npairs) {
  constexpr int WK_ = BK_ / 64;                      // waves along the output columns
  constexpr int NT_ = (BN_ / (32 * FI)) * WK_ * 64;  // threads
  constexpr int RSA = BN_ * 2 + 64, RSB = BK_ * 2 + 64;     // LDS row strides (bytes): 64 mod 256
  constexpr int TILE_A = TN_BM * RSA, TILE_B = TN_BM * RSB, STAGE = TILE_A + TILE_B;
  constexpr int CA = BN_ / 8, CB = BK_ / 8;          // 16-byte chunks per tile row
  constexpr int RPA = NT_ / CA, RPB = NT_ / CB;      // rows covered by one pass of the workgroup
  constexpr int JA = TN_BM / RPA, JB = TN_BM / RPB;  // passes (loads per thread) per step
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][a tile, b tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // Workgroups are dealt to the 8 XCDs round-robin by their linear index.  XCD x takes the x-th eighth of the
  // (slice, tile) pairs in slice-major order, so the tiles of one slice run on one XCD (two at a boundary) and share
  // its L2: each 64-row step of a slice's rows is fetched from HBM about once instead of once per output tile, for
  // any slice count.
  const int per_xcd = (npairs + 7) >> 3;
  const int pair = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || pair >= npairs) return;   // block-uniform: the padding of the last eighth
  const int tiles = npairs / nslices;
  const int slice = pair / tiles, tile = pair - slice * tiles;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * BN_, k0 = tk * BK_;
  const int m_begin = slice * rows_per_slice;
  int m_end = m_begin + rows_per_slice;
  m_end = m_end < M ? m_end : M;
  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;

  // staging: 16-byte chunks fetched with raw buffer loads whose descriptor ends at the slice's last row: rows past
  // the slice (and the chunks of threads whose columns lie past N / K, which are given an offset beyond any
  // descriptor) come back as zeros from the range check, with no branch, no clamp and no 64-bit address arithmetic
  // in the loop.
  const int rowa = tid / CA, cha = tid % CA, rowb = tid / CB, chb = tid % CB;
  const int rows_here = m_end > m_begin ? m_end - m_begin : 0;
  const __amdgpu_buffer_rsrc_t ra_desc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a + (int64_t)m_begin * lda), 0, (int)((int64_t)rows_here * lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_desc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(b + (int64_t)m_begin * ldb), 0, (int)((int64_t)rows_here * ldb * 2), 0x00020000);
  const unsigned OOB = 0x80000000u;
  unsigned va = n0 + cha * 8 < N ? (unsigned)((rowa * lda + n0 + cha * 8) * 2) : OOB;
  unsigned vb = k0 + chb * 8 < K ? (unsigned)((rowb * ldb + k0 + chb * 8) * 2) : OOB;
  const unsigned ja = (unsigned)(RPA * lda * 2), jb = (unsigned)(RPB * ldb * 2);
  const int soffa = rowa * RSA + cha * 16, soffb = TILE_A + rowb * RSB + chb * 16;
  struct Stage {   // one thread's share of a 64-row step of both operands
    u32x4_t a[JA], b[JB];
  };
  auto fetch = [&]() {   // the next 64 rows; advances the offsets
    Stage st;
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      st.a[j] = __builtin_amdgcn_raw_buffer_load_b128(ra_desc, va, 0, 0);
      va += ja;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      st.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rb_desc, vb, 0, 0);
      vb += jb;
    }
    return st;
  };
  auto stash = [&](int buf, const Stage st) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < JA; ++j) *(u32x4_t*)(base + soffa + j * RPA * RSA) = st.a[j];
#pragma unroll
    for (int j = 0; j < JB; ++j) *(u32x4_t*)(base + soffb + j * RPB * RSB) = st.b[j];
  };

  f32x16_t acc[FI][2];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int g16 = lane >> 4, li = lane & 15;
  const int trow = 4 * hh + (li >> 2), tcol = (16 * (g16 & 1) + 4 * (li & 3)) * 2;
  const int wn = wave / WK_, wk = wave % WK_;
  const int a_off = trow * RSA + tcol + (32 * FI * wn) * 2, b_off = TILE_A + trow * RSB + tcol + (64 * wk) * 2;
  // column sums of a (the bias gradient) ride along on the workgroups of the first column tile: one more product
  // per a fragment, against a fragment of ones -- every column of that accumulator is sum_m a[m, n].  The waves of a
  // wave row share the duty, two fragments each (wave column wk takes fragments 2 wk, 2 wk + 1).
  const bool want_csum = csum != nullptr && tk == 0 && wk < FI / 2;   // wave-uniform
  f32x16_t cacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) cacc[i][e] = 0.f;
  u16x8_t ones_u;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones_u[j] = 0x3F80;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);
  auto compute = [&](int buf) {
    const char* base = smem + buf * STAGE;
#pragma unroll
    for (int s2 = 0; s2 < TN_BM / 16; ++s2) {
      const char* ta = base + 16 * s2 * RSA + a_off;
      const char* tb = base + 16 * s2 * RSB + b_off;
      bf16x8_t af[FI];
#pragma unroll
      for (int i = 0; i < FI; ++i) af[i] = tn_frag<RSA>(ta + 64 * i);
      const bf16x8_t b0 = tn_frag<RSB>(tb), b1 = tn_frag<RSB>(tb + 64);
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], b1, acc[i][1], 0, 0, 0);
      }
      if (want_csum) {
        if (FI == 2 || wk == 0) {
          cacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], ones, cacc[0], 0, 0, 0);
          cacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], ones, cacc[1], 0, 0, 0);
        } else {
          cacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[FI - 2], ones, cacc[0], 0, 0, 0);
          cacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[FI - 1], ones, cacc[1], 0, 0, 0);
        }
      }
    }
  };

  // LDS holds the current step (two buffers); registers hold the next DEPTH steps, so a load has DEPTH steps of
  // multiplying to arrive in (the operands come from L2 / the Infinity Cache at ~1.5 us under this load: with one
  // step of cover the loop ran at the speed of the loads alone).  Steps past the slice load zeros and are never used.
  if (nsteps > 0) {   // block-uniform
    Stage R[DEPTH];
    R[0] = fetch();
    stash(0, R[0]);
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) R[u] = fetch();      // R[u] <- step u + 1
    __syncthreads();
    // whole groups of DEPTH steps, no branch inside a group (the wait counts then stay exact); the up to DEPTH - 1
    // extra steps multiply zeros
    for (int step = 0; step < nsteps; step += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        compute((step + u) & 1);
        stash((step + u + 1) & 1, R[u]);                 // step + u + 1
        R[u] = fetch();                                  // step + u + 1 + DEPTH
        __syncthreads();
      }
    }
  }

  if (want_csum && r == 0) {   // one lane per row pair half: lane column 0 of both halves
    float* co = csum + (int64_t)slice * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + 32 * FI * wn + 32 * (2 * wk + i) + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (n < N) co[n] = cacc[i][e] * (row_scale ? row_scale[n] : 1.0f);
      }
  }
  float* out = dst + (int64_t)slice * slice_stride;
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + 64 * wk + 32 * j + r;
      if (kc >= K) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + 32 * FI * wn + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (n < N) out[(int64_t)n * ldd + kc] = acc[i][j][e] * (row_scale ? row_scale[n] : 1.0f);
      }
    }
}

// out[n, k] = sum over slices, in slice order; the same for the column sums when there are any
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int64_t slice_stride, int slices,
                                                        float* __restrict__ out, int64_t ldo, int N, int K,
                                                        const float* __restrict__ cpart, float* __restrict__ cout) {
  const int64_t total = (int64_t)N * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = part[i];
    for (int sl = 1; sl < slices; ++sl) s += part[i + sl * slice_stride];
    const int64_t n = i / K;
    out[n * ldo + (i - n * K)] = s;
  }
  if (cout != nullptr)
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
      float s = cpart[n];
      for (int sl = 1; sl < slices; ++sl) s += cpart[n + (int64_t)sl * N];
      cout[n] = s;
    }
}

struct TnPlan {
  int bn, bk, tiles_n, tiles_k, slices, rows_per_slice;   // bn = bk = 128 or 256
};
// Tile size and slice count.  256 x 256 tiles when the output is large enough; the slice count minimises a simple time
// model: rounds of resident workgroups x (steps
// per slice + a fixed per-workgroup cost) plus the fp32 partial tiles written and read back.
TnPlan tn_plan(int M, int N, int K) {
  TnPlan p;
  p.bn = p.bk = (N >= 512 && K >= 256) ? 256 : 128;
  p.tiles_n = (N + p.bn - 1) / p.bn;
  p.tiles_k = (K + p.bk - 1) / p.bk;
  const int64_t tiles = (int64_t)p.tiles_n * p.tiles_k;
  const int ncu = vitm_cu_count();
  const int64_t slots = (int64_t)ncu * (p.bn == 128 ? 2 : 1);
  const int64_t smax = (M + TN_MIN_ROWS - 1) / TN_MIN_ROWS;
  int best = 1;
  {
    const double step_us = p.bn == 128 ? 0.75 : 2.1;                     // one 64-row step of one workgroup
    const double part_us = 8.0 * (double)N * K / 4.0e6;                 // one slice's partial tile: write + read at 4 TB/s
    double best_cost = 1e30;
    for (int s = 1; s <= 64 && s <= smax; ++s) {
      const int64_t rounds = (tiles * s + slots - 1) / slots;
      const int64_t steps = ((M + s - 1) / s + TN_BM - 1) / TN_BM;
      const double cost = (double)rounds * ((double)steps * step_us + 4.0) + (s > 1 ? (double)s * part_us : 0.0);
      if (cost < best_cost) {
        best_cost = cost;
        best = s;
      }
    }
  }
  int rps = (M + best - 1) / best;
  rps = (rps + TN_BM - 1) / TN_BM * TN_BM;
  p.rows_per_slice = rps;
  p.slices = best;                               // trailing slices may own no rows: they write zeros
  return p;
}

template <int BN_, int BK_, int FI>
int tn_launch(const TnPlan& p, const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, float* dst, int64_t ldd,
              int64_t slice_stride, float* csum, const float* row_scale, int M, int N, int K, hipStream_t st) {
  constexpr int SMEM_ = 2 * TN_BM * ((BN_ * 2 + 64) + (BK_ * 2 + 64));
  constexpr int DEPTH_ = BN_ == 256 ? 1 : TN_DEPTH;   // the 256 x 256 tile's accumulators leave registers for one step
  auto kfn = gemm_tn_kernel<BN_, BK_, FI, DEPTH_>;
  VITM_LDS_OPTIN(kfn, SMEM_, "vitm_gemm_tn");
  const int npairs = p.tiles_n * p.tiles_k * p.slices;
  const dim3 grid((unsigned)(8 * ((npairs + 7) / 8)));
  hipLaunchKernelGGL(kfn, grid, dim3((BN_ / (32 * FI)) * (BK_ / 64) * 64), SMEM_, st, a, lda, b, ldb, dst, ldd, slice_stride, csum, row_scale, M,
                     N, K, p.rows_per_slice, p.tiles_k, p.slices, npairs);
  VITM_CHECK_LAUNCH("vitm_gemm_tn");
  return VITM_OK;
}

}  // namespace

extern "C" int64_t vitm_gemm_tn_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const TnPlan p = tn_plan(M, N, K);
  return p.slices > 1 ? (int64_t)p.slices * ((int64_t)N * K + N) : 0;
}

extern "C" int vitm_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t ldo,
                            float* colsum_out, const float* row_scale, int M, int N, int K, float* workspace,
                            int64_t workspace_floats, vitm_stream_t stream) {
  VITM_REQUIRE(a && b && out, "vitm_gemm_tn: null pointer");
  VITM_REQUIRE(M > 0 && N > 0 && K > 0, "vitm_gemm_tn: M, N, K must be positive");
  VITM_REQUIRE(lda >= N && ldb >= K && ldo >= K && lda % 8 == 0 && ldb % 8 == 0,
               "vitm_gemm_tn: lda >= N, ldb >= K (both multiples of 8), ldo >= K");
  VITM_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "vitm_gemm_tn: operands must be 16-byte aligned");
  const TnPlan p = tn_plan(M, N, K);
  VITM_REQUIRE((int64_t)p.tiles_n * p.tiles_k * p.slices < (1ll << 31), "vitm_gemm_tn: grid too large for one launch");
  VITM_REQUIRE(((int64_t)p.rows_per_slice + (2 * TN_DEPTH + 3) * TN_BM) * (lda > ldb ? lda : ldb) * 2 < (1ll << 31),
               "vitm_gemm_tn: a slice of rows must stay below 2 GiB");
  const int64_t need = p.slices > 1 ? (int64_t)p.slices * ((int64_t)N * K + N) : 0;
  VITM_REQUIRE(need == 0 || (workspace && workspace_floats >= need),
               "vitm_gemm_tn: workspace of %lld floats needed (vitm_gemm_tn_workspace_floats)", (long long)need);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *ap = (const bf16_t*)a, *bp = (const bf16_t*)b;
  const bool direct = p.slices == 1;
  float* dst = direct ? out : workspace;
  const int64_t ldd = direct ? ldo : (int64_t)K, sstride = direct ? 0 : (int64_t)N * K;
  float* cs = !colsum_out ? nullptr : (direct ? colsum_out : workspace + (int64_t)p.slices * N * K);
  const int rc = p.bn == 256 ? tn_launch<256, 256, 4>(p, ap, lda, bp, ldb, dst, ldd, sstride, cs, row_scale, M, N, K, st)
                             : tn_launch<128, 128, 2>(p, ap, lda, bp, ldb, dst, ldd, sstride, cs, row_scale, M, N, K, st);
  if (rc != VITM_OK || direct) return rc;
  const int64_t total = (int64_t)N * K;
  int64_t g = (total + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)workspace, (int64_t)N * K,
                     p.slices, out, ldo, N, K, (const float*)workspace + (int64_t)p.slices * N * K, colsum_out);
  VITM_CHECK_LAUNCH("vitm_gemm_tn");
  return VITM_OK;
}
