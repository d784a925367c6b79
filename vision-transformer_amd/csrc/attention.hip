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
#include <type_traits>
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

// HD is the head dimension the arithmetic runs at (a multiple of 32); hd <= HD is the real one (a multiple of 8).
// EXACT (hd == HD) compiles the padding away; otherwise columns hd .. HD-1 of Q, K and V are zeros in registers / LDS
// (they add nothing to the scores) and the matching output columns are not stored.
template <int HD, bool EXACT>
__global__ __launch_bounds__(WAVES * 64) void attn_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                          bf16_t* __restrict__ out, int64_t ldo, int N,
                                                          int H, float sl2e, int kchunk, int nqg,
                                                          float* __restrict__ lse, int hd) {
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
  if (EXACT) hd = HD;
  const int D = H * hd;
  const int q0 = (qg * WAVES + wave) * QB;
  const bool active = q0 < N;  // wave-uniform

  const bf16_t* base = qkv + (int64_t)b * N * ldq + h * hd;

  // Q fragments (B operand of S^T = K.Q^T): lane (r, hh) holds Q[q0 + r][16 s + 8 hh + 0..7]
  bf16x8_t qf[KS];
  {
    int qrow = q0 + r;
    qrow = qrow < N ? qrow : N - 1;  // rows past N are computed on a copy of the last row and never stored
    const bf16_t* qp = base + (int64_t)qrow * ldq + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (EXACT || 16 * s + 8 * hh < hd) qf[s] = *(const bf16x8_t*)(qp + 16 * s);
      else qf[s] = __builtin_bit_cast(bf16x8_t, (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0});
    }
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
      if (row < rows && (EXACT || ch * 8 < hd)) {
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
    bf16_t* op = out + ((int64_t)b * N + q0 + r) * ldo + h * hd;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack_bf2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
        w.y = pack_bf2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        if (EXACT || 32 * d + 8 * g + 4 * hh < hd) *(uint2*)(op + 32 * d + 8 * g + 4 * hh) = w;
      }
  }
}

template <int HD, bool EXACT>
int launch_attn(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale,
                float* lse, int hd, hipStream_t st) {
  const int npad = (N + KT - 1) / KT * KT;
  const int kchunk = npad < KCHUNK_MAX ? npad : KCHUNK_MAX;
  const int smem = 2 * kchunk * (HD * 2 + 16);
  auto kfn = attn_kernel<HD, EXACT>;
  VITM_LDS_OPTIN(kfn, 2 * KCHUNK_MAX * (HD * 2 + 16), "vitm_attention");   // once, for the largest chunk
  const int nqb = (N + QB - 1) / QB;
  const int nqg = (nqb + WAVES - 1) / WAVES;
  dim3 grid((unsigned)((int64_t)nqg * B * H));
  hipLaunchKernelGGL(kfn, grid, dim3(WAVES * 64), smem, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                     scale * 1.4426950408889634f, kchunk, nqg, lse, hd);
  VITM_CHECK_LAUNCH("vitm_attention");
  return VITM_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Short sequences (N <= 224 keys, head_dim 64: every 224-pixel ViT): persistent, exact two-pass softmax.
//
// One 8-wave workgroup per CU walks the (image, head) pairs.  Waves 0-6 each own a 32-query block; wave 7 only
// moves data.  All K and V rows of a pair fit LDS (2 x 28 KiB), so the whole score strip S^T[keys, 32 queries] of a
// wave (NT tiles x 16 registers) stays in registers: the row maximum is known before the first exponential, and the
// online-softmax machinery (a running maximum, a cross-half exchange and a rescale of O per tile) disappears.  A
// computing wave runs three long phases per pair:
//   1. NT x 4 MFMAs   S^T = K . Q^T          (K fragments: ds_read_b128, one tile ahead)
//   2. VALU only      max, exp2, bf16        (one exchange between the two 32-lane halves for the max)
//   3. NT x 6 MFMAs   O^T += V^T . P^T       (V^T fragments: ds_read_b64_tr_b16, one k-step ahead; the row sums of P
//                                            come from the same product against a fragment of ones, i.e. they are
//                                            the sums of the bf16 values that the numerator uses, and cost no VALU)
// and the two waves that share a SIMD are in different phases most of the time, so the matrix pipe and the VALU
// overlap across waves without any interleaving inside a wave.
// Every byte moves in whole 128-byte lines, 8 lines per wave instruction.  K and V rows of the workgroup's NEXT pair
// are brought by wave 7 with LDS-DMA (1-KiB pieces of 8 rows x 128 B, no registers) into the other half of a
// two-entry LDS ring while the current pair is computed: the address unit takes tens of cycles per piece, which a
// computing wave cannot afford between its MFMAs.  Each computing wave has a 4-KiB LDS region of its own that holds
// its Q rows (LDS-DMA, issued by the wave itself) until they are in registers, then turns the previous pair's
// output tile into rows for whole-line stores, then receives the next pair's Q rows.  One barrier per pair.
// Rows are unpadded (the DMA writes lane-linear); the 16-byte chunk index is XOR-ed, on the DMA's source address and
// on the reads, with (row >> 1) & 7 for K, Q and the output tile (ds_read_b128 of 16 consecutive rows: 16 distinct
// bank windows) and with 4 * ((row >> 1) & 1) for V (the 4 rows x 64 B of a transposing read cover all 64 banks).
namespace short_seq {

#ifdef ATTN_STAMP   // diagnostic build: lane 0 of every wave records s_memtime at phase boundaries into `lse`
#define ASTAMP(kk)                                                                                      \
  do {                                                                                                  \
    unsigned long long t__;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                          \
    if (lane == 0 && k < 12) ((unsigned long long*)lse)[(((size_t)blockIdx.x * 12 + k) * 8 + wave) * 8 + (kk)] = t__; \
  } while (0)
#else
#define ASTAMP(kk) do {} while (0)
#endif

constexpr int HD = 64, RS = 128, CWAVES = 7, NTH = (CWAVES + 1) * 64;
constexpr int MAX_TILES = 7;                    // 224 keys
constexpr int PARTB = MAX_TILES * 32 * RS;      // K (or V) rows of one pair: 28 KiB
constexpr int BUFB = 2 * PARTB;                 // one ring entry
constexpr int QBASE = 2 * BUFB;                 // per-wave Q / output regions follow the ring
constexpr int QREG = 32 * RS;                   // 4 KiB
constexpr int SMEM = QBASE + CWAVES * QREG;     // 140 KiB

// LDS-DMA of 16 bytes per lane (1 KiB per wave) to LDS byte address `lds_addr` (wave-uniform) + 16 * lane, written as
// inline asm so that the compiler does not know a load is in flight: with the builtin it drains every outstanding
// piece (s_waitcnt vmcnt(0)) in front of the next transposing LDS read or LDS store, which would put the whole
// memory latency of the prefetch into the middle of the compute phases.  Completion is this kernel's own business:
// one "s_waitcnt vmcnt(0)" before the barrier that ends a pair.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_addr)
               : "memory");
}
// 4 bytes per lane (256 B per wave) by LDS-DMA: used to pull 64 different cache lines towards L2 with one instruction;
// the bytes themselves land in a scratch corner of LDS and are never read
__device__ __forceinline__ void glds4(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_addr)
               : "memory");
}
// a value and that of lane ^ 32, without the LDS crossbar (v_permlane32_swap of two copies)
__device__ __forceinline__ void half_pair(float x, float& own, float& other) {
  own = x;
  other = x;
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(own), "+v"(other));
}
__device__ __forceinline__ void swap32u(unsigned& a, unsigned& b) {
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pair_done() {   // this wave's DMA landed, then everyone is done with the pair
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// The mover's copy of NPC 8-row pieces of K and NPC of V (rows k0 .. of one (image, head), clamped to row N - 1) into a
// ring entry, through registers: all 2 NPC 16-byte loads of the lane are issued first (up to 56 KiB per CU in flight),
// then written to LDS as they return.  An LDS-DMA piece occupies the issuing wave for ~165 cycles, which capped the one
// mover wave at ~6 B/clk -- less than the computing waves consume at 577 tokens (56 KiB per 224-key chunk); a 1-KiB
// register load plus its LDS write costs the wave a third of that, and the mover has the whole register budget of the
// kernel to itself.  Same byte placement as the DMA form: piece pl of an operand lands at pl * 1024 + 16 * lane.
template <int NPC>
__device__ __forceinline__ void stage_kv(const bf16_t* base, int64_t ldq, int D, int k0, int N, int lane, char* dst) {
  const int lr = lane >> 3;
  const int kcol0 = ((lane & 7) ^ ((lane >> 4) & 3)) * 8;        // K chunk swizzle: (row >> 1) & 7, row = 8 pl + lr
  const int kcol1 = ((lane & 7) ^ (((lane >> 4) & 3) | 4)) * 8;  //   odd pieces
  const int vcol = ((lane & 7) ^ (4 * ((lane >> 4) & 1))) * 8;   // V chunk swizzle: 4 * ((row >> 1) & 1)
  typedef __attribute__((ext_vector_type(4))) unsigned q4_t;   // (a HIP uint4 array is not promoted to registers)
  q4_t buf[2 * NPC];   // K pieces 0 .. NPC - 1, then V
#pragma unroll
  for (int p = 0; p < 2 * NPC; ++p) {
    const int pl = p < NPC ? p : p - NPC;
    int row = k0 + pl * 8 + lr;
    row = row < N ? row : N - 1;
    buf[p] = *(const q4_t*)(base + (int64_t)row * ldq + (p < NPC ? D + ((pl & 1) ? kcol1 : kcol0) : 2 * D + vcol));
  }
#pragma unroll
  for (int p = 0; p < 2 * NPC; ++p) {
    const int pl = p < NPC ? p : p - NPC;
    *(q4_t*)(dst + (p < NPC ? 0 : PARTB) + pl * 1024 + lane * 16) = buf[p];
  }
}
// The mover's copy of the 7 x 32 query rows q00 .. of one (image, head) into the computing waves' 4-KiB regions at `qdst`
// (blocks that start past N are skipped), in the layout those waves read: piece j of a block = rows 8 j .. 8 j + 7, 16-byte
// chunks XOR-ed with (row >> 1) & 7.
__device__ __forceinline__ void stage_q(const bf16_t* base, int64_t ldq, int q00, int N, int lane, char* qdst, int nblocks, int region) {
  typedef __attribute__((ext_vector_type(4))) unsigned q4_t;
  q4_t qb[28];
#pragma unroll
  for (int w = 0; w < 7; ++w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rl = 8 * j + (lane >> 3);
      int row = q00 + 32 * w + rl;
      row = row < N ? row : N - 1;
      if (w < nblocks && q00 + 32 * w < N) qb[4 * w + j] = *(const q4_t*)(base + (int64_t)row * ldq + (((lane & 7) ^ ((rl >> 1) & 7)) * 8));
    }
  }
#pragma unroll
  for (int w = 0; w < 7; ++w) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (w < nblocks && q00 + 32 * w < N) *(q4_t*)(qdst + w * region + j * 1024 + lane * 16) = qb[4 * w + j];
  }
}
// the mover's end of a pair / chunk: its LDS writes are done, then everyone is done with the previous entry
__device__ __forceinline__ void mover_done() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// WITH_LSE: also write the log-sum-exp of every query row (training); it is formed from the fp32 probabilities
// (a VALU sum on top), while the output is normalised by the sum of the bf16 ones either way.
template <int NT, bool WITH_LSE>   // key tiles of 32: 32 (NT - 1) < N <= 32 NT
__global__ __launch_bounds__(NTH, 2) void attn_short_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                            bf16_t* __restrict__ out, int64_t ldo, int N, int H,
                                                            float sl2e, int nitems, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD;
  const int my_items = blockIdx.x < nitems ? (nitems - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  if (my_items == 0) return;
  const unsigned lds0 = (unsigned)(unsigned long)LPTR(smem);   // LDS byte address of the dynamic region
  constexpr int NP = NT * 4;                                   // 8-row pieces per operand
  auto pair_base = [&](int k) -> const bf16_t* {
    const int item = blockIdx.x + k * gridDim.x;
    const int b = item / H, h = item - b * H;
    return qkv + (int64_t)b * N * ldq + h * HD;
  };

  if (wave == CWAVES) {
    // ---- the mover: K rows then V rows of pair k + 1 into ring entry (k + 1) & 1 while pair k is computed.  Rows
    // past N repeat row N - 1 (finite values: their scores are masked and their probabilities are exactly 0).
    // By LDS-DMA: here the mover has a whole pair's time for 56 pieces, and staged through registers (stage_kv, what
    // the chunked kernel below does) the same copy measured 4 % slower at 197 tokens.
    const int lr = lane >> 3;
    const int kcol0 = ((lane & 7) ^ ((lane >> 4) & 3)) * 8;        // K chunk swizzle: (row >> 1) & 7, row = 8 pl + lr
    const int kcol1 = ((lane & 7) ^ (((lane >> 4) & 3) | 4)) * 8;  //   odd pieces
    const int vcol = ((lane & 7) ^ (4 * ((lane >> 4) & 1))) * 8;   // V chunk swizzle: 4 * ((row >> 1) & 1)
    auto move_pair = [&](int k) {
      const bf16_t* base = pair_base(k);
      const unsigned dst = lds0 + (k & 1) * BUFB;
#pragma unroll 4
      for (int pl = 0; pl < NP; ++pl) {
        int row = pl * 8 + lr;
        row = row < N ? row : N - 1;
        glds16(base + (int64_t)row * ldq + D + ((pl & 1) ? kcol1 : kcol0), dst + pl * 1024);
      }
#pragma unroll 4
      for (int pl = 0; pl < NP; ++pl) {
        int row = pl * 8 + lr;
        row = row < N ? row : N - 1;
        glds16(base + (int64_t)row * ldq + 2 * D + vcol, dst + PARTB + pl * 1024);
      }
    };
    move_pair(0);
    pair_done();
    for (int k = 0; k < my_items; ++k) {
      if (k + 1 < my_items) move_pair(k + 1);
      pair_done();
    }
    return;
  }

  const int r = lane & 31, hh = lane >> 5;
  const int q0 = wave * 32;
  const bool active = q0 < N;   // wave-uniform
  if (!active) {                // no queries for this wave: keep the barrier count
    pair_done();
    for (int k = 0; k < my_items; ++k) pair_done();
    return;
  }

  // lane-constant offsets
  const int fk = (r >> 1) & 7;
  int koff[4];                  // fragment reads of K / Q rows and the output tile's writes: row r, chunk 2 s + hh
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = r * RS + (((2 * s + hh) ^ fk) * 16);
  const int g16 = lane >> 4, li = lane & 15;
  int voff[2];                  // transposing reads of V: 16-lane group = 4 keys x 16 dims
#pragma unroll
  for (int d = 0; d < 2; ++d)
    voff[d] = (4 * hh + (li >> 2)) * RS + 64 * (d ^ ((li >> 3) & 1)) + 32 * (g16 & 1) + 8 * (li & 3);
  int ooff[4];                  // output tile as rows: lane -> row 8 i + (lane >> 3), 16-byte chunk lane & 7
  int64_t qsrc[4];              // Q pieces: element offset of this lane's 16 bytes of row q0 + 8 j + (lane >> 3)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    const int sw = (row >> 1) & 7;
    ooff[i] = row * RS + (((lane & 7) ^ sw) * 16);
    const int grow = q0 + row < N ? q0 + row : N - 1;
    qsrc[i] = (int64_t)grow * ldq + ((lane & 7) ^ sw) * 8;
  }
  char* const qreg = smem + QBASE + wave * QREG;
  const unsigned qreg_lds = lds0 + QBASE + wave * QREG;
  auto issue_q = [&](const bf16_t* base) {
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(base + qsrc[j], qreg_lds + j * 1024);
  };
  // fragment of ones for the row sums: A-operand rows 0 and 4 (accumulator register 0 of the lower / upper half)
  const unsigned one2 = ((lane & 31) == 0 || (lane & 31) == 4) ? 0x3f803f80u : 0u;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(one2, one2, one2, one2));

  uint4 ost0, ost1, ost2, ost3;   // packed output of the previous pair: ostJ = dims 16 J + 8 hh .. + 7 of query r
  ost0 = ost1 = ost2 = ost3 = make_uint4(0, 0, 0, 0);
  // previous pair's tile -> rows (through this wave's region) -> global memory in whole lines; `dst` = the
  // tile's first row, out + (b N + q0) ldo + h HD
#define VITM_FLUSH_OUTPUT(dst)                                                                        \
  do {                                                                                                \
    *(uint4*)(qreg + koff[0]) = ost0;   /* dims 16 j + 8 hh .. + 7 = chunk 2 j + hh of row r */       \
    *(uint4*)(qreg + koff[1]) = ost1;                                                                 \
    *(uint4*)(qreg + koff[2]) = ost2;                                                                 \
    *(uint4*)(qreg + koff[3]) = ost3;                                                                 \
    wait_lds();                                                                                       \
    const uint4 w0 = *(const uint4*)(qreg + ooff[0]), w1 = *(const uint4*)(qreg + ooff[1]);           \
    const uint4 w2 = *(const uint4*)(qreg + ooff[2]), w3 = *(const uint4*)(qreg + ooff[3]);           \
    wait_lds();                                                                                       \
    bf16_t* op_ = (dst) + (int64_t)(lane >> 3) * ldo + (lane & 7) * 8;                                \
    const int qr_ = q0 + (lane >> 3);                                                                 \
    if (qr_ < N) *(uint4*)op_ = w0;                                                                   \
    if (qr_ + 8 < N) *(uint4*)(op_ + 8 * ldo) = w1;                                                   \
    if (qr_ + 16 < N) *(uint4*)(op_ + 16 * ldo) = w2;                                                 \
    if (qr_ + 24 < N) *(uint4*)(op_ + 24 * ldo) = w3;                                                 \
  } while (0)
  auto out_base = [&](int k) -> bf16_t* {
    const int item = blockIdx.x + k * gridDim.x;
    const int b = item / H, h = item - b * H;
    return out + ((int64_t)b * N + q0) * ldo + h * HD;
  };

  issue_q(pair_base(0));
  pair_done();

  for (int k = 0; k < my_items; ++k) {
    const bool more = k + 1 < my_items;
    const char* Kb = smem + (k & 1) * BUFB;
    const char* Vb = Kb + PARTB;
    ASTAMP(0);
    // ---- Q fragments out of the region
    bf16x8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8_t*)(qreg + koff[s]);
    bf16x8_t kf[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[0][s] = *(const bf16x8_t*)(Kb + koff[s]);
    ASTAMP(1);
    // ---- phase 1: S^T strip; K fragments one tile ahead
    f32x16_t sacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[(t + 1) & 1][s] = *(const bf16x8_t*)(Kb + (t + 1) * 32 * RS + koff[s]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t & 1][s], qf[s], sacc[t], 0, 0, 0);
    }
    if (NT * 32 > N) {   // ragged last tile: keys past N contribute nothing
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (NT - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (key >= N) sacc[NT - 1][e] = -1e30f;
      }
    }
    ASTAMP(2);
    // ---- the region is free (Q is in registers): previous output through it, then the next pair's Q into it
    if (k > 0) VITM_FLUSH_OUTPUT(out_base(k - 1));
    if (more) issue_q(pair_base(k + 1));
    // ---- phase 2: exact softmax of the strip
    float m = sacc[0][0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, sacc[t][e]);
    float m_own, m_other;
    half_pair(m, m_own, m_other);
    m = fmaxf(m_own, m_other);
    const float mb = m * sl2e;
    bf16x8_t pb[NT][2];
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u16x8_t pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[t][8 * s2 + j], sl2e, -mb));
          if (WITH_LSE) lsum += pv;
          pk[j] = f2bf(pv);
        }
        pb[t][s2] = __builtin_bit_cast(bf16x8_t, pk);
      }
    }
    ASTAMP(3);
    // ---- phase 3: O^T += V^T . P^T ; element j of the A fragment is key 16 s2 + 8 (j >> 2) + 4 hh + (j & 3);
    // V^T fragments one k-step (16 keys) ahead
    f32x16_t o[2], osum;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = osum[e] = 0.f;
    u16x4_t vf[2][4];   // [parity of the k-step][2 d + (lo, hi)]
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      vf[0][2 * d] = tr_read(Vb + voff[d]);
      vf[0][2 * d + 1] = tr_read(Vb + voff[d] + 8 * RS);
    }
#pragma unroll
    for (int ks = 0; ks < 2 * NT; ++ks) {
      if (ks + 1 < 2 * NT) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          vf[(ks + 1) & 1][2 * d] = tr_read(Vb + (ks + 1) * 16 * RS + voff[d]);
          vf[(ks + 1) & 1][2 * d + 1] = tr_read(Vb + (ks + 1) * 16 * RS + voff[d] + 8 * RS);
        }
      }
      osum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb[ks >> 1][ks & 1], osum, 0, 0, 0);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const u16x4_t lo = vf[ks & 1][2 * d], hi = vf[ks & 1][2 * d + 1];
        u16x8_t tv;
        tv[0] = lo[0]; tv[1] = lo[1]; tv[2] = lo[2]; tv[3] = lo[3];
        tv[4] = hi[0]; tv[5] = hi[1]; tv[6] = hi[2]; tv[7] = hi[3];
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, tv), pb[ks >> 1][ks & 1], o[d], 0, 0, 0);
      }
    }
    ASTAMP(4);
    // ---- normalise, pack; pairs of 4-dim groups of the two halves are exchanged so a lane holds 8 consecutive
    // dims (16 bytes): lower half dims 32 d + 16 p .. + 7, upper half dims 32 d + 16 p + 8 .. + 15
    const float l = osum[0];    // sum over all keys of this lane's query (rows 0 / 4 of the ones product)
    const float inv = __builtin_amdgcn_rcpf(l);
#ifndef ATTN_STAMP
    if (WITH_LSE) {
      float ls_own, ls_other;
      half_pair(lsum, ls_own, ls_other);
      const int item = blockIdx.x + k * gridDim.x;
      if (hh == 0 && q0 + r < N) lse[(int64_t)item * N + q0 + r] = mb + __log2f(ls_own + ls_other);   // item = b H + h
    }
#endif
#define VITM_PACK_OUT(dst, d, pq)                                                          \
  do {                                                                                    \
    unsigned a0 = pack_bf2(o[d][8 * pq + 0] * inv, o[d][8 * pq + 1] * inv);               \
    unsigned a1 = pack_bf2(o[d][8 * pq + 2] * inv, o[d][8 * pq + 3] * inv);               \
    unsigned b0 = pack_bf2(o[d][8 * pq + 4] * inv, o[d][8 * pq + 5] * inv);               \
    unsigned b1 = pack_bf2(o[d][8 * pq + 6] * inv, o[d][8 * pq + 7] * inv);               \
    swap32u(a0, b0);                                                                      \
    swap32u(a1, b1);                                                                      \
    dst = make_uint4(a0, a1, b0, b1);                                                     \
  } while (0)
    VITM_PACK_OUT(ost0, 0, 0);
    VITM_PACK_OUT(ost1, 0, 1);
    VITM_PACK_OUT(ost2, 1, 0);
    VITM_PACK_OUT(ost3, 1, 1);
    ASTAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(6);
    __builtin_amdgcn_s_barrier();
    ASTAMP(7);
  }
  VITM_FLUSH_OUTPUT(out_base(my_items - 1));
#undef VITM_FLUSH_OUTPUT
#undef VITM_PACK_OUT
}

// ---- longer sequences (head_dim 64, any N): the same machinery with the keys in chunks of 224 ----------------------
// An item is (image, head, group of 7 query blocks); its K and V rows stream through the two-entry ring in chunks of up
// to 224 keys (the mover runs one chunk ahead, across items).  Within a chunk the softmax is the exact two-pass one of
// the strip; across chunks a running maximum rescales O and the row sums once per chunk (2-3 times per item at 577
// tokens, against 18 times per item for a tile-by-tile online softmax).  The query groups of one (image, head) are
// dealt to workgroups of one XCD (blockIdx & 7), so their K / V chunks are fetched from memory once and then from L2.
#ifdef ATTN_STAMP   // chunked kernel: record [workgroup][step < 16][wave][slot < 8]
#define CSTAMP(kk)                                                                                      \
  do {                                                                                                  \
    unsigned long long t__;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                          \
    if (lane == 0 && step < 16) ((unsigned long long*)lse)[(((size_t)blockIdx.x * 16 + step) * 8 + wave) * 8 + (kk)] = t__; \
  } while (0)
#else
#define CSTAMP(kk) do {} while (0)
#endif
template <int NTL, bool WITH_LSE>   // NTL: key tiles of the LAST chunk, 32 (NTL - 1) < N - 224 (nchunks - 1) <= 32 NTL
__global__ __launch_bounds__(NTH, 2) void attn_chunked_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                              bf16_t* __restrict__ out, int64_t ldo, int N, int H,
                                                              float sl2e, int nunits, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CK = 32 * MAX_TILES;   // keys per chunk, queries per group
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD;
  const int nchunks = (N + CK - 1) / CK, QG = nchunks;
  // items of this workgroup: XCD group x = blockIdx & 7 owns the units (image, head) u = x, x + 8, ...; its local items
  // (unit, query group), unit-major, are dealt round-robin to the group's workgroups
  const int xg = blockIdx.x & 7, lb = blockIdx.x >> 3, nloc = (gridDim.x + 7 - xg) >> 3;
  const int units_x = nunits > xg ? (nunits - xg + 7) >> 3 : 0;
  const int loc_items = units_x * QG;
  const int my_items = lb < loc_items ? (loc_items - lb + nloc - 1) / nloc : 0;
  if (my_items == 0) return;
  const unsigned lds0 = (unsigned)(unsigned long)LPTR(smem);
  auto item_of = [&](int k, int& unit, int& qg) {
    const int li = lb + k * nloc;
    const int ul = li / QG;
    qg = li - ul * QG;
    unit = ul * 8 + xg;
  };
  auto unit_base = [&](int unit) -> const bf16_t* {
    const int b = unit / H, h = unit - b * H;
    return qkv + (int64_t)b * N * ldq + h * HD;
  };

  if (wave == CWAVES) {
    // ---- the mover: chunk (step % nchunks) of item (step / nchunks) into ring entry step & 1, one step ahead
    const int nsteps = my_items * nchunks;
      auto move_step = [&](int step) {
      const int k = step / nchunks, c = step - k * nchunks;
      int unit, qg;
      item_of(k, unit, qg);
      char* dst = smem + (step & 1) * BUFB;
      if (c + 1 < nchunks) stage_kv<4 * MAX_TILES>(unit_base(unit), ldq, D, c * CK, N, lane, dst);
      else stage_kv<4 * NTL>(unit_base(unit), ldq, D, c * CK, N, lane, dst);
    };
    // Q rows of item k for all computing waves.  The regions are free from the flush of the previous item's output
    // (after the S phase of an item's first chunk) to the start of the next item, so the rows of item k + 1 are brought
    // during the second chunk of item k, behind that chunk's barrier
    auto move_q = [&](int k) {
      int unit, qg;
      item_of(k, unit, qg);
      stage_q(unit_base(unit), ldq, qg * CK, N, lane, smem + QBASE, CWAVES, QREG);
    };
    move_step(0);
    move_q(0);
    mover_done();
    for (int step = 0; step < nsteps; ++step) {
      if (step + 1 < nsteps) move_step(step + 1);
      const int k = step / nchunks;
      if (step - k * nchunks == 1 && k + 1 < my_items) move_q(k + 1);
      mover_done();
    }
    return;
  }

  const int r = lane & 31, hh = lane >> 5;
  const int fk = (r >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = r * RS + (((2 * s + hh) ^ fk) * 16);
  const int g16 = lane >> 4, li = lane & 15;
  int voff[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
    voff[d] = (4 * hh + (li >> 2)) * RS + 64 * (d ^ ((li >> 3) & 1)) + 32 * (g16 & 1) + 8 * (li & 3);
  int ooff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    const int sw = (row >> 1) & 7;
    ooff[i] = row * RS + (((lane & 7) ^ sw) * 16);
  }
  char* const qreg = smem + QBASE + wave * QREG;
  const unsigned qreg_lds = lds0 + QBASE + wave * QREG;
  uint4 ost0, ost1, ost2, ost3;
  ost0 = ost1 = ost2 = ost3 = make_uint4(0, 0, 0, 0);
  bf16_t* pend = nullptr;   // first row of the packed tile waiting in ost0..3 (wave-uniform), with its first query
  int pend_q0 = 0;
#define VITM_FLUSH_PENDING()                                                                          \
  do {                                                                                                \
    *(uint4*)(qreg + koff[0]) = ost0;                                                                 \
    *(uint4*)(qreg + koff[1]) = ost1;                                                                 \
    *(uint4*)(qreg + koff[2]) = ost2;                                                                 \
    *(uint4*)(qreg + koff[3]) = ost3;                                                                 \
    wait_lds();                                                                                       \
    const uint4 w0 = *(const uint4*)(qreg + ooff[0]), w1 = *(const uint4*)(qreg + ooff[1]);           \
    const uint4 w2 = *(const uint4*)(qreg + ooff[2]), w3 = *(const uint4*)(qreg + ooff[3]);           \
    wait_lds();                                                                                       \
    bf16_t* op_ = pend + (int64_t)(lane >> 3) * ldo + (lane & 7) * 8;                                 \
    const int qr_ = pend_q0 + (lane >> 3);                                                            \
    if (qr_ < N) *(uint4*)op_ = w0;                                                                   \
    if (qr_ + 8 < N) *(uint4*)(op_ + 8 * ldo) = w1;                                                   \
    if (qr_ + 16 < N) *(uint4*)(op_ + 16 * ldo) = w2;                                                 \
    if (qr_ + 24 < N) *(uint4*)(op_ + 24 * ldo) = w3;                                                 \
    pend = nullptr;                                                                                   \
  } while (0)

  pair_done();

  int step = 0;
  for (int k = 0; k < my_items; ++k) {
    int unit, qg;
    item_of(k, unit, qg);
    const int q0 = qg * CK + wave * 32;
    const bool active = q0 < N;   // wave-uniform
    const bool more = k + 1 < my_items;
    if (!active) {                // no queries for this wave in this item: keep the barrier count
      if (pend != nullptr) VITM_FLUSH_PENDING();
      for (int c = 0; c < nchunks; ++c, ++step) pair_done();
      continue;
    }
    bf16x8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8_t*)(qreg + koff[s]);
    wait_lds();
    CSTAMP(5);
    // the region is free now (Q is in registers): the previous item's output goes through it after the S phase of the
    // first chunk, and the mover refills it with the next item's Q rows during the second chunk
    f32x16_t o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = 0.f;
    float m_run = -1e30f, lsum = 0.f;   // lsum: this half's share of the row sum (fp32 probabilities)

    // one chunk of NTC key tiles out of ring entry step & 1
    auto chunk = [&](auto ntc_tag, int keys, bool first) {
      constexpr int NTC = decltype(ntc_tag)::value;
      const char* Kb = smem + (step & 1) * BUFB;
      const char* Vb = Kb + PARTB;
      CSTAMP(0);
      // K fragments half a tile ahead (two of a tile's four k-steps): the registers that a whole tile ahead would
      // take are held by O here
      bf16x8_t kf[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) kf[0][s] = *(const bf16x8_t*)(Kb + koff[s]);
      f32x16_t sacc[NTC];
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {   // half tile hs: k-steps 2 hs, 2 hs + 1
          const int nx = 2 * t + hs + 1;   // the next half tile
          if (nx < 2 * NTC) {
#pragma unroll
            for (int s = 0; s < 2; ++s) kf[nx & 1][s] = *(const bf16x8_t*)(Kb + (nx >> 1) * 32 * RS + koff[2 * (nx & 1) + s]);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
            sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[hs][s], qf[2 * hs + s], sacc[t], 0, 0, 0);
        }
      }
      if (NTC * 32 > keys) {   // ragged last tile of the last chunk
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = (NTC - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= keys) sacc[NTC - 1][e] = -1e30f;
        }
      }
      CSTAMP(1);
      if (first && pend != nullptr) VITM_FLUSH_PENDING();   // the stores go out while the softmax runs
      CSTAMP(6);
      float m = sacc[0][0];
#pragma unroll
      for (int t = 0; t < NTC; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, sacc[t][e]);
      float m_own, m_other;
      half_pair(m, m_own, m_other);
      CSTAMP(2);
      const float m_new = fmaxf(m_run, fmaxf(m_own, m_other));
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2e);   // 0 for the first chunk
      m_run = m_new;
      const float mb = m_new * sl2e;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        o[0][e] *= alpha;
        o[1][e] *= alpha;
      }
      lsum *= alpha;
      bf16x8_t pb[NTC][2];
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u16x8_t pk;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[t][8 * s2 + j], sl2e, -mb));
            lsum += pv;
            pk[j] = f2bf(pv);
          }
          pb[t][s2] = __builtin_bit_cast(bf16x8_t, pk);
        }
      }
      u16x4_t vf[2][4];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        vf[0][2 * d] = tr_read(Vb + voff[d]);
        vf[0][2 * d + 1] = tr_read(Vb + voff[d] + 8 * RS);
      }
#pragma unroll
      for (int ks = 0; ks < 2 * NTC; ++ks) {
        if (ks + 1 < 2 * NTC) {
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            vf[(ks + 1) & 1][2 * d] = tr_read(Vb + (ks + 1) * 16 * RS + voff[d]);
            vf[(ks + 1) & 1][2 * d + 1] = tr_read(Vb + (ks + 1) * 16 * RS + voff[d] + 8 * RS);
          }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const u16x4_t lo = vf[ks & 1][2 * d], hi = vf[ks & 1][2 * d + 1];
          u16x8_t tv;
          tv[0] = lo[0]; tv[1] = lo[1]; tv[2] = lo[2]; tv[3] = lo[3];
          tv[4] = hi[0]; tv[5] = hi[1]; tv[6] = hi[2]; tv[7] = hi[3];
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, tv), pb[ks >> 1][ks & 1], o[d], 0, 0, 0);
        }
      }
    };

    for (int c = 0; c < nchunks; ++c, ++step) {
      if (c + 1 < nchunks) {
        chunk(std::integral_constant<int, MAX_TILES>{}, CK, c == 0);
        CSTAMP(3);
        pair_done();
        CSTAMP(4);
      } else {
        chunk(std::integral_constant<int, NTL>{}, N - c * CK, false);
        // ---- the item is complete: normalise, pack (see attn_short_kernel)
        float ls_own, ls_other;
        half_pair(lsum, ls_own, ls_other);
        const float l = ls_own + ls_other;
        const float inv = __builtin_amdgcn_rcpf(l);
#ifndef ATTN_STAMP
        if (WITH_LSE) {
          if (hh == 0 && q0 + r < N) lse[(int64_t)unit * N + q0 + r] = m_run * sl2e + __log2f(l);
        }
#endif
#define VITM_PACK_OUT(dst, d, pq)                                                          \
  do {                                                                                    \
    unsigned a0 = pack_bf2(o[d][8 * pq + 0] * inv, o[d][8 * pq + 1] * inv);               \
    unsigned a1 = pack_bf2(o[d][8 * pq + 2] * inv, o[d][8 * pq + 3] * inv);               \
    unsigned b0 = pack_bf2(o[d][8 * pq + 4] * inv, o[d][8 * pq + 5] * inv);               \
    unsigned b1 = pack_bf2(o[d][8 * pq + 6] * inv, o[d][8 * pq + 7] * inv);               \
    swap32u(a0, b0);                                                                      \
    swap32u(a1, b1);                                                                      \
    dst = make_uint4(a0, a1, b0, b1);                                                     \
  } while (0)
        VITM_PACK_OUT(ost0, 0, 0);
        VITM_PACK_OUT(ost1, 0, 1);
        VITM_PACK_OUT(ost2, 1, 0);
        VITM_PACK_OUT(ost3, 1, 1);
#undef VITM_PACK_OUT
        const int b = unit / H, h = unit - b * H;
        pend = out + ((int64_t)b * N + q0) * ldo + h * HD;
        pend_q0 = q0;
        CSTAMP(3);
        pair_done();
        CSTAMP(4);
      }
    }
  }
  if (pend != nullptr) VITM_FLUSH_PENDING();
#undef VITM_FLUSH_PENDING
}

template <int NTL>
int launch_chunked(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale, float* lse,
                   hipStream_t st) {
  const int nunits = B * H;
  const int QG = (N + 32 * MAX_TILES - 1) / (32 * MAX_TILES);
  const int64_t items = (int64_t)nunits * QG;
  const int ncu = vitm_cu_count() & ~7;
  int grid = (int)(items < ncu ? (items + 7) / 8 * 8 : ncu);
  grid = grid < 8 ? 8 : grid;
  const float sl2e = scale * 1.4426950408889634f;
  if (lse != nullptr) {
    auto kfn = attn_chunked_kernel<NTL, true>;
    VITM_LDS_OPTIN(kfn, SMEM, "vitm_attention_lse");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NTH), SMEM, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                       sl2e, nunits, lse);
  } else {
    auto kfn = attn_chunked_kernel<NTL, false>;
    VITM_LDS_OPTIN(kfn, SMEM, "vitm_attention");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NTH), SMEM, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                       sl2e, nunits, lse);
  }
  VITM_CHECK_LAUNCH("vitm_attention (chunked keys)");
  return VITM_OK;
}


// ---- 225 .. 608 tokens (head_dim 64; ViT-B/16 at 384 pixels: 577): all K and V rows of an (image, head) stay in LDS ----
// 19 key tiles x 32 rows x 128 B x 2 operands = 152 KiB: with nothing else in LDS a whole (image, head) fits, so a
// workgroup loads its K / V once (the chunked kernel's three query-group workgroups each streamed them) and, more
// important, there is no per-chunk barrier: a wave takes the 32-query blocks w, w + 8, w + 16 of the unit and runs them
// at its own pace.  The chunked kernel's barrier after every 224 keys re-aligned the phases of the two waves of a SIMD at
// every step (both on the matrix pipe for S, then both on the VALU for the softmax) and made the four older waves wait
// ~3 k cycles of every 8.3 k for the younger ones; left alone the waves drift apart and the pipes overlap.
// All 8 waves compute.  Q fragments come straight from global memory into registers (the next block's rows are requested
// right after the last S phase of the current block, when the fragment registers fall free), and the output tile is
// stored from registers as 32-byte segments (four instructions fill a row's 128-byte line).  Per unit: barrier, the 8
// waves bring K and V by LDS-DMA (19 pieces each) -- the first 224 keys first --, barrier when those are in, each wave's
// first strip, barrier when the rest is in.  Within a block the softmax is the strip-wise one of the
// chunked kernel: exact over 224 keys, a running maximum and one rescale of O per further 224.
constexpr int RES_MAX_TILES = 19;
constexpr int RES_OPB = RES_MAX_TILES * 32 * RS;      // K (or V) rows of one unit: 76 KiB
constexpr int RES_SMEM = 2 * RES_OPB + 8 * 256;       // 152 KiB + a 256-byte sink per wave for the prefetch

template <int NTL, bool WITH_LSE>   // NTL: key tiles of the LAST 224-key strip, as in attn_chunked_kernel
__global__ __launch_bounds__(512, 2) void attn_resident_kernel(const bf16_t* __restrict__ qkv, int64_t ldq,
                                                               bf16_t* __restrict__ out, int64_t ldo, int N, int H,
                                                               float sl2e, int nunits, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CK = 32 * MAX_TILES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD;
  const int nchunks = (N + CK - 1) / CK;
  const int ntiles = (nchunks - 1) * MAX_TILES + NTL;   // key tiles of a unit (<= 19)
  const int nqb = (N + 31) >> 5;                        // 32-query blocks of a unit
  const unsigned lds0 = (unsigned)(unsigned long)LPTR(smem);
  const int r = lane & 31, hh = lane >> 5;
  const int fk = (r >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = r * RS + (((2 * s + hh) ^ fk) * 16);
  const int g16 = lane >> 4, li = lane & 15;
  int voff[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
    voff[d] = (4 * hh + (li >> 2)) * RS + 64 * (d ^ ((li >> 3) & 1)) + 32 * (g16 & 1) + 8 * (li & 3);
  // loader constants (see the mover of attn_short_kernel)
  const int lr = lane >> 3;
  const int kcol0 = ((lane & 7) ^ ((lane >> 4) & 3)) * 8, kcol1 = ((lane & 7) ^ (((lane >> 4) & 3) | 4)) * 8;
  const int vcol = ((lane & 7) ^ (4 * ((lane >> 4) & 1))) * 8;
  const int npieces = 4 * ntiles;   // per operand

#ifdef ATTN_STAMP
  int ui = 0, bi = 0;
#define RSTAMP(kk)                                                                                      \
  do {                                                                                                  \
    unsigned long long t__;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                          \
    if (lane == 0 && ui < 4 && bi < 3) ((unsigned long long*)lse)[((((size_t)blockIdx.x * 4 + ui) * 8 + wave) * 3 + bi) * 8 + (kk)] = t__; \
  } while (0)
#else
#define RSTAMP(kk) do {} while (0)
#endif
  // The next unit's K and V rows are touched while this unit is computed (one line per lane, 2 N lines = 19
  // instructions per unit, dealt over the waves): loading a unit is not overlapped with anything, and read from HBM
  // its 152 KiB took a CU ~17 k cycles (~10 B/clk) of a unit's ~70 k.
  auto prefetch_unit = [&](int nu, int i) {   // instruction i of unit nu: lines 64 i .. 64 i + 63 (line = K row, then V row)
    if (nu >= nunits) return;
    const int idx = 64 * i + lane;
    const int nb = nu / H, nh = nu - nb * H;
    int row = idx < N ? idx : idx - N;
    row = row < N ? row : N - 1;
    const bf16_t* src = qkv + (int64_t)nb * N * ldq + nh * HD + (int64_t)row * ldq + (idx < N ? D : 2 * D);
    glds4(src, lds0 + 2 * RES_OPB + wave * 256);
  };
  const int npf = (2 * N + 63) >> 6;   // prefetch instructions per unit
  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const int b = unit / H, h = unit - b * H;
#ifdef ATTN_STAMP
    bi = 0;
    RSTAMP(0);
#endif
    const bf16_t* base = qkv + (int64_t)b * N * ldq + h * HD;
    // ---- everyone is done with the previous unit's rows; K and V of this one, piece p by wave p & 7
    __builtin_amdgcn_s_barrier();
    auto bring = [&](int pl) {
      int row = pl * 8 + lr;
      row = row < N ? row : N - 1;
      glds16(base + (int64_t)row * ldq + D + ((pl & 1) ? kcol1 : kcol0), lds0 + pl * 1024);
      glds16(base + (int64_t)row * ldq + 2 * D + vcol, lds0 + RES_OPB + pl * 1024);
    };
    // the first 224 keys (what the first strip of every wave's first block needs), then Q, then the rest: the first strip
    // starts when the first part is in and runs under the arrival of the second
    constexpr int P0 = 4 * MAX_TILES;   // pieces of the first strip
    int pl = wave;
    for (; pl < P0; pl += 8) bring(pl);
    // this wave's first Q fragments: row q0 + r, 16-byte chunk 2 s + hh
    int qb = wave;
    bf16x8_t qf[4];
    auto load_q = [&](int blk) {
      int row = 32 * blk + r;
      row = row < N ? row : N - 1;
      const bf16_t* qp = base + (int64_t)row * ldq + hh * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8_t*)(qp + 16 * s);
    };
    if (qb < nqb) load_q(qb);
    int later = 0;   // DMA instructions of the second part issued by this wave
    for (; pl < npieces; pl += 8, later += 2) bring(pl);
    switch (later) {   // the counter takes an immediate
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    RSTAMP(1);
    __builtin_amdgcn_s_barrier();
    RSTAMP(2);
    bool rest_pending = true;   // the second part has not been waited for yet (first strip of the wave's first block)

    for (; qb < nqb; qb += 8) {
      const int q0 = 32 * qb;
      RSTAMP(3);
      f32x16_t o[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = 0.f;
      float m_run = -1e30f, lsum = 0.f;   // lsum: this half's share of the row sum (fp32 probabilities)

      // one strip of NTC key tiles starting at key k0; `last`: request the next block's Q once the S phase is issued
      auto strip = [&](auto ntc_tag, int k0, int keys, bool last) {
        constexpr int NTC = decltype(ntc_tag)::value;
        const char* Kb = smem + k0 * RS;
        const char* Vb = smem + RES_OPB + k0 * RS;
        bf16x8_t kf[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) kf[0][s] = *(const bf16x8_t*)(Kb + koff[s]);
        f32x16_t sacc[NTC];
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
#pragma unroll
          for (int e = 0; e < 16; ++e) sacc[t][e] = 0.f;
#pragma unroll
          for (int hs = 0; hs < 2; ++hs) {
            const int nx = 2 * t + hs + 1;
            if (nx < 2 * NTC) {
#pragma unroll
              for (int s = 0; s < 2; ++s) kf[nx & 1][s] = *(const bf16x8_t*)(Kb + (nx >> 1) * 32 * RS + koff[2 * (nx & 1) + s]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
              sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[hs][s], qf[2 * hs + s], sacc[t], 0, 0, 0);
          }
        }
        if (last && qb + 8 < nqb) load_q(qb + 8);   // the fragment registers are free from here on
        if (NTC * 32 > keys) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = (NTC - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (key >= keys) sacc[NTC - 1][e] = -1e30f;
          }
        }
        float m = sacc[0][0];
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) m = fmaxf(m, sacc[t][e]);
        float m_own, m_other;
        half_pair(m, m_own, m_other);
        const float m_new = fmaxf(m_run, fmaxf(m_own, m_other));
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2e);   // 0 for the first strip
        m_run = m_new;
        const float mb = m_new * sl2e;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          o[0][e] *= alpha;
          o[1][e] *= alpha;
        }
        float ssum = 0.f;   // this strip's share of the row sum; folded into the running one by ONE explicit fma below (left
                            // as "lsum *= alpha; lsum += p" the compiler contracts the two differently per instantiation)
        bf16x8_t pb[NTC][2];
        const f32x2_t sc2 = {sl2e, sl2e}, nmb2 = {-mb, -mb};
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            unsigned pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x2_t x = __builtin_elementwise_fma((f32x2_t){sacc[t][8 * s2 + 2 * j], sacc[t][8 * s2 + 2 * j + 1]}, sc2, nmb2);
              const float p0 = __builtin_amdgcn_exp2f(x.x), p1 = __builtin_amdgcn_exp2f(x.y);
              ssum += p0 + p1;
              pk[j] = pack_bf2(p0, p1);
            }
            pb[t][s2] = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
          }
        }
        lsum = __builtin_fmaf(lsum, alpha, ssum);
        u16x4_t vf[2][4];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          vf[0][2 * d] = tr_read(Vb + voff[d]);
          vf[0][2 * d + 1] = tr_read(Vb + voff[d] + 8 * RS);
        }
#pragma unroll
        for (int ks = 0; ks < 2 * NTC; ++ks) {
          if (ks + 1 < 2 * NTC) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              vf[(ks + 1) & 1][2 * d] = tr_read(Vb + (ks + 1) * 16 * RS + voff[d]);
              vf[(ks + 1) & 1][2 * d + 1] = tr_read(Vb + (ks + 1) * 16 * RS + voff[d] + 8 * RS);
            }
          }
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const u16x4_t lo = vf[ks & 1][2 * d], hi = vf[ks & 1][2 * d + 1];
            u16x8_t tv;
            tv[0] = lo[0]; tv[1] = lo[1]; tv[2] = lo[2]; tv[3] = lo[3];
            tv[4] = hi[0]; tv[5] = hi[1]; tv[6] = hi[2]; tv[7] = hi[3];
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, tv), pb[ks >> 1][ks & 1], o[d], 0, 0, 0);
          }
        }
      };

      for (int c = 0; c + 1 < nchunks; ++c) {
        strip(std::integral_constant<int, MAX_TILES>{}, c * CK, CK, false);
        if (rest_pending) {   // (every wave has a first block: 8 <= nqb; its first strip is a whole one: N > 224)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          rest_pending = false;
        }
#ifndef ATTN_NO_PREFETCH
        if (c == 0) {   // this wave's share of the next unit's lines: instructions pfi, pfi + 8, ... over its blocks
          const int blk_i = (qb - wave) >> 3;             // 0, 1, 2: which of this wave's blocks
          for (int i = wave + 8 * blk_i; i < npf; i += 24) prefetch_unit(unit + gridDim.x, i);
        }
#endif
        RSTAMP(4 + c);
      }
      strip(std::integral_constant<int, NTL>{}, (nchunks - 1) * CK, N - (nchunks - 1) * CK, true);
      RSTAMP(6);

      // ---- normalise, pack (see attn_short_kernel), store: lane (r, hh) holds dims 16 J + 8 hh .. + 7 of query q0 + r
      float ls_own, ls_other;
      half_pair(lsum, ls_own, ls_other);
      const float l = ls_own + ls_other;
      const float inv = __builtin_amdgcn_rcpf(l);
#ifndef ATTN_STAMP
      if (WITH_LSE) {
        if (hh == 0 && q0 + r < N) lse[(int64_t)unit * N + q0 + r] = m_run * sl2e + __log2f(l);
      }
#endif
      uint4 ost0, ost1, ost2, ost3;
#define VITM_PACK_OUT(dst, d, pq)                                                          \
  do {                                                                                    \
    unsigned a0 = pack_bf2(o[d][8 * pq + 0] * inv, o[d][8 * pq + 1] * inv);               \
    unsigned a1 = pack_bf2(o[d][8 * pq + 2] * inv, o[d][8 * pq + 3] * inv);               \
    unsigned b0 = pack_bf2(o[d][8 * pq + 4] * inv, o[d][8 * pq + 5] * inv);               \
    unsigned b1 = pack_bf2(o[d][8 * pq + 6] * inv, o[d][8 * pq + 7] * inv);               \
    swap32u(a0, b0);                                                                      \
    swap32u(a1, b1);                                                                      \
    dst = make_uint4(a0, a1, b0, b1);                                                     \
  } while (0)
      VITM_PACK_OUT(ost0, 0, 0);
      VITM_PACK_OUT(ost1, 0, 1);
      VITM_PACK_OUT(ost2, 1, 0);
      VITM_PACK_OUT(ost3, 1, 1);
#undef VITM_PACK_OUT
      if (q0 + r < N) {
        bf16_t* op = out + ((int64_t)b * N + q0 + r) * ldo + h * HD + 8 * hh;
        *(uint4*)op = ost0;
        *(uint4*)(op + 16) = ost1;
        *(uint4*)(op + 32) = ost2;
        *(uint4*)(op + 48) = ost3;
      }
      RSTAMP(7);
#ifdef ATTN_STAMP
      ++bi;
#endif
    }
#ifdef ATTN_STAMP
    ++ui;
#endif
  }
#undef RSTAMP
}

template <int NTL>
int launch_resident(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale, float* lse,
                    hipStream_t st) {
  const int nunits = B * H;
  const int ncu = vitm_cu_count();
  const int grid = nunits < ncu ? nunits : ncu;
  const float sl2e = scale * 1.4426950408889634f;
  if (lse != nullptr) {
    auto kfn = attn_resident_kernel<NTL, true>;
    VITM_LDS_OPTIN(kfn, RES_SMEM, "vitm_attention_lse");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), RES_SMEM, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                       sl2e, nunits, lse);
  } else {
    auto kfn = attn_resident_kernel<NTL, false>;
    VITM_LDS_OPTIN(kfn, RES_SMEM, "vitm_attention");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), RES_SMEM, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                       sl2e, nunits, lse);
  }
  VITM_CHECK_LAUNCH("vitm_attention (resident keys)");
  return VITM_OK;
}

int launch_long(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale, float* lse,
                hipStream_t st) {
  const int last = N - (N - 1) / (32 * MAX_TILES) * (32 * MAX_TILES);   // keys of the last chunk, 1 .. 224
#ifndef ATTN_NO_RESIDENT
  if (N <= 32 * RES_MAX_TILES) {   // every K / V row of an (image, head) fits LDS
    switch ((last + 31) / 32) {
      case 1: return launch_resident<1>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
      case 2: return launch_resident<2>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
      case 3: return launch_resident<3>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
      case 4: return launch_resident<4>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
      case 5: return launch_resident<5>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
      case 6: return launch_resident<6>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
      default: return launch_resident<7>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    }
  }
#endif
  switch ((last + 31) / 32) {
    case 1: return launch_chunked<1>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 2: return launch_chunked<2>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 3: return launch_chunked<3>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 4: return launch_chunked<4>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 5: return launch_chunked<5>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 6: return launch_chunked<6>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    default: return launch_chunked<7>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
  }
}

template <int NT>
int launch_nt(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale, float* lse,
              hipStream_t st) {
  const int nitems = B * H;
  const int ncu = vitm_cu_count();
  const int grid = nitems < ncu ? nitems : ncu;
  const float sl2e = scale * 1.4426950408889634f;
  if (lse != nullptr) {
    auto kfn = attn_short_kernel<NT, true>;
    VITM_LDS_OPTIN(kfn, SMEM, "vitm_attention_lse");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NTH), SMEM, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                       sl2e, nitems, lse);
  } else {
#ifdef ATTN_STAMP
    auto kfn = attn_short_kernel<NT, true>;
#else
    auto kfn = attn_short_kernel<NT, false>;
#endif
    VITM_LDS_OPTIN(kfn, SMEM, "vitm_attention");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NTH), SMEM, st, (const bf16_t*)qkv, ldq, (bf16_t*)out, ldo, N, H,
                       sl2e, nitems, lse);
  }
  VITM_CHECK_LAUNCH("vitm_attention (short sequences)");
  return VITM_OK;
}

int launch(const void* qkv, int64_t ldq, void* out, int64_t ldo, int B, int N, int H, float scale, float* lse,
           hipStream_t st) {
  switch ((N + 31) / 32) {
    case 1: return launch_nt<1>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 2: return launch_nt<2>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 3: return launch_nt<3>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 4: return launch_nt<4>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 5: return launch_nt<5>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    case 6: return launch_nt<6>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
    default: return launch_nt<7>(qkv, ldq, out, ldo, B, N, H, scale, lse, st);
  }
}

}  // namespace short_seq

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
  if (head_dim == 64 && (int64_t)B * H < (1ll << 28)) {
    if (N <= 32 * short_seq::MAX_TILES) return short_seq::launch(qkv, ldqkv, out, ldo, B, N, H, scale, lse, st);
    return short_seq::launch_long(qkv, ldqkv, out, ldo, B, N, H, scale, lse, st);
  }
  if (head_dim <= 0 || head_dim % 8 != 0 || head_dim > 128) {
    vitm_set_error("vitm_attention: head_dim=%d unsupported (a multiple of 8, at most 128)", head_dim);
    return VITM_EUNSUPPORTED;
  }
  const int hd = head_dim;
  switch ((hd + 31) / 32) {   // the arithmetic runs at the next multiple of 32
    case 1: return hd == 32 ? launch_attn<32, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                            : launch_attn<32, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
    case 2: return hd == 64 ? launch_attn<64, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                            : launch_attn<64, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
    case 3: return hd == 96 ? launch_attn<96, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                            : launch_attn<96, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
    default: return hd == 128 ? launch_attn<128, true>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st)
                              : launch_attn<128, false>(qkv, ldqkv, out, ldo, B, N, H, scale, lse, hd, st);
  }
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
