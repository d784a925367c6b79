// Big-tile bf16 MFMA GEMM for gfx950, persistent, with a two-group ping-pong schedule:
//     out = epilogue(A[M,K] . Wp[N,K]^T)
//
// One 512-thread workgroup (8 waves) per CU walks a list of 256x256 output tiles.  Waves 0-3 (group 0) own output
// rows 0-127 of the tile, waves 4-7 (group 1) rows 128-255; wave (g, c) computes the 128x64 block at columns 64c:
// 8x4 MFMA 16x16x32 tiles = 128 accumulator registers.  K is consumed in 64-deep K-tiles, each in two 32-deep
// sub-steps; per sub-step a wave has a LOAD segment (12 ds_read_b128 operand fragments -> registers, plus its
// share of the LDS-DMA for later K-tiles) and a COMPUTE segment (32 back-to-back MFMAs on those registers).
// Each SIMD hosts one wave of each group and the groups run one segment apart, so the matrix pipe of every SIMD
// always has one wave feeding it while the other wave does its memory work:
//
//     segment (mod 4):    0          1          2          3          0
//     group 0:        | L(t,0)   | C(t,0)   | L(t,1)   | C(t,1)   | L(t+1,0) ...
//     group 1:        | C(t-1,1) | L(t,0)   | C(t,0)   | L(t,1)   | C(t,1)   ...
//
// Segments are separated by raw s_barrier.  At the start of a tile group 1 waits one barrier (group 0's first
// LOAD) and at its end group 0 waits one (group 1's last COMPUTE); both groups then run the epilogue together.
//
// Operands go global -> LDS by LDS-DMA in 1-KiB pieces of 8 rows x 128 B (whole cache lines).  LDS holds a ring
// of 3 A tiles and 2 W tiles (256 rows x 128 B = 32 KiB each, 160 KiB in all); activations come from HBM / the
// Infinity Cache and get the deeper ring, weights sit in L2.  The K-tile stream runs straight across tile
// boundaries: while a tile's last K-tiles are being multiplied, the first K-tiles of the NEXT tile of this
// workgroup are already landing, and since the epilogue touches no LDS there is no prologue bubble between tiles.
// A piece blocks its wave for 60-180 cycles while the address unit takes it, so DMA is issued only from LOAD
// segments (never between MFMAs).  vmcnt retires in order, so the issue order is A(0) W(0) A(1), then per K-tile
// g: W(g+1) in L(g,0) and A(g+2) in L(g,1); "s_waitcnt vmcnt(4)" at the end of segment 3 leaves the 4 pieces of
// A(g+2) in flight and guarantees A(g+1), W(g+1) and everything older (output stores included) are done; the
// barrier closing segment 3 publishes them just before group 0 starts L(g+1,0).  Weights are in flight for ~4
// segments, activations for ~6 (about 3000 MFMA cycles).
//
// 128-byte LDS rows: the 16-byte chunk index is XOR-ed with (row & 7), on the DMA's per-lane SOURCE address (its
// LDS destination is lane-linear) and on the fragment reads; every ds_read_b128 lane group then hits 16 distinct
// 16-byte bank slots.
//
// Tile-count quantisation.  With T tiles on G workgroups the last round is only partly filled (591 tiles on 256
// CUs: 2.31 rounds cost 3).  When that wastes enough, a workgroup runs its whole tiles and then, in the same launch,
// a SLICE pass over the leftover tiles: each XCD group shares its leftover tiles, S = min(4, workgroups /
// leftover tiles) workgroups per tile, each taking NBLK = ceil(8 / S) of the tile's 16-row blocks (block range
// [i0, i0 + NBLK) of BOTH wave groups, i.e. rows 16 i .. and 128 + 16 i ..).  A slice moves only its own A rows
// and has NBLK x 4 MFMAs per segment; every output element is still produced by exactly one wave with the same
// K order, so results are bit-identical to the unshared schedule and need no workspace.
//
// Narrow last column tile.  N = 384 (ViT-S/16) is one and a half column tiles: computed as a 256-column tile, every
// second tile of the proj / fc2 GEMMs spends half of its MFMAs, operand bytes and stores on columns that do not exist
// (394 tiles = two full rounds of the chip for 1.15 rounds of work).  When N leaves at most 128 columns in the last of
// two column tiles, the launch runs the whole tiles first and then, in the same kernel, a NARROW pass over the last
// column: the 8 waves tile 256 rows x 128 columns as 4 row groups x 2 column waves (wave (g, c): rows 128 g + 64 (c >> 1),
// columns 64 (c & 1); 4 row blocks = 16 MFMAs per segment, same segment schedule), only 128 weight rows are moved, and
// the narrow tiles are dealt to the workgroups in reverse order, so that whoever got the fewest whole tiles gets the
// narrow ones first.
//
// Epilogue, in registers only.  The weight tile is the MFMA A operand, so a lane holds 4 consecutive output
// COLUMNS of one output row per accumulator tile; bias / folded-LayerNorm fix-up / GELU / LayerScale run on the
// fp32 accumulators in place, v_permlane16_swap pairs the quads of two neighbouring column tiles into 8
// consecutive columns per lane, a DPP row rotate pairs rows r and r+8, and each store instruction then writes 8 rows
// x 128 contiguous bytes (whole cache lines: 16 rows x 64 bytes measured 3x slower to drain).  Stores are never
// waited for inside the tile that issued them.
#include <stdlib.h>
#include "gemm_common.h"

#ifndef PP_ABLATE
#define PP_ABLATE 0   // timing-only diagnostic builds (wrong results): 1 no DMA after the prologue, 4 no output stores
#endif

#ifdef PP_STAMP  // diagnostic build: s_memtime at phase boundaries into p.resid, 20 slots per record: 0 tile start,
                 // 1 K loop end, 2 epilogue end, 3 realtime, 4 + t after K-tile t < 16.  -DPP_STAMP: one record per
                 // (workgroup, tile), written by thread 0; -DPP_STAMP=2: one per (workgroup, tile, wave), by lane 0
#if (PP_STAMP + 0) == 2
#define STAMP_AT ((((size_t)blockIdx.x * 12 + it) * 8 + wave) * 20)
#define STAMP_WHO (lane == 0)
#else
#define STAMP_AT (((size_t)blockIdx.x * 12 + it) * 20)
#define STAMP_WHO (tid == 0)
#endif
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    unsigned long long t__;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                    \
    if (!SLICE && STAMP_WHO && it < 12 && p.resid) ((unsigned long long*)p.resid)[STAMP_AT + (k)] = t__; \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

namespace {
using namespace vitm_gemm_detail;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NT = 512;
constexpr int PART = 256 * 128;             // bytes of one operand tile: 256 rows x 128 B
constexpr int NA = 3;                       // LDS ring depth of A tiles; W tiles: 2
constexpr int WBASE = NA * PART;            // W tiles follow the A tiles
constexpr int SMEM = (NA + 2) * PART;       // 163840 B = all of a CU's LDS
static_assert(SMEM <= 160 * 1024, "LDS budget");

// v_permlane16_swap: lanes 16-31 / 48-63 of `a` trade places with lanes 0-15 / 32-47 of `b`.  Afterwards a lane of
// an even 16-lane row holds (its own a, the a of lane+16) and a lane of an odd row (the b of lane-16, its own b).
// The builtin is used on unsigned operands only: fed float values cast on the fly, hipcc (ROCm 7.2) folded its second
// result into the first; floats therefore go through explicit bit casts into unsigned variables first.
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {   // integer operands: the builtin is compiled
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);   // correctly and places the wait states itself
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void swap16(float& a, float& b) {   // through the integer form (see the note above)
  unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  swap16(ua, ub);
  a = __builtin_bit_cast(float, ua);
  b = __builtin_bit_cast(float, ub);
}

// v_permlane32_swap: lanes 32-63 of `a` trade places with lanes 0-31 of `b`; afterwards a lane of the lower half holds
// (its own a, the a of lane+32) and a lane of the upper half (the b of lane-32, its own b).
__device__ __forceinline__ void swap32(float& a, float& b) {
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float ror8(float x) {   // the value of lane ^ 8
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, x), 0x128, 0xf, 0xf, false));
}

// Lanes l and l^8 (output rows r and r+8 of a 16-row accumulator tile) trade 16-byte chunks so that one store
// instruction covers 8 rows x 128 contiguous bytes (whole cache lines) instead of 16 rows x 64 bytes.  In: x0 / x1
// = this lane's row, columns [c, c+8) / [c+32, c+40).  Out: x0 goes to row (r&7) and x1 to row 8+(r&7), both at
// columns c + (r >= 8 ? 32 : 0).
__device__ __forceinline__ void rowpair_exchange1(unsigned& x0, unsigned& x1, bool /*hi8*/) {
  // two DPP moves whose bank masks pick the receiving lanes (a 16-lane row has four banks of four lanes; lanes 8-15 =
  // banks 2, 3): lanes 8-15 take x1 of lane ^ 8 as their x0, lanes 0-7 take x0 of lane ^ 8 as their x1
  const unsigned n0 = __builtin_amdgcn_update_dpp(x0, x1, 0x128, 0xf, 0xc, false);   // row_ror:8 -> value of lane ^ 8
  const unsigned n1 = __builtin_amdgcn_update_dpp(x1, x0, 0x128, 0xf, 0x3, false);
  x0 = n0;
  x1 = n1;
}
__device__ __forceinline__ void rowpair_exchange(uint4& x0, uint4& x1, bool hi8) {
  rowpair_exchange1(x0.x, x1.x, hi8);
  rowpair_exchange1(x0.y, x1.y, hi8);
  rowpair_exchange1(x0.z, x1.z, hi8);
  rowpair_exchange1(x0.w, x1.w, hi8);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// 16-byte non-temporal store of an output chunk: the output (hundreds of MB) is not read again by this kernel, and
// written with the default policy it pushes the weight tiles out of L2 (measured: +7..9 % on the K = 768 shapes)
__device__ __forceinline__ void store_out(void* ptr, unsigned a, unsigned b, unsigned c, unsigned d) {
  __builtin_nontemporal_store((u32x4_t){a, b, c, d}, (u32x4_t*)ptr);
}

__device__ __forceinline__ void seg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// all DMA of this wave landed except its `keep` youngest pieces (0, 2 or 4: the A pieces it issued last)
__device__ __forceinline__ void wait_dma(int keep) {
  if (keep == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// How the tiles of one XCD group (a contiguous run of row-major tiles) are dealt to its workgroups.
struct Deal {
  int base, wpx, full, rem, S;   // first tile of the run, workgroups in the group, whole rounds, leftover tiles, sharers per leftover tile
};
__host__ __device__ inline Deal deal_tiles(int T, int nwg, int xg, bool share, int force_S = 0) {
  Deal d;
  const int q = T >> 3, r = T & 7;
  const int rlen = q + (xg < r ? 1 : 0);
  d.wpx = (nwg - xg + 7) >> 3;
  d.base = xg * q + (xg < r ? xg : r);
  d.full = rlen / d.wpx;
  d.rem = rlen - d.full * d.wpx;
  d.S = 1;
  if (share && d.rem > 0) {
    d.S = d.wpx / d.rem;
    d.S = d.S > 4 ? 4 : (d.S < 1 ? 1 : d.S);
    if (force_S > 0) d.S = force_S;   // the launch-wide value (the smallest over the XCD groups)
  }
  return d;
}

// SLICE = false: NBLK = 8, whole tiles (p.share: the leftover tiles are left to the SLICE pass).
// SLICE = true:  one slice of NBLK 16-row blocks (of both wave groups) of one leftover tile per workgroup.
// STATS (RESIDUAL epilogue only): also write the per-row raw moments of this tile's output values (p.stats_out).
// IM2COL (patch embedding): the A operand is never materialised -- row m of A is patch (or class-token slot) m of the
// NCHW bf16 image batch p.img and the DMA gathers it: a lane's 16-byte chunk is 8 consecutive pixels of one patch row
// (patch width a power of two >= 8, so a chunk never straddles patch rows and the pixel column of a lane is the same in
// every K-tile); from one K-tile to the next a lane moves down 64 / pw image rows, wrapping into the next channel.
// NARROW (NBLK = 4, not SLICE): the pass over the last, at most 128 columns wide, column tile (see the head of the file).
template <int EPI, int NBLK, bool SLICE, bool STATS = false, bool IM2COL = false, bool NARROW = false>
__device__ __forceinline__ void tile_loop(const GemmArgs& p, char* smem) {
  static_assert(!NARROW || (NBLK == 4 && !SLICE && !IM2COL), "narrow tiles: 4 row blocks per wave, whole tiles only");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int wcol = NARROW ? (wc & 1) : wc;   // which 64-column slab of the tile this wave computes

  // ---- this workgroup's tiles.  Blocks b and b+8 share an XCD (and its L2): XCD group x = b & 7 takes a
  // contiguous run of row-major tiles and its workgroups walk that run side by side, so at any time the 32 tiles
  // in flight on an XCD share a few A row panels and W column panels.
  const int T = NARROW ? p.tilesM : p.tilesM * p.tilesN;   // NARROW: one tile per row panel, all in the last column
  const int xg = blockIdx.x & 7;
  const Deal d = deal_tiles(T, p.ncu, xg, p.share != 0, p.S);
  const int lg = NARROW ? d.wpx - 1 - (int)(blockIdx.x >> 3) : (int)(blockIdx.x >> 3);   // NARROW: dealt in reverse
  int ntl;
  int i0 = NARROW ? 4 * (wc >> 1) : 0;   // first 16-row block this wave computes (SLICE: of both wave groups alike)
  int is0 = 0, is1 = 8;                  // SLICE: blocks [is0, is1) are the ones this slice owns and stores
  int lo_tile = -1;
  if (!SLICE) {
    ntl = d.full;
    if (d.S == 1 && lg < d.rem) {   // unshared leftover tile: an ordinary last tile of this workgroup
      lo_tile = d.base + d.full * d.wpx + lg;
      ++ntl;
    }
  } else {
    ntl = 0;
    const int idx = lg / d.S;
    if (d.S > 1 && idx < d.rem) {
      const int part = lg - idx * d.S;
      lo_tile = d.base + d.full * d.wpx + idx;
      is0 = part * 8 / d.S;
      is1 = (part + 1) * 8 / d.S;
      i0 = is0 + NBLK > 8 ? 8 - NBLK : is0;
      ntl = 1;
    }
  }
  const int full = SLICE ? 0 : d.full;
  const int wpx = d.wpx;
  auto tile_of = [&](int it) { return it < full ? d.base + lg + it * wpx : lo_tile; };
  // Tile order: column bands of p.band N-tiles, row-major inside a band (the last band may be narrower).  The weight
  // rows of a band (band x 256 x K) fit an XCD's L2, so they are fetched once per XCD and band instead of once per
  // A row panel; the A panels are read once per band.
  auto tile_mn = [&](int w, int& mt, int& nt) {   // unsigned: the divisions by run-time values are half as long
    if constexpr (NARROW) {
      mt = w;
      nt = p.tilesN - 1;
      return;
    }
    const unsigned per_band = (unsigned)p.tilesM * (unsigned)p.band;
    const unsigned b = (unsigned)w / per_band, r = (unsigned)w - b * per_band;
    unsigned bw = (unsigned)p.tilesN - b * (unsigned)p.band;
    bw = bw < (unsigned)p.band ? bw : (unsigned)p.band;
    const unsigned m = r / bw;
    mt = (int)m;
    nt = (int)(b * (unsigned)p.band + (r - m * bw));
  };
  // (mt, nt) of a tile of this workgroup's list, worked out once per tile and ahead of time: the DMA stream reaches
  // the next tile in the last K-tiles of the current one, inside LOAD segments, where two run-time divisions (~700
  // cycles of dependent scalar work, twice: A and W) used to sit on the critical path
  int prep_it = -1, prep_mt = 0, prep_nt = 0;
  auto prep_tile = [&](int it) {
    tile_mn(tile_of(it), prep_mt, prep_nt);
    prep_it = it;
  };
  const int nk = p.K / BK;

  // ---- DMA: this wave moves rows 32w..32w+31 of every A tile and of every W tile, as 4 + 4 pieces of 8 rows x
  // 128 B; lane -> row lane>>3, physical chunk lane&7, logical chunk = physical ^ (row & 7)
  const int lr = lane >> 3;
  const int lc = (lane & 7) ^ lr;
  // operand addresses = a wave-uniform base (first row of the tile's rows) + a 32-bit per-lane element offset;
  // the gathering loader keeps per-lane pointers (its rows belong to different images)
  const bf16_t* asrc[IM2COL ? 4 : 1];
  const char* abase = (const char*)p.A;
  const char* wbase = (const char*)p.W;
  // byte offsets of this lane's 16 bytes inside a tile's rows: the same for every tile except a ragged last one,
  // whose rows past the end are clamped to the last row (a_rag / w_rag: the offsets currently hold clamped values)
  unsigned aob[4], wob[4];
  bool a_rag = true, w_rag = true;   // "true" forces the first set_a / set_w to fill the offsets
  // this wave's A pieces q = 0..3 carry blocks fa, fa, fa + 1, fa + 1 of its wave group's half; a slice moves only
  // the pieces [a_lo, a_hi) that lie in its block range (0, 2 or 4 of them)
  const int fa = 2 * (wave & 3);
  int a_lo = 0, a_hi = 4;
  if (SLICE) {
    a_lo = fa >= i0 ? 0 : (fa + 1 >= i0 ? 2 : 4);
    a_hi = fa + 1 < i0 + NBLK ? 4 : (fa < i0 + NBLK ? 2 : 0);
    a_hi = a_hi < a_lo ? a_lo : a_hi;
  }
  // IM2COL: this lane's pixel offset inside a patch-row-major K index, at K-tile 0 (ky0 rows down, channel folded in)
  // and as the A stream advances (a_ky, a_koff)
  int ky0 = 0, koff0 = 0, a_ky = 0, a_koff = 0;
  const int kstep = IM2COL ? (BK >> p.pw_shift) : 0;           // image rows per K-tile
  if (IM2COL) {
    ky0 = (lc * 8) >> p.pw_shift;
    koff0 = ky0 * p.Wd + ((lc * 8) & (p.pw - 1));
    while (ky0 >= p.ph) {
      ky0 -= p.ph;
      koff0 += (p.H - p.ph) * p.Wd;
    }
  }
  auto set_a = [&](int it) {
    if (it != prep_it) prep_tile(it);
    const int am0 = prep_mt * BM;
    if constexpr (!IM2COL) {
      abase = (const char*)(p.A + (int64_t)am0 * p.lda);
      const bool rag = am0 + BM > p.M;
      if (rag || a_rag) {   // wave-uniform and rare after the first tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int r = (4 * wave + q) * 8 + lr;
          r = am0 + r < p.M ? r : p.M - 1 - am0;
          aob[q] = ((unsigned)r * (unsigned)p.lda + lc * 8) * 2u;
        }
      }
      a_rag = rag;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int am = am0 + (4 * wave + q) * 8 + lr;
      am = am < p.M ? am : p.M - 1;
      if constexpr (IM2COL) {
        const int ntok = p.npatch + p.cls;
        const int b = am / ntok;
        int t = am - b * ntok - p.cls;       // a class-token slot reads patch 0; its output row is not stored
        t = t < 0 ? 0 : t;
        const int py = t / p.gw, px = t - py * p.gw;
        asrc[q] = (const bf16_t*)p.img + (((int64_t)b * p.C * p.H + py * p.ph) * p.Wd + px * p.pw);
      }
    }
    a_ky = ky0;
    a_koff = koff0;
  };
  auto advance_a_k = [&]() {   // IM2COL: one K-tile further down the patch
    a_ky += kstep;
    a_koff += kstep * p.Wd;
    while (a_ky >= p.ph) {
      a_ky -= p.ph;
      a_koff += (p.H - p.ph) * p.Wd;
    }
  };
  auto set_w = [&](int it) {
    if (it != prep_it) prep_tile(it);
    const int wn0 = prep_nt * BN;
    wbase = (const char*)(p.W + (int64_t)wn0 * p.ldw);
    const bool rag = wn0 + BN > p.N;
    if (rag || w_rag) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int r = (4 * wave + q) * 8 + lr;
        r = wn0 + r < p.N ? r : p.N - 1 - wn0;
        wob[q] = ((unsigned)r * (unsigned)p.ldw + lc * 8) * 2u;
      }
    }
    w_rag = rag;
  };
  set_a(0);
  set_w(0);
  int a_it = 0, a_t = 0, a_slot = 0;   // next A K-tile to issue: tile a_it of this workgroup, K-tile a_t, ring slot
  int w_it = 0, w_t = 0, w_slot = 0;
  auto issueA = [&]() -> int {   // returns the number of pieces issued
    if (a_it >= ntl) return 0;
    char* sb = smem + a_slot * PART + 4 * wave * 1024;
    const int ao = IM2COL ? a_koff : a_t * BK;
    const char* ab = abase + (size_t)a_t * (BK * 2);   // wave-uniform: the address is scalar base + 32-bit lane offset
    auto src = [&](int q) -> const void* {
      if constexpr (IM2COL) return asrc[q] + ao;
      else return ab + aob[q];
    };
    int n = 4;
    if (!SLICE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(src(q), sb + q * 1024);
    } else {
      n = a_hi - a_lo;
      if (a_lo == 0 && n >= 2) {
        glds16(src(0), sb);
        glds16(src(1), sb + 1024);
      }
      if (a_hi == 4 && n >= 2) {
        glds16(src(2), sb + 2048);
        glds16(src(3), sb + 3072);
      }
    }
    a_slot = a_slot == NA - 1 ? 0 : a_slot + 1;
    if (++a_t == nk) {
      a_t = 0;
      if (++a_it < ntl) set_a(a_it);
    } else if (IM2COL) {
      advance_a_k();
    }
    return n;
  };
  // first / second half of the next A K-tile (pieces 0-1 / 2-3); the second call advances the stream
  auto issueA_half = [&](int half) -> int {
    if (a_it >= ntl) return 0;
    char* sb = smem + a_slot * PART + 4 * wave * 1024;
    const char* ab = abase + (size_t)a_t * (BK * 2);
    glds16(ab + aob[2 * half], sb + (2 * half) * 1024);
    glds16(ab + aob[2 * half + 1], sb + (2 * half + 1) * 1024);
    if (half == 1) {
      a_slot = a_slot == NA - 1 ? 0 : a_slot + 1;
      if (++a_t == nk) {
        a_t = 0;
        if (++a_it < ntl) set_a(a_it);
      }
    }
    return 2;
  };
  auto issueW = [&]() {
    if (w_it >= ntl) return;
    char* sb = smem + WBASE + w_slot * PART + 4 * wave * 1024;
    const char* wb = wbase + (size_t)w_t * (BK * 2);
    if (!NARROW || wave < 4) {   // a narrow tile has 128 weight rows: waves 0-3 move them
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(wb + wob[q], sb + q * 1024);
    }
    w_slot ^= 1;
    if (++w_t == nk) {
      w_t = 0;
      if (++w_it < ntl) set_w(w_it);
    }
  };

  // ---- fragment addressing (tile-relative): lane reads row (lane&15) of a 16-row block, logical chunk 4kk + lane>>4
  int offA[2], offB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int pch = ((kk * 4 + (lane >> 4)) ^ (lane & 7)) * 16;
    offA[kk] = (grp * 128 + i0 * 16 + (lane & 15)) * 128 + pch;
    offB[kk] = WBASE + (wcol * 64 + (lane & 15)) * 128 + pch;
  }
  int ca = 0, cw = 0;  // ring slots of the K-tile being consumed

  f32x4_t acc[NBLK][4];
  bf16x8_t a[NBLK], b[4];
  auto load_frags = [&](int kk) {
    const char* sa = smem + ca * PART;
    const char* sw = smem + cw * PART;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8_t*)(sw + offB[kk] + j * 2048);
#pragma unroll
    for (int i = 0; i < NBLK; ++i) a[i] = *(const bf16x8_t*)(sa + offA[kk] + i * 2048);
  };
  // the weight tile is the MFMA A operand: a lane ends up with 4 consecutive output COLUMNS of one output row
  auto compute = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue of the K-tile stream: A(0) W(0) A(1) in flight, K-tile 0 landed and published
  issueA();
  issueW();
  wait_dma(issueA());
  seg_barrier();
  const bool dma = (PP_ABLATE & 1) == 0;

  for (int it = 0; it < ntl; ++it) {
    if (it != prep_it) prep_tile(it);          // normally prepared while the previous tile ran
    const int mt_ = prep_mt, nt_ = prep_nt;
    const int m0 = mt_ * BM;
    const int n0 = nt_ * BN;
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (it + 1 < ntl) prep_tile(it + 1);   // for the DMA stream, which gets there in this tile's last K-tiles
    STAMP(0);
#ifdef PP_STAMP
    if (!SLICE && STAMP_WHO && it < 12 && p.resid) ((unsigned long long*)p.resid)[STAMP_AT + 3] = __builtin_amdgcn_s_memrealtime();
#endif
    if (grp == 1) seg_barrier();   // group 0's first LOAD of the tile
    for (int t = 0; t < nk; ++t) {
      // ---- L(t,0)
      load_frags(0);
#if (PP_ABLATE & 8)
      issueA_half(0);
#else
      if (dma) issueW();           // W(g+1)
#endif
      wait_lds();
      seg_barrier();
      // ---- C(t,0)
      compute();
      seg_barrier();
      // ---- L(t,1)
      load_frags(1);
#if (PP_ABLATE & 8)
      const int keep = 2 * issueA_half(1);
#else
      const int keep = dma ? issueA() : 0;   // A(g+2)
#endif
      wait_lds();
      if (grp == 1) wait_dma(keep);   // group 1: its L(t,1) IS segment 3
      seg_barrier();
      // ---- C(t,1)
      compute();
      if (grp == 0) wait_dma(keep);   // group 0: its C(t,1) is segment 3
      seg_barrier();
      ca = ca == NA - 1 ? 0 : ca + 1;
      cw ^= 1;
#ifdef PP_STAMP
      if (t < 16) STAMP(4 + t);   // after K-tile t
#endif
    }
    if (grp == 0) seg_barrier();   // group 1's last COMPUTE of the tile
    STAMP(1);

    // ---- epilogue (registers only).  acc[i][j][r] = out[m0 + 128 grp + 16 i + (lane & 15)][n0 + 64 wc + 16 j + 4 q + r],
    // q = lane >> 4.  After the permlane16 swap of tiles (j, j+1), j even, a lane with q even holds columns
    // 16 j + 4 q .. +7 and a lane with q odd holds columns 16 (j+1) + 4 (q-1) .. +7.
    const int q4 = lane >> 4;
    const bool hi8 = (lane & 8) != 0;
    const int mrow = m0 + grp * 128 + i0 * 16 + (lane & 15);  // + 16 i  (accumulator layout)
    const int mst = m0 + grp * 128 + i0 * 16 + (lane & 7);    // + 16 i (+ 8 for the second store)
    const int ncol = n0 + wcol * 64 + 4 * q4;                 // + 16 j   (accumulator layout)
    const int nsw = n0 + wcol * 64 + ((q4 & 1) ? 16 + 4 * (q4 - 1) : 4 * q4);   // + 32 jp: columns held after the swap
    const int nst = nsw + (hi8 ? 32 : 0);                                       // columns stored after the row-pair exchange
    float4 cq[4], sq[4], gq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = ncol + 16 * j;
      n = n < p.N ? n : p.N - 4;   // N % 8 == 0: a clamped quad is computed but never stored
      if (EPI != VITM_EPI_DGELU) cq[j] = *(const float4*)(p.c + n);
      if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) sq[j] = *(const float4*)(p.s + n);
      if (EPI == VITM_EPI_RESIDUAL) gq[j] = *(const float4*)(p.g + n);
    }
    float2 st[NBLK];
    if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) {
      // the (wave-uniform) layout test sits outside the row loop: inside it every row block became a basic block of
      // its own, i.e. load, wait for everything in flight, combine, eight times in a row
      if (!p.stats_partial) {
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          int m = mrow + 16 * i;
          m = m < p.M ? m : p.M - 1;
          st[i] = *(const float2*)(p.stats + 2 * (int64_t)m);
        }
      } else {
        // raw moments per 256-column tile of the producing GEMM: combine, then mean and rstd.  The four 16-lane rows
        // of a wave need the same 16 x NBLK output rows, so each takes two of the row blocks (4 loads instead of 16,
        // a quarter of the arithmetic) and the results go round by cross-lane reads
        float2 mine[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          int i = 2 * q4 + e;
          i = i < NBLK ? i : NBLK - 1;
          int m = mrow + 16 * i;
          m = m < p.M ? m : p.M - 1;
          const float4 u = *(const float4*)(p.stats + 2 * VITM_STATS_PARTS * (int64_t)m);
          const float4 w = *(const float4*)(p.stats + 2 * VITM_STATS_PARTS * (int64_t)m + 4);
          const float mean = ((u.x + u.z) + (w.x + w.z)) * p.ln_inv_dim;
          const float var = fmaf(-mean, mean, ((u.y + u.w) + (w.y + w.w)) * p.ln_inv_dim);
          mine[e] = make_float2(mean, __builtin_amdgcn_rsqf(fmaxf(var, 0.f) + p.ln_eps));
        }
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
          const int src = (i >> 1) * 16 + (lane & 15);
          st[i] = make_float2(__shfl(mine[i & 1].x, src, 64), __shfl(mine[i & 1].y, src, 64));
        }
      }
    }
    const bool stores_on = (PP_ABLATE & 4) == 0;
    if (EPI != VITM_EPI_RESIDUAL && EPI != VITM_EPI_DGELU) {
      // The two waves of a SIMD (one per wave group) run this epilogue side by side, and at equal priority the older
      // one (group 0) wins every issue slot: it finishes ~5 k cycles before group 1 and then waits for it at the first
      // barrier of the next tile.  Group 1 therefore goes first for three quarters of its blocks (per-wave stamps:
      // fc1 launch -1.5 %)
      if (grp == 1) __builtin_amdgcn_s_setprio(2);
      // a tile with no ragged edge stores without per-lane tests, from one pointer advanced by 16 rows per block (the
      // general path costs every block ~14 VALU instructions of address arithmetic and compares and four branches)
      const bool fast = !SLICE && stores_on && m0 + BM <= p.M && n0 + (NARROW ? 128 : BN) <= p.N && (!IM2COL || p.cls == 0);
      // The exchanges above leave lane l with row (l & 7) and 16-byte chunk CH(l >> 3) of the wave's 128-byte rows.  Stored
      // like that, the 8 lanes of a row are 8 lanes apart, and the memory path takes 3.5 x as long over the instruction as
      // when consecutive lanes write consecutive chunks (8.7 k against 2.5 k cycles per 128-KiB tile on a CU by itself).
      // Full tiles therefore turn the 8 x 8 lane grid round first (4 cross-lane reads per store): lane l ends up with row
      // l >> 3, chunk l & 7.
      // Not where the GELU keeps the VALU busier than the stores keep the memory path (the stores hide behind it).
      constexpr bool TURN = EPI == VITM_EPI_BIAS || EPI == VITM_EPI_LN;
      const int xsrc = 4 * ((lane >> 3) + 8 * ((lane & 7) >> 2) + 16 * (((lane & 1) << 1) | ((lane >> 1) & 1)));
      bf16_t* ofast = TURN ? (bf16_t*)p.out + (int64_t)(m0 + grp * 128 + i0 * 16 + (lane >> 3)) * p.ldo + (n0 + wcol * 64 + 8 * (lane & 7))
                           : (bf16_t*)p.out + (int64_t)mst * p.ldo + nst;
      auto turn = [&](unsigned x) { return TURN ? (unsigned)__builtin_amdgcn_ds_bpermute(xsrc, (int)x) : x; };
      bf16_t* o2fast = EPI == VITM_EPI_BIAS_GELU2 ? const_cast<bf16_t*>(p.resid) + (int64_t)mst * p.ldr + nst : nullptr;
#pragma unroll
      for (int i = 0; i < NBLK; ++i) {
        if (i == (3 * NBLK + 3) / 4 && grp == 1) __builtin_amdgcn_s_setprio(0);
        f32x2_t v[8];  // the 16 values of row i: pairs (j, half)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2_t c01 = {cq[j].x, cq[j].y}, c23 = {cq[j].z, cq[j].w};
          v[2 * j] = (f32x2_t){acc[i][j][0], acc[i][j][1]};
          v[2 * j + 1] = (f32x2_t){acc[i][j][2], acc[i][j][3]};
          if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) {
            const f32x2_t s01 = {sq[j].x, sq[j].y}, s23 = {sq[j].z, sq[j].w};
            v[2 * j] = (v[2 * j] - s01 * st[i].x) * st[i].y + c01;       // rstd * (acc - mean * s) + c
            v[2 * j + 1] = (v[2 * j + 1] - s23 * st[i].x) * st[i].y + c23;
          } else {
            v[2 * j] += c01;
            v[2 * j + 1] += c23;
          }
        }
        if (EPI == VITM_EPI_LN_GELU || EPI == VITM_EPI_BIAS_GELU) gelu_erf2_n<8>(v);
        // chunk jp = this lane's row, columns nsw + 32 jp .. +7, as dwords (a, b, c, d)
        unsigned a0 = pack_bf2(v[0].x, v[0].y), b0 = pack_bf2(v[1].x, v[1].y);
        unsigned c0 = pack_bf2(v[2].x, v[2].y), d0 = pack_bf2(v[3].x, v[3].y);
        unsigned a1 = pack_bf2(v[4].x, v[4].y), b1 = pack_bf2(v[5].x, v[5].y);
        unsigned c1 = pack_bf2(v[6].x, v[6].y), d1 = pack_bf2(v[7].x, v[7].y);
        swap16(a0, c0);   // (tile 0 quad lo, tile 1 quad lo) -> 8 consecutive columns: a b | c d
        swap16(b0, d0);
        swap16(a1, c1);
        swap16(b1, d1);
        rowpair_exchange1(a0, a1, hi8);
        rowpair_exchange1(b0, b1, hi8);
        rowpair_exchange1(c0, c1, hi8);
        rowpair_exchange1(d0, d1, hi8);
        if (fast) {
          store_out(ofast, turn(a0), turn(b0), turn(c0), turn(d0));
          store_out(ofast + 8 * p.ldo, turn(a1), turn(b1), turn(c1), turn(d1));
          ofast += 16 * p.ldo;
        } else if (stores_on && nst < p.N && (!SLICE || (i0 + i >= is0 && i0 + i < is1))) {
          const int m1 = mst + 16 * i, m2 = m1 + 8;
          bf16_t* o = (bf16_t*)p.out + (int64_t)m1 * p.ldo + nst;
          if (!IM2COL || p.cls == 0) {
            if (m1 < p.M) store_out(o, a0, b0, c0, d0);
            if (m2 < p.M) store_out(o + 8 * p.ldo, a1, b1, c1, d1);
          } else {   // rows that are class-token slots are left untouched
            const int ntok = p.npatch + 1;
            if (m1 < p.M && m1 % ntok != 0) store_out(o, a0, b0, c0, d0);
            if (m2 < p.M && m2 % ntok != 0) store_out(o + 8 * p.ldo, a1, b1, c1, d1);
          }
        }
        if (EPI == VITM_EPI_BIAS_GELU2) {
          // second output: the GELU of the same values, in the same lane layout, to p.resid
          gelu_erf2_n<8>(v);
          a0 = pack_bf2(v[0].x, v[0].y), b0 = pack_bf2(v[1].x, v[1].y);
          c0 = pack_bf2(v[2].x, v[2].y), d0 = pack_bf2(v[3].x, v[3].y);
          a1 = pack_bf2(v[4].x, v[4].y), b1 = pack_bf2(v[5].x, v[5].y);
          c1 = pack_bf2(v[6].x, v[6].y), d1 = pack_bf2(v[7].x, v[7].y);
          swap16(a0, c0);
          swap16(b0, d0);
          swap16(a1, c1);
          swap16(b1, d1);
          rowpair_exchange1(a0, a1, hi8);
          rowpair_exchange1(b0, b1, hi8);
          rowpair_exchange1(c0, c1, hi8);
          rowpair_exchange1(d0, d1, hi8);
          if (fast) {
            store_out(o2fast, a0, b0, c0, d0);
            store_out(o2fast + 8 * p.ldr, a1, b1, c1, d1);
            o2fast += 16 * p.ldr;
          } else if (stores_on && nst < p.N && (!SLICE || (i0 + i >= is0 && i0 + i < is1))) {
            const int m1 = mst + 16 * i, m2 = m1 + 8;
            bf16_t* o2 = const_cast<bf16_t*>(p.resid) + (int64_t)m1 * p.ldr + nst;
            if (m1 < p.M) store_out(o2, a0, b0, c0, d0);
            if (m2 < p.M) store_out(o2 + 8 * p.ldr, a1, b1, c1, d1);
          }
        }
      }
      if (grp == 1) __builtin_amdgcn_s_setprio(0);   // (NBLK < 2: the reset inside the loop did not run)
    } else {
      // residual: out = resid + (g * acc + c), one rounding.  The fp32 values are swapped (4 dwords per quad) so
      // the residual is read and the result written as 16 bytes per lane.  Two halves of 4 rows keep the
      // residual values of a half in registers before its first store.  DGELU uses the same layout with
      // out = acc * gelu'(resid).
      constexpr int I4 = NBLK < 4 ? NBLK : 4, IH = NBLK / I4;
      float rs[8][2], rq[8][2];   // STATS: moments of this lane's 8 values of rows (16 i + 8 h + (lane & 7))
      if (STATS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rs[i][0] = rs[i][1] = rq[i][0] = rq[i][1] = 0.f;
      }
      // full tiles: residual and output through one advancing pointer each, no per-lane tests (see the bias branch)
      const bool fast = !SLICE && stores_on && m0 + BM <= p.M && n0 + (NARROW ? 128 : BN) <= p.N;
      // (no lane-grid turn here: with the residual read and the output written, the 8 cross-lane reads per 16-row block
      // cost more than the faster stores return: proj +3.6 %, fc2 +1.9 % in the kernel trace, A/B of the whole forward)
      const bf16_t* rfast = p.resid + (int64_t)mst * p.ldr + nst;
      bf16_t* ofast = (bf16_t*)p.out + (int64_t)mst * p.ldo + nst;
#pragma unroll
      for (int ih = 0; ih < IH; ++ih) {
        uint4 rres[I4][2];   // residual in store layout: [.][0] row (r&7), [.][1] row 8 + (r&7), columns nst .. +7
        if (fast) {
#pragma unroll
          for (int i4 = 0; i4 < I4; ++i4) {
            rres[i4][0] = *(const uint4*)rfast;
            rres[i4][1] = *(const uint4*)(rfast + 8 * p.ldr);
            rfast += 16 * p.ldr;
          }
        } else {
#pragma unroll
          for (int i4 = 0; i4 < I4; ++i4)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              int m = mst + 16 * (I4 * ih + i4) + 8 * h;
              m = m < p.M ? m : p.M - 1;
              const int n = nst < p.N ? nst : p.N - 8;
              rres[i4][h] = *(const uint4*)(p.resid + (int64_t)m * p.ldr + n);
            }
        }
#pragma unroll
        for (int i4 = 0; i4 < I4; ++i4) {
          const int i = I4 * ih + i4;
          unsigned lo0[4], hi0[4], lo1[4], hi1[4];   // fp32 bits of chunk 0 / chunk 1: columns +0..3 (lo) and +4..7 (hi)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g0[4] = {gq[0].x, gq[0].y, gq[0].z, gq[0].w}, cc0[4] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w};
            const float g1[4] = {gq[1].x, gq[1].y, gq[1].z, gq[1].w}, cc1[4] = {cq[1].x, cq[1].y, cq[1].z, cq[1].w};
            const float g2[4] = {gq[2].x, gq[2].y, gq[2].z, gq[2].w}, cc2[4] = {cq[2].x, cq[2].y, cq[2].z, cq[2].w};
            const float g3[4] = {gq[3].x, gq[3].y, gq[3].z, gq[3].w}, cc3[4] = {cq[3].x, cq[3].y, cq[3].z, cq[3].w};
            float l0, h0, l1, h1;
            if (EPI == VITM_EPI_DGELU) {
              l0 = acc[i][0][e]; h0 = acc[i][1][e]; l1 = acc[i][2][e]; h1 = acc[i][3][e];
            } else {
              l0 = fmaf(g0[e], acc[i][0][e], cc0[e]); h0 = fmaf(g1[e], acc[i][1][e], cc1[e]);
              l1 = fmaf(g2[e], acc[i][2][e], cc2[e]); h1 = fmaf(g3[e], acc[i][3][e], cc3[e]);
            }
            swap16(l0, h0);
            swap16(l1, h1);
            lo0[e] = __builtin_bit_cast(unsigned, l0);
            hi0[e] = __builtin_bit_cast(unsigned, h0);
            lo1[e] = __builtin_bit_cast(unsigned, l1);
            hi1[e] = __builtin_bit_cast(unsigned, h1);
            rowpair_exchange1(lo0[e], lo1[e], hi8);
            rowpair_exchange1(hi0[e], hi1[e], hi8);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint4 rv = rres[i4][h];
            const unsigned* l4 = h ? lo1 : lo0;
            const unsigned* h4 = h ? hi1 : hi0;
            uint4 o;
            if (EPI == VITM_EPI_DGELU) {
              const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
              unsigned ow[4];
#pragma unroll
              for (int w2 = 0; w2 < 4; ++w2) {
                const unsigned* a4 = w2 < 2 ? l4 : h4;
                const f32x2_t d = dgelu_erf2((f32x2_t){__builtin_bit_cast(float, rw[w2] << 16),
                                                       __builtin_bit_cast(float, rw[w2] & 0xffff0000u)});
                ow[w2] = pack_bf2(__builtin_bit_cast(float, a4[2 * (w2 & 1)]) * d.x,
                                  __builtin_bit_cast(float, a4[2 * (w2 & 1) + 1]) * d.y);
              }
              o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else {
              const f32x2_t f0 = (f32x2_t){__builtin_bit_cast(float, rv.x << 16), __builtin_bit_cast(float, rv.x & 0xffff0000u)} +
                                 (f32x2_t){__builtin_bit_cast(float, l4[0]), __builtin_bit_cast(float, l4[1])};
              const f32x2_t f1 = (f32x2_t){__builtin_bit_cast(float, rv.y << 16), __builtin_bit_cast(float, rv.y & 0xffff0000u)} +
                                 (f32x2_t){__builtin_bit_cast(float, l4[2]), __builtin_bit_cast(float, l4[3])};
              const f32x2_t f2 = (f32x2_t){__builtin_bit_cast(float, rv.z << 16), __builtin_bit_cast(float, rv.z & 0xffff0000u)} +
                                 (f32x2_t){__builtin_bit_cast(float, h4[0]), __builtin_bit_cast(float, h4[1])};
              const f32x2_t f3 = (f32x2_t){__builtin_bit_cast(float, rv.w << 16), __builtin_bit_cast(float, rv.w & 0xffff0000u)} +
                                 (f32x2_t){__builtin_bit_cast(float, h4[2]), __builtin_bit_cast(float, h4[3])};
              o.x = pack_bf2(f0.x, f0.y);
              o.y = pack_bf2(f1.x, f1.y);
              o.z = pack_bf2(f2.x, f2.y);
              o.w = pack_bf2(f3.x, f3.y);
              if (STATS) {
                // moments of the fp32 values before the bf16 rounding (the rounding moves the mean by ~1e-4 of the
                // row's standard deviation and the variance by ~1e-5 of itself); columns past N count as nothing
                const f32x2_t sp = (f0 + f1) + (f2 + f3);
                const f32x2_t qp = f0 * f0 + f1 * f1 + (f2 * f2 + f3 * f3);
                const bool colok = nst < p.N;
                rs[i][h] = colok ? sp.x + sp.y : 0.f;
                rq[i][h] = colok ? qp.x + qp.y : 0.f;
              }
            }
            const int m = mst + 16 * i + 8 * h;
            if (fast)
              store_out(ofast + (h ? 8 * p.ldo : 0), o.x, o.y, o.z, o.w);
            else if (stores_on && m < p.M && nst < p.N && (!SLICE || (i0 + i >= is0 && i0 + i < is1)))
              store_out((bf16_t*)p.out + (int64_t)m * p.ldo + nst, o.x, o.y, o.z, o.w);
          }
          ofast += 16 * p.ldo;
        }
      }
      if (STATS) {
        // Sum each row's moments over the 8 lanes that hold its 64 columns (lane bits 3, 4, 5) with a halving
        // exchange: at every step a lane gives away half of its rows and completes the other half.  Afterwards
        // lane l holds rows 16 i + 8 h + (l & 7), i = 4 j + 2 (l >> 5 & 1) + (l >> 4 & 1), h = l >> 3 & 1, j = 0, 1.
        float as[8], aq[8], bs[4], bq[4], cs[2], cq2[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          as[i] = (hi8 ? rs[i][1] : rs[i][0]) + ror8(hi8 ? rs[i][0] : rs[i][1]);
          aq[i] = (hi8 ? rq[i][1] : rq[i][0]) + ror8(hi8 ? rq[i][0] : rq[i][1]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          swap16(as[2 * k], as[2 * k + 1]);
          swap16(aq[2 * k], aq[2 * k + 1]);
          bs[k] = as[2 * k] + as[2 * k + 1];
          bq[k] = aq[2 * k] + aq[2 * k + 1];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          swap32(bs[2 * j], bs[2 * j + 1]);
          swap32(bq[2 * j], bq[2 * j + 1]);
          cs[j] = bs[2 * j] + bs[2 * j + 1];
          cq2[j] = bq[2 * j] + bq[2 * j + 1];
        }
        // the four waves of a wave group meet in the A ring slot that the last K-tile just left (nothing lands there
        // before the next tile's second LOAD segment): [row 0..255][wave column 0..3] (sum, sum of squares)
        char* sb = smem + (ca == 0 ? NA - 1 : ca - 1) * PART;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ib = 4 * j + 2 * ((lane >> 5) & 1) + ((lane >> 4) & 1);
          const int R = 16 * ib + (hi8 ? 8 : 0) + (lane & 7);
          if constexpr (NARROW) {   // rows are absolute inside the wave group's half (this wave owns blocks i0 .. i0 + 3); two column waves
            if (ib < NBLK) *(float2*)(sb + ((grp * 128 + i0 * 16 + R) * 4 + wcol) * 8) = make_float2(cs[j], cq2[j]);
          } else {
            *(float2*)(sb + ((grp * 128 + R) * 4 + wc) * 8) = make_float2(cs[j], cq2[j]);
          }
        }
        wait_lds();
        seg_barrier();
        if (tid < 256) {
          const float4 u = *(const float4*)(sb + tid * 32), w = *(const float4*)(sb + tid * 32 + 16);
          const int R = tid & 127, ib = R >> 4;
          const int m = NARROW ? m0 + (tid >> 7) * 128 + R : m0 + (tid >> 7) * 128 + i0 * 16 + R;
          if ((NARROW || ib < NBLK) && m < p.M && (!SLICE || (i0 + ib >= is0 && i0 + ib < is1))) {
            float2* so = (float2*)(p.stats_out + ((int64_t)m * VITM_STATS_PARTS + nt_) * 2);
            *so = NARROW ? make_float2(u.x + u.z, u.y + u.w) : make_float2((u.x + u.z) + (w.x + w.z), (u.y + u.w) + (w.y + w.w));
            // the last column tile also clears the slots of the column tiles this shape does not have: the consumer adds
            // all VITM_STATS_PARTS slots, and the buffer then needs no clearing by whoever allocated it
            if (nt_ == p.stats_last)
              for (int z = nt_ + 1; z < VITM_STATS_PARTS; ++z) so[z - nt_] = make_float2(0.f, 0.f);
          }
        }
      }
    }
    STAMP(2);
  }
}

// Whole tiles only (every tile of the launch when nothing is shared).  KDEEP (here and below) changes no code: it
// gives the K > 1024 launches of the moments-writing RESIDUAL epilogue (fc2) a kernel name of their own, so that a
// kernel trace shows them apart from the K <= 1024 ones (proj), which otherwise share every template argument.
template <int EPI, bool STATS = false, bool IM2COL = false, bool KDEEP = false>
__global__ __launch_bounds__(NT, 2) void gemm_pp_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tile_loop<EPI, 8, false, STATS, IM2COL>(p, smem);
}

// Whole rounds, then this workgroup's slice of a shared leftover tile (NBLK_S 16-row blocks of both wave groups), in
// one launch: a workgroup moves on to its slice as soon as its own whole tiles are done, instead of waiting at a kernel
// boundary for the slowest workgroup of the chip.
template <int EPI, int NBLK_S, bool STATS = false, bool IM2COL = false, bool KDEEP = false>
__global__ __launch_bounds__(NT, 2) void gemm_pp_shared_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tile_loop<EPI, 8, false, STATS, IM2COL>(p, smem);
  wait_lds();       // the LDS reads of the last whole tile (moments combine) are over before the ring is refilled
  seg_barrier();
  tile_loop<EPI, NBLK_S, true, STATS, IM2COL>(p, smem);
}

// Whole tiles of every column but the last, then the narrow last column (see the head of the file).
template <int EPI, bool STATS = false, bool KDEEP = false>
__global__ __launch_bounds__(NT, 2) void gemm_pp_narrow_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmArgs q = p;             // the whole-tile pass sees a matrix without the last column tile
  q.tilesN = p.tilesN - 1;
  q.N = q.tilesN * BN;
  q.band = q.tilesN;
  tile_loop<EPI, 8, false, STATS, false>(q, smem);
  wait_lds();
  seg_barrier();
  tile_loop<EPI, 4, false, STATS, false, true>(p, smem);
}

template <int EPI, int NBLK_S, bool STATS = false, bool IM2COL = false, bool KDEEP = false>   // NBLK_S = 8: nothing shared
int launch_one(const GemmArgs& b, int grid, hipStream_t st) {
  if constexpr (NBLK_S == 8) {
    auto kfn = gemm_pp_kernel<EPI, STATS, IM2COL, KDEEP>;
    VITM_LDS_OPTIN(kfn, SMEM, "vitm_gemm");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NT), SMEM, st, b);
  } else {
    auto kfn = gemm_pp_shared_kernel<EPI, NBLK_S, STATS, IM2COL, KDEEP>;
    VITM_LDS_OPTIN(kfn, SMEM, "vitm_gemm");
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NT), SMEM, st, b);
  }
  VITM_CHECK_LAUNCH("vitm_gemm (256x256 tile)");
  return VITM_OK;
}

// S workgroups per leftover tile -> blocks per slice
template <int EPI, bool STATS, bool IM2COL, bool KDEEP = false>
int launch_shared(const GemmArgs& b, int grid, int S, hipStream_t st) {
  switch (S) {
    case 2: return launch_one<EPI, 4, STATS, IM2COL, KDEEP>(b, grid, st);
    case 3: return launch_one<EPI, 3, STATS, IM2COL, KDEEP>(b, grid, st);
    default: return launch_one<EPI, 2, STATS, IM2COL, KDEEP>(b, grid, st);
  }
}

template <int EPI>
int launch(const GemmArgs& a, hipStream_t st) {
  GemmArgs b = a;
  b.tilesN = (a.N + BN - 1) / BN;
  b.tilesM = (a.M + BM - 1) / BM;
  {
    // weight bytes of one N-tile: BN x K x 2; a band is what stays resident in a 4 MiB L2 next to the A stream
    static const int64_t band_bytes = [] { const char* e = getenv("VITM_PP_BAND_KB"); return (int64_t)(e ? atoi(e) : 2560) * 1024; }();
    int64_t band = band_bytes / ((int64_t)BN * a.K * 2);
    band = band < 1 ? 1 : band;
    // even bands, and none when a single band would hold more than half of the tiles anyway
    const int64_t nb = (b.tilesN + band - 1) / band;
    band = (b.tilesN + nb - 1) / nb;
    b.band = (band_bytes > 0 && a.K <= 1024 && nb > 1) ? (int)band : b.tilesN;
  }
  const int ncu = vitm_cu_count();
  const int64_t tiles = (int64_t)b.tilesM * b.tilesN;
  int grid = (int)(tiles < ncu ? tiles : ncu);   // one persistent workgroup per CU
#ifdef PP_STAMP
  if (const char* e = getenv("VITM_PP_GRID")) grid = atoi(e);   // diagnostic: fewer workgroups than CUs
#endif
  b.ncu = grid;
  // Share the leftover tiles of a partly filled last round when that round would be less than ~70 % full and
  // there is at least one whole round to amortise the slice pass over.
  const int rounds = (int)((tiles + grid - 1) / grid);
  const int64_t left = tiles - (int64_t)(rounds - 1) * grid;
  b.share = (rounds >= 2 && left * 10 < (int64_t)grid * 7 && (grid & 7) == 0) ? 1 : 0;
  int S = 1;
  if (b.share) {
    // every XCD group must agree on S (the SLICE instantiation is chosen per launch): take the smallest
    S = 4;
    for (int x = 0; x < 8; ++x) {
      const Deal d = deal_tiles((int)tiles, grid, x, true);
      if (d.rem > 0 && d.S < S) S = d.S;
    }
    if (S < 2) b.share = 0;
  }
  b.S = S;
  b.stats_last = b.tilesN - 1;
  // two column tiles of which the second is at most half full (N = 257 .. 384): whole tiles + narrow pass, nothing shared
  const int nrest = a.N - (b.tilesN - 1) * BN;
  if (b.tilesN == 2 && nrest <= 128 && b.img == nullptr && (grid & 7) == 0) {
    b.share = 0;
    b.S = 1;
    auto go = [&](auto kfn) {
      VITM_LDS_OPTIN(kfn, SMEM, "vitm_gemm");
      hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NT), SMEM, st, b);
      VITM_CHECK_LAUNCH("vitm_gemm (256x256 tile, narrow last column)");
      return VITM_OK;
    };
    if constexpr (EPI == VITM_EPI_RESIDUAL) {
      if (b.stats_out != nullptr) return a.K > 1024 ? go(gemm_pp_narrow_kernel<EPI, true, true>) : go(gemm_pp_narrow_kernel<EPI, true, false>);
    }
    return go(gemm_pp_narrow_kernel<EPI, false, false>);
  }
  if (EPI == VITM_EPI_RESIDUAL && b.stats_out != nullptr) {
    constexpr int E = VITM_EPI_RESIDUAL;   // the only epilogue whose output feeds a LayerNorm
    if (a.K > 1024) return b.share ? launch_shared<E, true, false, true>(b, grid, S, st) : launch_one<E, 8, true, false, true>(b, grid, st);
    return b.share ? launch_shared<E, true, false>(b, grid, S, st) : launch_one<E, 8, true>(b, grid, st);
  }
  if constexpr (EPI == VITM_EPI_BIAS) {
    if (b.img != nullptr)   // patch embedding: A gathered from the image
      return b.share ? launch_shared<EPI, false, true>(b, grid, S, st) : launch_one<EPI, 8, false, true>(b, grid, st);
  }
  return b.share ? launch_shared<EPI, false, false>(b, grid, S, st) : launch_one<EPI, 8>(b, grid, st);
}

}  // namespace

bool vitm_gemm_pp_wanted(int M, int N, int K) {
  // worth it once the 256x256 tiles can occupy a good part of the 256 CUs; small problems keep the 128x128 kernel
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  return K % BK == 0 && M >= 1024 && N >= 256 && tiles >= 128;
}

int vitm_gemm_pp_launch(const GemmArgs& a, int epilogue, hipStream_t st) {
  switch (epilogue) {
    case VITM_EPI_BIAS: return launch<VITM_EPI_BIAS>(a, st);
    case VITM_EPI_BIAS_GELU: return launch<VITM_EPI_BIAS_GELU>(a, st);
    case VITM_EPI_LN: return launch<VITM_EPI_LN>(a, st);
    case VITM_EPI_LN_GELU: return launch<VITM_EPI_LN_GELU>(a, st);
    case VITM_EPI_RESIDUAL: return launch<VITM_EPI_RESIDUAL>(a, st);
    case VITM_EPI_DGELU: return launch<VITM_EPI_DGELU>(a, st);
    case VITM_EPI_BIAS_GELU2: return launch<VITM_EPI_BIAS_GELU2>(a, st);
    default: break;
  }
  vitm_set_error("vitm_gemm: unknown epilogue %d", epilogue);
  return VITM_EINVAL;
}
