// Big-tile bf16 MFMA GEMM for gfx950 with a two-group ping-pong schedule:  out = epilogue(A[M,K] . Wp[N,K]^T)
//
// Tile 256x256, K-tile 64, 512 threads = 8 waves, one workgroup per CU.  Waves 0-3 (group 0) own output rows
// 0-127, waves 4-7 (group 1) rows 128-255; wave (g, c) computes the 128x64 block at columns 64c: 8x4 MFMA
// 16x16x32 tiles = 128 accumulator registers.  A K-tile is consumed in two 32-deep sub-steps; per sub-step a
// wave has a LOAD segment (12 ds_read_b128 operand fragments -> registers) and a COMPUTE segment (32
// back-to-back MFMAs on those registers).  Each SIMD hosts one wave of each group and the groups run one
// segment apart, so the matrix pipe of every SIMD always has one wave feeding it while the other does its
// memory work:
//
//     segment (mod 4):    0          1          2          3          0
//     group 0:        | L(t,0)   | C(t,0)   | L(t,1)   | C(t,1)   | L(t+1,0) ...
//     group 1:        | C(t-1,1) | L(t,0)   | C(t,0)   | L(t,1)   | C(t,1)   ...
//
// Segments are separated by raw s_barrier; group 1 idles through segment 0 and group 0 through the last one.
//
// LDS holds two K-tile slots (A part 256 rows x 128 B, W part 256 rows x 128 B = 64 KiB per slot).  Operands
// come global -> LDS by LDS-DMA in 1-KiB pieces of 8 rows x 128 B: whole cache lines, half as many address-unit
// cycles as 16 x 64 B pieces.  Slot (t+1)&1 is free once group 1 has read L(t-1,1), i.e. after segment 3 of
// K-tile t-1; every wave then issues its A pieces of K-tile t+1 in segment 0 and its W pieces in segment 1 (inside
// whatever role it has there), and retires them with s_waitcnt vmcnt(0) at the end of segment 3, three to four
// segments (1500-2000 MFMA cycles) later; the barrier closing segment 3 publishes them just before group 0
// starts L(t+1,0).  No wait for DMA sits on the critical path in steady state.
//
// 128-byte LDS rows: the 16-byte chunk index is XOR-ed with (row & 7), on the DMA's per-lane SOURCE address (its
// LDS destination is lane-linear) and on the fragment reads; every ds_read_b128 lane group then hits 16 distinct
// 16-byte bank slots.
#include "gemm_common.h"

#ifndef PP_ABLATE
#define PP_ABLATE 0   // timing-only diagnostic builds (wrong results): 1 no DMA in the loop, 2 no fragment reads, 4 no output stores
#endif

#ifdef PP_STAMP  // diagnostic build: wave 0 of each workgroup records s_memtime at phase boundaries into p.resid
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    unsigned long long t__;                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                    \
    if (tid == 0) ((unsigned long long*)p.resid)[(size_t)blockIdx.x * 8 + (k)] = t__;              \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

namespace {
using namespace vitm_gemm_detail;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NT = 512;
constexpr int PART = 256 * 128;             // bytes of one operand tile: 256 rows x 128 B
constexpr int NA = 3, NW = 2;               // LDS ring depth: A tiles (HBM / Infinity Cache latency), W tiles (L2)
constexpr int WBASE = NA * PART;            // W tiles follow the A tiles
constexpr int CS = BN + 4;                  // fp32 staging row stride (floats)
constexpr int SMEM = (NA + NW) * PART;      // 163840 B = all of a CU's LDS
static_assert(SMEM >= 256 * 528 && SMEM >= 128 * CS * 4 && SMEM <= 160 * 1024, "epilogue staging must fit");

__device__ __forceinline__ void seg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_dma_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int EPI, int OUT_F32>
__global__ __launch_bounds__(NT, 2) void gemm_pp_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;

  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wgid / p.tilesN) * BM;
  const int n0 = (wgid % p.tilesN) * BN;

  // ---- DMA set-up: this wave moves rows 32w..32w+31 of the A part and of the W part of every K-tile, as 4 + 4
  // pieces of 8 rows x 128 B; lane -> row lane>>3, physical chunk lane&7, logical chunk = physical ^ (row & 7)
  const int lr = lane >> 3;
  const int lc = (lane & 7) ^ lr;
  const bf16_t* asrc[4];
  const bf16_t* wsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (4 * wave + q) * 8 + lr;
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    asrc[q] = p.A + (int64_t)am * p.lda + lc * 8;
    int wn = n0 + row;
    wn = wn < p.N ? wn : p.N - 1;
    wsrc[q] = p.W + (int64_t)wn * p.ldw + lc * 8;
  }
  auto issueA = [&](int t) {
    char* sb = smem + (t % NA) * PART + 4 * wave * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(asrc[q] + t * BK, sb + q * 1024);
  };
  auto issueW = [&](int t) {
    char* sb = smem + WBASE + (t & 1) * PART + 4 * wave * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(wsrc[q] + t * BK, sb + q * 1024);
  };

  // ---- fragment addressing (slot-relative): lane reads row (lane&15) of a 16-row tile, logical chunk 4kk + lane>>4
  int offA[2], offB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int pch = ((kk * 4 + (lane >> 4)) ^ (lane & 7)) * 16;
    offA[kk] = (grp * 128 + (lane & 15)) * 128 + pch;
    offB[kk] = WBASE + (wc * 64 + (lane & 15)) * 128 + pch;
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  bf16x8_t a[8], b[4];
  auto load_frags = [&](int t, int kk) {
    if ((PP_ABLATE & 2) && (t | kk)) return;
    const char* sa = smem + (t % NA) * PART;
    const char* sw = smem + (t & 1) * PART;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8_t*)(sw + offB[kk] + j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8_t*)(sa + offA[kk] + i * 2048);
  };
  // the weight tile is the MFMA A operand: a lane ends up with 4 consecutive output COLUMNS of one output row
  auto compute = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = p.K / BK;
  const bool dma = (PP_ABLATE & 1) == 0;

  STAMP(0);
  // ---- prologue.  Issue order is what makes the counted waits work (vmcnt retires in order): A(0) W(0) A(1),
  // then per K-tile t: W(t+1) in the wave's L(t,0), A(t+2) in its L(t,1); "vmcnt(4)" at the end of segment 3 leaves the 4
  // pieces of A(t+2) in flight and guarantees A(t+1), W(t+1) and everything older have landed.
  issueA(0);
  issueW(0);
  if (nk > 1) {
    issueA(1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    wait_dma_all();
  }
  seg_barrier();
  STAMP(1);

  // DMA is issued only from LOAD segments: a piece blocks its wave for 60-180 cycles while the address unit takes
  // it, which inside a COMPUTE segment would stall the matrix pipe directly.
  if (grp == 0) {
    for (int t = 0; t < nk; ++t) {
      load_frags(t, 0);
      if (dma && t + 1 < nk && !(PP_ABLATE & 8)) issueW(t + 1);    // segment 0: L(t,0)
      wait_lds();
      seg_barrier();
      compute();                                                   // segment 1: C(t,0)
      seg_barrier();
      load_frags(t, 1);
      if (dma && t + 2 < nk && !(PP_ABLATE & 16)) issueA(t + 2);   // segment 2: L(t,1)
      wait_lds();
      seg_barrier();
      compute();                                                   // segment 3: C(t,1)
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else wait_dma_all();
      seg_barrier();
    }
    seg_barrier();                     // group 1's last COMPUTE
  } else {
    seg_barrier();                     // group 0's first LOAD
    for (int t = 0; t < nk; ++t) {
      load_frags(t, 0);
      if (dma && t + 1 < nk && !(PP_ABLATE & 8)) issueW(t + 1);    // segment 1: L(t,0)
      wait_lds();
      seg_barrier();
      compute();                                                   // segment 2: C(t,0)
      seg_barrier();
      load_frags(t, 1);
      if (dma && t + 2 < nk && !(PP_ABLATE & 16)) issueA(t + 2);   // segment 3: L(t,1)
      wait_lds();
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else wait_dma_all();
      seg_barrier();
      compute();                                                   // segment 0 of K-tile t+1: C(t,1)
      seg_barrier();
    }
  }

  STAMP(2);
  // ---- epilogue.  The MFMAs ran with the weight tile as the A operand, so a lane holds, for each (i, j),
  // 4 CONSECUTIVE output columns of one output row:
  //   acc[i][j][r] = out[m0 + 128 grp + 16 i + (lane & 15)][n0 + 64 wc + 16 j + 4 (lane >> 4) + r]
  const int mrow = grp * 128 + (lane & 15);        // + 16 i
  const int ncol = wc * 64 + 4 * (lane >> 4);      // + 16 j
  if (EPI != VITM_EPI_RESIDUAL && !OUT_F32) {
    // Fast path: bias / LayerNorm fix-up / GELU on the accumulators in place (fp32), one rounding to bf16,
    // then the whole 256x256 bf16 tile crosses LDS once (8-byte writes, 16-byte reads) so that global stores
    // are full 512-byte row segments.  Stores are left in flight: nothing waits for them.
    float4 cq[4], sq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + ncol + 16 * j;
      n = n < p.N ? n : p.N - 4;   // N % 8 == 0: a clamped quad is read but never stored
      cq[j] = *(const float4*)(p.c + n);
      if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) sq[j] = *(const float4*)(p.s + n);
    }
    constexpr int TS = 528;  // bytes per LDS row of the bf16 tile image (512 + 16)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float2 st = make_float2(0.f, 1.f);
      if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) {
        int m = m0 + mrow + 16 * i;
        m = m < p.M ? m : p.M - 1;
        st = *(const float2*)(p.stats + 2 * (int64_t)m);
      }
      f32x2_t v[8];  // the 16 values of row i: pairs (j, half)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2_t c01 = {cq[j].x, cq[j].y}, c23 = {cq[j].z, cq[j].w};
        v[2 * j] = (f32x2_t){acc[i][j][0], acc[i][j][1]};
        v[2 * j + 1] = (f32x2_t){acc[i][j][2], acc[i][j][3]};
        if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) {
          const f32x2_t s01 = {sq[j].x, sq[j].y}, s23 = {sq[j].z, sq[j].w};
          v[2 * j] = (v[2 * j] - s01 * st.x) * st.y + c01;       // rstd * (acc - mean * s) + c
          v[2 * j + 1] = (v[2 * j + 1] - s23 * st.x) * st.y + c23;
        } else {
          v[2 * j] += c01;
          v[2 * j + 1] += c23;
        }
      }
      if (EPI == VITM_EPI_LN_GELU || EPI == VITM_EPI_BIAS_GELU) gelu_erf2_n<8>(v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 w;
        w.x = pack_bf2(v[2 * j].x, v[2 * j].y);
        w.y = pack_bf2(v[2 * j + 1].x, v[2 * j + 1].y);
        *(uint2*)(smem + (mrow + 16 * i) * TS + (ncol + 16 * j) * 2) = w;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    seg_barrier();
    STAMP(3);
    const int ch = tid & 31, rp = tid >> 5;   // 16-byte chunk of the row, row within a 16-row pass
    const int n = n0 + ch * 8;
    if (n < p.N && (PP_ABLATE & 4) == 0) {
      bf16_t* obase = (bf16_t*)p.out + n;
#pragma unroll
      for (int pass = 0; pass < 16; ++pass) {
        const int row = pass * 16 + rp;
        const int m = m0 + row;
        if (m < p.M) {
          const uint4 w = *(const uint4*)(smem + row * TS + ch * 16);
          *(uint4*)(obase + (int64_t)m * p.ldo) = w;
        }
      }
    }
    STAMP(4);
#ifdef PP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
#endif
    return;
  }

  // Staged path (residual add, or fp32 output): 128 rows at a time through LDS in fp32 (16-byte writes), then
  // the row-contiguous fix-up reads the residual with 16-byte loads and rounds once.
  float* Cs = (float*)smem;
  const int cch = tid & 31, rp = tid >> 5;  // thread -> 8 columns cch*8.., row rp of each 16-row pass
  const int n = n0 + cch * 8;
  EpiCols<EPI> cols;
  if (n < p.N) cols.load(p, n);
  // every load the fix-up needs is issued before the first store, so no wait ever covers a store
  u16x8_t rres[16];
  float2 rst[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    int m = m0 + q * 16 + rp;
    m = m < p.M ? m : p.M - 1;
    rres[q] = (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    rst[q] = make_float2(0.f, 1.f);
    if (n < p.N) {
      if (EPI == VITM_EPI_RESIDUAL) rres[q] = *(const u16x8_t*)(p.resid + (int64_t)m * p.ldr + n);
      if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) rst[q] = *(const float2*)(p.stats + 2 * (int64_t)m);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (grp == h) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(f32x4_t*)(Cs + ((lane & 15) + 16 * i) * CS + ncol + 16 * j) = acc[i][j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    seg_barrier();
    if (n < p.N && (PP_ABLATE & 4) == 0) {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 16 + rp;
        const int m = m0 + h * 128 + row;
        if (m < p.M) {
          float v[8];
          const float4 a0 = *(const float4*)(Cs + row * CS + cch * 8);
          const float4 a1 = *(const float4*)(Cs + row * CS + cch * 8 + 4);
          v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w;
          v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
          cols.template finish_pre<OUT_F32>(p, m, n, v, rres[h * 8 + pass], rst[h * 8 + pass]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    seg_barrier();
  }
}

template <int EPI, int OUT_F32>
int launch(const GemmArgs& a, hipStream_t st) {
  auto kfn = gemm_pp_kernel<EPI, OUT_F32>;
  if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
    vitm_set_error("vitm_gemm: cannot reserve %d bytes of LDS", SMEM);
    (void)hipGetLastError();
    return VITM_ELAUNCH;
  }
  GemmArgs b = a;
  b.tilesN = (a.N + BN - 1) / BN;
  const int tilesM = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(tilesM * b.tilesN)), dim3(NT), SMEM, st, b);
  VITM_CHECK_LAUNCH("vitm_gemm (256x256 tile)");
  return VITM_OK;
}

}  // namespace

bool vitm_gemm_pp_wanted(int M, int N, int K) {
  // worth it once the 256x256 tiles can occupy a good part of the 256 CUs; small problems keep the 128x128 kernel
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  return K % BK == 0 && M >= 1024 && N >= 256 && tiles >= 128;
}

int vitm_gemm_pp_launch(const GemmArgs& a, int epilogue, int out_f32, hipStream_t st) {
  if (out_f32) {
    switch (epilogue) {
      case VITM_EPI_BIAS: return launch<VITM_EPI_BIAS, 1>(a, st);
      case VITM_EPI_LN: return launch<VITM_EPI_LN, 1>(a, st);
      default: break;
    }
    vitm_set_error("vitm_gemm: fp32 output supports only the BIAS and LN epilogues");
    return VITM_EUNSUPPORTED;
  }
  switch (epilogue) {
    case VITM_EPI_BIAS: return launch<VITM_EPI_BIAS, 0>(a, st);
    case VITM_EPI_BIAS_GELU: return launch<VITM_EPI_BIAS_GELU, 0>(a, st);
    case VITM_EPI_LN: return launch<VITM_EPI_LN, 0>(a, st);
    case VITM_EPI_LN_GELU: return launch<VITM_EPI_LN_GELU, 0>(a, st);
    case VITM_EPI_RESIDUAL: return launch<VITM_EPI_RESIDUAL, 0>(a, st);
    default: break;
  }
  vitm_set_error("vitm_gemm: unknown epilogue %d", epilogue);
  return VITM_EINVAL;
}
