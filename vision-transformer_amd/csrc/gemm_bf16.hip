// bf16 MFMA GEMM family for the ViT forward on gfx950:  out = epilogue(A[M,K] . Wp[N,K]^T)
//
// One kernel template covers every dense contraction of the forward:
//   patch embedding   A gathered from the NCHW image (im2col fused into the tile loader)
//   qkv / fc1         LayerNorm folded in: row statistics + pre-scaled weights, fixed up in the epilogue
//   proj / fc2        bias + LayerScale + residual add in the epilogue
//   classifier head   LayerNorm-folded, fp32 logits out
//
// Tile 128x128x64, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// A and W tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier
// per K-step.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row & 7) -- applied
// to the per-lane SOURCE address of the DMA (its LDS destination is lane-linear) and again on the
// fragment reads -- which makes the ds_read_b128 fragment reads bank-conflict free.
// The accumulators leave through LDS (fp32) so the epilogue runs row-contiguous: 16-byte loads of the
// residual, 16-byte stores of the output, per-thread column constants.
#include "gemm_common.h"

namespace {
using namespace vitm_gemm_detail;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;       // one operand tile: 16 KiB
constexpr int BUF_BYTES = 2 * TILE_BYTES;     // A + W
constexpr int CS_STRIDE = BN + 4;             // fp32 staging row stride (floats): +4 keeps ds_write_b32 conflict free
constexpr int SMEM_BYTES = BM * CS_STRIDE * 4;  // 67584 >= 2 * BUF_BYTES
static_assert(SMEM_BYTES >= 2 * BUF_BYTES, "epilogue staging must cover the double buffer");

enum { A_DIRECT = 0, A_IM2COL_F32 = 1, A_IM2COL_BF16 = 2 };

template <int EPI, int AMODE, int OUT_F32>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wgid / p.tilesN) * BM;
  const int n0 = (wgid % p.tilesN) * BN;

  // ---- loader set-up --------------------------------------------------------------------------
  // DMA piece i of this wave covers LDS rows (wave*4+i)*8 .. +7; lane -> row lane>>3, physical chunk lane&7.
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;  // logical 16-B chunk fetched into physical chunk lane&7
  const bf16_t* wsrc[4];
  const bf16_t* asrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    int wn = n0 + row;
    wn = wn < p.N ? wn : p.N - 1;
    wsrc[i] = p.W + (int64_t)wn * p.ldw + lchunk * 8;
    if (AMODE == A_DIRECT) {
      int am = m0 + row;
      am = am < p.M ? am : p.M - 1;
      asrc[i] = p.A + (int64_t)am * p.lda + lchunk * 8;
    }
  }
  // im2col: thread owns 4 chunks per K-step: chunk id = tid + 256*i -> row cid>>3, physical chunk cid&7
  int64_t ibase[4];
  int ilds[4], ilch[4];
  if (AMODE != A_DIRECT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cid = tid + NTHREADS * i;
      const int row = cid >> 3, pch = cid & 7;
      int am = m0 + row;
      am = am < p.M ? am : p.M - 1;
      const int b = am / p.npatch, rem = am - b * p.npatch;
      const int py = rem / p.gw, px = rem - py * p.gw;
      ibase[i] = ((int64_t)b * p.C * p.H + (int64_t)py * p.ph) * p.Wd + (int64_t)px * p.pw;
      ilds[i] = row * 128 + pch * 16;
      ilch[i] = pch ^ (row & 7);
    }
  }
  float ireg[4][8];  // im2col staging registers (fp32 values of 8 consecutive k)

  auto stage_issue = [&](int buf, int kt) {
    char* base = smem + buf * BUF_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(wsrc[i] + k0, base + TILE_BYTES + (wave * 4 + i) * 1024);
    if (AMODE == A_DIRECT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(asrc[i] + k0, base + (wave * 4 + i) * 1024);
    } else {
      const int pp = p.ph * p.pw;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + ilch[i] * 8;
        if (p.vec && k + 8 <= p.kreal) {
          const int c = k / pp, kr = k - c * pp;
          const int ky = kr / p.pw, kx = kr - ky * p.pw;
          const int64_t off = ibase[i] + ((int64_t)c * p.H + ky) * p.Wd + kx;
          if (AMODE == A_IM2COL_F32) {
            const float4* s4 = (const float4*)((const float*)p.img + off);
            const float4 lo = s4[0], hi = s4[1];
            ireg[i][0] = lo.x; ireg[i][1] = lo.y; ireg[i][2] = lo.z; ireg[i][3] = lo.w;
            ireg[i][4] = hi.x; ireg[i][5] = hi.y; ireg[i][6] = hi.z; ireg[i][7] = hi.w;
          } else {
            const u16x8_t v = *(const u16x8_t*)((const bf16_t*)p.img + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) ireg[i][e] = bf2f(v[e]);
          }
        } else {  // general patch width or the zero-padded K tail: element-wise gather
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ke = k + e;
            float v = 0.f;
            if (ke < p.kreal) {
              const int c = ke / pp, kr = ke - c * pp;
              const int ky = kr / p.pw, kx = kr - ky * p.pw;
              const int64_t off = ibase[i] + ((int64_t)c * p.H + ky) * p.Wd + kx;
              v = (AMODE == A_IM2COL_F32) ? ((const float*)p.img)[off] : bf2f(((const bf16_t*)p.img)[off]);
            }
            ireg[i][e] = v;
          }
        }
      }
    }
  };
  auto stage_commit = [&](int buf) {
    if (AMODE != A_DIRECT) {
      char* base = smem + buf * BUF_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 v;
        v.x = pack_bf2(ireg[i][0], ireg[i][1]);
        v.y = pack_bf2(ireg[i][2], ireg[i][3]);
        v.z = pack_bf2(ireg[i][4], ireg[i][5]);
        v.w = pack_bf2(ireg[i][6], ireg[i][7]);
        *(uint4*)(base + ilds[i]) = v;
      }
    }
  };

  // ---- fragment read offsets ------------------------------------------------------------------
  int offA[2], offB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int pch = ((kk * 4 + (lane >> 4)) ^ (lane & 7)) * 16;
    offA[kk] = (wr * 64 + (lane & 15)) * 128 + pch;
    offB[kk] = TILE_BYTES + (wc * 64 + (lane & 15)) * 128 + pch;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_issue(0, 0);
  stage_commit(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) stage_issue(cur ^ 1, kt + 1);
    const char* base = smem + cur * BUF_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)(base + offA[kk] + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8_t*)(base + offB[kk] + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stage_commit(cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> row-contiguous fix-up and store ------------------
  float* Cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr * 64 + i * 16 + (lane >> 4) * 4 + r) * CS_STRIDE + wc * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
  __syncthreads();

  const int cch = tid & 15, rp = tid >> 4;
  const int n = n0 + cch * 8;
  if (n >= p.N) return;
  EpiCols<EPI> cols;
  cols.load(p, n);
#pragma unroll
  for (int pass = 0; pass < BM / 16; ++pass) {
    const int row = pass * 16 + rp;
    const int m = m0 + row;
    if (m >= p.M) break;
    float v[8];
    const float4 a0 = *(const float4*)(Cs + row * CS_STRIDE + cch * 8);
    const float4 a1 = *(const float4*)(Cs + row * CS_STRIDE + cch * 8 + 4);
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w;
    v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    cols.template finish<OUT_F32>(p, m, n, v);
  }
}

template <int EPI, int AMODE, int OUT_F32>
int launch(const GemmArgs& a, hipStream_t st) {
  auto kfn = gemm_kernel<EPI, AMODE, OUT_F32>;
  // > 64 KiB of dynamic LDS needs the opt-in; it is per function and device, cheap and idempotent.
  VITM_LDS_OPTIN(kfn, SMEM_BYTES, "vitm_gemm");
  const int tilesM = (a.M + BM - 1) / BM;
  dim3 grid((unsigned)(tilesM * a.tilesN));
  hipLaunchKernelGGL(kfn, grid, dim3(NTHREADS), SMEM_BYTES, st, a);
  VITM_CHECK_LAUNCH("vitm_gemm");
  return VITM_OK;
}

template <int AMODE>
int dispatch(const GemmArgs& a, int epi, int out_f32, hipStream_t st) {
  if (out_f32) {
    switch (epi) {
      case VITM_EPI_BIAS: return launch<VITM_EPI_BIAS, AMODE, 1>(a, st);
      case VITM_EPI_LN: return launch<VITM_EPI_LN, AMODE, 1>(a, st);
      default: break;
    }
    vitm_set_error("vitm_gemm: fp32 output supports only the BIAS and LN epilogues");
    return VITM_EUNSUPPORTED;
  }
  switch (epi) {
    case VITM_EPI_BIAS: return launch<VITM_EPI_BIAS, AMODE, 0>(a, st);
    case VITM_EPI_BIAS_GELU: return launch<VITM_EPI_BIAS_GELU, AMODE, 0>(a, st);
    case VITM_EPI_LN: return launch<VITM_EPI_LN, AMODE, 0>(a, st);
    case VITM_EPI_LN_GELU: return launch<VITM_EPI_LN_GELU, AMODE, 0>(a, st);
    case VITM_EPI_RESIDUAL: return launch<VITM_EPI_RESIDUAL, AMODE, 0>(a, st);
    case VITM_EPI_DGELU: return launch<VITM_EPI_DGELU, AMODE, 0>(a, st);
    case VITM_EPI_BIAS_GELU2: return launch<VITM_EPI_BIAS_GELU2, AMODE, 0>(a, st);
    default: break;
  }
  vitm_set_error("vitm_gemm: unknown epilogue %d", epi);
  return VITM_EINVAL;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
constexpr int BN_BIG = 256;   // column tile of the big-tile kernel (one statistics slot per tile)

}  // namespace

extern "C" int vitm_gemm_k_multiple(void) { return BK; }
extern "C" int vitm_gemm_uses_big_tile(int M, int N, int K) { return vitm_gemm_pp_wanted(M, N, K) ? 1 : 0; }

static int gemm_impl(const void* A, int64_t lda, const void* Wp, int64_t ldw, void* out, int64_t ldo,
                     int out_dtype, int M, int N, int K, int epilogue, const float* stats,
                     const float* s, const float* c, const float* g, const void* resid,
                     int64_t ldr, int row_group, float* stats_out, int stats_partial, int ln_dim, float ln_eps,
                     vitm_stream_t stream) {
  VITM_REQUIRE(A && Wp && out && c, "vitm_gemm: null A/Wp/out/c pointer");
  VITM_REQUIRE(M > 0 && N > 0 && K > 0, "vitm_gemm: M, N, K must be positive (got %d, %d, %d)", M, N, K);
  VITM_REQUIRE(K % BK == 0, "vitm_gemm: K=%d must be a multiple of %d (pad the packed weight and A)", K, BK);
  VITM_REQUIRE(N % 8 == 0, "vitm_gemm: N=%d must be a multiple of 8", N);
  VITM_REQUIRE(lda >= K && ldw >= K && ldo >= N, "vitm_gemm: leading dimension smaller than the row");
  VITM_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0, "vitm_gemm: leading dimensions must be multiples of 8");
  VITM_REQUIRE(aligned16(A) && aligned16(Wp) && aligned16(out) && aligned16(c), "vitm_gemm: pointers must be 16-byte aligned");
  VITM_REQUIRE(out_dtype == VITM_BF16 || out_dtype == VITM_F32, "vitm_gemm: bad out_dtype %d", out_dtype);
  const bool ln = epilogue == VITM_EPI_LN || epilogue == VITM_EPI_LN_GELU;
  VITM_REQUIRE(!ln || (stats && s && aligned16(s) && ((uintptr_t)stats & 7) == 0),
               "vitm_gemm: LN epilogue needs stats and s (16-byte aligned)");
  VITM_REQUIRE(epilogue != VITM_EPI_RESIDUAL || (resid && g && aligned16(resid) && aligned16(g) && ldr >= N && ldr % 8 == 0),
               "vitm_gemm: RESIDUAL epilogue needs resid (ldr >= N, multiple of 8) and g");
  VITM_REQUIRE(epilogue != VITM_EPI_DGELU || (resid && aligned16(resid) && ldr >= N && ldr % 8 == 0 && out_dtype == VITM_BF16),
               "vitm_gemm: DGELU epilogue needs resid (the GELU input, ldr >= N, multiple of 8) and bf16 output");
  VITM_REQUIRE(epilogue != VITM_EPI_BIAS_GELU2 ||
                   (resid && resid != out && aligned16(resid) && ldr >= N && ldr % 8 == 0 && out_dtype == VITM_BF16 && row_group == 0),
               "vitm_gemm: BIAS_GELU2 epilogue needs resid (the second output, ldr >= N, multiple of 8), bf16 output, row_group 0");
  VITM_REQUIRE(row_group >= 0, "vitm_gemm: row_group must be >= 0");
  // the big-tile kernel addresses a tile's operand rows by 32-bit byte offsets from the tile's first row
  const bool big = row_group == 0 && out_dtype == VITM_BF16 && vitm_gemm_pp_wanted(M, N, K) && lda < (1 << 22) && ldw < (1 << 22);
  if (stats_out != nullptr || stats_partial) {
    VITM_REQUIRE(big, "vitm_gemm_stats: carried LayerNorm statistics need the 256x256-tile route (vitm_gemm_uses_big_tile)");
    VITM_REQUIRE(stats_out == nullptr || (epilogue == VITM_EPI_RESIDUAL && N <= VITM_STATS_PARTS * BN_BIG && aligned16(stats_out)),
                 "vitm_gemm_stats: stats_out needs the RESIDUAL epilogue, N <= %d and a 16-byte aligned buffer", VITM_STATS_PARTS * BN_BIG);
    VITM_REQUIRE(!stats_partial || (ln && aligned16(stats) && ln_dim > 0 && ln_eps > 0.f),
                 "vitm_gemm_stats: partial statistics need an LN epilogue, a 16-byte aligned stats buffer, ln_dim and ln_eps");
  }
  GemmArgs a{};
  a.stats_out = stats_out; a.stats_partial = stats_partial ? 1 : 0;
  a.ln_inv_dim = ln_dim > 0 ? 1.0f / (float)ln_dim : 0.f; a.ln_eps = ln_eps;
  a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)Wp; a.ldw = ldw;
  a.out = out; a.ldo = ldo; a.stats = stats; a.s = s; a.c = c; a.g = g;
  a.resid = (const bf16_t*)resid; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.tilesN = (N + BN - 1) / BN; a.row_group = row_group;
  if (big) return vitm_gemm_pp_launch(a, epilogue, (hipStream_t)stream);
  return dispatch<A_DIRECT>(a, epilogue, out_dtype == VITM_F32, (hipStream_t)stream);
}

extern "C" int vitm_gemm(const void* A, int64_t lda, const void* Wp, int64_t ldw, void* out, int64_t ldo,
                         int out_dtype, int M, int N, int K, int epilogue, const float* stats,
                         const float* s, const float* c, const float* g, const void* resid,
                         int64_t ldr, int row_group, vitm_stream_t stream) {
  return gemm_impl(A, lda, Wp, ldw, out, ldo, out_dtype, M, N, K, epilogue, stats, s, c, g, resid, ldr, row_group,
                   nullptr, 0, 0, 0.f, stream);
}

extern "C" int vitm_gemm_stats(const void* A, int64_t lda, const void* Wp, int64_t ldw, void* out, int64_t ldo,
                               int M, int N, int K, int epilogue, const float* stats, const float* s,
                               const float* c, const float* g, const void* resid, int64_t ldr,
                               float* stats_out, int stats_partial, int ln_dim, float ln_eps, vitm_stream_t stream) {
  return gemm_impl(A, lda, Wp, ldw, out, ldo, VITM_BF16, M, N, K, epilogue, stats, s, c, g, resid, ldr, 0,
                   stats_out, stats_partial, ln_dim, ln_eps, stream);
}

extern "C" int vitm_patch_embed(const void* img, int img_dtype, int B, int C, int H, int W, int ph, int pw,
                                const void* Wp, int64_t ldw, int kpad, const float* c, void* tokens,
                                int64_t ldo, int D, int cls_slot, vitm_stream_t stream) {
  VITM_REQUIRE(img && Wp && c && tokens, "vitm_patch_embed: null pointer");
  VITM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && D > 0, "vitm_patch_embed: bad shape");
  VITM_REQUIRE(img_dtype == VITM_BF16 || img_dtype == VITM_F32, "vitm_patch_embed: bad img_dtype %d", img_dtype);
  const int gh = H / ph, gw = W / pw;
  VITM_REQUIRE(gh > 0 && gw > 0, "vitm_patch_embed: patch larger than the image");
  const int kreal = C * ph * pw;
  VITM_REQUIRE(kpad % BK == 0 && kpad >= kreal, "vitm_patch_embed: kpad=%d must be a multiple of %d and >= C*ph*pw=%d", kpad, BK, kreal);
  VITM_REQUIRE(D % 8 == 0 && ldo >= D && ldo % 8 == 0 && ldw >= kpad && ldw % 8 == 0, "vitm_patch_embed: D/ldo/ldw must be multiples of 8");
  VITM_REQUIRE(aligned16(Wp) && aligned16(tokens) && aligned16(c), "vitm_patch_embed: pointers must be 16-byte aligned");
  // the 8-wide vector gather needs 8 consecutive kx to be 16/32-byte aligned in the image
  const bool vec_ok = (pw % 8 == 0) && (W % 8 == 0) && (((uintptr_t)img & 31) == 0);
  GemmArgs a{};
  a.W = (const bf16_t*)Wp; a.ldw = ldw; a.out = tokens; a.ldo = ldo; a.c = c;
  a.M = B * gh * gw; a.N = D; a.K = kpad; a.tilesN = (D + BN - 1) / BN;
  a.row_group = cls_slot ? gh * gw : 0;
  a.img = img; a.C = C; a.H = H; a.Wd = W; a.ph = ph; a.gw = gw; a.npatch = gh * gw; a.kreal = kreal;
  a.pw = pw;
  a.vec = vec_ok ? 1 : 0;  // otherwise the loader gathers element by element
  // large batch, bf16 image, power-of-two patch width: the 256x256 persistent kernel with the gather in its DMA
  // addresses (no patch matrix in memory).  Its rows include the class-token slots, so the row map is the identity.
  const int rows = B * (gh * gw + (cls_slot ? 1 : 0));
  if (img_dtype == VITM_BF16 && vec_ok && kreal == kpad && (pw & (pw - 1)) == 0 && pw <= BK &&
      vitm_gemm_pp_wanted(rows, D, kpad) && (int64_t)C * H * W < (1ll << 31)) {   // per-image offsets are 32-bit
    GemmArgs b = a;
    b.M = rows; b.row_group = 0; b.cls = cls_slot ? 1 : 0;
    b.pw_shift = __builtin_ctz((unsigned)pw);
    return vitm_gemm_pp_launch(b, VITM_EPI_BIAS, (hipStream_t)stream);
  }
  if (img_dtype == VITM_F32) return launch<VITM_EPI_BIAS, A_IM2COL_F32, 0>(a, (hipStream_t)stream);
  return launch<VITM_EPI_BIAS, A_IM2COL_BF16, 0>(a, (hipStream_t)stream);
}
