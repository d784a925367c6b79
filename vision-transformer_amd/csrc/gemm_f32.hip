// fp32 GEMM family of the ViT forward on gfx950:  out = epilogue(A[M,K] . Wp[N,K]^T), all operands fp32.
//
// The full-precision twin of gemm_bf16.hip, used when the caller's tensors are fp32: fp32 operands, the
// fp32 matrix instruction (v_mfma_f32_16x16x4_f32) and an erf-accurate GELU.  LayerNorm is not folded
// here: the fp32 path normalises explicitly (vitm_layernorm) and then calls the BIAS epilogues.
//
// Tile 128x128x32, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 = 4x4 MFMA tiles.  An LDS row
// is 32 floats = 128 B, so the tile loader (LDS-DMA, XOR-swizzled 16-byte chunks, double buffered) and the
// fp32 staging of the epilogue have the same byte geometry as the bf16 kernel: a lane's 16-byte fragment
// read holds k = 4*(lane>>4) .. +3 of a 16-wide K group and feeds four successive MFMAs.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 4;       // 16 KiB
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int CS_STRIDE = BN + 4;
constexpr int SMEM_BYTES = BM * CS_STRIDE * 4;
static_assert(SMEM_BYTES >= 2 * BUF_BYTES, "epilogue staging must cover the double buffer");

struct Args {
  const float* A;
  int64_t lda;
  const float* W;
  int64_t ldw;
  float* out;
  int64_t ldo;
  const float* c;
  const float* g;
  const float* resid;
  int64_t ldr;
  int M, N, K, tilesN, row_group;
  const void* img;
  int C, H, Wd, ph, pw, gw, npatch, kreal, vec;
};

enum { A_DIRECT = 0, A_IM2COL_F32 = 1, A_IM2COL_BF16 = 2 };

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(l), 16, 0, 0);
}

template <int EPI, int AMODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f32_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = (blockIdx.x / p.tilesN) * BM;
  const int n0 = (blockIdx.x % p.tilesN) * BN;

  // DMA piece i of this wave covers LDS rows (wave*4+i)*8 .. +7; lane -> row lane>>3, physical chunk lane&7
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const float* wsrc[4];
  const float* asrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    int wn = n0 + row;
    wn = wn < p.N ? wn : p.N - 1;
    wsrc[i] = p.W + (int64_t)wn * p.ldw + lchunk * 4;
    if (AMODE == A_DIRECT) {
      int am = m0 + row;
      am = am < p.M ? am : p.M - 1;
      asrc[i] = p.A + (int64_t)am * p.lda + lchunk * 4;
    }
  }
  // im2col: thread owns 4 chunks (of 4 consecutive k) per K-step: chunk id = tid + 256*i
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
  float ireg[4][4];

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
        const int k = k0 + ilch[i] * 4;
        if (p.vec && k + 4 <= p.kreal) {
          const int c = k / pp, kr = k - c * pp;
          const int ky = kr / p.pw, kx = kr - ky * p.pw;
          const int64_t off = ibase[i] + ((int64_t)c * p.H + ky) * p.Wd + kx;
          if (AMODE == A_IM2COL_F32) {
            const float4 v = *(const float4*)((const float*)p.img + off);
            ireg[i][0] = v.x; ireg[i][1] = v.y; ireg[i][2] = v.z; ireg[i][3] = v.w;
          } else {
            const u16x4_t v = *(const u16x4_t*)((const bf16_t*)p.img + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) ireg[i][e] = bf2f(v[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
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
      for (int i = 0; i < 4; ++i) *(float4*)(base + ilds[i]) = make_float4(ireg[i][0], ireg[i][1], ireg[i][2], ireg[i][3]);
    }
  };

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
      f32x4_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const f32x4_t*)(base + offA[kk] + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *(const f32x4_t*)(base + offB[kk] + j * 2048);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    if (more) stage_commit(cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS -> row-contiguous fix-up and 16-byte stores -----------------
  float* Cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr * 64 + i * 16 + (lane >> 4) * 4 + r) * CS_STRIDE + wc * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
  __syncthreads();

  const int cch = tid & 31, rp = tid >> 5;   // 32 column chunks of 4 floats, 8 rows per pass
  const int n = n0 + cch * 4;
  if (n >= p.N) return;
  const float4 cv = *(const float4*)(p.c + n);
  float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (EPI == VITM_EPI_RESIDUAL) gv = *(const float4*)(p.g + n);
#pragma unroll 4
  for (int pass = 0; pass < BM / 8; ++pass) {
    const int row = pass * 8 + rp;
    const int m = m0 + row;
    if (m >= p.M) break;
    float4 v = *(const float4*)(Cs + row * CS_STRIDE + cch * 4);
    if (EPI == VITM_EPI_RESIDUAL) {
      const float4 r = *(const float4*)(p.resid + (int64_t)m * p.ldr + n);
      v.x = r.x + fmaf(gv.x, v.x, cv.x);
      v.y = r.y + fmaf(gv.y, v.y, cv.y);
      v.z = r.z + fmaf(gv.z, v.z, cv.z);
      v.w = r.w + fmaf(gv.w, v.w, cv.w);
    } else {
      v.x += cv.x; v.y += cv.y; v.z += cv.z; v.w += cv.w;
      if (EPI == VITM_EPI_BIAS_GELU) {
        v.x = gelu_erf_accurate(v.x);
        v.y = gelu_erf_accurate(v.y);
        v.z = gelu_erf_accurate(v.z);
        v.w = gelu_erf_accurate(v.w);
      }
    }
    const int64_t om = p.row_group > 0 ? (int64_t)m + m / p.row_group + 1 : (int64_t)m;
    *(float4*)(p.out + om * p.ldo + n) = v;
  }
}

template <int EPI, int AMODE>
int launch(const Args& a, hipStream_t st) {
  auto kfn = gemm_f32_kernel<EPI, AMODE>;
  VITM_LDS_OPTIN(kfn, SMEM_BYTES, "vitm_gemm_f32");
  const int tilesM = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(tilesM * a.tilesN)), dim3(NTHREADS), SMEM_BYTES, st, a);
  VITM_CHECK_LAUNCH("vitm_gemm_f32");
  return VITM_OK;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int vitm_gemm_f32(const float* A, int64_t lda, const float* Wp, int64_t ldw, float* out, int64_t ldo,
                             int M, int N, int K, int epilogue, const float* c, const float* g,
                             const float* resid, int64_t ldr, int row_group, vitm_stream_t stream) {
  VITM_REQUIRE(A && Wp && out && c, "vitm_gemm_f32: null A/Wp/out/c pointer");
  VITM_REQUIRE(M > 0 && N > 0 && K > 0, "vitm_gemm_f32: M, N, K must be positive (got %d, %d, %d)", M, N, K);
  VITM_REQUIRE(K % BK == 0, "vitm_gemm_f32: K=%d must be a multiple of %d", K, BK);
  VITM_REQUIRE(N % 4 == 0, "vitm_gemm_f32: N=%d must be a multiple of 4", N);
  VITM_REQUIRE(lda >= K && ldw >= K && ldo >= N, "vitm_gemm_f32: leading dimension smaller than the row");
  VITM_REQUIRE(lda % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0, "vitm_gemm_f32: leading dimensions must be multiples of 4");
  VITM_REQUIRE(aligned16(A) && aligned16(Wp) && aligned16(out) && aligned16(c), "vitm_gemm_f32: pointers must be 16-byte aligned");
  VITM_REQUIRE(epilogue != VITM_EPI_RESIDUAL || (resid && g && aligned16(resid) && aligned16(g) && ldr >= N && ldr % 4 == 0),
               "vitm_gemm_f32: RESIDUAL epilogue needs resid (ldr >= N, multiple of 4) and g");
  VITM_REQUIRE(row_group >= 0, "vitm_gemm_f32: row_group must be >= 0");
  Args a{};
  a.A = A; a.lda = lda; a.W = Wp; a.ldw = ldw; a.out = out; a.ldo = ldo; a.c = c; a.g = g;
  a.resid = resid; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.tilesN = (N + BN - 1) / BN; a.row_group = row_group;
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case VITM_EPI_BIAS: return launch<VITM_EPI_BIAS, A_DIRECT>(a, st);
    case VITM_EPI_BIAS_GELU: return launch<VITM_EPI_BIAS_GELU, A_DIRECT>(a, st);
    case VITM_EPI_RESIDUAL: return launch<VITM_EPI_RESIDUAL, A_DIRECT>(a, st);
    default: break;
  }
  vitm_set_error("vitm_gemm_f32: epilogue %d is not available in fp32 (BIAS, BIAS_GELU, RESIDUAL; LayerNorm is "
                 "applied explicitly on the fp32 path)", epilogue);
  return VITM_EUNSUPPORTED;
}

extern "C" int vitm_patch_embed_f32(const void* img, int img_dtype, int B, int C, int H, int W, int ph, int pw,
                                    const float* Wp, int64_t ldw, int kpad, const float* c, float* tokens,
                                    int64_t ldo, int D, int cls_slot, vitm_stream_t stream) {
  VITM_REQUIRE(img && Wp && c && tokens, "vitm_patch_embed_f32: null pointer");
  VITM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && D > 0, "vitm_patch_embed_f32: bad shape");
  VITM_REQUIRE(img_dtype == VITM_BF16 || img_dtype == VITM_F32, "vitm_patch_embed_f32: bad img_dtype %d", img_dtype);
  const int gh = H / ph, gw = W / pw;
  VITM_REQUIRE(gh > 0 && gw > 0, "vitm_patch_embed_f32: patch larger than the image");
  const int kreal = C * ph * pw;
  VITM_REQUIRE(kpad % BK == 0 && kpad >= kreal, "vitm_patch_embed_f32: kpad=%d must be a multiple of %d and >= C*ph*pw=%d", kpad, BK, kreal);
  VITM_REQUIRE(D % 4 == 0 && ldo >= D && ldo % 4 == 0 && ldw >= kpad && ldw % 4 == 0, "vitm_patch_embed_f32: D/ldo/ldw must be multiples of 4");
  VITM_REQUIRE(aligned16(Wp) && aligned16(tokens) && aligned16(c), "vitm_patch_embed_f32: pointers must be 16-byte aligned");
  const bool vec_ok = (pw % 4 == 0) && (W % 4 == 0) && (((uintptr_t)img & 15) == 0);
  Args a{};
  a.W = Wp; a.ldw = ldw; a.out = tokens; a.ldo = ldo; a.c = c;
  a.M = B * gh * gw; a.N = D; a.K = kpad; a.tilesN = (D + BN - 1) / BN;
  a.row_group = cls_slot ? gh * gw : 0;
  a.img = img; a.C = C; a.H = H; a.Wd = W; a.ph = ph; a.pw = pw; a.gw = gw; a.npatch = gh * gw; a.kreal = kreal;
  a.vec = vec_ok ? 1 : 0;
  if (img_dtype == VITM_F32) return launch<VITM_EPI_BIAS, A_IM2COL_F32>(a, (hipStream_t)stream);
  return launch<VITM_EPI_BIAS, A_IM2COL_BF16>(a, (hipStream_t)stream);
}
