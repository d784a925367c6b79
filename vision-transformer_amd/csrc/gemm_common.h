// Argument block and helpers shared by the GEMM kernels of libvitm_gfx950.
#pragma once
#include "common.h"

namespace vitm_gemm_detail {

struct GemmArgs {
  const bf16_t* A;
  int64_t lda;
  const bf16_t* W;
  int64_t ldw;
  void* out;
  int64_t ldo;
  const float* stats;
  const float* s;
  const float* c;
  const float* g;
  const bf16_t* resid;   // RESIDUAL / DGELU: input; BIAS_GELU2: the second output (written)
  int64_t ldr;
  int M, N, K, tilesM, tilesN, row_group;
  int ncu, share, S;   // big-tile kernel: persistent workgroups, leftover tiles shared (0/1) by S workgroups each
  int band;            // big-tile kernel: N-tiles per column band of the tile order (tilesN: plain row-major)
  // LayerNorm statistics carried between GEMMs (big-tile kernel).  stats_out: the RESIDUAL epilogue also writes, per
  // output row m and 256-column tile t, the raw moments (sum, sum of squares) of that tile's values of the row to
  // stats_out[(m * VITM_STATS_PARTS + t) * 2].  stats_partial: the LN epilogues read `stats` in that layout (all
  // VITM_STATS_PARTS slots are summed; unused ones must be zero) and form mean / rstd over ln_dim values themselves.
  float* stats_out;
  int stats_partial;
  int stats_last;      // the column tile whose epilogue also zero-fills the unused slots of stats_out (the real last one)
  float ln_inv_dim, ln_eps;
  // im2col source (patch embedding).  Big-tile kernel: img != nullptr selects the gathering loader; rows are
  // (image, token) with `cls` leading class-token slots per image, pw = 1 << pw_shift
  const void* img;
  int C, H, Wd, ph, pw, gw, npatch, kreal, vec;
  int cls, pw_shift;
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(l), 16, 0, 0);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
// row-major tiles so the A row panel and the W panels it needs stay in that L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Per-thread column constants of the row-contiguous epilogue (8 consecutive output columns).
template <int EPI>
struct EpiCols {
  float cv[8], sv[8], gv[8];
  __device__ __forceinline__ void load(const GemmArgs& p, int n) {
    const float4 c0 = *(const float4*)(p.c + n), c1 = *(const float4*)(p.c + n + 4);
    cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w;
    cv[4] = c1.x; cv[5] = c1.y; cv[6] = c1.z; cv[7] = c1.w;
    if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) {
      const float4 s0 = *(const float4*)(p.s + n), s1 = *(const float4*)(p.s + n + 4);
      sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w;
      sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
    }
    if (EPI == VITM_EPI_RESIDUAL) {
      const float4 g0 = *(const float4*)(p.g + n), g1 = *(const float4*)(p.g + n + 4);
      gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w;
      gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
    }
  }
  // v[8] = fp32 accumulators of row m, columns n..n+7 -> fixed up and stored
  template <int OUT_F32>
  __device__ __forceinline__ void finish(const GemmArgs& p, int m, int n, float* v) const {
    u16x8_t rv = {0, 0, 0, 0, 0, 0, 0, 0};
    float2 st = make_float2(0.f, 1.f);
    if (EPI == VITM_EPI_RESIDUAL || EPI == VITM_EPI_DGELU) rv = *(const u16x8_t*)(p.resid + (int64_t)m * p.ldr + n);
    if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) st = *(const float2*)(p.stats + 2 * (int64_t)m);
    finish_pre<OUT_F32>(p, m, n, v, rv, st);
  }
  // same with the row's residual values / LayerNorm statistics already in registers
  template <int OUT_F32>
  __device__ __forceinline__ void finish_pre(const GemmArgs& p, int m, int n, float* v, u16x8_t rv, float2 st) const {
    if (EPI == VITM_EPI_LN || EPI == VITM_EPI_LN_GELU) {
      // rstd * (acc - mean * s) + c as two plain v_fma_f32 per value, issued from inline asm on purpose.  Left to the
      // compiler this becomes v_pk_fma_f32 with the row's mean / rstd broadcast to both halves, and that form was seen
      // to lose the product of its LOW half in lanes 48..63 now and then (the output came out as c[n] bit for bit, with
      // every operand register correct when read back afterwards) whenever a second workgroup of this kernel shared the
      // CU: 100 .. 400 wrong elements per launch at M = 6304, N = 1152, K = 384, none with one workgroup per CU, none
      // with the unpacked instruction (tests/test_gpu_repeatable.py::test_gemm_small_tile_is_repeatable).
      const float nmean = -st.x;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(sv[e]), "v"(nmean), "v"(v[e]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[e]) : "v"(t), "v"(st.y), "v"(cv[e]));
      }
    } else if (EPI == VITM_EPI_BIAS || EPI == VITM_EPI_BIAS_GELU || EPI == VITM_EPI_BIAS_GELU2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += cv[e];
    }
    if (EPI == VITM_EPI_BIAS_GELU2) {   // the pre-activation to `out`, its GELU (below) to the second output
      uint4 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      o.z = pack_bf2(v[4], v[5]);
      o.w = pack_bf2(v[6], v[7]);
      *(uint4*)((bf16_t*)p.out + (int64_t)m * p.ldo + n) = o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2_t gl = gelu_erf2((f32x2_t){v[e], v[e + 1]});
        o.x = e == 0 ? pack_bf2(gl.x, gl.y) : o.x;
        o.y = e == 2 ? pack_bf2(gl.x, gl.y) : o.y;
        o.z = e == 4 ? pack_bf2(gl.x, gl.y) : o.z;
        o.w = e == 6 ? pack_bf2(gl.x, gl.y) : o.w;
      }
      *(uint4*)(const_cast<bf16_t*>(p.resid) + (int64_t)m * p.ldr + n) = o;
      return;
    }
    if (EPI == VITM_EPI_LN_GELU || EPI == VITM_EPI_BIAS_GELU) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2_t gl = gelu_erf2((f32x2_t){v[e], v[e + 1]});
        v[e] = gl.x;
        v[e + 1] = gl.y;
      }
    }
    if (EPI == VITM_EPI_RESIDUAL) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf2f(rv[e]) + fmaf(gv[e], v[e], cv[e]);
    }
    if (EPI == VITM_EPI_DGELU) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2_t d = dgelu_erf2((f32x2_t){bf2f(rv[e]), bf2f(rv[e + 1])});
        v[e] *= d.x;
        v[e + 1] *= d.y;
      }
    }
    const int64_t om = p.row_group > 0 ? (int64_t)m + m / p.row_group + 1 : (int64_t)m;
    if (OUT_F32) {
      float* o = (float*)p.out + om * p.ldo + n;
      *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      uint4 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      o.z = pack_bf2(v[4], v[5]);
      o.w = pack_bf2(v[6], v[7]);
      *(uint4*)((bf16_t*)p.out + om * p.ldo + n) = o;
    }
  }
};

}  // namespace vitm_gemm_detail

// big-tile persistent kernel (gemm_pp.hip): bf16 output, row_group == 0, arguments already validated
int vitm_gemm_pp_launch(const vitm_gemm_detail::GemmArgs& a, int epilogue, hipStream_t st);
bool vitm_gemm_pp_wanted(int M, int N, int K);
