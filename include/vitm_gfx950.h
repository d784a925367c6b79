/*
 * libvitm_gfx950 -- C ABI of the MI355X (gfx950) forward path for the vitm Vision Transformer.
 *
 * The reference (MohitKumar020291/Vision-Transformer, package `vitm`) has no FFI of its own: its
 * boundary is the Python nn.Module API of vitm/models/__init__.py:1-12.  These entry points are
 * what a binding for that path calls, one per fused step of the forward; each comment names the
 * reference code the step replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; the library never allocates or frees;
 *  - matrices are row-major with an explicit leading dimension in ELEMENTS;
 *  - activations are bf16 unless an `*_dtype` argument says otherwise or the entry point is one of
 *    the `*_f32` family (the full-precision path: fp32 operands, fp32 matrix instructions, erf-accurate
 *    GELU; it serves fp32 callers, who expect 1e-4 agreement with an fp32 reference);
 *  - small per-channel vectors (biases, LayerNorm affine, LayerScale, folded-LN sums) are fp32;
 *  - calls are asynchronous on `stream` (a hipStream_t; NULL = the default stream) and run on
 *    the calling thread's current HIP device; the library keeps no mutable global state;
 *  - return 0 on success, a negative VITM_E* code otherwise; vitm_last_error() returns a
 *    thread-local message for the last failure on the calling thread.
 */
#ifndef VITM_GFX950_H
#define VITM_GFX950_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vitm_stream_t; /* hipStream_t */

enum vitm_dtype { VITM_BF16 = 0, VITM_F32 = 1 };

enum vitm_status {
  VITM_OK = 0,
  VITM_EINVAL = -1,   /* bad argument (null pointer, unsupported size/alignment) */
  VITM_ELAUNCH = -2,  /* the HIP runtime refused the launch */
  VITM_EUNSUPPORTED = -3
};

/* Epilogues of vitm_gemm (acc = sum_k A[m,k] * Wp[n,k], fp32). */
enum vitm_epilogue {
  VITM_EPI_BIAS = 0,      /* out = acc + c[n]                                                  */
  VITM_EPI_BIAS_GELU = 1, /* out = gelu_erf(acc + c[n])                                        */
  VITM_EPI_LN = 2,        /* out = rstd[m] * (acc - mean[m] * s[n]) + c[n]    (LayerNorm folded) */
  VITM_EPI_LN_GELU = 3,   /* out = gelu_erf(rstd[m] * (acc - mean[m] * s[n]) + c[n])            */
  VITM_EPI_RESIDUAL = 4,  /* out = resid[m,n] + g[n] * acc + c[n]   (bias + LayerScale + residual) */
  VITM_EPI_DGELU = 5,     /* out = acc * gelu_erf'(resid[m,n]): the GELU backward fused into the input-gradient
                             GEMM of the Linear that follows it (training; resid = the GELU's input, c unused)  */
  VITM_EPI_BIAS_GELU2 = 6 /* out = acc + c[n] AND resid[m,n] = gelu_erf(acc + c[n]): `resid` is a second OUTPUT here
                             (bf16, leading dimension ldr, written), so the training forward keeps the pre-activation
                             for the backward without a separate GELU pass (models.py:127-128)                 */
};

const char* vitm_version(void);
const char* vitm_last_error(void);

/* K granularity the GEMM needs: packed weights and A operands have K padded to a multiple of this. */
int vitm_gemm_k_multiple(void);

/*
 * One-time weight packing for a Linear / patch-projection layer, optionally folding the LayerNorm
 * that precedes it and the LayerScale that follows it:
 *   wp[n,k] = wp_dtype(w[n,k] * ln_gamma[k])       k < K, zero for K <= k < kpad
 *   s[n]    = sum_k float(wp[n,k])                 (only when s != NULL)
 *   c[n]    = (sum_k ln_beta[k] * w[n,k] + bias[n]) * out_scale[n]
 *   g[n]    = out_scale[n]                         (only when g != NULL)
 * Null ln_gamma / ln_beta / bias / out_scale mean 1 / 0 / 0 / 1.  `w` is [N,K] row-major in
 * `w_dtype` (nn.Linear.weight, or Conv2d.weight viewed as (D, C*ph*pw)); `wp` is [N,kpad] in
 * `wp_dtype` (bf16 for vitm_gemm, fp32 for vitm_gemm_f32); vectors are fp32.
 * Replaces nothing at run time: it prepares nn.Linear weights (models.py:51,57,121,124),
 * nn.LayerNorm affines (models.py:189,196) and LayerScale.gamma (models.py:135) for the fused GEMMs.
 */
int vitm_prepack_linear(const void* w, int w_dtype, const float* bias, const float* ln_gamma,
                        const float* ln_beta, const float* out_scale, int N, int K, int kpad,
                        void* wp, int wp_dtype, float* s, float* c, float* g, vitm_stream_t stream);

/*
 * Row statistics of LayerNorm: stats[m] = (mean, 1/sqrt(var + eps)) over x[m, 0:D], x bf16 or f32, D <= 8192
 * (the limit of vitm_layernorm and vitm_embed_finalize as well: a row is held in registers).
 * First half of nn.LayerNorm (models.py:189,196,310,324); the affine half is folded into the
 * consuming GEMM (VITM_EPI_LN*).
 */
int vitm_ln_stats(const void* x, int x_dtype, int64_t ldx, int M, int D, float eps, float* stats,
                  vitm_stream_t stream);

/*
 * out[M,N] = epilogue(A[M,K] . Wp[N,K]^T).  A and Wp are bf16, K a multiple of
 * vitm_gemm_k_multiple() (pad columns of A must be finite, pad columns of Wp zero), N a multiple
 * of 8.  `stats` is [M][2] (VITM_EPI_LN*), `s`,`c`,`g` are [N] fp32, `resid` is bf16 [M,ldr]
 * and may alias `out` (VITM_EPI_RESIDUAL).  `out_dtype` is VITM_BF16 or VITM_F32.
 * When row_group > 0 output row m goes to row m + m / row_group + 1 (token rows of a (B, 1+Np, D)
 * buffer, leaving row 0 of every image for the class token).
 * Replaces nn.Linear in Attention.forward (models.py:76,94), MLP.forward with its GELU
 * (models.py:126-130), LayerScale + the residual adds of Block.forward (models.py:209-210) and the
 * classifier head (classifier.py:38).
 */
int vitm_gemm(const void* A, int64_t lda, const void* Wp, int64_t ldw, void* out, int64_t ldo,
              int out_dtype, int M, int N, int K, int epilogue, const float* stats, const float* s,
              const float* c, const float* g, const void* resid, int64_t ldr, int row_group,
              vitm_stream_t stream);

/*
 * vitm_gemm (bf16 output, row_group 0) on the 256x256-tile route with the LayerNorm row statistics carried from
 * the GEMM that writes the residual stream to the GEMM that normalises it, instead of a vitm_ln_stats pass over
 * the stream in between:
 *  - stats_out != NULL (VITM_EPI_RESIDUAL, N <= VITM_STATS_PARTS * 256): besides `out`, the epilogue writes for every
 *    row m and 256-column tile t the raw moments of that tile's output values of the row,
 *    stats_out[(m * VITM_STATS_PARTS + t) * 2 + {0, 1}] = (sum, sum of squares); slots t >= ceil(N / 256) are
 *    not written and must be zero (clear the [M][VITM_STATS_PARTS][2] buffer once).
 *  - stats_partial != 0 (VITM_EPI_LN*): `stats` is such a buffer; the epilogue sums the slots and forms
 *    mean = sum / ln_dim, rstd = 1 / sqrt(sumsq / ln_dim - mean^2 + ln_eps) itself.
 * Fails with VITM_EINVAL when the shape does not take the 256x256-tile route (ask vitm_gemm_uses_big_tile first).
 * Replaces the statistics half of the two nn.LayerNorm calls of Block.forward (models.py:189,196,209-210).
 */
#define VITM_STATS_PARTS 4
int vitm_gemm_stats(const void* A, int64_t lda, const void* Wp, int64_t ldw, void* out, int64_t ldo, int M, int N,
                    int K, int epilogue, const float* stats, const float* s, const float* c, const float* g,
                    const void* resid, int64_t ldr, float* stats_out, int stats_partial, int ln_dim, float ln_eps,
                    vitm_stream_t stream);

/*
 * Patch embedding as an im2col-fused GEMM: tokens[b, 1 + py*gw + px, :] =
 *   sum_{c,ky,kx} img[b,c,py*ph+ky,px*pw+kx] * Wp[:, (c,ky,kx)] + c[:]
 * img is NCHW (bf16 or f32) contiguous, Wp is the packed (D, kpad) projection weight.  With
 * cls_slot = 1 the output buffer is (B, 1+Np, D) and row 0 of each image is left untouched;
 * with cls_slot = 0 it is (B, Np, D).
 * Replaces PatchEmbed.forward's Conv2d + flatten + transpose (patch_embeddings.py:115-117).
 */
int vitm_patch_embed(const void* img, int img_dtype, int B, int C, int H, int W, int ph, int pw,
                     const void* Wp, int64_t ldw, int kpad, const float* c, void* tokens,
                     int64_t ldo, int D, int cls_slot, vitm_stream_t stream);

/*
 * 1 when vitm_gemm (bf16 output, row_group 0) runs this shape on the persistent 256x256-tile kernel, 0 when on
 * the 128x128 one.  vitm_patch_embed gathers the patches in the tile loader of either kernel for bf16 images with a
 * power-of-two patch width; for other inputs it always takes the 128x128 one, and a caller with a large batch does
 * better with vitm_im2col + vitm_gemm (one extra pass over the patches, then the faster kernel).
 */
int vitm_gemm_uses_big_tile(int M, int N, int K);

/*
 * Patch matrix for vitm_gemm: out[(b, cls_slot + patch), k] = img[b, c, py*ph+ky, px*pw+kx], k = (c, ky, kx) in
 * Conv2d.weight.reshape(D, -1) order, bf16 or fp32 (out_dtype), K padded with zeros to kpad; with cls_slot = 1 row 0 of every image
 * is a zero row (the GEMM then leaves bias there; vitm_embed_finalize overwrites it with the class token).
 * img is NCHW bf16 or fp32.  First half of PatchEmbed.forward's Conv2d (patch_embeddings.py:115) at large batch.
 */
int vitm_im2col(const void* img, int img_dtype, int B, int C, int H, int W, int ph, int pw, void* out,
                int out_dtype, int64_t ldo, int kpad, int cls_slot, vitm_stream_t stream);

/*
 * Full-precision GEMM: as vitm_gemm with A, Wp, out and resid all fp32 and the fp32 matrix
 * instruction.  K a multiple of vitm_gemm_k_multiple(), N and the leading dimensions multiples of 4.
 * Epilogues: VITM_EPI_BIAS, VITM_EPI_BIAS_GELU (erf to 1.5e-7), VITM_EPI_RESIDUAL; LayerNorm is not
 * folded on this path (call vitm_layernorm first).  Replaces the same reference lines as vitm_gemm
 * for fp32 tensors.
 */
int vitm_gemm_f32(const float* A, int64_t lda, const float* Wp, int64_t ldw, float* out, int64_t ldo,
                  int M, int N, int K, int epilogue, const float* c, const float* g,
                  const float* resid, int64_t ldr, int row_group, vitm_stream_t stream);

/* vitm_patch_embed with an fp32 packed weight and fp32 tokens (patch_embeddings.py:115-117). */
int vitm_patch_embed_f32(const void* img, int img_dtype, int B, int C, int H, int W, int ph, int pw,
                         const float* Wp, int64_t ldw, int kpad, const float* c, float* tokens,
                         int64_t ldo, int D, int cls_slot, vitm_stream_t stream);

/*
 * In place on x (B, 1+Np, D), bf16 or fp32 (x_dtype): row 0 <- class token; token rows optionally LayerNorm'ed with
 * (en_w,en_b) (PatchEmbed.norm_layer); then every row optionally LayerNorm'ed with (np_w,np_b)
 * (norm_pre), then + pos[n,:].  Null pointers skip the step.
 * Replaces patch_embeddings.py:120 and VisionTransformer.forward_features (models.py:356-360).
 */
int vitm_embed_finalize(void* x, int x_dtype, int64_t ldx, int B, int ntok, int D, const float* cls,
                        const float* pos, const float* en_w, const float* en_b, float en_eps,
                        const float* np_w, const float* np_b, float np_eps, vitm_stream_t stream);

/*
 * vitm_embed_finalize that also leaves stats[m] = (mean, 1/sqrt(var + stats_eps)) of every finished row (the layout of
 * vitm_ln_stats, and the same numbers: they are taken from the values as stored, i.e. after rounding to bf16 when x is
 * bf16): the LayerNorm folded into the first GEMM that reads x (norm1 of block 0, models.py:209) then needs no pass of
 * its own.  stats may be null.
 */
int vitm_embed_finalize_stats(void* x, int x_dtype, int64_t ldx, int B, int ntok, int D, const float* cls,
                              const float* pos, const float* en_w, const float* en_b, float en_eps,
                              const float* np_w, const float* np_b, float np_eps, float* stats,
                              float stats_eps, vitm_stream_t stream);

/*
 * Fused multi-head self-attention on a packed qkv activation: qkv is (B*N, 3*D) bf16 laid out as
 * nn.Linear(dim, 3*dim) writes it ([q | k | v], each D = H*hd wide, head-major inside), out is
 * (B*N, D) bf16 token-major:  out = softmax(scale * q k^T) v  per (image, head).
 * head_dim is a multiple of 8 up to 128 (64 takes the persistent kernels; dims other than 32 / 96 / 128 run padded
 * with zeros to the next multiple of 32 inside the kernel, e.g. 80 of ViT-H).
 * Replaces models.py:79-93 (reshape/permute, q*scale, q@k^T, softmax, @v, transpose/reshape).
 */
int vitm_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int N, int H,
                   int head_dim, float scale, vitm_stream_t stream);

/*
 * Per-head LayerNorm of q and k, in place on the packed qkv activation (bf16 or fp32): for every row and
 * head, q[h*hd : (h+1)*hd] and k[h*hd : (h+1)*hd] are normalised over head_dim with (q_w, q_b, q_eps) /
 * (k_w, k_b, k_eps); v is untouched (head_dim 32, 64, 96 or 128).  Run it between the qkv GEMM and vitm_attention.
 * Replaces q_norm / k_norm of Attention(qk_norm=True) (models.py:54-55,82).
 */
int vitm_qk_norm(void* qkv, int dtype, int64_t ldqkv, int M, int H, int head_dim, const float* q_w,
                 const float* q_b, float q_eps, const float* k_w, const float* k_b, float k_eps,
                 vitm_stream_t stream);

/*
 * Same attention with fp32 qkv and out and fp32 arithmetic throughout (leading dimensions multiples
 * of 4; head_dim a multiple of 4 up to 128).  Replaces models.py:79-93 for fp32 tensors.
 */
int vitm_attention_f32(const float* qkv, int64_t ldqkv, float* out, int64_t ldo, int B, int N, int H,
                       int head_dim, float scale, vitm_stream_t stream);

/*
 * Full LayerNorm: out[m,:] = (x[m,:] - mean) * rstd * w + b over D; x and out each bf16 or f32.
 * Replaces norm_final (models.py:410), PatchEmbed.norm_layer when used stand-alone, and norm1 /
 * norm2 (models.py:209-210) on the fp32 path.
 */
int vitm_layernorm(const void* x, int x_dtype, int64_t ldx, void* out, int out_dtype, int64_t ldo,
                   int M, int D, const float* w, const float* b, float eps, vitm_stream_t stream);

/* out[m,n] = x[m,n] * gamma[n]  (LayerScale.forward stand-alone, models.py:138-139); x and out in `dtype`. */
int vitm_scale_channels(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D,
                        const float* gamma, vitm_stream_t stream);

/*
 * out[m,n] = gelu_erf(x[m,n]), x and out in `dtype`: nn.GELU() stand-alone, as the ViT-22B block's MLP applies
 * it to an already projected hidden activation (parallel_block.py:121-124).
 */
int vitm_gelu(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D, vitm_stream_t stream);

/*
 * out = act(x) elementwise for an MLP built with an activation other than the exact GELU (MLP's act_layer argument,
 * models.py:106,122: the reference instantiates whatever class it is given).  x, out as in vitm_gelu (may alias).
 * The exact GELU stays fused in the fc1 epilogue; these run as a pass of their own between the two GEMMs.
 */
enum vitm_activation_kind {
  VITM_ACT_GELU_ERF = 0, /* nn.GELU()                   */
  VITM_ACT_RELU = 1,     /* nn.ReLU()                   */
  VITM_ACT_GELU_TANH = 2,/* nn.GELU(approximate="tanh") */
  VITM_ACT_SILU = 3,     /* nn.SiLU()                   */
  VITM_ACT_SIGMOID = 4,  /* nn.Sigmoid()                */
  VITM_ACT_TANH = 5      /* nn.Tanh()                   */
};
int vitm_activation(const void* x, int64_t ldx, void* out, int64_t ldo, int dtype, int M, int D, int kind,
                    vitm_stream_t stream);

/* dst (bf16) <- src (f32), n elements; used to bring fp32 images/activations onto the bf16 path. */
int vitm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, vitm_stream_t stream);
/* dst (f32) <- src (bf16), n elements. */
int vitm_cast_bf16_to_f32(const void* src, float* dst, int64_t n, vitm_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Training step (vitm/training/train.py:208-241).  The GEMMs of the backward pass are vitm_gemm / vitm_gemm_f32
 * calls on transposed operands; the entry points below are the rest.  Activation-sized operands are fp32 (the
 * reference's training precision) or bf16 (`dtype`: mixed precision, fp32 statistics and reductions).
 * ------------------------------------------------------------------------------------------------------ */

/*
 * vitm_attention_f32 that also writes lse[b, h, n] = log2 of the softmax denominator of query n (scores taken
 * times scale*log2(e)), the one per-row number the backward pass needs to recompute the probabilities.
 */
int vitm_attention_f32_lse(const float* qkv, int64_t ldqkv, float* out, int64_t ldo, float* lse, int B, int N,
                           int H, int head_dim, float scale, vitm_stream_t stream);

/* vitm_attention (bf16) that also writes the same lse array (mixed-precision training forward). */
int vitm_attention_lse(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, float* lse, int B, int N, int H,
                       int head_dim, float scale, vitm_stream_t stream);

/*
 * Backward of models.py:88-93: dqkv (B*N, 3*D; [dq | dk | dv], every element written) from qkv, the forward
 * output o, d_o and lse.  `delta` is a (B*H*N) fp32 scratch array (row sums of d_o * o).
 */
int vitm_attention_bwd_f32(const float* qkv, int64_t ldqkv, const float* o, int64_t ldo, const float* d_o,
                           int64_t lddo, const float* lse, float* dqkv, int64_t lddqkv, float* delta, int B, int N,
                           int H, int head_dim, float scale, vitm_stream_t stream);

/*
 * vitm_attention_bwd_f32 for bf16 operands on the bf16 matrix cores (fp32 accumulation, fp32 lse / delta): qkv, o, d_o
 * and dqkv are bf16 with leading dimensions that are multiples of 8; lse is what vitm_attention_lse wrote.
 */
int vitm_attention_bwd(const void* qkv, int64_t ldqkv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                       const float* lse, void* dqkv, int64_t lddqkv, float* delta, int B, int N, int H, int head_dim,
                       float scale, vitm_stream_t stream);

/*
 * Weight gradient of a Linear layer without transposed copies: out[n, k] (fp32) = sum_m a[m, n] * b[m, k] for bf16
 * row-major a (M x N, dy) and b (M x K, the layer's input); backward of models.py:76,94,127,129 and classifier.py:38.
 * colsum_out (nullable, N floats) receives sum_m a[m, n], the bias gradient, from the same pass over a.
 * row_scale (nullable, N floats) multiplies row n of out and colsum_out[n]: a LayerScale between the Linear and the
 * gradient that arrives (dy = upstream * gamma) without materialising dy.
 * lda and ldb are multiples of 8.  The reduction over m is split over workgroup slices whose fp32 partial results go to
 * `workspace` (vitm_gemm_tn_workspace_floats(M, N, K) floats, 0 when one slice is enough) and are added in a fixed order.
 */
int vitm_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t ldo, float* colsum_out,
                 const float* row_scale, int M, int N, int K, float* workspace, int64_t workspace_floats,
                 vitm_stream_t stream);
int64_t vitm_gemm_tn_workspace_floats(int M, int N, int K);

/*
 * dst[n, m] = src[m, n] for m < M, zero for M <= m < m_pad (K padding of a GEMM operand), converting src_dtype to
 * dst_dtype on the way: W^T of an fp32 master weight as the bf16 operand of the input-gradient GEMM in one pass.
 * row_scale (nullable, M floats) multiplies source row m (a LayerScale folded into W^T).
 */
int vitm_transpose(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd, int M, int N,
                   int m_pad, const float* row_scale, vitm_stream_t stream);

/*
 * out[c, d] = sum over the rows m of chunk c (nchunks equal chunks of the M rows) of a[m, d] * (b ? b[m, d] : 1).
 * Bias gradients (b = NULL), LayerScale gradients (a = d_y, b = branch output), second stage of LayerNorm
 * parameter gradients.  a and b are in `dtype`, out is fp32.  Call once with nchunks > 1 into a partial array, then
 * once (dtype fp32) with nchunks = 1 over it.
 */
int vitm_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, int dtype, float* out, int64_t ldo, int M,
                int D, int nchunks, vitm_stream_t stream);

/*
 * out_a[d] = sum_m a[m, d] and out_b[d] = sum_m b[m, d] for two contiguous fp32 (rows, D) arrays (b / out_b may both
 * be null), one launch, fixed
 * summation order: reduces the dw / db partial sums of vitm_layernorm_bwd to the gradients of nn.LayerNorm's weight
 * and bias (models.py:189,196,310,324 under loss.backward(), train.py:219).
 */
int vitm_colsum_pair(const float* a, const float* b, int rows, int D, float* out_a, float* out_b, vitm_stream_t stream);

/*
 * LayerScale gradient out of the weight gradient of the Linear in front of it (x + gamma * Linear(h), models.py:209-210,
 * 138-139): on entry dw (N, K) / db (N, nullable together with bias) hold the Linear's gradients taken WITHOUT gamma; on
 * return dgamma[n] = sum_k w[n,k] dw[n,k] + bias[n] db[n] (= sum_m dy[m,n] z[m,n], so the Linear's output z need not be
 * kept) and dw, db are multiplied by gamma[n] in place: the true gradients.  round_bf16: w is rounded to bf16 first, as
 * the bf16 forward multiplied by it.
 */
int vitm_layerscale_grads(const float* w, const float* bias, float* dw, float* db, const float* gamma, float* dgamma, int N,
                          int K, int round_bf16, vitm_stream_t stream);

/*
 * LayerNorm backward: dx[m,:] = (dx_add ? dx_add[m,:] : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;
 * dw_partial / db_partial are (nblocks, D) fp32 per-workgroup sums of dy * xhat / dy (reduce them with vitm_colsum_pair);
 * x, dy, dx_add and dx are in `dtype`.  Backward of nn.LayerNorm (models.py:189,196,310,324).
 */
int vitm_layernorm_bwd(const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* w, float eps,
                       const void* dx_add, int64_t ldadd, void* dx, int64_t lddx, int dtype, float* dw_partial,
                       float* db_partial, int nblocks, int M, int D, vitm_stream_t stream);

/* out[m,d] = resid[m,d] + gamma[d] * z[m,d]: LayerScale + residual add with the branch output kept (models.py:209-210). */
int vitm_residual_scale(const void* resid, int64_t ldr, const void* z, int64_t ldz, const float* gamma, void* out,
                        int64_t ldo, int dtype, int M, int D, vitm_stream_t stream);

/*
 * Backward of the per-head LayerNorm of q and k (models.py:54-55,82; parallel_block.py:86).  qkv holds the values
 * BEFORE the norm; dqkv holds d(loss)/d(normalised q, k) on entry and d(loss)/d(q, k) on return (the v third is left
 * alone).  partial is [4][nblocks][head_dim] fp32: per-workgroup sums of (dq_w, dq_b, dk_w, dk_b), to be reduced over
 * nblocks with vitm_colsum.
 */
int vitm_qk_norm_bwd(const void* qkv, int64_t ldqkv, void* dqkv, int64_t lddqkv, int dtype, int M, int H, int head_dim,
                     const float* q_w, float q_eps, const float* k_w, float k_eps, float* partial, int nblocks,
                     vitm_stream_t stream);

/* dpre = dh * d/dx gelu_erf(pre): backward of nn.GELU() (models.py:128). */
int vitm_gelu_bwd(const void* pre, int64_t ldp, const void* dh, int64_t ldh, void* dpre, int64_t ldd, int dtype, int M,
                  int D, vitm_stream_t stream);

/*
 * loss[0] = mean over the batch of softmax cross-entropy(logits[b,:], labels[b]); dlogits (optional) its gradient.
 * nn.CrossEntropyLoss() with class-index targets (train.py:187,220).  row_loss: B floats of scratch (the per-image
 * losses, added in index order).  Labels must lie in [0, C): a label outside it (ignore_index -100 included, which is
 * not implemented) gives that row a NaN loss -- hence a NaN mean -- and a zero gradient row, never an out-of-bounds read.
 */
int vitm_cross_entropy_f32(const float* logits, int64_t ld, const int64_t* labels, int B, int C, float* loss,
                           float* dlogits, int64_t lddl, float* row_loss, vitm_stream_t stream);

/*
 * vitm_cross_entropy_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * or training loops, harnesses that forbid hipMalloc): no counterpart in the reference.  Same arithmetic and results
 * as vitm_cross_entropy_f32, bit for bit; the per-row losses go to a device array inside the library (64 KiB per GPU,
 * held until the process ends) instead of `row_loss`, so B is limited to vitm_cross_entropy_noalloc_max_rows()
 * (16384) -- a larger B is refused with VITM_EINVAL.  The call allocates and frees nothing.  The first call on a
 * device looks the array's address up (hipGetSymbolAddress): make that call outside a graph capture.  All calls on one
 * GPU share the array: calls that overlap on one GPU (different streams) race; calls on one stream are ordered.
 * `stream` must belong to the calling thread's current device (each GPU has its own copy of the array and the kernels
 * must get the copy of the GPU they run on): a stream of another device is refused with VITM_EINVAL.
 */
int vitm_cross_entropy_f32_noalloc(const float* logits, int64_t ld, const int64_t* labels, int B, int C, float* loss,
                                   float* dlogits, int64_t lddl, vitm_stream_t stream);
int vitm_cross_entropy_noalloc_max_rows(void);

/*
 * torch.nn.utils.clip_grad_norm_(parameters, max_norm) (train.py:222) on one flat gradient buffer of n floats:
 * norm_and_coef[0] = L2 norm, [1] = min(1, max_norm / (norm + 1e-6)), grads *= [1].  `scratch` holds
 * vitm_grad_clip_scratch_floats() floats.
 */
int vitm_grad_clip_f32(float* grads, int64_t n, float max_norm, float* scratch, float* norm_and_coef,
                       vitm_stream_t stream);
int vitm_grad_clip_scratch_floats(void);

/* One torch.optim.AdamW update (train.py:68-72) of a parameter tensor of n floats; `step` counts from 1. */
int vitm_adamw_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, vitm_stream_t stream);

/*
 * The same update with its scalars in DEVICE memory: hyper[7] = { lr, beta1, beta2, eps, weight_decay, 1 - beta1^step,
 * sqrt(1 - beta2^step) }.  A launch captured into a HIP graph then follows the learning-rate schedule and the step count
 * of train.py:223-224 (optimizer.step(); scheduler.step()) by rewriting those seven floats before each replay.
 */
int vitm_adamw_f32_hyper(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper,
                         vitm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITM_GFX950_H */
