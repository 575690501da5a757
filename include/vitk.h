/*
 * vitk.h -- C-ABI of libvitk.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the ViT
 * training step of PaulParizacaMozo/VisionTransformer.
 *
 * This is the accelerator boundary: plain device pointers, sizes and a HIP stream handle; no C++
 * or torch types.  It plays the role the reference's own `*_cuda(const Tensor&)` op library plays
 * in its CUDA tree (cuda_Vit/include/utils/CudaUtils.cuh:15-53), except that tensors stay
 * device-resident, nothing here allocates or frees memory, and nothing synchronises the stream.
 * Each entry point cites the reference CPU function (file:line in the reference tree) whose
 * arithmetic it reproduces.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; vitk_last_error() gives the
 *     message of the calling thread's last failure (the C++ host layer turns it into
 *     std::runtime_error, the reference's error convention, app/main.cpp:76-79);
 *   - all tensors are float32, row-major, in device memory (hipMalloc or equivalent);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - workspaces are caller-provided; the *_workspace_bytes() helpers give their sizes.
 */
#ifndef VITK_H
#define VITK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *vitk_stream_t;

/* ---- library -------------------------------------------------------------------------------- */
int vitk_version(void);
const char *vitk_last_error(void);
/* Select `device` for the calling thread and verify that it is a gfx950 part. */
int vitk_init(int device);

/* ---- dense contractions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) --------------------
 * C[b] = epilogue(alpha * op(A[b]) * op(B[b])) for b in [0, batch).
 * Replaces matrixMultiply (src/core/Tensor.cpp:472-504) and batchMatrixMultiply (:510-544) and,
 * through the epilogue, the addBroadcast bias add (:435-445, Dense.cpp:43,53), GELU forward and
 * backward (src/activations/GELU.cpp:13-42, 44-81) and the residual adds of
 * TransformerEncoderBlock (src/model/TransformerEncoderBlock.cpp:28,34).  Transposed operands
 * are addressed in place -- the reference's transpose()+contiguous() copies never materialise.
 *
 *   trans_a = 0: A is [M][K] (lda >= K)     trans_a = 1: A is [K][M] (lda >= M)
 *   trans_b = 0: B is [K][N] (ldb >= N)     trans_b = 1: B is [N][K] (ldb >= K)
 *   batch index b = bo * batch_inner + bi; operand X of batch b starts at
 *   X + bo * stride_x_outer + bi * stride_x_inner (elements).  batch = batch_outer*batch_inner.
 *
 * Epilogue, in this order, per element (m, n):
 *   v = alpha * acc
 *   act == VITK_ACT_GELU_BWD:  v = v * gelu'(aux[m][n])               (aux is an input)
 *   bias != NULL:              v += bias[n]
 *   act == VITK_ACT_GELU_FWD:  if (aux) aux[m][n] = v;  v = gelu(v)   (aux is an output)
 *   residual != NULL:          v += residual[m][n]
 *   C[m][n] = v
 * split_k > 1 (or 0 = let the library choose) splits the contraction over several workgroups;
 * partial tiles go to `workspace` and are summed in a fixed order by a second kernel, so results
 * do not depend on scheduling.  Split-K requires alpha == 1 and no bias/act/residual.
 */
enum { VITK_ACT_NONE = 0, VITK_ACT_GELU_FWD = 1, VITK_ACT_GELU_BWD = 2 };

typedef struct vitk_gemm_args {
  int32_t trans_a, trans_b;
  int32_t M, N, K;
  const float *A;
  int64_t lda;
  const float *B;
  int64_t ldb;
  float *C;
  int64_t ldc;
  float alpha;
  int32_t act;
  const float *bias;
  const float *residual;
  int64_t ldr;
  float *aux;
  int64_t ldaux;
  int32_t batch_outer, batch_inner; /* 0 is treated as 1 */
  int64_t stride_a_outer, stride_a_inner;
  int64_t stride_b_outer, stride_b_inner;
  int64_t stride_c_outer, stride_c_inner;
  int32_t split_k; /* 1 = none, 0 = auto */
  int32_t reserved; /* 0 = the library's tile plan; 1 / 2 / 3 force 64 x 64 / 128 x 64 / 64 x 128 output tiles
                     * (measurements and tests; the rectangular ones only where the unguarded kernel applies) */
  float *workspace;
  size_t workspace_bytes;
  /* optional: colsum_b[n] = sum_k B[k][n] (trans_b = 0, batch = 1 only), accumulated from the B
   * tiles the contraction loads anyway -- the bias gradient sum(0) of Dense::backward
   * (src/layers/Dense.cpp:95) comes for free with dW = X^T dY.  Fixed summation order. */
  float *colsum_b;
} vitk_gemm_args;

int vitk_gemm_f32(const vitk_gemm_args *args, vitk_stream_t stream);
/* Bytes of workspace vitk_gemm_f32 needs for these arguments (0 when it will not split K). */
size_t vitk_gemm_workspace_bytes(const vitk_gemm_args *args);
/* The two products of Dense::backward (src/layers/Dense.cpp:90-98) as one call: dx_args describes
 * dX = dY W^T (trans_a = 0, trans_b = 1; any epilogue), dw_args describes dW = X^T dY (trans_a = 1,
 * trans_b = 0, optionally colsum_b).  Neither product may read what the other writes.  When both
 * are problems of the small-problem kernel (below) they run as ONE launch -- on the reference's
 * own small models a launch costs more than its arithmetic; otherwise this is vitk_gemm_f32 on
 * dw_args, then on dx_args, each with its own workspace requirement (pass the larger).  Results are
 * bitwise those of the two separate calls. */
int vitk_gemm_pair_f32(const vitk_gemm_args *dx_args, const vitk_gemm_args *dw_args, vitk_stream_t stream);

/* Measurement aid for bench.py (not used by the training path): while enabled, every main
 * contraction launch is bracketed by two HIP events on its stream.  variant = 16*colsum + 8*full +
 * 4*(tile==64) + 2*trans_a + trans_b (0..31) identifies the instantiation of the tiled kernel and
 * 32 + 2*trans_a + trans_b the small-problem kernel, 36 .. 51 the rectangular tiles, 52 the paired
 * small-problem launch of vitk_gemm_pair_f32 (vitk_gemm_variant_name gives the name a
 * profiler shows); vitk_gemm_profile_read returns the summed event time, the summed algorithmic
 * FLOPs (2*M*N*K*batch) and the launch count of one variant, optionally clearing all records. */
int vitk_gemm_profile_enable(int on);
int vitk_gemm_profile_read(int variant, double *total_ms, double *total_flops, int64_t *launches, int reset);
const char *vitk_gemm_variant_name(int variant);

/* Problems of at most 8 x (compute units) 16 x 16 output tiles with K <= 1024, no batch and no
 * forced split run on a kernel of their own (one wave per tile, operands straight from global
 * memory: on a nearly idle chip the tiled kernel's slab pipeline is pure latency).  Same
 * arguments, epilogue and colsum_b; the summation order over k differs, so results agree with the
 * tiled kernel to rounding, not bitwise.  In the weight-gradient form (trans_a, not trans_b) with K > 128 and SIMDs
 * to spare, up to eight waves share a tile, 128 k each (one round trip to memory per wave), and add their
 * results in wave order.  on = 0 keeps everything on the tiled kernel, on = 2 keeps the small kernel at one wave
 * per tile (tests, measurements); the default is 1.  Process-wide. */
int vitk_gemm_small_enable(int on);
/* Several launch plans size themselves by the device's CU count: which problems take the
 * small-problem kernel, the split count of a long contraction, the tile shape, heads per attention
 * workgroup.  Split counts and kernel routes change the order in which a sum is formed, so the same
 * model gives results that differ in the last bits between devices with different CU counts (a
 * partitioned MI355X, for one).  vitk_set_planning_cu_count(n) makes every plan use n instead of
 * the device's count (n = 256 repeats a whole MI355X's bits anywhere; any n is correct, only
 * slower or faster); 0 restores the device's own.  Process-wide. */
int vitk_set_planning_cu_count(int cus);

/* ---- column sum: out[j] = sum_i x[i][j]  (Tensor::sum(0), src/core/Tensor.cpp:385-430; the
 * bias gradient of Dense::backward, src/layers/Dense.cpp:95, and dpos/dcls of
 * Embeddings::backward, src/layers/Embeddings.cpp:43,50).  Two-stage, fixed order.
 * workspace: vitk_colsum_workspace_bytes(rows, cols). */
size_t vitk_colsum_workspace_bytes(int64_t rows, int64_t cols);
int vitk_colsum_f32(const float *x, int64_t rows, int64_t cols, int64_t ldx, float *out, float *workspace,
                    size_t workspace_bytes, vitk_stream_t stream);

/* ---- LayerNorm (src/layers/LayerNorm.cpp:24-89 forward, :91-163 backward) ---------------------
 * forward: two-pass mean / biased variance per row, inv_std = 1/sqrt(var+eps),
 * y = gamma * (x-mean)*inv_std + beta.  mean / inv_std ([rows], may be NULL for inference) are
 * what backward needs; x-hat is recomputed there instead of being stored.
 * backward: dx = (1/D)*inv_std*(D*dxhat - sum(dxhat) - xhat*sum(dxhat*xhat)), dxhat = dy*gamma;
 * if `residual` is non-NULL it is added to dx (the skip branch of the block's backward,
 * src/model/TransformerEncoderBlock.cpp:52,63).  dgamma/dbeta are reduced over rows in two
 * fixed-order stages (per-workgroup partials in the workspace, then a reduction launch); with at
 * most 1024 rows and aligned operands everything happens in one launch (a fixed-order sum per
 * column, the workspace is not touched) unless vitk_layernorm_small_enable(0) turned that off
 * (tests, measurements; process-wide).  Rows are `row_stride` elements apart (row_stride >= D). */
size_t vitk_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D);
int vitk_layernorm_small_enable(int on);
int vitk_layernorm_fwd(const float *x, int64_t rows, int64_t D, int64_t row_stride, const float *gamma,
                       const float *beta, float eps, float *y, float *mean, float *inv_std, vitk_stream_t stream);
int vitk_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *inv_std, const float *gamma,
                       int64_t rows, int64_t D, int64_t row_stride, const float *residual, float *dx, float *dgamma,
                       float *dbeta, float *workspace, size_t workspace_bytes, vitk_stream_t stream);

/* vitk_layernorm_bwd for callers that cannot allocate or pass a workspace.  It has no counterpart
 * in the reference; it exists for callers that must not call hipMalloc: code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation.  Same arithmetic, kernels and
 * results (bit for bit) as vitk_layernorm_bwd, but the per-workgroup dgamma/dbeta partials live
 * in a fixed 8 MiB array inside the library's device code (one copy per GPU, held until the
 * process ends) instead of a workspace: no hipMalloc/hipFree, no allocating call of any kind.
 * Limit: D <= vitk_layernorm_bwd_static_max_d() (1024); larger D is rejected with an error.
 * vitk_init(device) resolves the array's address for that device; without it the first call on
 * a device does (one hipGetSymbolAddress).  All calls on one GPU share the array: calls that
 * overlap on one GPU (different streams or threads) race -- serialise them. */
size_t vitk_layernorm_bwd_static_max_d(void);
int vitk_layernorm_bwd_static(const float *dy, const float *x, const float *mean, const float *inv_std,
                              const float *gamma, int64_t rows, int64_t D, int64_t row_stride, const float *residual,
                              float *dx, float *dgamma, float *dbeta, vitk_stream_t stream);

/* ---- GELU, tanh approximation (src/activations/GELU.cpp:9,13-42,44-81) ------------------------ */
int vitk_gelu_fwd(const float *x, int64_t n, float *y, vitk_stream_t stream);
int vitk_gelu_bwd(const float *x, const float *dy, int64_t n, float *dx, vitk_stream_t stream);

/* ---- inverted dropout (src/layers/Dropout.cpp:21-59 forward, :61-87 backward) -----------------
 * mask[i] = 0 with probability `rate`, else 1/(1-rate); y = x*mask.  The reference draws from a
 * random_device-seeded mt19937, so only the distribution can be matched: the device stream is
 * Philox4x32-10 keyed by (seed, offset + element index / 4). */
int vitk_dropout_fwd(const float *x, int64_t n, float rate, uint64_t seed, uint64_t offset, float *y, float *mask,
                     vitk_stream_t stream);
int vitk_dropout_bwd(const float *dy, const float *mask, int64_t n, float *dx, vitk_stream_t stream);
/* The same forward with the skip connection folded in, y = residual + x*mask (residual may be
 * NULL), and with the mask optional: the factors are a pure function of (seed, offset, element
 * index), so the backward can regenerate them instead of reading a stored mask:
 * vitk_dropout_stream_bwd gives dx = dy*mask for the mask of the forward call with the same
 * (rate, seed, offset).  (x + dropout(...) in src/model/TransformerEncoderBlock.cpp:28,34.) */
int vitk_dropout_residual_fwd(const float *x, const float *residual, int64_t n, float rate, uint64_t seed,
                              uint64_t offset, float *y, float *mask, vitk_stream_t stream);
int vitk_dropout_stream_bwd(const float *dy, int64_t n, float rate, uint64_t seed, uint64_t offset, float *dx,
                            vitk_stream_t stream);

/* The same factors applied to a compact [rows][cols] slice of the tensor a dense call would have
 * covered: element (r, c) takes the factor of flat position r*idx_stride + c (read from `mask`
 * when it is not NULL, else regenerated from (rate, seed, offset)); y = x*factor (+ residual).
 * Forward and backward of the dropout around the one token per sequence that the last encoder
 * block computes, with the masks the dense block would have drawn. */
int vitk_dropout_rows(const float *x, const float *residual, const float *mask, int64_t rows, int64_t cols,
                      int64_t idx_stride, float rate, uint64_t seed, uint64_t offset, float *y, vitk_stream_t stream);

/* ---- element-wise helpers (Tensor::operator+, src/core/Tensor.cpp:316-351; fill :128-137) ----- */
int vitk_add_f32(const float *a, const float *b, int64_t n, float *out, vitk_stream_t stream);
int vitk_add3_f32(const float *a, const float *b, const float *c, int64_t n, float *out, vitk_stream_t stream);
int vitk_scale_f32(const float *x, float s, int64_t n, float *out, vitk_stream_t stream);
int vitk_fill_f32(float *x, float value, int64_t n, vitk_stream_t stream);
/* x[r][c] += v[c] in place: Tensor::addBroadcast ({1,N} over {M,N} and {1,N,D} over {B,N,D},
 * src/core/Tensor.cpp:435-461) outside a GEMM epilogue. */
int vitk_add_rowvec_f32(float *x, int64_t rows, int64_t cols, const float *v, vitk_stream_t stream);
/* strided 2-D copy: dst[r][c] = src[r][c], r < rows, c < cols (contiguous()/slice copies,
 * src/core/Tensor.cpp:227-254). */
int vitk_copy2d_f32(const float *src, int64_t ld_src, float *dst, int64_t ld_dst, int64_t rows, int64_t cols,
                    vitk_stream_t stream);
/* strided 2-D accumulate: dst[r][c] += src[r][c] (a gradient that exists for some rows only, added
 * into the rows of a full tensor) */
int vitk_add2d_f32(const float *src, int64_t ld_src, float *dst, int64_t ld_dst, int64_t rows, int64_t cols,
                   vitk_stream_t stream);
/* gather of a strided view of up to 4 dimensions (shape4/strides4 in elements, host arrays of 4;
 * pad leading dimensions with extent 1) into a contiguous buffer: the general form of
 * Tensor::contiguous() for transposed / sliced / stride-0 expanded views. */
int vitk_copy4d_f32(const float *src, const int64_t *shape4, const int64_t *strides4, float *dst,
                    vitk_stream_t stream);

/* ---- row softmax over the last axis and its backward (src/layers/MultiHeadAttention.cpp:123-156
 * and :279-308).  Rows are ld elements apart; columns [n, ld) of the output are written as 0.
 * backward: dx = scale * p * (g - sum_k g_k p_k). */
int vitk_softmax_rows(const float *x, int64_t rows, int64_t n, int64_t ld, float *p, vitk_stream_t stream);
int vitk_softmax_rows_bwd(const float *g, const float *p, int64_t rows, int64_t n, int64_t ld, float scale, float *dx,
                          vitk_stream_t stream);

/* ---- attention key-gradient layout of the reference --------------------------------------------
 * MultiHeadAttention::backward computes Q^T dS as a [dh][N] buffer per (batch, head) and then
 * re-reads that buffer as if it were [N][dh] (src/layers/MultiHeadAttention.cpp:207-222), so the
 * key gradient the reference trains with is a permutation of the true dK = dS^T Q.
 * VITK_DK_REFERENCE reproduces it; VITK_DK_EXACT is the mathematically correct gradient. */
enum { VITK_DK_REFERENCE = 0, VITK_DK_EXACT = 1 };
/* The re-read itself: t is [B*heads][dh][N] (= Q^T dS per head); writes
 * dk[(b*N + n')*ld_dk + head*dh + d'] = t[bh][n'*dh + d' (flat index into the [dh][N] buffer)]. */
int vitk_dk_reference_scatter(const float *t, int32_t B, int32_t heads, int32_t N, int32_t dh, float *dk,
                              int64_t ld_dk, vitk_stream_t stream);

/* ---- fused multi-head scaled-dot-product attention ---------------------------------------------
 * One launch for  scores = Q K^T * scale -> softmax over keys -> P V  per (batch, head)
 * (src/layers/MultiHeadAttention.cpp:96-156; cuda_Vit scaledDotProductAttention_cuda,
 * cuda_Vit/include/utils/CudaUtils.cuh:24), and its backward (src/layers/MultiHeadAttention.cpp:
 * 160-308) without materialising the [B*H][N][N] matrices.  q, k, v (and dq, dk, dv) address
 * element (b, n, head, d) at ptr[(b*N + n)*ld + head*dh + d], i.e. head split / merge is address
 * arithmetic on the packed projections; out / dout likewise with ld_out.
 * stats (optional in the forward, required by the backward) is [B*H][N][2] = {row maximum of the
 * scaled scores, row sum of exp(score - maximum)}: the backward recomputes the probabilities from
 * it.  dk_mode selects the key-gradient layout (VITK_DK_*).  Supported: N <= 256, dh <= 64
 * (vitk_sdpa_supported); callers fall back to vitk_gemm_f32 + vitk_softmax_rows otherwise.
 * workspace: vitk_sdpa_bwd_workspace_bytes(B, H, N) bytes. */
int vitk_sdpa_supported(int32_t N, int32_t dh);
/* Head dimensions 33..64 with 16-byte-aligned operands run the forward on 16-row tiles with four
 * heads per workgroup (197 tokens = 13 tiles instead of 7 x 32 = 224 rows, every wave busy in
 * every round).  on = 0 keeps them on the 32-row kernel that serves the other head sizes (tests,
 * measurements); the default is on; on >= 2 also fixes the heads per workgroup (otherwise chosen
 * from the number of heads and the device size).  Same results to rounding.  Process-wide. */
int vitk_sdpa_tile16_enable(int on);
/* The same switch for the backward (one launch per call in both forms; the 16-row form deals the
 * key blocks of several heads to the waves of a workgroup: as many heads as it takes for every
 * workgroup to be resident at once, or exactly `on` heads when on >= 2 -- for tests). */
int vitk_sdpa_bwd_tile16_enable(int on);
/* Sequences of at most 32 tokens with dh <= 32 and 16-byte-aligned operands (the 5- and 17-token
 * models) are one 32 x 32 tile per head: one wave does a whole head from registers, forward in
 * one launch and backward in one launch, four heads per workgroup.  on = 0 keeps them on the
 * tiled kernels (tests, measurements); the default is on.  Same results to rounding.  Process-wide. */
int vitk_sdpa_head_kernels_enable(int on);
int vitk_sdpa_fwd(const float *q, const float *k, const float *v, int64_t ld_qkv, int32_t B, int32_t H, int32_t N,
                  int32_t dh, float scale, float *out, int64_t ld_out, float *stats, vitk_stream_t stream);
size_t vitk_sdpa_bwd_workspace_bytes(int32_t B, int32_t H, int32_t N);
int vitk_sdpa_bwd(const float *q, const float *k, const float *v, int64_t ld_qkv, const float *out, const float *dout,
                  int64_t ld_out, const float *stats, int32_t B, int32_t H, int32_t N, int32_t dh, float scale,
                  int32_t dk_mode, float *dq, float *dk, float *dv, int64_t ld_dqkv, float *workspace,
                  size_t workspace_bytes, vitk_stream_t stream);

/* The same attention core for ONE query token per (batch, head) -- all that the last encoder block
 * needs, because only its CLS token reaches the classification head
 * (src/model/VisionTransformer.cpp:27-37, 49-56).  q / out / dout / dq address element (b, h, d) at
 * ptr[b*ld + h*dh + d] (one row per batch entry); k, v, dk, dv as in vitk_sdpa_fwd.  stats is
 * [B*H][2].  The backward writes dk / dv for ALL keys (outer products) and dq for the one row.
 * Supported: N <= 256, dh <= 64 (vitk_sdpa_row_supported). */
int vitk_sdpa_row_supported(int32_t N, int32_t dh);
int vitk_sdpa_row_fwd(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, int32_t B, int32_t H,
                      int32_t N, int32_t dh, float scale, float *out, int64_t ld_out, float *stats,
                      vitk_stream_t stream);
int vitk_sdpa_row_bwd(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *out,
                      const float *dout, int64_t ld_out, const float *stats, int32_t B, int32_t H, int32_t N, int32_t dh,
                      float scale, int32_t dk_mode, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                      vitk_stream_t stream);

/* ---- batch assembly: dst[r][:] = src[index[r]][:]  (the per-batch gather of shuffled samples,
 * src/model/Trainer.cpp:163-199, done on the device from a dataset that stays resident in HBM).
 * index is a device array of `rows` int32 values in [0, src_rows). */
int vitk_gather_rows_f32(const float *src, int64_t src_rows, const int32_t *index, int64_t rows, int64_t row_len,
                         float *dst, vitk_stream_t stream);

/* ---- data augmentation (DataAugmentation::apply, src/utils/DataAugmentation.cpp:4-143) --------
 * dst[b] = zoom(translate(rotate(src[index ? index[b] : b]))), each stage present only when its
 * flag is set: the reference's three nearest-neighbour resamplings with zero fill, composed into
 * one gather per output pixel (bit-identical to applying them one after another).  The random
 * draws stay on the host (DataAugmentation in the host layer draws them in the reference's
 * order); the kernel receives one record per image.  With `index` (device, B int32 values in
 * [0, src_images)) the per-batch gather of shuffled samples is folded into the same pass.
 *   rotate:    source = R(theta) (pixel - centre) + centre, cos/sin of the drawn angle (:42-64)
 *   translate: source = pixel - (ty, tx), the drawn shifts truncated to int (:87-97)
 *   zoom:      source = (pixel - centre) / zoom + centre (:127-128)
 * src and dst are [.][C][H][W]; must not alias. */
enum { VITK_AUG_ROTATE = 1, VITK_AUG_TRANSLATE = 2, VITK_AUG_ZOOM = 4 };
typedef struct vitk_augment_params {
  int32_t flags;
  float cos_theta, sin_theta;
  int32_t tx, ty;
  float zoom;
  int32_t reserved[2];
} vitk_augment_params; /* 32 bytes */
int vitk_augment_f32(const float *src, int64_t src_images, const int32_t *index, const vitk_augment_params *params,
                     int64_t B, int32_t C, int32_t H, int32_t W, float *dst, vitk_stream_t stream);

/* ---- patch extraction (src/layers/PatchEmbedding.cpp:31-52) -----------------------------------
 * img is [B][C][HW][HW]; patches is [B*Np][P*P*C], Np = (HW/P)^2.
 * VITK_PATCH_REFERENCE: what the reference CPU path does -- patch (ph,pw) is the P*P*C CONTIGUOUS
 * floats starting at flat offset ph*P*HW + pw*P of the image (not an im2col; for C > 1 only
 * channel 0 is ever read).  VITK_PATCH_IM2COL: true ViT patches, element order (c, i, j), the
 * layout the reference's CUDA tree uses (cuda_Vit/src/layers/Conv2d.cu:49-57). */
enum { VITK_PATCH_REFERENCE = 0, VITK_PATCH_IM2COL = 1 };
int vitk_patch_gather(const float *img, int32_t B, int32_t C, int32_t HW, int32_t P, int32_t mode, float *patches,
                      vitk_stream_t stream);

/* Its backward (src/layers/PatchEmbedding.cpp:74-103): the image gradient dimg [B][C][HW][HW] from
 * the patch-row gradients dpatches [B*Np][P*P*C] (= dY W^T of the projection).  VITK_PATCH_IM2COL:
 * each pixel takes element (c, i, j) of its own patch's row.  VITK_PATCH_REFERENCE: the
 * reference reads every patch's row through a pointer that ignores the row offset (:86-87), so all
 * patches of all images receive row 0; reproduced as is.  (Trainer discards this gradient,
 * src/model/Trainer.cpp:212.) */
int vitk_patch_scatter_bwd(const float *dpatches, int32_t B, int32_t C, int32_t HW, int32_t P, int32_t mode,
                           float *dimg, vitk_stream_t stream);

/* ---- embeddings (src/layers/Embeddings.cpp:24-40 forward, :42-74 backward) --------------------
 * assemble: out[b][0][:] = cls + pos[0]; out[b][1+p][:] = proj[b*Np+p][:] + pos[1+p].
 * split (backward): dproj[b*Np+p][:] = dy[b][1+p][:]   (dpos/dcls come from vitk_colsum_f32 over
 * dy viewed as [B][N*D]). */
int vitk_embed_assemble(const float *proj, const float *cls, const float *pos, int32_t B, int32_t Np, int32_t D,
                        float *out, vitk_stream_t stream);
int vitk_embed_split_bwd(const float *dy, int32_t B, int32_t Np, int32_t D, float *dproj, vitk_stream_t stream);

/* ---- CLS token slice and its backward (src/model/VisionTransformer.cpp:35, :49-56) ------------
 * gather: cls_rows[b][:] = x[b][0][:];  scatter: dx = 0 except dx[b][0][:] = dcls[b][:]. */
int vitk_cls_gather(const float *x, int32_t B, int32_t N, int32_t D, float *cls_rows, vitk_stream_t stream);
int vitk_cls_scatter(const float *dcls, int32_t B, int32_t N, int32_t D, float *dx, vitk_stream_t stream);

/* ---- softmax cross-entropy, forward + backward fused (src/losses/CrossEntropy.cpp:20-111) -----
 * probs = softmax(logits); loss = (1/B) sum_i -w[c_i] * log(probs[i][c_i] + 1e-12) over the
 * entries where y == 1 (class weights scale the LOSS only); dlogits = (probs - y) / grad_batch
 * (grad_batch = B on one device; the global batch under data parallelism).  Also counts rows
 * whose first-maximum logit is a labelled class (Trainer.cpp:23-50).
 * out2[0] = loss, out2[1] = number of correct rows (as float).  workspace: 2*B floats. */
int vitk_softmax_ce(const float *logits, const float *y_onehot, int32_t B, int32_t K, const float *class_w,
                    float grad_batch, float *dlogits, float *probs, float *out2, float *workspace,
                    size_t workspace_bytes, vitk_stream_t stream);
/* first index of the row maximum (app/test.cpp:103-112) */
int vitk_argmax_rows(const float *x, int32_t B, int32_t K, int32_t *out, vitk_stream_t stream);

/* ---- Adam with coupled L2 over a flat arena (src/optimizers/Adam.cpp:29-60) -------------------
 * g' = g + wd*p (if wd > 0); m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2;
 * p -= lr * (m/(1-beta1_t)) / (sqrt(v/(1-beta2_t)) + eps).  beta?_t = float(pow(beta, t)) is
 * computed by the caller, as the reference does (:32-33).  grad_scale multiplies g first
 * (1 normally; 1/world when averaging all-reduced gradients). */
int vitk_adam_l2(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                 float beta1_t, float beta2_t, float eps, float wd, float grad_scale, vitk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
