/*
 * TEST INFRASTRUCTURE ONLY -- see vit_oracle.h.  CPU restatement of the reference's ViT
 * training step.  Citations are file:line in the reference tree (PaulParizacaMozo/VisionTransformer).
 * Parity status: PINNED against oracle/_ref (the compiled reference) and tests/golden/.
 */
#include "vit_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float *falloc(uint64_t n) {
  float *p = (float *)calloc(n ? n : 1, sizeof(float));
  if (!p) abort();
  return p;
}

/* ------------------------------------------------------------------------------------------
 * matrixMultiply, src/core/Tensor.cpp:472-504: C[i][j] = sum_k a(i,k)*b(k,j), k sequential,
 * rows parallel.  Strided operands (lda/ldb + transposes) cover the transposed views the
 * reference passes from Dense::backward (src/layers/Dense.cpp:92-98).
 * ------------------------------------------------------------------------------------------ */
static void mm_strided(const float *a, int64_t a_rs, int64_t a_cs, const float *b, int64_t b_rs, int64_t b_cs,
                       uint64_t m, uint64_t n, uint64_t p, float *out) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)m; ++i) {
    for (uint64_t j = 0; j < p; ++j) {
      float sum = 0.0f;
      for (uint64_t k = 0; k < n; ++k) sum += a[i * a_rs + (int64_t)k * a_cs] * b[(int64_t)k * b_rs + (int64_t)j * b_cs];
      out[i * (int64_t)p + (int64_t)j] = sum;
    }
  }
}

void vo_matmul(const float *a, const float *b, uint64_t m, uint64_t n, uint64_t p, float *out) {
  mm_strided(a, (int64_t)n, 1, b, (int64_t)p, 1, m, n, p, out);
}

/* batchMatrixMultiply, src/core/Tensor.cpp:510-544: parallel over batch only. */
static void bmm_strided(const float *a, int64_t a_bs, int64_t a_rs, int64_t a_cs, const float *b, int64_t b_bs,
                        int64_t b_rs, int64_t b_cs, uint64_t bs, uint64_t m, uint64_t n, uint64_t p, float *out) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)bs; ++i) {
    for (uint64_t j = 0; j < m; ++j) {
      for (uint64_t k = 0; k < p; ++k) {
        float sum = 0.0f;
        for (uint64_t l = 0; l < n; ++l)
          sum += a[i * a_bs + (int64_t)j * a_rs + (int64_t)l * a_cs] * b[i * b_bs + (int64_t)l * b_rs + (int64_t)k * b_cs];
        out[(i * (int64_t)m + (int64_t)j) * (int64_t)p + (int64_t)k] = sum;
      }
    }
  }
}

void vo_bmm(const float *a, const float *b, uint64_t bs, uint64_t m, uint64_t n, uint64_t p, float *out) {
  bmm_strided(a, (int64_t)(m * n), (int64_t)n, 1, b, (int64_t)(n * p), (int64_t)p, 1, bs, m, n, p, out);
}

/* Dense::forward, src/layers/Dense.cpp:25-57: Y = X W (matrixMultiply) then addBroadcast of the
 * {1,out} bias (src/core/Tensor.cpp:435-445). */
void vo_dense_forward(const float *x, uint64_t rows, uint64_t in, uint64_t out, const float *W, const float *b,
                      float *y) {
  vo_matmul(x, W, rows, in, out, y);
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)rows; ++i)
    for (uint64_t j = 0; j < out; ++j) y[i * (int64_t)out + (int64_t)j] += b[j];
}

/* Dense::backward, src/layers/Dense.cpp:62-106: dW = X^T dY (sum over rows in order),
 * db = dY.sum(0) (src/core/Tensor.cpp:397-409, sequential over rows), dX = dY W^T. */
void vo_dense_backward(const float *x, const float *dy, uint64_t rows, uint64_t in, uint64_t out, const float *W,
                       float *dW, float *db, float *dx) {
  mm_strided(x, 1, (int64_t)in, dy, (int64_t)out, 1, in, rows, out, dW);
  for (uint64_t j = 0; j < out; ++j) {
    float s = 0.0f;
    for (uint64_t k = 0; k < rows; ++k) s += dy[k * out + j];
    db[j] = s;
  }
  if (dx) mm_strided(dy, (int64_t)out, 1, W, 1, (int64_t)out, rows, out, in, dx);
}

/* LayerNorm::forward, src/layers/LayerNorm.cpp:24-89: two-pass mean / biased variance,
 * inv_std = 1/sqrt(var+eps), xhat saved, y = gamma*xhat + beta. */
void vo_layernorm_forward(const float *x, uint64_t rows, uint64_t D, const float *gamma, const float *beta, float eps,
                          float *y, float *xhat, float *inv_std) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)rows; ++i) {
    const float *xi = x + i * (int64_t)D;
    float mean = 0.0f;
    for (uint64_t j = 0; j < D; ++j) mean += xi[j];
    mean /= D;
    float var = 0.0f;
    for (uint64_t j = 0; j < D; ++j) {
      float diff = xi[j] - mean;
      var += diff * diff;
    }
    var /= D;
    float inv = 1.0f / sqrtf(var + eps);
    if (inv_std) inv_std[i] = inv;
    for (uint64_t j = 0; j < D; ++j) {
      float xh = (xi[j] - mean) * inv;
      if (xhat) xhat[i * (int64_t)D + (int64_t)j] = xh;
      y[i * (int64_t)D + (int64_t)j] = gamma[j] * xh + beta[j];
    }
  }
}

/* LayerNorm::backward, src/layers/LayerNorm.cpp:91-163: sequential over rows; dgamma/dbeta
 * accumulate in fp32 in row order; dx = (1/D)*inv_std*(D*dxhat - sum(dxhat) - xhat*sum(dxhat*xhat)). */
void vo_layernorm_backward(const float *dy, const float *xhat, const float *inv_std, const float *gamma,
                           uint64_t rows, uint64_t D, float *dgamma, float *dbeta, float *dx) {
  for (uint64_t j = 0; j < D; ++j) dgamma[j] = 0.0f, dbeta[j] = 0.0f;
  for (uint64_t i = 0; i < rows; ++i) {
    float inv = inv_std[i];
    float dot = 0, sum = 0;
    for (uint64_t j = 0; j < D; ++j) {
      float g = dy[i * D + j];
      float xh = xhat[i * D + j];
      dgamma[j] += g * xh;
      dbeta[j] += g;
      float dxh = g * gamma[j];
      sum += dxh;
      dot += dxh * xh;
    }
    for (uint64_t j = 0; j < D; ++j) {
      float dxh = dy[i * D + j] * gamma[j];
      float xh = xhat[i * D + j];
      float term1 = D * dxh;
      float term2 = sum;
      float term3 = xh * dot;
      dx[i * D + j] = (1.0f / D) * inv * (term1 - term2 - term3);
    }
  }
}

/* GELU::forward / backward, src/activations/GELU.cpp:9,13-42,44-81 (tanh approximation). */
static const float SQRT_2_OVER_PI = 0.7978845608028654f;

void vo_gelu_forward(const float *x, uint64_t n, float *y) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    float v = x[i];
    float x_cubed = v * v * v;
    float inner = SQRT_2_OVER_PI * (v + 0.044715f * x_cubed);
    y[i] = 0.5f * v * (1.0f + tanhf(inner));
  }
}

void vo_gelu_backward(const float *x, const float *dy, uint64_t n, float *dx) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    float v = x[i];
    float x_squared = v * v;
    float x_cubed = x_squared * v;
    float inner = SQRT_2_OVER_PI * (v + 0.044715f * x_cubed);
    float tanh_inner = tanhf(inner);
    float d_inner_dx = SQRT_2_OVER_PI * (1.0f + 3.0f * 0.044715f * x_squared);
    float sech_squared = 1.0f - tanh_inner * tanh_inner;
    float d = 0.5f * (1.0f + tanh_inner) + 0.5f * v * sech_squared * d_inner_dx;
    dx[i] = d * dy[i];
  }
}

/* softmax(t, 2), src/layers/MultiHeadAttention.cpp:123-156, and the 2-D softmax of
 * src/losses/CrossEntropy.cpp:20-50 (same row algorithm): max, exp(x-max), sum, divide. */
void vo_softmax_rows(const float *x, uint64_t rows, uint64_t n, float *p) {
#pragma omp parallel for
  for (int64_t r = 0; r < (int64_t)rows; ++r) {
    const float *xr = x + r * (int64_t)n;
    float *pr = p + r * (int64_t)n;
    float mx = -INFINITY;
    for (uint64_t d = 0; d < n; ++d)
      if (xr[d] > mx) mx = xr[d];
    float sum = 0.0f;
    for (uint64_t d = 0; d < n; ++d) {
      float e = expf(xr[d] - mx);
      pr[d] = e;
      sum += e;
    }
    for (uint64_t d = 0; d < n; ++d) pr[d] /= sum;
  }
}

/* softmax_backward, src/layers/MultiHeadAttention.cpp:279-308: dS_i = P_i (dP_i - sum_k dP_k P_k). */
void vo_softmax_rows_backward(const float *g, const float *p, uint64_t rows, uint64_t n, float *dx) {
#pragma omp parallel for
  for (int64_t r = 0; r < (int64_t)rows; ++r) {
    const float *gr = g + r * (int64_t)n;
    const float *pr = p + r * (int64_t)n;
    float dot = 0.0f;
    for (uint64_t k = 0; k < n; ++k) dot += gr[k] * pr[k];
    for (uint64_t i = 0; i < n; ++i) dx[r * (int64_t)n + (int64_t)i] = pr[i] * (gr[i] - dot);
  }
}

/* {B,N,h,dh} <-> {B,h,N,dh}: the reshape/transpose(1,2)/contiguous pairs of
 * src/layers/MultiHeadAttention.cpp:41-59 (split) and :73-84 (merge). */
static void split_heads(const float *x, uint64_t B, uint64_t N, uint64_t h, uint64_t dh, float *out) {
  for (uint64_t b = 0; b < B; ++b)
    for (uint64_t n = 0; n < N; ++n)
      for (uint64_t k = 0; k < h; ++k)
        memcpy(out + ((b * h + k) * N + n) * dh, x + ((b * N + n) * h + k) * dh, dh * sizeof(float));
}
static void merge_heads(const float *x, uint64_t B, uint64_t N, uint64_t h, uint64_t dh, float *out) {
  for (uint64_t b = 0; b < B; ++b)
    for (uint64_t n = 0; n < N; ++n)
      for (uint64_t k = 0; k < h; ++k)
        memcpy(out + ((b * N + n) * h + k) * dh, x + ((b * h + k) * N + n) * dh, dh * sizeof(float));
}

typedef struct {
  const float *x; /* borrowed */
  float *q, *k, *v, *P, *ctx;
} mha_cache;

static void mha_cache_free(mha_cache *c) {
  free(c->q); free(c->k); free(c->v); free(c->P); free(c->ctx);
}

/* MultiHeadAttention::forward, src/layers/MultiHeadAttention.cpp:24-88 and
 * scaledDotProductAttention :90-119. */
static void mha_forward(uint64_t D, uint64_t h, const float *prm, const float *x, uint64_t B, uint64_t N, float *y,
                        mha_cache *c) {
  const uint64_t dh = D / h, M = B * N, BH = B * h;
  const float *qW = prm, *qb = qW + D * D, *kW = qb + D, *kb = kW + D * D, *vW = kb + D, *vb = vW + D * D,
              *oW = vb + D, *ob = oW + D * D;
  float *t = falloc(M * D);
  c->x = x;
  c->q = falloc(M * D); c->k = falloc(M * D); c->v = falloc(M * D);
  vo_dense_forward(x, M, D, D, qW, qb, t); split_heads(t, B, N, h, dh, c->q);
  vo_dense_forward(x, M, D, D, kW, kb, t); split_heads(t, B, N, h, dh, c->k);
  vo_dense_forward(x, M, D, D, vW, vb, t); split_heads(t, B, N, h, dh, c->v);
  /* scores = Q K^T (k transposed view), * 1/sqrt(dh), softmax, @ V */
  float *scores = falloc(BH * N * N);
  bmm_strided(c->q, (int64_t)(N * dh), (int64_t)dh, 1, c->k, (int64_t)(N * dh), 1, (int64_t)dh, BH, N, dh, N, scores);
  float scale = 1.0f / sqrtf((float)dh);
  for (uint64_t i = 0; i < BH * N * N; ++i) scores[i] *= scale;
  c->P = falloc(BH * N * N);
  vo_softmax_rows(scores, BH * N, N, c->P);
  free(scores);
  vo_bmm(c->P, c->v, BH, N, N, dh, t);
  c->ctx = falloc(M * D);
  merge_heads(t, B, N, h, dh, c->ctx);
  free(t);
  vo_dense_forward(c->ctx, M, D, D, oW, ob, y);
}

/* MultiHeadAttention::backward, src/layers/MultiHeadAttention.cpp:160-249. */
static void mha_backward(uint64_t D, uint64_t h, const float *prm, const mha_cache *c, uint64_t B, uint64_t N,
                         const float *dy, float *g, float *dx) {
  const uint64_t dh = D / h, M = B * N, BH = B * h, blk = D * D + D;
  const float *qW = prm, *kW = qW + blk, *vW = kW + blk, *oW = vW + blk;
  float *gq = g, *gk = g + blk, *gv = g + 2 * blk, *go = g + 3 * blk;
  float *t = falloc(M * D), *dctx = falloc(M * D);
  vo_dense_backward(c->ctx, dy, M, D, D, oW, go, go + D * D, t);
  split_heads(t, B, N, h, dh, dctx);
  /* dP = dctx V^T ; dV = P^T dctx */
  float *dP = falloc(BH * N * N), *dS = falloc(BH * N * N);
  bmm_strided(dctx, (int64_t)(N * dh), (int64_t)dh, 1, c->v, (int64_t)(N * dh), 1, (int64_t)dh, BH, N, dh, N, dP);
  float *dV = falloc(M * D), *dQ = falloc(M * D), *dK = falloc(M * D);
  bmm_strided(c->P, (int64_t)(N * N), 1, (int64_t)N, dctx, (int64_t)(N * dh), (int64_t)dh, 1, BH, N, N, dh, dV);
  vo_softmax_rows_backward(dP, c->P, BH * N, N, dS);
  float scale = 1.0f / sqrtf((float)dh);
  for (uint64_t i = 0; i < BH * N * N; ++i) dS[i] *= scale;
  vo_bmm(dS, c->k, BH, N, N, dh, dQ);
  bmm_strided(c->q, (int64_t)(N * dh), 1, (int64_t)dh, dS, (int64_t)(N * N), (int64_t)N, 1, BH, dh, N, N, t);
  /* t is {BH, dh, N}: dK = Q^T dS gives {BH,dh,N}?  No: reference computes Q_T{BH,dh,N} @ dS{BH,N,N}
   * = {BH,dh,N} and then reshapes that buffer as {B,h,N,dh} (src/layers/MultiHeadAttention.cpp:207-222). */
  memcpy(dK, t, M * D * sizeof(float));
  free(dP); free(dS);
  float *m1 = falloc(M * D), *dxq = falloc(M * D), *dxk = falloc(M * D), *dxv = falloc(M * D);
  merge_heads(dQ, B, N, h, dh, m1);
  vo_dense_backward(c->x, m1, M, D, D, qW, gq, gq + D * D, dxq);
  merge_heads(dK, B, N, h, dh, m1);
  vo_dense_backward(c->x, m1, M, D, D, kW, gk, gk + D * D, dxk);
  merge_heads(dV, B, N, h, dh, m1);
  vo_dense_backward(c->x, m1, M, D, D, vW, gv, gv + D * D, dxv);
  for (uint64_t i = 0; i < M * D; ++i) dx[i] = (dxq[i] + dxk[i]) + dxv[i];
  free(t); free(dctx); free(dV); free(dQ); free(dK); free(m1); free(dxq); free(dxk); free(dxv);
}

void vo_mha(uint64_t D, uint64_t h, const float *params, const float *x, uint64_t B, uint64_t N, const float *dy,
            float *y, float *grads, float *dx) {
  mha_cache c;
  mha_forward(D, h, params, x, B, N, y, &c);
  if (dy) mha_backward(D, h, params, &c, B, N, dy, grads, dx);
  mha_cache_free(&c);
}

typedef struct {
  const float *x;
  float *pre, *act;
} ffn_cache;

/* Dropout::forward with the mask given instead of drawn, src/layers/Dropout.cpp:42-51: the mask
 * holds 0 or 1/(1-rate); a dropped element is written as +0, a kept one as in*scale. */
static void dropout_apply(float *x, const float *mask, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) x[i] = mask[i] == 0.0f ? 0.0f : x[i] * mask[i];
}

/* FeedForward::forward, src/layers/FeedForward.cpp:18-24.  mask == NULL: dropout rate 0 or
 * inference (identity, src/layers/Dropout.cpp:23); otherwise the {M,D} mask of its Dropout. */
static void ffn_forward(uint64_t D, uint64_t H, const float *prm, const float *x, uint64_t M, const float *mask,
                        float *y, ffn_cache *c) {
  const float *W1 = prm, *b1 = W1 + D * H, *W2 = b1 + H, *b2 = W2 + H * D;
  c->x = x;
  c->pre = falloc(M * H);
  c->act = falloc(M * H);
  vo_dense_forward(x, M, D, H, W1, b1, c->pre);
  vo_gelu_forward(c->pre, M * H, c->act);
  vo_dense_forward(c->act, M, H, D, W2, b2, y);
  if (mask) dropout_apply(y, mask, M * D);
}

/* FeedForward::backward, src/layers/FeedForward.cpp:29-35: dense2.backward receives the
 * ORIGINAL outputGradient (the dropout-masked gradient is computed on line 30 and discarded on
 * line 31), so the mask does not appear here at any dropout rate. */
static void ffn_backward(uint64_t D, uint64_t H, const float *prm, const ffn_cache *c, uint64_t M, const float *dy,
                         float *g, float *dx) {
  const float *W1 = prm, *W2 = W1 + D * H + H;
  float *g1 = g, *g2 = g + D * H + H;
  float *dact = falloc(M * H), *dpre = falloc(M * H);
  vo_dense_backward(c->act, dy, M, H, D, W2, g2, g2 + H * D, dact);
  vo_gelu_backward(c->pre, dact, M * H, dpre);
  vo_dense_backward(c->x, dpre, M, D, H, W1, g1, g1 + D * H, dx);
  free(dact); free(dpre);
}

void vo_ffn(uint64_t D, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N, const float *dy,
            float *y, float *grads, float *dx) {
  ffn_cache c;
  ffn_forward(D, H, params, x, B * N, NULL, y, &c);
  if (dy) ffn_backward(D, H, params, &c, B * N, dy, grads, dx);
  free(c.pre); free(c.act);
}

typedef struct {
  float *xhat1, *inv1, *n1, *xhat2, *inv2, *n2, *res1;
  mha_cache mha;
  ffn_cache ffn;
} block_cache;

static uint64_t block_params(uint64_t D, uint64_t H) { return 2 * D + 4 * (D * D + D) + 2 * D + (D * H + H) + (H * D + D); }

/* TransformerEncoderBlock::forward, src/model/TransformerEncoderBlock.cpp:14-35 (pre-norm).
 * masks == NULL: no dropout.  Otherwise 2 x {M,D}: the mask of attention_dropout (:23), then the
 * mask of the FeedForward's dropout (src/layers/FeedForward.cpp:22). */
static void block_forward(uint64_t D, uint64_t h, uint64_t H, const float *prm, const float *x, uint64_t B, uint64_t N,
                          const float *masks, float *y, block_cache *c) {
  const uint64_t M = B * N;
  const float *g1 = prm, *b1 = g1 + D, *mha = b1 + D, *g2 = mha + 4 * (D * D + D), *b2 = g2 + D, *ffn = b2 + D;
  c->xhat1 = falloc(M * D); c->inv1 = falloc(M); c->n1 = falloc(M * D);
  c->xhat2 = falloc(M * D); c->inv2 = falloc(M); c->n2 = falloc(M * D);
  c->res1 = falloc(M * D);
  float *t = falloc(M * D);
  vo_layernorm_forward(x, M, D, g1, b1, 1e-5f, c->n1, c->xhat1, c->inv1);
  mha_forward(D, h, mha, c->n1, B, N, t, &c->mha);
  if (masks) dropout_apply(t, masks, M * D);
  for (uint64_t i = 0; i < M * D; ++i) c->res1[i] = x[i] + t[i];
  vo_layernorm_forward(c->res1, M, D, g2, b2, 1e-5f, c->n2, c->xhat2, c->inv2);
  ffn_forward(D, H, ffn, c->n2, M, masks ? masks + M * D : NULL, t, &c->ffn);
  for (uint64_t i = 0; i < M * D; ++i) y[i] = c->res1[i] + t[i];
  free(t);
}

/* TransformerEncoderBlock::backward, src/model/TransformerEncoderBlock.cpp:40-64.  The attention
 * branch multiplies by its dropout mask (:58, Dropout::backward src/layers/Dropout.cpp:79-81); the
 * feed-forward branch does not (see ffn_backward). */
static void block_backward(uint64_t D, uint64_t h, uint64_t H, const float *prm, const block_cache *c, uint64_t B,
                           uint64_t N, const float *masks, const float *dy, float *g, float *dx) {
  const uint64_t M = B * N;
  const float *g1 = prm, *mha = g1 + 2 * D, *g2 = mha + 4 * (D * D + D), *ffn = g2 + 2 * D;
  float *dg1 = g, *dmha = dg1 + 2 * D, *dg2 = dmha + 4 * (D * D + D), *dffn = dg2 + 2 * D;
  float *t = falloc(M * D), *u = falloc(M * D), *grad1 = falloc(M * D);
  ffn_backward(D, H, ffn, &c->ffn, M, dy, dffn, t);
  vo_layernorm_backward(t, c->xhat2, c->inv2, g2, M, D, dg2, dg2 + D, u);
  for (uint64_t i = 0; i < M * D; ++i) grad1[i] = dy[i] + u[i];
  const float *gm = grad1;
  float *masked = NULL;
  if (masks) {
    masked = falloc(M * D);
    for (uint64_t i = 0; i < M * D; ++i) masked[i] = grad1[i] * masks[i];
    gm = masked;
  }
  mha_backward(D, h, mha, &c->mha, B, N, gm, dmha, t);
  free(masked);
  vo_layernorm_backward(t, c->xhat1, c->inv1, g1, M, D, dg1, dg1 + D, u);
  for (uint64_t i = 0; i < M * D; ++i) dx[i] = grad1[i] + u[i];
  free(t); free(u); free(grad1);
}

static void block_cache_free(block_cache *c) {
  free(c->xhat1); free(c->inv1); free(c->n1); free(c->xhat2); free(c->inv2); free(c->n2); free(c->res1);
  mha_cache_free(&c->mha);
  free(c->ffn.pre); free(c->ffn.act);
}

void vo_block_masked(uint64_t D, uint64_t h, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N,
                     const float *masks, const float *dy, float *y, float *grads, float *dx) {
  block_cache c;
  block_forward(D, h, H, params, x, B, N, masks, y, &c);
  if (dy) block_backward(D, h, H, params, &c, B, N, masks, dy, grads, dx);
  block_cache_free(&c);
}

void vo_block(uint64_t D, uint64_t h, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N,
              const float *dy, float *y, float *grads, float *dx) {
  vo_block_masked(D, h, H, params, x, B, N, NULL, dy, y, grads, dx);
}

/* PatchEmbedding::forward gather, src/layers/PatchEmbedding.cpp:31-52.  NOT an im2col: for
 * patch (ph,pw) of image b it copies patch_dim = P*P*C CONTIGUOUS floats starting at flat offset
 * b*C*HW*HW + ph*P*HW + pw*P (view offset; the view's strides are ignored by std::copy). */
void vo_patch_gather(const float *img, uint64_t B, uint64_t C, uint64_t HW, uint64_t P, float *patches) {
  const uint64_t np = HW / P, pd = P * P * C;
  uint64_t idx = 0;
  for (uint64_t b = 0; b < B; ++b)
    for (uint64_t ph = 0; ph < np; ++ph)
      for (uint64_t pw = 0; pw < np; ++pw) {
        memcpy(patches + idx * pd, img + b * C * HW * HW + ph * P * HW + pw * P, pd * sizeof(float));
        ++idx;
      }
}

/* Embeddings::forward, src/layers/Embeddings.cpp:24-40 and ::backward :42-74 (parameter
 * gradients only: the image gradient is discarded by Trainer, src/model/Trainer.cpp:212).
 * params/grads: W_pe{pd,D}, b_pe{1,D}, cls{1,1,D}, pos{1,N,D}. */
static void embeddings_forward(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *prm, const float *x,
                               uint64_t B, float *y, float **patches_out) {
  const uint64_t np = (img / P) * (img / P), N = np + 1, pd = P * P * C;
  const float *W = prm, *b = W + pd * D, *cls = b + D, *pos = cls + D;
  float *patches = falloc(B * np * pd), *proj = falloc(B * np * D);
  vo_patch_gather(x, B, C, img, P, patches);
  vo_dense_forward(patches, B * np, pd, D, W, b, proj);
  for (uint64_t bi = 0; bi < B; ++bi) {
    for (uint64_t d = 0; d < D; ++d) y[(bi * N) * D + d] = cls[d];
    memcpy(y + (bi * N + 1) * D, proj + bi * np * D, np * D * sizeof(float));
  }
  for (uint64_t bi = 0; bi < B; ++bi)
    for (uint64_t n = 0; n < N; ++n)
      for (uint64_t d = 0; d < D; ++d) y[(bi * N + n) * D + d] += pos[n * D + d];
  free(proj);
  if (patches_out) *patches_out = patches; else free(patches);
}

static void embeddings_backward(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *prm,
                                const float *patches, uint64_t B, const float *dy, float *g) {
  const uint64_t np = (img / P) * (img / P), N = np + 1, pd = P * P * C;
  const float *W = prm;
  float *dW = g, *db = dW + pd * D, *dcls = db + D, *dpos = dcls + D;
  /* dpos = dY.sum(0), dcls = dY[:,0:1].sum(0): sequential over batch (src/core/Tensor.cpp:410-424) */
  for (uint64_t n = 0; n < N; ++n)
    for (uint64_t d = 0; d < D; ++d) {
      float s = 0.0f;
      for (uint64_t bi = 0; bi < B; ++bi) s += dy[(bi * N + n) * D + d];
      dpos[n * D + d] = s;
    }
  for (uint64_t d = 0; d < D; ++d) {
    float s = 0.0f;
    for (uint64_t bi = 0; bi < B; ++bi) s += dy[(bi * N) * D + d];
    dcls[d] = s;
  }
  float *gp = falloc(B * np * D);
  for (uint64_t bi = 0; bi < B; ++bi) memcpy(gp + bi * np * D, dy + (bi * N + 1) * D, np * D * sizeof(float));
  vo_dense_backward(patches, gp, B * np, pd, D, W, dW, db, NULL);
  free(gp);
}

/* The image gradient Embeddings::backward returns (src/layers/Embeddings.cpp:57-73 ->
 * PatchEmbedding::backward, src/layers/PatchEmbedding.cpp:65-104): dpatches = dY[:,1:,:] W^T, then
 * the "un-patch" loop -- which takes every patch's row through getData(), a pointer that ignores
 * the slice offset (:86-87), so the (c, h, w)-ordered values written into every patch of every
 * image are those of row 0 (patch 0 of image 0). */
void vo_embeddings_input_grad(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *params, uint64_t B,
                              const float *dy, float *dimg) {
  const uint64_t nph = img / P, np = nph * nph, N = np + 1, pd = P * P * C;
  const float *W = params;
  float *gp = falloc(B * np * D), *dpatch = falloc(B * np * pd), *dW = falloc(pd * D), *db = falloc(D);
  float *zeros = falloc(B * np * pd);
  for (uint64_t bi = 0; bi < B; ++bi) memcpy(gp + bi * np * D, dy + (bi * N + 1) * D, np * D * sizeof(float));
  vo_dense_backward(zeros, gp, B * np, pd, D, W, dW, db, dpatch);
  for (uint64_t b = 0; b < B; ++b)
    for (uint64_t ph = 0; ph < nph; ++ph)
      for (uint64_t pw = 0; pw < nph; ++pw)
        for (uint64_t c = 0; c < C; ++c)
          for (uint64_t h = 0; h < P; ++h)
            for (uint64_t w = 0; w < P; ++w)
              dimg[((b * C + c) * img + ph * P + h) * img + pw * P + w] = dpatch[c * P * P + h * P + w];
  free(gp); free(dpatch); free(dW); free(db); free(zeros);
}

void vo_embeddings(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *params, const float *x, uint64_t B,
                   const float *dy, float *y, float *grads) {
  float *patches = NULL;
  embeddings_forward(img, P, C, D, params, x, B, y, &patches);
  if (dy) embeddings_backward(img, P, C, D, params, patches, B, dy, grads);
  free(patches);
}

/* CrossEntropy::calculate / backward, src/losses/CrossEntropy.cpp:55-111: loss is class-weighted
 * (-w*log(p+1e-12) where yTrue==1), mean over batch; gradient (softmax - y)/B is NOT weighted. */
float vo_cross_entropy(const float *logits, const float *y, uint64_t B, uint64_t K, const float *w, uint64_t n_w,
                       float *grad) {
  float *p = falloc(B * K);
  vo_softmax_rows(logits, B, K, p);
  float total = 0.0f;
  const float epsilon = 1e-12;
  for (uint64_t i = 0; i < B; ++i)
    for (uint64_t j = 0; j < K; ++j)
      if (y[i * K + j] == 1.0f) {
        float weight = n_w ? w[j] : 1.0f;
        total += -weight * logf(p[i * K + j] + epsilon);
      }
  if (grad)
    for (uint64_t i = 0; i < B; ++i)
      for (uint64_t j = 0; j < K; ++j) grad[i * K + j] = (p[i * K + j] - y[i * K + j]) / B;
  free(p);
  return total / B;
}

/* Adam::update inner loop, src/optimizers/Adam.cpp:47-60: coupled L2 (g += wd*p), eps outside
 * the sqrt; beta1_t/beta2_t = float(pow(beta, t)) computed by the caller (:32-33). */
void vo_adam_update(float *p, const float *g_in, float *m, float *v, uint64_t n, float lr, float beta1, float beta2,
                    float beta1_t, float beta2_t, float eps, float wd) {
#pragma omp parallel for
  for (int64_t j = 0; j < (int64_t)n; ++j) {
    float g = g_in[j];
    if (wd > 0.0f) g += wd * p[j];
    m[j] = beta1 * m[j] + (1.0f - beta1) * g;
    v[j] = beta2 * v[j] + (1.0f - beta2) * (g * g);
    float m_hat = m[j] / (1.0f - beta1_t);
    float v_hat = v[j] / (1.0f - beta2_t);
    p[j] -= lr * m_hat / (sqrtf(v_hat) + eps);
  }
}

/* cosine_warmup_lr, include/optimizers/Scheduler.hpp:5-16 (cosine evaluated in double). */
float vo_cosine_warmup_lr(long long step, long long warmup, long long total, float lr_max) {
  if (step < warmup) {
    long long w = warmup > 1 ? warmup : 1;
    return lr_max * (float)step / w;
  }
  long long den = (total - warmup) > 1 ? (total - warmup) : 1;
  float progress = (float)(step - warmup) / den;
  return lr_max * 0.5f * (1.0f + cos(M_PI * progress));
}

/* argmax with strict '>' from -inf: first index of the maximum (app/test.cpp:103-112,
 * src/model/Trainer.cpp:33-42). */
void vo_argmax_rows(const float *logits, uint64_t B, uint64_t K, int32_t *out) {
  for (uint64_t i = 0; i < B; ++i) {
    float best = -INFINITY;
    int32_t pred = -1;
    for (uint64_t j = 0; j < K; ++j)
      if (logits[i * K + j] > best) { best = logits[i * K + j]; pred = (int32_t)j; }
    out[i] = pred;
  }
}

/* ---------------------------------------------------------------------------- whole model
 * Parameter order = VisionTransformer::getParameters, src/model/VisionTransformer.cpp:74-87:
 * embeddings (W_pe, b_pe, cls, pos: src/layers/Embeddings.cpp:76-83), L blocks of 16 tensors
 * (src/model/TransformerEncoderBlock.cpp:69-80), final LN (gamma, beta), head (W, b). */
static uint64_t tokens(const vo_cfg *c) { return 1 + (c->image_size / c->patch_size) * (c->image_size / c->patch_size); }
static uint64_t patch_dim(const vo_cfg *c) { return c->patch_size * c->patch_size * c->in_channels; }
static uint64_t emb_params(const vo_cfg *c) {
  return patch_dim(c) * c->embedding_dim + c->embedding_dim + c->embedding_dim + tokens(c) * c->embedding_dim;
}

/* ---------------------------------------------------------------- data augmentation ---------
 * std::mt19937 (ISO C++ [rand.predef]: MT19937, 32-bit, seeded by the linear recurrence
 * 1812433253) and GCC 11 libstdc++'s generate_canonical<float, 24> (bits/random.tcc): one 32-bit
 * draw converted to float (round to nearest), divided by 2^32, clamped below 1;
 * uniform_real_distribution<float>(a, b) = canonical * (b - a) + a. */
typedef struct { uint32_t mt[624]; int idx; } vo_mt;
static void vo_mt_seed(vo_mt *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t vo_mt_next(vo_mt *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
static float vo_uniform(vo_mt *g, float a, float b) {
  float r = (float)vo_mt_next(g) / 4294967296.0f;
  if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
  return r * (b - a) + a;
}

/* one nearest-neighbour resampling stage over all channels: out(h, w) = in(sh, sw) or 0 */
static void vo_aug_rotate(const float *in, float *out, uint64_t C, uint64_t H, uint64_t W, float angle) {
  const float rad = (float)((double)angle * M_PI / 180.0f); /* :45, evaluated in double */
  const float ct = cosf(rad), st = sinf(rad);
  const float cx = W / 2.0f, cy = H / 2.0f;
  for (uint64_t h = 0; h < H; ++h)
    for (uint64_t w = 0; w < W; ++w) {
      const float x = w - cx, y = h - cy;
      const float sx = ct * x + st * y + cx;  /* :63-64 */
      const float sy = -st * x + ct * y + cy;
      const int ok = sx >= 0 && sx < W && sy >= 0 && sy < H;
      for (uint64_t c = 0; c < C; ++c)
        out[(c * H + h) * W + w] = ok ? in[(c * H + (uint64_t)sy) * W + (uint64_t)sx] : 0.0f;
    }
}
static void vo_aug_translate(const float *in, float *out, uint64_t C, uint64_t H, uint64_t W, float tx, float ty) {
  const int height = (int)H, width = (int)W;
  for (int h = 0; h < height; ++h)
    for (int w = 0; w < width; ++w) {
      const int sh = h - (int)ty, sw = w - (int)tx; /* :96-97 */
      const int ok = sh >= 0 && sh < height && sw >= 0 && sw < width;
      for (uint64_t c = 0; c < C; ++c)
        out[(c * H + (uint64_t)h) * W + (uint64_t)w] = ok ? in[(c * H + (uint64_t)sh) * W + (uint64_t)sw] : 0.0f;
    }
}
static void vo_aug_zoom(const float *in, float *out, uint64_t C, uint64_t H, uint64_t W, float zoom) {
  const float cx = W / 2.0f, cy = H / 2.0f;
  for (uint64_t h = 0; h < H; ++h)
    for (uint64_t w = 0; w < W; ++w) {
      const float x = (w - cx) / zoom + cx; /* :127-128 */
      const float y = (h - cy) / zoom + cy;
      const int ok = x >= 0 && x < W && y >= 0 && y < H;
      for (uint64_t c = 0; c < C; ++c)
        out[(c * H + h) * W + w] = ok ? in[(c * H + (uint64_t)y) * W + (uint64_t)x] : 0.0f;
    }
}

void vo_augment(const float *x, uint64_t B, uint64_t C, uint64_t H, uint64_t W, uint32_t seed, const float *cfg,
                float *out, float *draws) {
  const uint64_t sample = C * H * W;
  vo_mt g;
  vo_mt_seed(&g, seed);
  float *a = (float *)malloc(sample * sizeof(float)), *b = (float *)malloc(sample * sizeof(float));
  for (uint64_t i = 0; i < B; ++i) { /* :8-32 */
    float rec[6] = {0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    int flags = 0;
    memcpy(a, x + i * sample, sample * sizeof(float));
    if (vo_uniform(&g, 0.0f, 1.0f) < cfg[0]) {
      rec[1] = vo_uniform(&g, -cfg[3], cfg[3]);
      vo_aug_rotate(a, b, C, H, W, rec[1]);
      { float *t = a; a = b; b = t; }
      flags |= 1;
    }
    if (vo_uniform(&g, 0.0f, 1.0f) < cfg[1]) {
      rec[2] = vo_uniform(&g, -cfg[4], cfg[4]) * W; /* x first, then y (:87-88) */
      rec[3] = vo_uniform(&g, -cfg[4], cfg[4]) * H;
      vo_aug_translate(a, b, C, H, W, rec[2], rec[3]);
      { float *t = a; a = b; b = t; }
      flags |= 2;
    }
    if (vo_uniform(&g, 0.0f, 1.0f) < cfg[2]) {
      rec[4] = vo_uniform(&g, cfg[5], cfg[6]);
      vo_aug_zoom(a, b, C, H, W, rec[4]);
      { float *t = a; a = b; b = t; }
      flags |= 4;
    }
    rec[0] = (float)flags;
    memcpy(out + i * sample, a, sample * sizeof(float));
    if (draws) memcpy(draws + 6 * i, rec, sizeof rec);
  }
  free(a);
  free(b);
}


uint64_t vo_param_count(const vo_cfg *c) {
  return emb_params(c) + c->num_layers * block_params(c->embedding_dim, c->mlp_hidden_dim) + 2 * c->embedding_dim +
         c->embedding_dim * c->num_classes + c->num_classes;
}
uint64_t vo_param_tensors(const vo_cfg *c) { return 4 + 16 * c->num_layers + 4; }

uint64_t vo_param_shape(const vo_cfg *c, uint64_t idx, uint64_t *dims) {
  const uint64_t D = c->embedding_dim, H = c->mlp_hidden_dim, L = c->num_layers;
  dims[0] = dims[1] = dims[2] = dims[3] = 0;
  if (idx == 0) { dims[0] = patch_dim(c); dims[1] = D; return 2; }
  if (idx == 1) { dims[0] = 1; dims[1] = D; return 2; }
  if (idx == 2) { dims[0] = 1; dims[1] = 1; dims[2] = D; return 3; }
  if (idx == 3) { dims[0] = 1; dims[1] = tokens(c); dims[2] = D; return 3; }
  idx -= 4;
  if (idx < 16 * L) {
    uint64_t k = idx % 16;
    dims[0] = 1; dims[1] = D;
    switch (k) {
      case 2: case 4: case 6: case 8: dims[0] = D; dims[1] = D; break;
      case 12: dims[0] = D; dims[1] = H; break;
      case 13: dims[0] = 1; dims[1] = H; break;
      case 14: dims[0] = H; dims[1] = D; break;
      default: break;
    }
    return 2;
  }
  idx -= 16 * L;
  dims[0] = 1; dims[1] = D;
  if (idx == 2) { dims[0] = D; dims[1] = c->num_classes; }
  if (idx == 3) { dims[0] = 1; dims[1] = c->num_classes; }
  return 2;
}

typedef struct {
  float *patches, *emb;          /* emb = embeddings output {B,N,D} */
  float **blk_out;               /* L outputs */
  block_cache *blk;
  float *fxhat, *finv, *fnorm, *cls_rows;
} model_cache;

/* VisionTransformer::forward, src/model/VisionTransformer.cpp:20-38. */
static void model_forward(const vo_cfg *c, const float *prm, const float *x, uint64_t B, const float *masks,
                          float *logits, model_cache *mc) {
  const uint64_t D = c->embedding_dim, H = c->mlp_hidden_dim, L = c->num_layers, N = tokens(c), M = B * N,
                 K = c->num_classes;
  const float *emb = prm, *blk0 = emb + emb_params(c), *fn = blk0 + L * block_params(D, H), *head = fn + 2 * D;
  mc->emb = falloc(M * D);
  embeddings_forward(c->image_size, c->patch_size, c->in_channels, D, emb, x, B, mc->emb, &mc->patches);
  mc->blk_out = (float **)calloc(L ? L : 1, sizeof(float *));
  mc->blk = (block_cache *)calloc(L ? L : 1, sizeof(block_cache));
  const float *cur = mc->emb;
  for (uint64_t l = 0; l < L; ++l) {
    mc->blk_out[l] = falloc(M * D);
    block_forward(D, c->num_heads, H, blk0 + l * block_params(D, H), cur, B, N,
                  masks ? masks + l * 2 * M * D : NULL, mc->blk_out[l], &mc->blk[l]);
    cur = mc->blk_out[l];
  }
  mc->fxhat = falloc(M * D); mc->finv = falloc(M); mc->fnorm = falloc(M * D);
  vo_layernorm_forward(cur, M, D, fn, fn + D, 1e-5f, mc->fnorm, mc->fxhat, mc->finv);
  mc->cls_rows = falloc(B * D);
  for (uint64_t b = 0; b < B; ++b) memcpy(mc->cls_rows + b * D, mc->fnorm + b * N * D, D * sizeof(float));
  vo_dense_forward(mc->cls_rows, B, D, K, head, head + D * K, logits);
}

/* VisionTransformer::backward, src/model/VisionTransformer.cpp:43-69. */
static void model_backward(const vo_cfg *c, const float *prm, const model_cache *mc, uint64_t B, const float *masks,
                           const float *dlogits, float *g, float *dacts) {
  const uint64_t D = c->embedding_dim, H = c->mlp_hidden_dim, L = c->num_layers, N = tokens(c), M = B * N,
                 K = c->num_classes, bp = block_params(D, H);
  const float *blk0 = prm + emb_params(c), *fn = blk0 + L * bp, *head = fn + 2 * D;
  float *gemb = g, *gblk0 = g + emb_params(c), *gfn = gblk0 + L * bp, *ghead = gfn + 2 * D;
  float *dcls = falloc(B * D);
  vo_dense_backward(mc->cls_rows, dlogits, B, D, K, head, ghead, ghead + D * K, dcls);
  float *grad_seq = falloc(M * D), *grad = falloc(M * D), *next = falloc(M * D);
  for (uint64_t b = 0; b < B; ++b) memcpy(grad_seq + b * N * D, dcls + b * D, D * sizeof(float));
  vo_layernorm_backward(grad_seq, mc->fxhat, mc->finv, fn, M, D, gfn, gfn + D, grad);
  if (dacts) { memcpy(dacts, grad, M * D * sizeof(float)); dacts += M * D; }
  for (int64_t l = (int64_t)L - 1; l >= 0; --l) {
    block_backward(D, c->num_heads, H, blk0 + l * bp, &mc->blk[l], B, N, masks ? masks + l * 2 * M * D : NULL, grad,
                   gblk0 + l * bp, next);
    float *t = grad; grad = next; next = t;
    if (dacts) { memcpy(dacts, grad, M * D * sizeof(float)); dacts += M * D; }
  }
  embeddings_backward(c->image_size, c->patch_size, c->in_channels, D, prm, mc->patches, B, grad, gemb);
  free(dcls); free(grad_seq); free(grad); free(next);
}

static void model_cache_free(const vo_cfg *c, model_cache *mc) {
  free(mc->patches); free(mc->emb);
  for (uint64_t l = 0; l < c->num_layers; ++l) { free(mc->blk_out[l]); block_cache_free(&mc->blk[l]); }
  free(mc->blk_out); free(mc->blk);
  free(mc->fxhat); free(mc->finv); free(mc->fnorm); free(mc->cls_rows);
}

static void copy_acts(const vo_cfg *c, const model_cache *mc, uint64_t B, float *acts) {
  const uint64_t n = B * tokens(c) * c->embedding_dim;
  memcpy(acts, mc->emb, n * sizeof(float)); acts += n;
  for (uint64_t l = 0; l < c->num_layers; ++l) { memcpy(acts, mc->blk_out[l], n * sizeof(float)); acts += n; }
  memcpy(acts, mc->fnorm, n * sizeof(float));
}

void vo_forward(const vo_cfg *c, const float *params, const float *x, uint64_t B, float *logits, float *acts) {
  model_cache mc;
  model_forward(c, params, x, B, NULL, logits, &mc);
  if (acts) copy_acts(c, &mc, B, acts);
  model_cache_free(c, &mc);
}

float vo_forward_backward_masked(const vo_cfg *c, const float *params, const float *x, const float *y, uint64_t B,
                                 const float *class_w, uint64_t n_w, const float *masks, float *logits, float *grads,
                                 float *acts, float *dacts) {
  model_cache mc;
  const uint64_t K = c->num_classes;
  float *lg = logits ? logits : falloc(B * K);
  float *dl = falloc(B * K);
  model_forward(c, params, x, B, masks, lg, &mc);
  if (acts) copy_acts(c, &mc, B, acts);
  float loss = vo_cross_entropy(lg, y, B, K, class_w, n_w, dl);
  model_backward(c, params, &mc, B, masks, dl, grads, dacts);
  model_cache_free(c, &mc);
  free(dl);
  if (!logits) free(lg);
  return loss;
}

float vo_forward_backward(const vo_cfg *c, const float *params, const float *x, const float *y, uint64_t B,
                          const float *class_w, uint64_t n_w, float *logits, float *grads, float *dacts) {
  return vo_forward_backward_masked(c, params, x, y, B, class_w, n_w, NULL, logits, grads, NULL, dacts);
}

typedef struct {
  vo_cfg cfg;
  uint64_t n;
  float *p, *g, *m, *v, *cw;
  uint64_t n_cw;
  long long t;
  float beta1, beta2, eps, wd;
} session;

void *vo_session_create(const vo_cfg *c, float beta1, float beta2, float eps, float wd) {
  session *s = (session *)calloc(1, sizeof(session));
  s->cfg = *c;
  s->n = vo_param_count(c);
  s->p = falloc(s->n); s->g = falloc(s->n); s->m = falloc(s->n); s->v = falloc(s->n);
  s->beta1 = beta1; s->beta2 = beta2; s->eps = eps; s->wd = wd;
  return s;
}
void vo_session_destroy(void *h) {
  session *s = (session *)h;
  free(s->p); free(s->g); free(s->m); free(s->v); free(s->cw); free(s);
}
void vo_session_set_params(void *h, const float *flat) { session *s = (session *)h; memcpy(s->p, flat, s->n * sizeof(float)); }
void vo_session_get_params(void *h, float *flat) { session *s = (session *)h; memcpy(flat, s->p, s->n * sizeof(float)); }
void vo_session_set_class_weights(void *h, const float *w, uint64_t n) {
  session *s = (session *)h;
  free(s->cw);
  s->cw = falloc(n);
  memcpy(s->cw, w, n * sizeof(float));
  s->n_cw = n;
}

/* Body of Trainer::train_epoch after batching (src/model/Trainer.cpp:201-227) with Adam::update
 * (src/optimizers/Adam.cpp:29-33: t++, beta^t via double pow cast to float). */
float vo_session_train_step_masked(void *h, const float *x, const float *y, uint64_t B, float lr_now,
                                   const float *masks, float *logits_out, float *grads_out);
float vo_session_train_step(void *h, const float *x, const float *y, uint64_t B, float lr_now, float *logits_out,
                            float *grads_out) {
  return vo_session_train_step_masked(h, x, y, B, lr_now, NULL, logits_out, grads_out);
}

float vo_session_train_step_masked(void *h, const float *x, const float *y, uint64_t B, float lr_now,
                                   const float *masks, float *logits_out, float *grads_out) {
  session *s = (session *)h;
  float loss = vo_forward_backward_masked(&s->cfg, s->p, x, y, B, s->cw, s->n_cw, masks, logits_out, s->g, NULL, NULL);
  if (grads_out) memcpy(grads_out, s->g, s->n * sizeof(float));
  s->t++;
  const float b1t = (float)pow((double)s->beta1, (double)s->t);
  const float b2t = (float)pow((double)s->beta2, (double)s->t);
  vo_adam_update(s->p, s->g, s->m, s->v, s->n, lr_now, s->beta1, s->beta2, b1t, b2t, s->eps, s->wd);
  return loss;
}
