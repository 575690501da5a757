/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's ViT training step.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported baseline.  The product path (libvitk / libvit_host)
 * never links, loads or calls it.
 *
 * Parity status: PINNED.  Every function is checked (tests/test_oracle_*.py) against
 * (a) oracle/_ref/libvit_ref.so -- the reference's own sources compiled in place -- when that
 * library is present, and (b) the golden vectors under tests/golden/ that were generated from it.
 *
 * All tensors are contiguous row-major float32 host buffers.  Loop nests and accumulation
 * order follow the reference file:line cited at each function in vit_oracle.c so results are
 * reproducible to the last bit wherever the reference itself is deterministic.
 */
#ifndef VIT_ORACLE_H
#define VIT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t image_size, patch_size, in_channels, num_classes, embedding_dim, num_heads, num_layers, mlp_hidden_dim;
  float dropout_rate; /* informational: the oracle never draws dropout masks (the reference seeds them from
                       * random_device); dropout is applied only by the *_masked entry points, from given masks */
} vo_cfg;

/* ---- free functions (src/core/Tensor.cpp) ---- */
void vo_matmul(const float *a, const float *b, uint64_t m, uint64_t n, uint64_t p, float *out);
void vo_bmm(const float *a, const float *b, uint64_t bs, uint64_t m, uint64_t n, uint64_t p, float *out);

/* ---- layers ---- */
void vo_dense_forward(const float *x, uint64_t rows, uint64_t in, uint64_t out, const float *W, const float *b,
                      float *y);
void vo_dense_backward(const float *x, const float *dy, uint64_t rows, uint64_t in, uint64_t out, const float *W,
                       float *dW, float *db, float *dx);
void vo_layernorm_forward(const float *x, uint64_t rows, uint64_t D, const float *gamma, const float *beta, float eps,
                          float *y, float *xhat, float *inv_std);
void vo_layernorm_backward(const float *dy, const float *xhat, const float *inv_std, const float *gamma,
                           uint64_t rows, uint64_t D, float *dgamma, float *dbeta, float *dx);
void vo_gelu_forward(const float *x, uint64_t n, float *y);
void vo_gelu_backward(const float *x, const float *dy, uint64_t n, float *dx);
void vo_softmax_rows(const float *x, uint64_t rows, uint64_t n, float *p);
void vo_softmax_rows_backward(const float *g, const float *p, uint64_t rows, uint64_t n, float *dx);
/* params: q W,b, k W,b, v W,b, out W,b (flat, getParameters() order).  grads/dx may be NULL with dy NULL. */
void vo_mha(uint64_t D, uint64_t h, const float *params, const float *x, uint64_t B, uint64_t N, const float *dy,
            float *y, float *grads, float *dx);
void vo_ffn(uint64_t D, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N, const float *dy,
            float *y, float *grads, float *dx);
void vo_block(uint64_t D, uint64_t h, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N,
              const float *dy, float *y, float *grads, float *dx);
/* vo_block with dropout at a given draw: masks = 2 x {B*N,D} holding 0 or 1/(1-rate) -- the mask of
 * the block's attention_dropout (src/model/TransformerEncoderBlock.cpp:23), then the mask of its
 * FeedForward's dropout (src/layers/FeedForward.cpp:22).  The backward applies the first mask on the
 * attention branch (:58) and none on the feed-forward branch (src/layers/FeedForward.cpp:30-31).
 * masks == NULL is vo_block. */
void vo_block_masked(uint64_t D, uint64_t h, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N,
                     const float *masks, const float *dy, float *y, float *grads, float *dx);
void vo_patch_gather(const float *img, uint64_t B, uint64_t C, uint64_t HW, uint64_t P, float *patches);
void vo_embeddings(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *params, const float *x, uint64_t B,
                   const float *dy, float *y, float *grads);

/* image gradient {B,C,img,img} that Embeddings::backward returns (row 0 of the patch gradients
 * un-patched into every patch: src/layers/PatchEmbedding.cpp:86-94) */
void vo_embeddings_input_grad(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *params, uint64_t B,
                              const float *dy, float *dimg);

/* ---- loss / optimiser / schedule ---- */
float vo_cross_entropy(const float *logits, const float *y, uint64_t B, uint64_t K, const float *w, uint64_t n_w,
                       float *grad);
void vo_adam_update(float *p, const float *g, float *m, float *v, uint64_t n, float lr, float beta1, float beta2,
                    float beta1_t, float beta2_t, float eps, float wd);
float vo_cosine_warmup_lr(long long step, long long warmup, long long total, float lr_max);
void vo_argmax_rows(const float *logits, uint64_t B, uint64_t K, int32_t *out);

/* ---- data augmentation (src/utils/DataAugmentation.cpp:4-143) ----
 * DataAugmentation(cfg).apply(x) with the engine seeded by `seed` instead of random_device.
 * cfg = {rotation_prob, translate_prob, zoom_prob, rotation_factor, translate_factor, zoom_min,
 * zoom_max} (include/utils/DataAugmentation.hpp:11-19).  x, out are [B][C][H][W].  draws
 * (optional) receives 6 values per image: flags (1 rotate, 2 translate, 4 zoom), angle in degrees,
 * tx, ty (before truncation), zoom, 0.  The random numbers restate std::mt19937 and GCC 11
 * libstdc++'s uniform_real_distribution<float> / generate_canonical<float, 24> (one 32-bit draw
 * per number); the image stages are applied one after another as the reference does. */
void vo_augment(const float *x, uint64_t B, uint64_t C, uint64_t H, uint64_t W, uint32_t seed, const float *cfg,
                float *out, float *draws);

/* ---- whole model ---- */
uint64_t vo_param_count(const vo_cfg *c);
uint64_t vo_param_tensors(const vo_cfg *c);
/* shape of parameter tensor idx in getParameters() order; dims holds 4 entries; returns rank */
uint64_t vo_param_shape(const vo_cfg *c, uint64_t idx, uint64_t *dims);

/* forward only (eval or train, identical arithmetic at dropout 0).  acts (optional) receives
 * (L+2) x {B,N,D}: embeddings, each block output, final norm output. */
void vo_forward(const vo_cfg *c, const float *params, const float *x, uint64_t B, float *logits, float *acts);
/* forward + CE + backward: loss returned, grads (flat, getGradients() order) written.
 * dacts (optional) receives (L+1) x {B,N,D}: d(final-norm input), d(block L-1 input) ... d(block 0 input). */
float vo_forward_backward(const vo_cfg *c, const float *params, const float *x, const float *y, uint64_t B,
                          const float *class_w, uint64_t n_w, float *logits, float *grads, float *dacts);

/* The same with dropout at a given draw: masks = L x 2 x {B*N,D} (per block as vo_block_masked), or
 * NULL.  acts (optional) as in vo_forward. */
float vo_forward_backward_masked(const vo_cfg *c, const float *params, const float *x, const float *y, uint64_t B,
                                 const float *class_w, uint64_t n_w, const float *masks, float *logits, float *grads,
                                 float *acts, float *dacts);

/* session = params + Adam state + step counter, to run the Trainer's per-batch body */
void *vo_session_create(const vo_cfg *c, float beta1, float beta2, float eps, float wd);
void vo_session_destroy(void *s);
void vo_session_set_params(void *s, const float *flat);
void vo_session_get_params(void *s, float *flat);
void vo_session_set_class_weights(void *s, const float *w, uint64_t n);
float vo_session_train_step(void *s, const float *x, const float *y, uint64_t B, float lr_now, float *logits_out,
                            float *grads_out);
float vo_session_train_step_masked(void *s, const float *x, const float *y, uint64_t B, float lr_now,
                                   const float *masks, float *logits_out, float *grads_out);

#ifdef __cplusplus
}
#endif
#endif
