/* TEST INFRASTRUCTURE ONLY.  Drives the C restatement (vit_oracle.c) through every entry point a
 * parity test uses -- a whole model forward / backward / Adam with and without dropout masks, the
 * single layers, the augmentation -- under AddressSanitizer and UndefinedBehaviorSanitizer (host
 * code only; `make -C oracle sanitize`).  Any report makes the run fail. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vit_oracle.h"

static uint32_t rng_state = 12345u;
static float rnd(void) { /* xorshift: uniform in [-1, 1) */
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 17;
  rng_state ^= rng_state << 5;
  return (float)(rng_state >> 8) / 8388608.0f - 1.0f;
}
static float *filled(size_t n, float scale) {
  float *p = (float *)malloc((n ? n : 1) * sizeof(float));
  for (size_t i = 0; i < n; ++i) p[i] = rnd() * scale;
  return p;
}
static int finite_all(const float *p, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(p[i])) return 0;
  return 1;
}

int main(void) {
  int bad = 0;
  const vo_cfg cfgs[2] = {{28, 14, 1, 10, 32, 4, 2, 64, 0.0f}, {28, 7, 3, 8, 24, 3, 2, 48, 0.25f}};
  for (int ci = 0; ci < 2; ++ci) {
    const vo_cfg *c = &cfgs[ci];
    const uint64_t B = 3, n = vo_param_count(c), np = (c->image_size / c->patch_size) * (c->image_size / c->patch_size);
    const uint64_t N = np + 1, D = c->embedding_dim, K = c->num_classes, L = c->num_layers;
    float *prm = filled(n, 0.05f), *x = filled(B * c->in_channels * c->image_size * c->image_size, 1.0f);
    float *y = (float *)calloc(B * K, sizeof(float));
    for (uint64_t b = 0; b < B; ++b) y[b * K + b % K] = 1.0f;
    float *logits = filled(B * K, 0.f), *grads = filled(n, 0.f), *acts = filled((L + 2) * B * N * D, 0.f);
    float *dacts = filled((L + 1) * B * N * D, 0.f), *w = filled(K, 0.f);
    for (uint64_t k = 0; k < K; ++k) w[k] = 1.0f / (float)K;
    vo_forward(c, prm, x, B, logits, acts);
    float loss = vo_forward_backward(c, prm, x, y, B, w, K, logits, grads, dacts);
    bad |= !isfinite(loss) || !finite_all(grads, n) || !finite_all(acts, (L + 2) * B * N * D);
    float *masks = filled(L * 2 * B * N * D, 0.f);
    for (uint64_t i = 0; i < L * 2 * B * N * D; ++i) masks[i] = (i % 4 == 0) ? 0.0f : 1.0f / 0.75f;
    loss = vo_forward_backward_masked(c, prm, x, y, B, NULL, 0, masks, logits, grads, acts, dacts);
    bad |= !isfinite(loss) || !finite_all(grads, n);
    void *s = vo_session_create(c, 0.9f, 0.999f, 1e-8f, 1e-4f);
    vo_session_set_params(s, prm);
    vo_session_set_class_weights(s, w, K);
    for (int step = 0; step < 2; ++step) {
      loss = vo_session_train_step(s, x, y, B, vo_cosine_warmup_lr(step + 1, 1, 4, 1e-3f), logits, grads);
      loss += vo_session_train_step_masked(s, x, y, B, 1e-3f, masks, logits, NULL);
    }
    vo_session_get_params(s, prm);
    bad |= !isfinite(loss) || !finite_all(prm, n);
    vo_session_destroy(s);
    int32_t top[3];
    vo_argmax_rows(logits, B, K, top);
    uint64_t dims[4];
    for (uint64_t t = 0; t < vo_param_tensors(c); ++t) bad |= vo_param_shape(c, t, dims) == 0;
    const float aug[7] = {0.5f, 0.5f, 0.5f, 10.0f, 0.1f, 0.9f, 1.1f};
    float *xo = filled(B * c->in_channels * c->image_size * c->image_size, 0.f), draws[18];
    vo_augment(x, B, c->in_channels, c->image_size, c->image_size, 7u + (uint32_t)ci, aug, xo, draws);
    bad |= !finite_all(xo, B * c->in_channels * c->image_size * c->image_size);
    free(prm); free(x); free(y); free(logits); free(grads); free(acts); free(dacts); free(w); free(masks); free(xo);
  }
  { /* single layers at ragged sizes */
    const uint64_t rows = 7, in = 13, out = 5;
    float *x = filled(rows * in, 1.f), *W = filled(in * out, 0.3f), *b = filled(out, 0.1f), *yv = filled(rows * out, 0.f);
    float *dW = filled(in * out, 0.f), *db = filled(out, 0.f), *dx = filled(rows * in, 0.f), *dy = filled(rows * out, 1.f);
    vo_dense_forward(x, rows, in, out, W, b, yv);
    vo_dense_backward(x, dy, rows, in, out, W, dW, db, dx);
    float *p = filled(rows * in, 0.f), *g = filled(rows * in, 1.f), *ds = filled(rows * in, 0.f);
    vo_softmax_rows(x, rows, in, p);
    vo_softmax_rows_backward(g, p, rows, in, ds);
    vo_gelu_forward(x, rows * in, p);
    vo_gelu_backward(x, g, rows * in, ds);
    float *xhat = filled(rows * in, 0.f), *inv = filled(rows, 0.f), *gam = filled(in, 1.f), *bet = filled(in, 1.f);
    vo_layernorm_forward(x, rows, in, gam, bet, 1e-5f, p, xhat, inv);
    float *dg = filled(in, 0.f), *dbeta = filled(in, 0.f);
    vo_layernorm_backward(g, xhat, inv, gam, rows, in, dg, dbeta, ds);
    bad |= !finite_all(ds, rows * in) || !finite_all(dW, in * out);
    free(x); free(W); free(b); free(yv); free(dW); free(db); free(dx); free(dy); free(p); free(g); free(ds);
    free(xhat); free(inv); free(gam); free(bet); free(dg); free(dbeta);
  }
  if (bad) {
    fprintf(stderr, "sanitize_main: a result was not finite\n");
    return 1;
  }
  printf("oracle under -fsanitize=address,undefined: clean\n");
  return 0;
}
