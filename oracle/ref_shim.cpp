// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// C-ABI shim over the *reference's own* CPU classes (compiled in place from
// /root/reference/src by oracle/Makefile into oracle/_ref/libvit_ref.so).
// It exists to (1) pin oracle/vit_oracle.c against the real reference and
// (2) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py)
// and (3) serve as the "reference" CPU baseline in bench.py.
//
// Nothing here restates reference arithmetic: every number comes out of the
// reference's Tensor / Layer / Loss / Optimizer objects.  The private-member
// access below is only used to read intermediate activations out of
// VisionTransformer (include/model/VisionTransformer.hpp:108-121) and the masks
// its Dropout layers drew (include/layers/Dropout.hpp:60).
#include <cstdint>
#include <cstring>
#include <random>
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>
#include <sstream>
#include <iostream>

// Pull in everything third-party / standard first so the access hack below only
// touches the reference's own class definitions.
#include <memory>
#include <cmath>
#include "utils/json.hpp"
#include "core/Tensor.hpp"
#define private public
#include "model/VisionTransformer.hpp"
#undef private
#define private public
#include "utils/DataAugmentation.hpp"  // private: the engine, to seed it (the class seeds it from random_device)
#undef private
#include "losses/CrossEntropy.hpp"
#include "optimizers/Adam.hpp"
#include "optimizers/Scheduler.hpp"
#include "utils/ModelUtils.hpp"
#include "layers/MultiHeadAttention.hpp"
#include "layers/FeedForward.hpp"
#include "activations/GELU.hpp"

#define TENSOR_SURFACE_FN vref_tensor_scenario
#include "../tests/tensor_surface.inc"
#include "utils/DataReader.hpp"
#define DATA_READER_FN vref_data_reader
#define CLASS_WEIGHTS_FN vref_class_weights
#include "../tests/data_reader.inc"

Tensor softmax(const Tensor &logits, int axis);
Tensor softmax_backward(const Tensor &grad_output, const Tensor &softmax_output);

namespace {

struct CoutSilencer {
  std::streambuf *old;
  std::ostringstream sink;
  CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~CoutSilencer() { std::cout.rdbuf(old); }
};

Tensor make(const std::vector<size_t> &shape, const float *src) {
  Tensor t(shape);
  if (src) std::memcpy(t.getData(), src, t.getSize() * sizeof(float));
  return t;
}

void dump(const Tensor &t, float *dst) {
  if (!dst) return;
  if (t.isContiguous()) {
    std::memcpy(dst, t.getData() + t.getDataOffset(), t.getSize() * sizeof(float));
  } else {
    Tensor c = t.contiguous();
    std::memcpy(dst, c.getData() + c.getDataOffset(), c.getSize() * sizeof(float));
  }
}

size_t total(const std::vector<Tensor *> &v) {
  size_t n = 0;
  for (auto *t : v) n += t->getSize();
  return n;
}

void set_flat(const std::vector<Tensor *> &v, const float *src) {
  for (auto *t : v) {
    std::memcpy(t->getData() + t->getDataOffset(), src, t->getSize() * sizeof(float));
    src += t->getSize();
  }
}

void get_flat(const std::vector<Tensor *> &v, float *dst) {
  for (auto *t : v) {
    dump(*t, dst);
    dst += t->getSize();
  }
}

struct Session {
  ViTConfig cfg;
  VisionTransformer model;
  CrossEntropy loss;
  Adam adam;
  Session(const ViTConfig &c, float lr, float b1, float b2, float eps, float wd)
      : cfg(c), model(c), adam(lr, b1, b2, eps, wd) {}
};

} // namespace

extern "C" {

struct vref_cfg {
  uint64_t image_size, patch_size, in_channels, num_classes, embedding_dim, num_heads, num_layers, mlp_hidden_dim;
  float dropout_rate;
};

static ViTConfig to_cfg(const vref_cfg *c) {
  ViTConfig k;
  k.image_size = c->image_size;
  k.patch_size = c->patch_size;
  k.in_channels = c->in_channels;
  k.num_classes = c->num_classes;
  k.embedding_dim = c->embedding_dim;
  k.num_heads = c->num_heads;
  k.num_layers = c->num_layers;
  k.mlp_hidden_dim = c->mlp_hidden_dim;
  k.dropout_rate = c->dropout_rate;
  return k;
}

// ---------------------------------------------------------------- whole model
void *vref_create(const vref_cfg *c, float lr, float b1, float b2, float eps, float wd) {
  return new Session(to_cfg(c), lr, b1, b2, eps, wd);
}
void vref_destroy(void *s) { delete static_cast<Session *>(s); }

uint64_t vref_param_count(void *s) { return total(static_cast<Session *>(s)->model.getParameters()); }
uint64_t vref_param_tensors(void *s) { return static_cast<Session *>(s)->model.getParameters().size(); }
// dims must hold 4 entries; returns the rank.
uint64_t vref_param_shape(void *s, uint64_t idx, uint64_t *dims) {
  auto p = static_cast<Session *>(s)->model.getParameters();
  const auto &sh = p[idx]->getShape();
  for (size_t i = 0; i < sh.size() && i < 4; ++i) dims[i] = sh[i];
  return sh.size();
}
void vref_set_params(void *s, const float *flat) { set_flat(static_cast<Session *>(s)->model.getParameters(), flat); }
void vref_get_params(void *s, float *flat) { get_flat(static_cast<Session *>(s)->model.getParameters(), flat); }
void vref_get_grads(void *s, float *flat) { get_flat(static_cast<Session *>(s)->model.getGradients(), flat); }
void vref_set_class_weights(void *s, const float *w, uint64_t n) {
  static_cast<Session *>(s)->loss.setClassWeights(std::vector<float>(w, w + n));
}

int vref_save_weights(void *s, const char *path) {
  CoutSilencer q;
  try {
    ModelUtils::save_weights(static_cast<Session *>(s)->model, path);
    return 0;
  } catch (const std::exception &) { return 1; }
}
int vref_load_weights(void *s, const char *path) {
  CoutSilencer q;
  try {
    ModelUtils::load_weights(static_cast<Session *>(s)->model, path);
    return 0;
  } catch (const std::exception &) { return 1; }
}
int vref_save_config(const vref_cfg *c, const char *path) {
  CoutSilencer q;
  try {
    ModelUtils::save_config(to_cfg(c), path);
    return 0;
  } catch (const std::exception &) { return 1; }
}
int vref_load_config(const char *path, vref_cfg *c) {
  CoutSilencer q;
  try {
    ViTConfig k = ModelUtils::load_config(path);
    c->image_size = k.image_size; c->patch_size = k.patch_size; c->in_channels = k.in_channels;
    c->num_classes = k.num_classes; c->embedding_dim = k.embedding_dim; c->num_heads = k.num_heads;
    c->num_layers = k.num_layers; c->mlp_hidden_dim = k.mlp_hidden_dim; c->dropout_rate = k.dropout_rate;
    return 0;
  } catch (const std::exception &) { return 1; }
}

// forward; x is {B,C,H,W}; logits {B,K}.
void vref_forward(void *s, const float *x, uint64_t B, int training, float *logits) {
  auto *S = static_cast<Session *>(s);
  Tensor in = make({B, S->cfg.in_channels, S->cfg.image_size, S->cfg.image_size}, x);
  Tensor out = S->model.forward(in, training != 0);
  dump(out, logits);
}

// forward in training mode through the model's own sub-layers, recording the
// token stream after embeddings, after each block and after the final norm:
// acts = (L+2) x {B,N,D}.  Same call sequence as VisionTransformer::forward
// (src/model/VisionTransformer.cpp:20-38).
void vref_forward_trace(void *s, const float *x, uint64_t B, float *acts, float *logits) {
  auto *S = static_cast<Session *>(s);
  auto &m = S->model;
  Tensor in = make({B, S->cfg.in_channels, S->cfg.image_size, S->cfg.image_size}, x);
  Tensor t = m.embeddings.forward(in, true);
  size_t n = t.getSize();
  dump(t, acts); acts += n;
  for (auto &blk : m.encoder_blocks) {
    t = blk.forward(t, true);
    dump(t, acts); acts += n;
  }
  t = m.final_norm.forward(t, true);
  m.final_norm_output = t;
  dump(t, acts);
  Tensor cls = t.slice(1, 0, 1).contiguous().reshape({(size_t)B, S->cfg.embedding_dim});
  Tensor out = m.mlp_head.forward(cls, true);
  dump(out, logits);
}

// backward through the model's own sub-layers (same sequence as
// VisionTransformer::backward, src/model/VisionTransformer.cpp:43-69),
// recording the gradient w.r.t. each block's input:
// dacts = (L+1) x {B,N,D}: [d(final-norm input), d(block L-1 input), ..., d(block 0 input)].
void vref_backward_trace(void *s, const float *dlogits, uint64_t B, float *dacts) {
  auto *S = static_cast<Session *>(s);
  auto &m = S->model;
  Tensor g = make({(size_t)B, S->cfg.num_classes}, dlogits);
  Tensor grad = m.mlp_head.backward(g);
  Tensor grad_seq({(size_t)B, m.num_tokens, S->cfg.embedding_dim});
  grad_seq.fill(0.0f);
  for (size_t b = 0; b < B; ++b)
    for (size_t d = 0; d < S->cfg.embedding_dim; ++d) grad_seq(b, 0, d) = grad(b, d);
  grad = m.final_norm.backward(grad_seq);
  size_t n = grad.getSize();
  dump(grad, dacts); dacts += n;
  for (int i = (int)m.encoder_blocks.size() - 1; i >= 0; --i) {
    grad = m.encoder_blocks[i].backward(grad);
    dump(grad, dacts); dacts += n;
  }
  m.embeddings.backward(grad);
}

float vref_loss(void *s, const float *logits, const float *y, uint64_t B) {
  auto *S = static_cast<Session *>(s);
  Tensor l = make({(size_t)B, S->cfg.num_classes}, logits);
  Tensor t = make({(size_t)B, S->cfg.num_classes}, y);
  return S->loss.calculate(l, t);
}

// One optimisation step exactly as Trainer::train_epoch's body does it after
// batching/augmentation (src/model/Trainer.cpp:201-226): forward(train) ->
// calculate -> backward -> model.backward -> setLearningRate -> update.
float vref_train_step(void *s, const float *x, const float *y, uint64_t B, float lr_now, float *logits_out,
                      float *grads_out) {
  auto *S = static_cast<Session *>(s);
  Tensor in = make({(size_t)B, S->cfg.in_channels, S->cfg.image_size, S->cfg.image_size}, x);
  Tensor t = make({(size_t)B, S->cfg.num_classes}, y);
  Tensor logits = S->model.forward(in, true);
  float loss = S->loss.calculate(logits, t);
  Tensor grad = S->loss.backward(logits, t);
  S->model.backward(grad);
  auto params = S->model.getParameters();
  auto grads = S->model.getGradients();
  dump(logits, logits_out);
  if (grads_out) get_flat(grads, grads_out);
  S->adam.setLearningRate(lr_now);
  S->adam.update(params, grads);
  return loss;
}

// forward + loss + backward only (no optimiser), gradients left in the model.
float vref_forward_backward(void *s, const float *x, const float *y, uint64_t B, float *logits_out) {
  auto *S = static_cast<Session *>(s);
  Tensor in = make({(size_t)B, S->cfg.in_channels, S->cfg.image_size, S->cfg.image_size}, x);
  Tensor t = make({(size_t)B, S->cfg.num_classes}, y);
  Tensor logits = S->model.forward(in, true);
  float loss = S->loss.calculate(logits, t);
  Tensor grad = S->loss.backward(logits, t);
  S->model.backward(grad);
  dump(logits, logits_out);
  return loss;
}

// ------------------------------------------------------------------ single ops
void vref_matmul(const float *a, const float *b, uint64_t m, uint64_t n, uint64_t p, float *out) {
  dump(matrixMultiply(make({m, n}, a), make({n, p}, b)), out);
}
void vref_bmm(const float *a, const float *b, uint64_t bs, uint64_t m, uint64_t n, uint64_t p, float *out) {
  dump(batchMatrixMultiply(make({bs, m, n}, a), make({bs, n, p}, b)), out);
}
void vref_softmax3d(const float *x, uint64_t b, uint64_t n, uint64_t d, float *out) {
  dump(softmax(make({b, n, d}, x), 2), out);
}
void vref_softmax3d_backward(const float *g, const float *p, uint64_t b, uint64_t n, uint64_t d, float *out) {
  dump(softmax_backward(make({b, n, d}, g), make({b, n, d}, p)), out);
}

// Dense on a {rows,in} input: y, then backward with dy.
void vref_dense(uint64_t in, uint64_t out, const float *W, const float *b, const float *x, uint64_t rows,
                const float *dy, float *y, float *dW, float *db, float *dx) {
  Dense L(in, out);
  auto p = L.getParameters();
  set_flat({p[0]}, W);
  set_flat({p[1]}, b);
  dump(L.forward(make({rows, in}, x), true), y);
  if (dy) {
    Tensor gx = L.backward(make({rows, out}, dy));
    auto g = L.getGradients();
    dump(*g[0], dW);
    dump(*g[1], db);
    dump(gx, dx);
  }
}

void vref_layernorm(uint64_t D, float eps, const float *gamma, const float *beta, const float *x, uint64_t rows,
                    const float *dy, float *y, float *dgamma, float *dbeta, float *dx) {
  LayerNorm L(D, eps);
  auto p = L.getParameters();
  set_flat({p[0]}, gamma);
  set_flat({p[1]}, beta);
  dump(L.forward(make({rows, D}, x), true), y);
  if (dy) {
    Tensor gx = L.backward(make({rows, D}, dy));
    auto g = L.getGradients();
    dump(*g[0], dgamma);
    dump(*g[1], dbeta);
    dump(gx, dx);
  }
}

void vref_gelu(const float *x, uint64_t n, const float *dy, float *y, float *dx) {
  GELU L;
  dump(L.forward(make({n}, x), true), y);
  if (dy) dump(L.backward(make({n}, dy)), dx);
}

// MultiHeadAttention: params flat in getParameters() order (q W,b, k W,b, v W,b, out W,b).
void vref_mha(uint64_t D, uint64_t h, const float *params, const float *x, uint64_t B, uint64_t N, const float *dy,
              float *y, float *grads, float *dx) {
  MultiHeadAttention L(D, h);
  set_flat(L.getParameters(), params);
  dump(L.forward(make({B, N, D}, x), true), y);
  if (dy) {
    Tensor gx = L.backward(make({B, N, D}, dy));
    get_flat(L.getGradients(), grads);
    dump(gx, dx);
  }
}

// FeedForward: params flat (dense1 W,b, dense2 W,b); dropout 0.
void vref_ffn(uint64_t D, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N, const float *dy,
              float *y, float *grads, float *dx) {
  FeedForward L(D, H, 0.0f);
  set_flat(L.getParameters(), params);
  dump(L.forward(make({B, N, D}, x), true), y);
  if (dy) {
    Tensor gx = L.backward(make({B, N, D}, dy));
    get_flat(L.getGradients(), grads);
    dump(gx, dx);
  }
}

// TransformerEncoderBlock: 16 tensors flat; dropout 0.
void vref_block(uint64_t D, uint64_t h, uint64_t H, const float *params, const float *x, uint64_t B, uint64_t N,
                const float *dy, float *y, float *grads, float *dx) {
  TransformerEncoderBlock L(D, h, H, 0.0f);
  set_flat(L.getParameters(), params);
  dump(L.forward(make({B, N, D}, x), true), y);
  if (dy) {
    Tensor gx = L.backward(make({B, N, D}, dy));
    get_flat(L.getGradients(), grads);
    dump(gx, dx);
  }
}

// The masks the block's two Dropout layers drew in their last training forward
// (attention_dropout, then ffn.dropout): 2 x {B*N*D}.  The reference draws them from a
// random_device-seeded engine (src/layers/Dropout.cpp:31-33); reading them back is the only way
// to know which elements a given run dropped.
static void dump_block_masks(TransformerEncoderBlock &blk, float *dst, size_t n) {
  dump(blk.attention_dropout.mask, dst);
  dump(blk.ffn.dropout.mask, dst + n);
}

// TransformerEncoderBlock at dropout `rate` (> 0): as vref_block, plus the drawn masks.
void vref_block_dropout(uint64_t D, uint64_t h, uint64_t H, float rate, const float *params, const float *x, uint64_t B,
                        uint64_t N, const float *dy, float *y, float *grads, float *dx, float *masks) {
  TransformerEncoderBlock L(D, h, H, rate);
  set_flat(L.getParameters(), params);
  dump(L.forward(make({B, N, D}, x), true), y);
  dump_block_masks(L, masks, B * N * D);
  if (dy) {
    Tensor gx = L.backward(make({B, N, D}, dy));
    get_flat(L.getGradients(), grads);
    dump(gx, dx);
  }
}

// Masks of every block after a training-mode forward of the session's model: L x 2 x {B*N*D}.
void vref_get_dropout_masks(void *s, uint64_t B, float *masks) {
  auto *S = static_cast<Session *>(s);
  const size_t n = B * S->model.num_tokens * S->cfg.embedding_dim;
  for (auto &blk : S->model.encoder_blocks) {
    dump_block_masks(blk, masks, n);
    masks += 2 * n;
  }
}

// Embeddings: params flat (W_pe, b_pe, cls, pos); x {B,C,H,W}; y {B,N,D}.
void vref_embeddings(uint64_t img, uint64_t P, uint64_t C, uint64_t D, const float *params, const float *x, uint64_t B,
                     const float *dy, float *y, float *grads, float *dx_img) {
  Embeddings L(img, img, P, C, D);
  set_flat(L.getParameters(), params);
  dump(L.forward(make({B, C, img, img}, x), true), y);
  if (dy) {
    size_t N = 1 + (img / P) * (img / P);
    Tensor gx = L.backward(make({B, N, D}, dy));
    get_flat(L.getGradients(), grads);
    dump(gx, dx_img);
  }
}

// CrossEntropy: loss value + gradient; weights may be null (n_w == 0).
float vref_cross_entropy(const float *logits, const float *y, uint64_t B, uint64_t K, const float *w, uint64_t n_w,
                         float *grad) {
  CrossEntropy L;
  if (n_w) L.setClassWeights(std::vector<float>(w, w + n_w));
  Tensor l = make({B, K}, logits), t = make({B, K}, y);
  float v = L.calculate(l, t);
  dump(L.backward(l, t), grad);
  return v;
}

// Adam on one {r,c} tensor for `steps` steps; grads = steps x r x c, lrs = steps.
void vref_adam(uint64_t r, uint64_t c, float b1, float b2, float eps, float wd, uint64_t steps, const float *lrs,
               const float *grads, float *param_inout) {
  Adam opt(0.0f, b1, b2, eps, wd);
  Tensor p = make({r, c}, param_inout);
  for (uint64_t s = 0; s < steps; ++s) {
    Tensor g = make({r, c}, grads + s * r * c);
    std::vector<Tensor *> ps{&p};
    std::vector<Tensor *> gs{&g};
    opt.setLearningRate(lrs[s]);
    opt.update(ps, gs);
  }
  dump(p, param_inout);
}

float vref_cosine_warmup_lr(long long step, long long warmup, long long total, float lr_max) {
  return cosine_warmup_lr(step, warmup, total, lr_max);
}

// Per-epoch sample order of Trainer::train_epoch (src/model/Trainer.cpp:151-154).
void vref_epoch_order(uint64_t n, int epoch, uint64_t *out) {
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::mt19937 rng(static_cast<uint32_t>(epoch));
  std::shuffle(idx.begin(), idx.end(), rng);
  for (uint64_t i = 0; i < n; ++i) out[i] = idx[i];
}

// DataAugmentation::apply (src/utils/DataAugmentation.cpp:4-35) with the engine re-seeded.
// cfg = the seven Config fields in declaration order.
void vref_augment(const float *x, uint64_t B, uint64_t C, uint64_t H, uint64_t W, uint32_t seed, const float *cfg,
                  float *out) {
  DataAugmentation::Config c;
  c.rotation_prob = cfg[0]; c.translate_prob = cfg[1]; c.zoom_prob = cfg[2];
  c.rotation_factor = cfg[3]; c.translate_factor = cfg[4]; c.zoom_min = cfg[5]; c.zoom_max = cfg[6];
  DataAugmentation aug(c);
  aug.rng.seed(seed);
  Tensor r = aug.apply(make({B, C, H, W}, x));
  dump(r, out);
}

// first-max argmax over rows as app/test.cpp:103-112 / Trainer.cpp:33-42 define it.
void vref_argmax(const float *logits, uint64_t B, uint64_t K, int32_t *out) {
  for (uint64_t i = 0; i < B; ++i) {
    float best = -std::numeric_limits<float>::infinity();
    int pred = -1;
    for (uint64_t j = 0; j < K; ++j)
      if (logits[i * K + j] > best) { best = logits[i * K + j]; pred = (int)j; }
    out[i] = pred;
  }
}

} // extern "C"
