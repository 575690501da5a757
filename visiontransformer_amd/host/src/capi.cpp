// C-ABI over the host layer (include/vit_host.h): opaque handles around the C++ classes, every
// exception turned into a status code + thread-local message.
#include "vit_host.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <random>
#include <string>

#include "core/Comm.hpp"
#include "core/Device.hpp"
#include "core/Ops.hpp"
#include "model/Trainer.hpp"
#include <fstream>

#include "optimizers/Scheduler.hpp"
#include "utils/ModelUtils.hpp"

namespace {

thread_local std::string g_err;

template <typename F>
int guard(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  } catch (...) {
    g_err = "unknown exception";
    return 1;
  }
}

ViTConfig to_cfg(const vith_config *c) {
  ViTConfig k;
  k.image_size = c->image_size; k.patch_size = c->patch_size; k.in_channels = c->in_channels;
  k.num_classes = c->num_classes; k.embedding_dim = c->embedding_dim; k.num_heads = c->num_heads;
  k.num_layers = c->num_layers; k.mlp_hidden_dim = c->mlp_hidden_dim; k.dropout_rate = c->dropout_rate;
  return k;
}

struct Model {
  VisionTransformer vit;
  explicit Model(const ViTConfig &c) : vit(c) {}
  Tensor input(const float *x_dev, uint64_t B) {
    const ViTConfig &c = vit.getConfig();
    return Tensor::wrapDevice(const_cast<float *>(x_dev), {(size_t)B, c.in_channels, c.image_size, c.image_size});
  }
  size_t tokens() const {
    const ViTConfig &c = vit.getConfig();
    return 1 + (c.image_size / c.patch_size) * (c.image_size / c.patch_size);
  }
};

struct TrainerState {
  Model *model;
  CrossEntropy loss;
  Adam adam;
  Tensor logits, labels;
  // gradient exchange of the staged step; created on the first step that finds a communicator
  std::unique_ptr<vit::GradientReducer> reducer;
  size_t bucket_bytes = 25u << 20;
  TrainerState(Model *m, float b1, float b2, float eps, float wd) : model(m), adam(0.0f, b1, b2, eps, wd) {}
  vit::GradientReducer *exchange() {
    if (!vit::comm_active()) {
      reducer.reset();
      return nullptr;
    }
    if (!reducer) reducer = std::make_unique<vit::GradientReducer>(model->vit, bucket_bytes);
    return reducer.get();
  }
};

std::vector<double> g_fit_epoch_ms;  // Trainer::epochTrainMs() of the last vith_fit* call

struct LayerBox {
  std::unique_ptr<Layer> layer;
};

Model *M(vith_model m) { return static_cast<Model *>(m); }
TrainerState *T(vith_trainer t) { return static_cast<TrainerState *>(t); }
LayerBox *L(vith_layer l) { return static_cast<LayerBox *>(l); }

void copy_out(const Tensor &t, float *dst_dev) {
  Tensor d = t.dense();
  vit::copy_on_device(dst_dev, d.deviceData(), d.getSize());
}

std::vector<size_t> shape_of(const uint64_t *s, uint64_t rank) {
  std::vector<size_t> v(rank);
  for (uint64_t i = 0; i < rank; ++i) v[i] = (size_t)s[i];
  return v;
}

}  // namespace

extern "C" {

const char *vith_last_error(void) { return g_err.c_str(); }
int vith_init(int device) { return guard([&] { vit::init(device); }); }
int vith_set_stream(void *s) { return guard([&] { vit::set_stream(s); }); }
int vith_synchronize(void) { return guard([&] { vit::synchronize(); }); }
int vith_seed(uint64_t seed) { return guard([&] { vit::seed(seed); }); }
int vith_set_fused_attention(int on) {
  return guard([&] { MultiHeadAttention::setDefaultFused(on != 0); });
}
int vith_comm_unique_id(void *id128) {
  return guard([&] {
    if (!id128) throw std::invalid_argument("vith_comm_unique_id: null pointer");
    vit::comm_unique_id(id128);
  });
}
int vith_comm_init(int rank, int world, const void *id128) {
  return guard([&] { vit::comm_init_rccl(rank, world, id128); });
}
int vith_comm_init_file(int rank, int world, const char *path, int timeout_s) {
  return guard([&] {
    if (!path) throw std::invalid_argument("vith_comm_init_file: null path");
    vit::comm_init_rccl_file(rank, world, path, timeout_s > 0 ? timeout_s : 120);
  });
}
int vith_comm_init_external(int rank, int world, vith_allreduce_fn allreduce, vith_broadcast_fn broadcast, void *user,
                            int stream_ordered) {
  return guard([&] { vit::comm_init_external(rank, world, allreduce, broadcast, user, stream_ordered != 0); });
}
int vith_comm_exchange_id_file(int rank, int world, const char *path, int timeout_s, void *id128) {
  return guard([&] {
    if (!path || !id128) throw std::invalid_argument("vith_comm_exchange_id_file: null pointer");
    vit::comm_exchange_id_file(rank, world, path, timeout_s > 0 ? timeout_s : 120, id128);
  });
}
int vith_comm_timing(int on) { return guard([&] { vit::comm_timing(on != 0); }); }
int64_t vith_comm_exposed_ms(double *total_ms) {
  int64_t n = -1;
  guard([&] { n = (int64_t)vit::comm_exposed_ms(total_ms); });
  return n;
}
int vith_comm_destroy(void) { return guard([&] { vit::comm_destroy(); }); }
int vith_comm_rank(void) { return vit::comm_rank(); }
int vith_comm_world(void) { return vit::comm_world(); }
int vith_comm_broadcast_params(vith_model m, int root) {
  return guard([&] {
    auto &p = M(m)->vit.arena().params();
    vit::comm_broadcast(p->dev_write(), p->count, root);
  });
}
int vith_comm_allreduce(float *dev, uint64_t count) {
  return guard([&] { vit::comm_allreduce(dev, (size_t)count); });
}
int vith_trainer_set_bucket_bytes(vith_trainer t, uint64_t bytes) {
  return guard([&] {
    T(t)->bucket_bytes = (size_t)bytes;
    T(t)->reducer.reset();
  });
}
uint64_t vith_trainer_num_buckets(vith_trainer t) {
  uint64_t n = 0;
  guard([&] {
    if (auto *r = T(t)->exchange()) n = r->buckets().size();
  });
  return n;
}
int64_t vith_plan_buckets(const uint64_t *stage_offset, const uint64_t *stage_count, uint64_t stages,
                          uint64_t bucket_bytes, uint64_t *out3, uint64_t out_capacity) {
  int64_t n = -1;
  guard([&] {
    std::vector<std::pair<size_t, size_t>> r(stages);
    for (uint64_t i = 0; i < stages; ++i) r[i] = {(size_t)stage_offset[i], (size_t)stage_count[i]};
    auto plan = vit::GradientReducer::planBuckets(r, (size_t)bucket_bytes);
    if (plan.size() > out_capacity) throw std::runtime_error("vith_plan_buckets: output too small");
    for (size_t i = 0; i < plan.size(); ++i) {
      out3[3 * i] = plan[i].last_stage;
      out3[3 * i + 1] = plan[i].offset;
      out3[3 * i + 2] = plan[i].count;
    }
    n = (int64_t)plan.size();
  });
  return n;
}
int vith_dropout_state(uint64_t *seed, uint64_t *offset) {
  return guard([&] {
    if (!seed || !offset) throw std::invalid_argument("vith_dropout_state: null pointer");
    vit::get_dropout_state(seed, offset);
  });
}
int vith_set_dropout_state(uint64_t seed, uint64_t offset) {
  return guard([&] { vit::set_dropout_state(seed, offset); });
}
int vith_dropout_replay(const float *masks_dev, uint64_t count) {
  return guard([&] { vit::set_dropout_replay(masks_dev, (size_t)count); });
}
int vith_set_first_token_tail(int on) {
  return guard([&] { VisionTransformer::setFirstTokenTail(on != 0); });
}
uint64_t vith_device_bytes_in_use(void) { return vit::device_bytes_in_use(); }

vith_model vith_model_create(const vith_config *cfg, int patch_mode, int dk_mode) {
  Model *m = nullptr;
  const int rc = guard([&] {
    if (!cfg) throw std::invalid_argument("vith_model_create: null config");
    PatchEmbedding::setDefaultPatchMode(patch_mode);
    MultiHeadAttention::setDefaultKeyGradMode(dk_mode);
    m = new Model(to_cfg(cfg));
    PatchEmbedding::setDefaultPatchMode(VITK_PATCH_REFERENCE);
    MultiHeadAttention::setDefaultKeyGradMode(VITK_DK_REFERENCE);
  });
  return rc ? nullptr : m;
}
void vith_model_destroy(vith_model m) { delete M(m); }

uint64_t vith_param_count(vith_model m) { return VisionTransformer::parameterCount(M(m)->vit.getConfig()); }
uint64_t vith_param_tensors(vith_model m) { return M(m)->vit.getParameters().size(); }
uint64_t vith_param_shape(vith_model m, uint64_t idx, uint64_t *dims) {
  auto p = M(m)->vit.getParameters();
  if (idx >= p.size()) return 0;
  const auto &s = p[idx]->getShape();
  for (size_t i = 0; i < s.size() && i < 4; ++i) dims[i] = s[i];
  return s.size();
}
int vith_get_params(vith_model m, float *flat) {
  return guard([&] {
    auto v = ModelUtils::flatten(M(m)->vit.getParameters());
    std::memcpy(flat, v.data(), v.size() * sizeof(float));
  });
}
int vith_set_params(vith_model m, const float *flat) {
  return guard([&] { ModelUtils::unflatten(M(m)->vit.getParameters(), flat, vith_param_count(m)); });
}
int vith_get_grads(vith_model m, float *flat) {
  return guard([&] {
    auto v = ModelUtils::flatten(M(m)->vit.getGradients());
    std::memcpy(flat, v.data(), v.size() * sizeof(float));
  });
}
float *vith_params_device(vith_model m) { return M(m)->vit.arena().params()->dev_write(); }
float *vith_grads_device(vith_model m) { return M(m)->vit.arena().grads()->dev_write(); }

int vith_save_weights(vith_model m, const char *path) {
  return guard([&] { ModelUtils::set_verbose(false); ModelUtils::save_weights(M(m)->vit, path); });
}
int vith_load_weights(vith_model m, const char *path) {
  return guard([&] { ModelUtils::set_verbose(false); ModelUtils::load_weights(M(m)->vit, path); });
}
int vith_save_config(const vith_config *cfg, const char *path) {
  return guard([&] { ModelUtils::set_verbose(false); ModelUtils::save_config(to_cfg(cfg), path); });
}
int vith_load_config(const char *path, vith_config *cfg) {
  return guard([&] {
    ModelUtils::set_verbose(false);
    ViTConfig k = ModelUtils::load_config(path);
    cfg->image_size = k.image_size; cfg->patch_size = k.patch_size; cfg->in_channels = k.in_channels;
    cfg->num_classes = k.num_classes; cfg->embedding_dim = k.embedding_dim; cfg->num_heads = k.num_heads;
    cfg->num_layers = k.num_layers; cfg->mlp_hidden_dim = k.mlp_hidden_dim; cfg->dropout_rate = k.dropout_rate;
  });
}

int vith_forward(vith_model m, const float *x_dev, uint64_t B, int training, float *logits_dev) {
  return guard([&] {
    Tensor out = M(m)->vit.forward(M(m)->input(x_dev, B), training != 0);
    copy_out(out, logits_dev);
  });
}

int vith_forward_trace(vith_model m, const float *x_dev, uint64_t B, float *acts_dev, float *logits_dev) {
  return guard([&] {
    VisionTransformer &v = M(m)->vit;
    v.setTrace(true);
    Tensor out = v.forward(M(m)->input(x_dev, B), true);
    v.setTrace(false);
    size_t off = 0;
    for (const Tensor &t : v.tracedActivations()) {
      copy_out(t, acts_dev + off);
      off += t.getSize();
    }
    copy_out(out, logits_dev);
  });
}

int vith_backward(vith_model m, const float *dlogits_dev, uint64_t B, float *dacts_dev) {
  return guard([&] {
    VisionTransformer &v = M(m)->vit;
    Tensor g = Tensor::wrapDevice(const_cast<float *>(dlogits_dev), {(size_t)B, v.getConfig().num_classes});
    v.setTrace(dacts_dev != nullptr);
    v.backward(g);
    v.setTrace(false);
    if (dacts_dev) {
      size_t off = 0;
      for (const Tensor &t : v.tracedGradients()) {
        copy_out(t, dacts_dev + off);
        off += t.getSize();
      }
    }
  });
}

int vith_backward_image(vith_model m, const float *dlogits_dev, uint64_t B, float *dimage_dev) {
  return guard([&] {
    VisionTransformer &v = M(m)->vit;
    if (!dimage_dev) throw std::invalid_argument("vith_backward_image: null output");
    Tensor g = Tensor::wrapDevice(const_cast<float *>(dlogits_dev), {(size_t)B, v.getConfig().num_classes});
    Tensor dimg = v.backward(g);
    if (!dimg.defined() || dimg.getSize() != (size_t)B * v.getConfig().in_channels * v.getConfig().image_size *
                                                 v.getConfig().image_size)
      throw std::runtime_error("vith_backward_image: the model's backward returned no image gradient");
    copy_out(dimg, dimage_dev);
  });
}

int vith_predict(vith_model m, const float *x_dev, uint64_t B, int32_t *pred_dev) {
  return guard([&] {
    Tensor logits = M(m)->vit.forward(M(m)->input(x_dev, B), false);
    Tensor probs = softmax(logits);
    vit::check(vitk_argmax_rows(probs.deviceData(), (int32_t)B, (int32_t)M(m)->vit.getConfig().num_classes, pred_dev,
                                vit::stream()),
               "vitk_argmax_rows");
  });
}

vith_trainer vith_trainer_create(vith_model m, float beta1, float beta2, float eps, float weight_decay,
                                 const float *class_weights, uint64_t n_weights) {
  TrainerState *t = nullptr;
  const int rc = guard([&] {
    t = new TrainerState(M(m), beta1, beta2, eps, weight_decay);
    if (class_weights && n_weights) t->loss.setClassWeights(std::vector<float>(class_weights, class_weights + n_weights));
  });
  return rc ? nullptr : t;
}
void vith_trainer_destroy(vith_trainer t) { delete T(t); }

int vith_step_forward(vith_trainer t, const float *x_dev, const float *y_dev, uint64_t B, float grad_batch) {
  return guard([&] {
    TrainerState *s = T(t);
    VisionTransformer &v = s->model->vit;
    s->logits = v.forward(s->model->input(x_dev, B), true);
    s->labels = Tensor::wrapDevice(const_cast<float *>(y_dev), {(size_t)B, v.getConfig().num_classes});
    s->loss.setGradientBatch(grad_batch);
    s->loss.enqueue(s->logits, s->labels);
    v.backwardBegin(s->loss.backward(s->logits, s->labels));
  });
}
uint64_t vith_num_stages(vith_trainer t) { return T(t)->model->vit.numBackwardStages(); }
int vith_step_backward_stage(vith_trainer t, uint64_t stage) {
  return guard([&] {
    T(t)->model->vit.backwardStage((size_t)stage);
    if (auto *r = T(t)->exchange()) r->afterStage((size_t)stage);
  });
}
int vith_stage_grad_range(vith_trainer t, uint64_t stage, uint64_t *offset, uint64_t *count) {
  return guard([&] {
    size_t o, c;
    T(t)->model->vit.stageGradientRange((size_t)stage, &o, &c);
    *offset = o;
    *count = c;
  });
}
int vith_step_update(vith_trainer t, float lr, float grad_scale) {
  return guard([&] {
    TrainerState *s = T(t);
    if (auto *r = s->exchange()) r->finish();
    auto p = s->model->vit.getParameters();
    auto g = s->model->vit.getGradients();
    s->adam.setLearningRate(lr);
    s->adam.setGradientScale(grad_scale);
    s->adam.update(p, g);
  });
}
static DataAugmentation::Config to_aug(const vith_augment_config *a) {
  DataAugmentation::Config c;
  c.rotation_prob = a->rotation_prob;
  c.translate_prob = a->translate_prob;
  c.zoom_prob = a->zoom_prob;
  c.rotation_factor = a->rotation_factor;
  c.translate_factor = a->translate_factor;
  c.zoom_min = a->zoom_min;
  c.zoom_max = a->zoom_max;
  return c;
}
int vith_augment(const vith_augment_config *cfg, uint32_t seed, const float *x_dev, uint64_t B, uint64_t C, uint64_t H,
                 uint64_t W, float *out_dev) {
  return guard([&] {
    if (!cfg || !x_dev || !out_dev) throw std::invalid_argument("vith_augment: null pointer");
    DataAugmentation augmentor(to_aug(cfg), seed);
    Tensor x = Tensor::wrapDevice(const_cast<float *>(x_dev), {(size_t)B, (size_t)C, (size_t)H, (size_t)W});
    Tensor y = augmentor.apply(x);
    vit::copy_on_device(out_dev, y.deviceData(), (size_t)(B * C * H * W));
  });
}
int vith_fit(vith_model m, const float *x, const float *y, uint64_t n, const float *x_test, const float *y_test,
             uint64_t n_test, int epochs, uint64_t batch_size, float lr_init, float weight_decay, float warmup_frac,
             float *history) {
  return vith_fit_augmented(m, x, y, n, x_test, y_test, n_test, epochs, batch_size, lr_init, weight_decay, warmup_frac,
                            nullptr, 0, history);
}
int vith_fit_augmented(vith_model m, const float *x, const float *y, uint64_t n, const float *x_test,
                       const float *y_test, uint64_t n_test, int epochs, uint64_t batch_size, float lr_init,
                       float weight_decay, float warmup_frac, const vith_augment_config *augment, uint32_t augment_seed,
                       float *history) {
  return guard([&] {
    VisionTransformer &v = M(m)->vit;
    const ViTConfig &c = v.getConfig();
    const size_t sample = c.in_channels * c.image_size * c.image_size, K = c.num_classes;
    auto wrap = [&](const float *px, const float *py, uint64_t rows) {
      return std::make_pair(
          Tensor({(size_t)rows, c.in_channels, c.image_size, c.image_size}, std::vector<float>(px, px + rows * sample)),
          Tensor({(size_t)rows, K}, std::vector<float>(py, py + rows * K)));
    };
    TrainerConfig tc;
    tc.epochs = epochs;
    tc.batch_size = (size_t)batch_size;
    tc.lr_init = lr_init;
    tc.learning_rate = lr_init;
    tc.weight_decay = weight_decay;
    tc.warmup_frac = warmup_frac;
    tc.augment = augment != nullptr;
    if (augment) {
      tc.augment_seed = augment_seed;
      tc.augment_config = to_aug(augment);
    }
    Trainer trainer(v, tc);
    trainer.setQuiet(true);
    trainer.train(wrap(x, y, n), wrap(x_test, y_test, n_test));
    g_fit_epoch_ms = trainer.epochTrainMs();
    const auto &h = trainer.history();
    for (size_t e = 0; e < h.size() && history; ++e)
      for (int k = 0; k < 4; ++k) history[4 * e + k] = h[e][k];
  });
}
int vith_fit_reference_defaults(vith_model m, const float *x, const float *y, uint64_t n, const float *x_test,
                                const float *y_test, uint64_t n_test, int epochs, uint64_t batch_size, float lr_init,
                                float weight_decay, float warmup_frac, float *history) {
  return guard([&] {
    VisionTransformer &v = M(m)->vit;
    const ViTConfig &c = v.getConfig();
    const size_t sample = c.in_channels * c.image_size * c.image_size, K = c.num_classes;
    auto wrap = [&](const float *px, const float *py, uint64_t rows) {
      return std::make_pair(
          Tensor({(size_t)rows, c.in_channels, c.image_size, c.image_size}, std::vector<float>(px, px + rows * sample)),
          Tensor({(size_t)rows, K}, std::vector<float>(py, py + rows * K)));
    };
    TrainerConfig tc;  // only the fields app/main.cpp:21-27 sets; everything else as constructed
    tc.epochs = epochs;
    tc.batch_size = (size_t)batch_size;
    tc.lr_init = lr_init;
    tc.weight_decay = weight_decay;
    tc.warmup_frac = warmup_frac;
    Trainer trainer(v, tc);
    trainer.setQuiet(true);
    trainer.train(wrap(x, y, n), wrap(x_test, y_test, n_test));
    g_fit_epoch_ms = trainer.epochTrainMs();
    const auto &h = trainer.history();
    for (size_t e = 0; e < h.size() && history; ++e)
      for (int k = 0; k < 4; ++k) history[4 * e + k] = h[e][k];
  });
}
int vith_fit_with_options(vith_model m, const float *x, const float *y, uint64_t n, const float *x_test,
                          const float *y_test, uint64_t n_test, int epochs, uint64_t batch_size, float lr_init,
                          float weight_decay, float warmup_frac, const vith_fit_options *opt, float *history) {
  return guard([&] {
    if (!opt) throw std::invalid_argument("vith_fit_with_options: null options");
    VisionTransformer &v = M(m)->vit;
    const ViTConfig &c = v.getConfig();
    const size_t sample = c.in_channels * c.image_size * c.image_size, K = c.num_classes;
    auto wrap = [&](const float *px, const float *py, uint64_t rows) {
      return std::make_pair(
          Tensor({(size_t)rows, c.in_channels, c.image_size, c.image_size}, std::vector<float>(px, px + rows * sample)),
          Tensor({(size_t)rows, K}, std::vector<float>(py, py + rows * K)));
    };
    TrainerConfig tc;
    tc.epochs = epochs;
    tc.batch_size = (size_t)batch_size;
    tc.lr_init = lr_init;
    tc.learning_rate = lr_init;
    tc.weight_decay = weight_decay;
    tc.warmup_frac = warmup_frac;
    tc.augment = false;
    if (opt->best_weights_path) tc.best_weights_path = opt->best_weights_path;
    if (opt->state_path) tc.state_path = opt->state_path;
    tc.start_epoch = opt->start_epoch;
    tc.stop_epoch = opt->stop_epoch;
    ModelUtils::set_verbose(false);
    if (opt->resume) {
      if (tc.state_path.empty()) throw std::invalid_argument("vith_fit_with_options: resume needs state_path");
      ModelUtils::load_weights(v, tc.state_path + ".weights");
    }
    Trainer trainer(v, tc);
    trainer.setQuiet(true);
    if (opt->resume) trainer.setStartEpoch(trainer.loadState(tc.state_path));
    trainer.train(wrap(x, y, n), wrap(x_test, y_test, n_test));
    g_fit_epoch_ms = trainer.epochTrainMs();
    const auto &h = trainer.history();
    for (size_t e = 0; e < h.size() && history; ++e)
      for (int k = 0; k < 4; ++k) history[4 * e + k] = h[e][k];
  });
}
uint64_t vith_next_seed(void) { return vit::next_seed(); }
uint64_t vith_last_fit_epoch_ms(double *ms, uint64_t capacity) {
  for (size_t e = 0; e < g_fit_epoch_ms.size() && e < capacity && ms; ++e) ms[e] = g_fit_epoch_ms[e];
  return g_fit_epoch_ms.size();
}
int vith_trainer_save_state(vith_trainer t, const char *path) {
  return guard([&] {
    if (!path) throw std::invalid_argument("vith_trainer_save_state: null path");
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error(std::string("No se pudo abrir el archivo de estado para escritura: ") + path);
    uint64_t dseed = 0, doff = 0;
    vit::get_dropout_base_state(&dseed, &doff);
    out.write("VITSTEP1", 8);
    out.write(reinterpret_cast<const char *>(&dseed), sizeof dseed);
    out.write(reinterpret_cast<const char *>(&doff), sizeof doff);
    T(t)->adam.writeState(out);
  });
}
int vith_trainer_load_state(vith_trainer t, const char *path) {
  return guard([&] {
    if (!path) throw std::invalid_argument("vith_trainer_load_state: null path");
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error(std::string("No se pudo abrir el archivo de estado para lectura: ") + path);
    char magic[8];
    uint64_t dseed = 0, doff = 0;
    in.read(magic, 8);
    in.read(reinterpret_cast<char *>(&dseed), sizeof dseed);
    in.read(reinterpret_cast<char *>(&doff), sizeof doff);
    if (!in || std::string(magic, 8) != "VITSTEP1")
      throw std::runtime_error(std::string("Archivo de estado con formato desconocido: ") + path);
    vit::set_dropout_base_state(dseed, doff);
    T(t)->adam.readState(in, T(t)->model->vit.getParameters());
  });
}
int vith_step_result(vith_trainer t, float *loss, float *correct) {
  return guard([&] {
    const float *r = T(t)->loss.resultBuffer().data();
    if (loss) *loss = r[0];
    if (correct) *correct = r[1];
  });
}
int vith_train_step(vith_trainer t, const float *x_dev, const float *y_dev, uint64_t B, float lr, float *loss,
                    float *correct) {
  if (vith_step_forward(t, x_dev, y_dev, B, 0.0f)) return 1;
  const uint64_t n = vith_num_stages(t);
  for (uint64_t s = 0; s < n; ++s)
    if (vith_step_backward_stage(t, s)) return 1;
  if (vith_step_update(t, lr, 1.0f)) return 1;
  if (loss || correct) return vith_step_result(t, loss, correct);
  return 0;
}
int vith_eval_batch(vith_trainer t, const float *x_dev, const float *y_dev, uint64_t B, float *loss, float *correct) {
  return guard([&] {
    TrainerState *s = T(t);
    VisionTransformer &v = s->model->vit;
    Tensor logits = v.forward(s->model->input(x_dev, B), false);
    Tensor y = Tensor::wrapDevice(const_cast<float *>(y_dev), {(size_t)B, v.getConfig().num_classes});
    s->loss.setGradientBatch(0.0f);
    const float l = s->loss.calculate(logits, y);
    if (loss) *loss = l;
    if (correct) *correct = s->loss.lastCorrect();
  });
}

float vith_cosine_warmup_lr(int64_t step, int64_t warmup, int64_t total, float lr_max) {
  return cosine_warmup_lr(step, warmup, total, lr_max);
}

int vith_epoch_order(uint64_t n, int32_t epoch, uint64_t *out) {
  return guard([&] {
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::mt19937 rng(static_cast<uint32_t>(epoch));
    std::shuffle(idx.begin(), idx.end(), rng);
    for (uint64_t i = 0; i < n; ++i) out[i] = idx[i];
  });
}

vith_layer vith_layer_create(int kind, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  return vith_layer_create_dropout(kind, a, b, c, d, 0.0f);
}
vith_layer vith_layer_create_dropout(int kind, uint64_t a, uint64_t b, uint64_t c, uint64_t d, float dropout_rate) {
  LayerBox *box = nullptr;
  const int rc = guard([&] {
    std::unique_ptr<Layer> l;
    switch (kind) {
      case 0: l = std::make_unique<Dense>(a, b); break;
      case 1: l = std::make_unique<LayerNorm>(a); break;
      case 2: l = std::make_unique<GELU>(); break;
      case 3: l = std::make_unique<MultiHeadAttention>(a, b); break;
      case 4: l = std::make_unique<FeedForward>(a, b, dropout_rate); break;
      case 5: l = std::make_unique<TransformerEncoderBlock>(a, b, c, dropout_rate); break;
      case 6: l = std::make_unique<Embeddings>(a, a, b, c, d); break;
      default: throw std::invalid_argument("vith_layer_create: unknown kind");
    }
    box = new LayerBox{std::move(l)};
  });
  return rc ? nullptr : box;
}
void vith_layer_destroy(vith_layer l) { delete L(l); }
uint64_t vith_layer_param_count(vith_layer l) {
  uint64_t n = 0;
  for (Tensor *t : L(l)->layer->getParameters()) n += t->getSize();
  return n;
}
int vith_layer_set_params(vith_layer l, const float *flat) {
  return guard([&] { ModelUtils::unflatten(L(l)->layer->getParameters(), flat, vith_layer_param_count(l)); });
}
int vith_layer_get_grads(vith_layer l, float *flat) {
  return guard([&] {
    auto v = ModelUtils::flatten(L(l)->layer->getGradients());
    std::memcpy(flat, v.data(), v.size() * sizeof(float));
  });
}
int vith_layer_forward(vith_layer l, const float *x, const uint64_t *x_shape, uint64_t x_rank, int training, float *y,
                       uint64_t y_count) {
  return guard([&] {
    auto shape = shape_of(x_shape, x_rank);
    size_t n = 1;
    for (size_t s : shape) n *= s;
    Tensor in(shape, std::vector<float>(x, x + n));
    Tensor out = L(l)->layer->forward(in, training != 0);
    if (out.getSize() != y_count) throw std::runtime_error("vith_layer_forward: output size mismatch");
    Tensor d = out.dense();
    std::memcpy(y, static_cast<const Tensor &>(d).data(), y_count * sizeof(float));
  });
}
int vith_layer_backward(vith_layer l, const float *dy, const uint64_t *dy_shape, uint64_t dy_rank, float *dx,
                        uint64_t dx_count) {
  return guard([&] {
    auto shape = shape_of(dy_shape, dy_rank);
    size_t n = 1;
    for (size_t s : shape) n *= s;
    Tensor g(shape, std::vector<float>(dy, dy + n));
    Tensor out = L(l)->layer->backward(g);
    if (!dx) return;
    if (!out.defined()) throw std::runtime_error("vith_layer_backward: this layer produces no input gradient");
    if (out.getSize() != dx_count) throw std::runtime_error("vith_layer_backward: gradient size mismatch");
    Tensor d = out.dense();
    std::memcpy(dx, static_cast<const Tensor &>(d).data(), dx_count * sizeof(float));
  });
}

}  // extern "C"
