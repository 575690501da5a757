// Adam with coupled L2 (reference: src/optimizers/Adam.cpp).
#include "optimizers/Adam.hpp"

#include <cmath>

#include "core/Ops.hpp"

Adam::Adam(float learningRate, float beta1, float beta2, float epsilon, float weight_decay)
    : Optimizer(learningRate), beta1(beta1), beta2(beta2), epsilon(epsilon), weight_decay(weight_decay), t(0),
      initialized(false) {}

namespace {
// true when the tensors are views that exactly tile one storage block (a model arena)
bool tiles_one_block(const std::vector<Tensor *> &ts) {
  if (ts.empty() || !ts[0]->defined()) return false;
  const auto &st = ts[0]->storage();
  size_t total = 0;
  for (const Tensor *t : ts) {
    if (t->storage() != st) return false;
    total += t->getSize();
  }
  return total == st->count;
}
}  // namespace

void Adam::update(std::vector<Tensor *> &parameters, const std::vector<Tensor *> &gradients) {
  if (parameters.size() != gradients.size())
    throw std::runtime_error("El numero de parametros y gradientes no coincide en Adam::update.");
  if (parameters.empty()) return;
  if (!initialized) {
    flat = tiles_one_block(parameters) && tiles_one_block(gradients);
    if (flat) {
      const size_t n = parameters[0]->storage()->count;
      m.emplace_back(std::vector<size_t>{n});
      v.emplace_back(std::vector<size_t>{n});
    } else {
      for (const Tensor *p : parameters) {
        if (!p->isContiguous()) throw std::runtime_error("Adam::update: parámetro no contiguo fuera de un arena.");
        m.emplace_back(p->getShape());
        v.emplace_back(p->getShape());
      }
    }
    initialized = true;
  }
  t++;
  // bias-correction terms exactly as the reference forms them: float(pow(double(beta), t))
  const float beta1_t = std::pow(beta1, t);
  const float beta2_t = std::pow(beta2, t);
  void *s = vit::stream();
  if (flat) {
    const auto &ps = parameters[0]->storage();
    const auto &gs = gradients[0]->storage();
    vit::check(vitk_adam_l2(ps->dev_write(), gs->dev_read(), m[0].deviceData(), v[0].deviceData(), (int64_t)ps->count,
                            learningRate, beta1, beta2, beta1_t, beta2_t, epsilon, weight_decay, grad_scale, s),
               "vitk_adam_l2");
    return;
  }
  for (size_t i = 0; i < parameters.size(); ++i) {
    Tensor g = gradients[i]->dense();
    vit::check(vitk_adam_l2(parameters[i]->deviceData(), g.deviceData(), m[i].deviceData(), v[i].deviceData(),
                            (int64_t)parameters[i]->getSize(), learningRate, beta1, beta2, beta1_t, beta2_t, epsilon,
                            weight_decay, grad_scale, s),
               "vitk_adam_l2");
  }
}
