// Adam with coupled L2 (reference: src/optimizers/Adam.cpp).
#include "optimizers/Adam.hpp"

#include <cmath>
#include <cstdint>
#include <istream>
#include <ostream>
#include <string>

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

void Adam::allocateMoments(const std::vector<Tensor *> &parameters, bool gradients_flat) {
  flat = gradients_flat && tiles_one_block(parameters);
  m.clear();
  v.clear();
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

// State record: "VITADAM1", i64 step count, u64 number of moment buffers, then per buffer u64 size,
// f32 m[size], f32 v[size] (native endian, like the .weights format).
void Adam::writeState(std::ostream &out) const {
  out.write("VITADAM1", 8);
  const int64_t steps = t;
  const uint64_t nbuf = initialized ? m.size() : 0;
  out.write(reinterpret_cast<const char *>(&steps), sizeof steps);
  out.write(reinterpret_cast<const char *>(&nbuf), sizeof nbuf);
  for (uint64_t i = 0; i < nbuf; ++i) {
    const uint64_t n = m[i].getSize();
    out.write(reinterpret_cast<const char *>(&n), sizeof n);
    out.write(reinterpret_cast<const char *>(m[i].data()), (std::streamsize)(n * sizeof(float)));
    out.write(reinterpret_cast<const char *>(v[i].data()), (std::streamsize)(n * sizeof(float)));
  }
  if (!out) throw std::runtime_error("Adam::writeState: error de escritura.");
}

void Adam::readState(std::istream &in, const std::vector<Tensor *> &parameters) {
  char magic[8];
  int64_t steps = 0;
  uint64_t nbuf = 0;
  in.read(magic, 8);
  in.read(reinterpret_cast<char *>(&steps), sizeof steps);
  in.read(reinterpret_cast<char *>(&nbuf), sizeof nbuf);
  if (!in || std::string(magic, 8) != "VITADAM1") throw std::runtime_error("Adam::readState: formato desconocido.");
  t = steps;
  if (nbuf == 0) {
    initialized = false;
    m.clear();
    v.clear();
    return;
  }
  // a model's parameters and gradients are two arenas of one layout: flat when there is one buffer
  allocateMoments(parameters, nbuf == 1);
  if (m.size() != nbuf) throw std::runtime_error("Adam::readState: el estado no corresponde a estos parámetros.");
  std::vector<float> buf;
  for (uint64_t i = 0; i < nbuf; ++i) {
    uint64_t n = 0;
    in.read(reinterpret_cast<char *>(&n), sizeof n);
    if (!in || n != m[i].getSize()) throw std::runtime_error("Adam::readState: tamaño de estado incompatible.");
    buf.resize(n);
    in.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)(n * sizeof(float)));
    m[i] = Tensor(m[i].getShape(), buf);
    in.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)(n * sizeof(float)));
    v[i] = Tensor(v[i].getShape(), buf);
    if (!in) throw std::runtime_error("Adam::readState: archivo truncado.");
  }
}

void Adam::update(std::vector<Tensor *> &parameters, const std::vector<Tensor *> &gradients) {
  if (parameters.size() != gradients.size())
    throw std::runtime_error("El numero de parametros y gradientes no coincide en Adam::update.");
  if (parameters.empty()) return;
  if (!initialized) allocateMoments(parameters, tiles_one_block(gradients));
  else if (flat && !tiles_one_block(gradients))
    throw std::runtime_error("Adam::update: estado plano pero gradientes fuera de un arena.");
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
