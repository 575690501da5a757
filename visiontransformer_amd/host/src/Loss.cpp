// CrossEntropy and the free softmax() helpers.
#include "losses/CrossEntropy.hpp"

#include "core/Ops.hpp"

namespace {
void *S() { return vit::stream(); }
}  // namespace

// 2-D row softmax (src/losses/CrossEntropy.cpp:20-50 of the reference)
Tensor softmax(const Tensor &logits) {
  const auto &s = logits.getShape();
  if (s.size() != 2) throw std::runtime_error("softmax(logits) espera un tensor 2D.");
  Tensor x = logits.dense();
  Tensor p = Tensor::deviceEmpty(s);
  vit::check(vitk_softmax_rows(x.deviceData(), (int64_t)s[0], (int64_t)s[1], (int64_t)s[1], p.deviceDataOverwrite(), S()),
             "vitk_softmax_rows");
  return p;
}

// softmax over the last axis of a 3-D tensor (src/layers/MultiHeadAttention.cpp:123-156)
Tensor softmax(const Tensor &logits, int axis) {
  const auto &s = logits.getShape();
  if (axis < 0) axis = (int)s.size() + axis;
  if (s.size() == 2 && axis == 1) return softmax(logits);
  if (!(axis == 2 && s.size() == 3)) throw std::runtime_error("Softmax a lo largo de este eje no está implementado.");
  Tensor x = logits.dense();
  Tensor p = Tensor::deviceEmpty(s);
  vit::check(vitk_softmax_rows(x.deviceData(), (int64_t)(s[0] * s[1]), (int64_t)s[2], (int64_t)s[2],
                               p.deviceDataOverwrite(), S()),
             "vitk_softmax_rows");
  return p;
}

void CrossEntropy::setClassWeights(const std::vector<float> &weights) {
  class_weights = weights;
  weightsDevice = weights.empty() ? Tensor() : Tensor({weights.size()}, weights);
}

void CrossEntropy::enqueue(const Tensor &yPred, const Tensor &yTrue) {
  if (yPred.getShape() != yTrue.getShape() || yPred.getShape().size() != 2)
    throw std::runtime_error("Las formas de predicción y etiquetas verdaderas no coinciden.");
  const size_t B = yPred.getShape()[0], K = yPred.getShape()[1];
  if (!class_weights.empty() && class_weights.size() != K)
    throw std::runtime_error("CrossEntropy: el número de pesos de clase no coincide con las clases.");
  Tensor lg = yPred.dense(), y = yTrue.dense();
  softmaxOutput = Tensor::deviceEmpty({B, K});
  gradient = Tensor::deviceEmpty({B, K});
  if (!out2.defined()) out2 = Tensor::deviceEmpty({2});
  if (scratch.getSize() < 2 * B) scratch = Tensor::deviceEmpty({2 * B});
  const float gb = grad_batch > 0.0f ? grad_batch : (float)B;
  vit::check(vitk_softmax_ce(lg.deviceData(), y.deviceData(), (int32_t)B, (int32_t)K,
                             class_weights.empty() ? nullptr : weightsDevice.deviceData(), gb,
                             gradient.deviceDataOverwrite(), softmaxOutput.deviceDataOverwrite(),
                             out2.deviceDataOverwrite(), scratch.deviceDataOverwrite(), scratch.getSize() * sizeof(float),
                             S()),
             "vitk_softmax_ce");
}

float CrossEntropy::calculate(const Tensor &yPred, const Tensor &yTrue) {
  enqueue(yPred, yTrue);
  const float *r = static_cast<const Tensor &>(out2).data();  // downloads and synchronises
  last_correct = r[1];
  return r[0];
}

Tensor CrossEntropy::backward(const Tensor & /*yPred*/, const Tensor &yTrue) {
  if (!gradient.defined() || gradient.getShape() != yTrue.getShape())
    throw std::runtime_error("CrossEntropy::backward requiere un calculate() previo con las mismas formas.");
  return gradient;
}
