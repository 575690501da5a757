// Dense, LayerNorm, GELU, Dropout, FeedForward on top of the libvitk kernels.
#include <memory>

#include "activations/GELU.hpp"
#include "layers/Dense.hpp"
#include "layers/Dropout.hpp"
#include "layers/FeedForward.hpp"
#include "layers/LayerNorm.hpp"

using vit::GemmOpts;

namespace {
void *S() { return vit::stream(); }
std::vector<size_t> with_last(const std::vector<size_t> &shape, size_t last) {
  std::vector<size_t> s = shape;
  s.back() = last;
  return s;
}
}  // namespace

// ------------------------------------------------------------------------------------ Dense
Dense::Dense(size_t inputSize, size_t outputSize)
    : Dense(inputSize, outputSize, *std::make_unique<vit::ParamArena>(inputSize * outputSize + outputSize)) {}

Dense::Dense(size_t inputSize, size_t outputSize, vit::ParamArena &arena) : in(inputSize), out(outputSize) {
  arena.take({in, out}, &weights, &weightGradients);
  arena.take({1, out}, &bias, &biasGradients);
  initialise();
}

Dense::Dense(size_t inputSize, size_t outputSize, Tensor W, Tensor b, Tensor dW, Tensor db)
    : in(inputSize), out(outputSize), weights(std::move(W)), bias(std::move(b)), weightGradients(std::move(dW)),
      biasGradients(std::move(db)) {}

void Dense::initialise() {
  // N(0, 0.02) weights, zero bias (src/layers/Dense.cpp:5-19 of the reference)
  Tensor w({in, out});
  w.randomizeNormal(0.0f, 0.02f);
  weights.copyFrom(w);
  Tensor z({1, out});
  bias.copyFrom(z);
}

Tensor Dense::forward(const Tensor &input, bool isTraining) {
  return forwardFused(input, isTraining, VITK_ACT_NONE, nullptr, nullptr);
}

Tensor Dense::forwardFused(const Tensor &input, bool isTraining, int act, Tensor *preact, const Tensor *residual) {
  const auto &shape = input.getShape();
  if (shape.size() != 2 && shape.size() != 3) throw std::runtime_error("Dense::forward solo soporta entradas 2D o 3D.");
  if (shape.back() != in) throw std::runtime_error("Dense::forward: dimensión de entrada incompatible.");
  Tensor x = input.dense();
  if (isTraining) inputTensor = x;
  const size_t rows = x.getSize() / in;
  Tensor y = Tensor::deviceEmpty(with_last(shape, out));
  GemmOpts o;
  o.bias = bias.deviceData();
  o.act = act;
  if (act == VITK_ACT_GELU_FWD && preact) {
    *preact = Tensor::deviceEmpty(with_last(shape, out));
    o.aux = preact->deviceDataOverwrite();
    o.ldaux = (int64_t)out;
  }
  Tensor r;
  if (residual) {
    r = residual->dense();
    o.residual = r.deviceData();
    o.ldr = (int64_t)out;
  }
  vit::gemm(false, false, (int64_t)rows, (int64_t)out, (int64_t)in, x.deviceData(), (int64_t)in, weights.deviceData(),
            (int64_t)weights.getStrides()[0], y.deviceDataOverwrite(), (int64_t)out, o);
  return y;
}

Tensor Dense::backward(const Tensor &outputGradient) {
  return backwardFused(outputGradient, VITK_ACT_NONE, nullptr, true);
}

Tensor Dense::backwardFused(const Tensor &outputGradient, int act, const Tensor *preact, bool needInputGrad) {
  if (!inputTensor.defined()) throw std::runtime_error("Dense::backward sin forward de entrenamiento previo.");
  Tensor dy = outputGradient.dense();
  const size_t rows = inputTensor.getSize() / in;
  if (dy.getSize() != rows * out) throw std::runtime_error("Dense::backward: forma de gradiente incompatible.");
  // dW = X^T dY : contraction over all rows; the library splits it over the chip.  db = dY.sum(0)
  // is accumulated by the same launch from the dY tiles it stages.
  if (!needInputGrad) {
    GemmOpts ow;
    ow.split_k = 0;
    ow.colsum_b = biasGradients.deviceData();
    vit::gemm(true, false, (int64_t)in, (int64_t)out, (int64_t)rows, inputTensor.deviceData(), (int64_t)in,
              dy.deviceData(), (int64_t)out, weightGradients.deviceData(), (int64_t)weightGradients.getStrides()[0], ow);
    return Tensor();
  }
  // ... together with dX = dY W^T (one launch for both on the reference's small models)
  Tensor dx = Tensor::deviceEmpty(inputTensor.getShape());
  GemmOpts ox;
  if (act == VITK_ACT_GELU_BWD) {
    if (!preact) throw std::runtime_error("Dense::backwardFused: GELU_BWD necesita la pre-activación.");
    ox.act = act;
    ox.aux = const_cast<float *>(preact->deviceData());
    ox.ldaux = (int64_t)in;
  }
  vit::dense_backward_gemms((int64_t)rows, (int64_t)in, (int64_t)out, inputTensor.deviceData(), dy.deviceData(),
                            weights.deviceData(), (int64_t)weights.getStrides()[0], weightGradients.deviceData(),
                            (int64_t)weightGradients.getStrides()[0], biasGradients.deviceData(),
                            dx.deviceDataOverwrite(), ox);
  return dx;
}

std::vector<Tensor *> Dense::getParameters() { return {&weights, &bias}; }
std::vector<Tensor *> Dense::getGradients() { return {&weightGradients, &biasGradients}; }

// -------------------------------------------------------------------------------- LayerNorm
LayerNorm::LayerNorm(size_t featureSize, float epsilon)
    : LayerNorm(featureSize, *std::make_unique<vit::ParamArena>(2 * featureSize), epsilon) {}

LayerNorm::LayerNorm(size_t featureSize, vit::ParamArena &arena, float epsilon)
    : featureSize(featureSize), epsilon(epsilon) {
  arena.take({1, featureSize}, &gamma, &gammaGradient);
  arena.take({1, featureSize}, &beta, &betaGradient);
  initialise();
}

void LayerNorm::initialise() {
  gamma.fill(1.0f);
  beta.fill(0.0f);
}

Tensor LayerNorm::forward(const Tensor &input, bool isTraining) {
  if (input.getShape().back() != featureSize)
    throw std::runtime_error("La última dimensión de la entrada no coincide con featureSize de LayerNorm.");
  Tensor x = input.dense();
  const size_t rows = x.getSize() / featureSize;
  Tensor y = Tensor::deviceEmpty(input.getShape());
  float *mu = nullptr, *inv = nullptr;
  if (isTraining) {
    inputTensor = x;
    mean = Tensor::deviceEmpty({rows, 1});
    variance = Tensor::deviceEmpty({rows, 1});
    mu = mean.deviceDataOverwrite();
    inv = variance.deviceDataOverwrite();
  }
  vit::check(vitk_layernorm_fwd(x.deviceData(), (int64_t)rows, (int64_t)featureSize, (int64_t)featureSize,
                                gamma.deviceData(), beta.deviceData(), epsilon, y.deviceDataOverwrite(), mu, inv, S()),
             "vitk_layernorm_fwd");
  return y;
}

Tensor LayerNorm::backward(const Tensor &outputGradient) { return backwardFused(outputGradient, nullptr); }

Tensor LayerNorm::backwardFused(const Tensor &outputGradient, const Tensor *residual) {
  if (!inputTensor.defined()) throw std::runtime_error("LayerNorm::backward sin forward de entrenamiento previo.");
  Tensor dy = outputGradient.dense();
  const size_t rows = inputTensor.getSize() / featureSize;
  if (dy.getSize() != rows * featureSize) throw std::runtime_error("LayerNorm::backward: forma incompatible.");
  Tensor r;
  if (residual) r = residual->dense();
  Tensor dx = Tensor::deviceEmpty(outputGradient.getShape());
  const size_t wsb = vitk_layernorm_bwd_workspace_bytes((int64_t)rows, (int64_t)featureSize);
  vit::check(vitk_layernorm_bwd(dy.deviceData(), inputTensor.deviceData(), mean.deviceData(), variance.deviceData(),
                                gamma.deviceData(), (int64_t)rows, (int64_t)featureSize, (int64_t)featureSize,
                                residual ? r.deviceData() : nullptr, dx.deviceDataOverwrite(),
                                gammaGradient.deviceData(), betaGradient.deviceData(), vit::workspace(wsb), wsb, S()),
             "vitk_layernorm_bwd");
  return dx;
}

std::vector<Tensor *> LayerNorm::getParameters() { return {&gamma, &beta}; }
std::vector<Tensor *> LayerNorm::getGradients() { return {&gammaGradient, &betaGradient}; }

// ------------------------------------------------------------------------------------- GELU
GELU::GELU() {}

Tensor GELU::forward(const Tensor &input, bool isTraining) {
  if (!input.isContiguous()) throw std::runtime_error("GELU::forward solo implementado para tensores contiguos.");
  if (isTraining) inputTensor = input;
  Tensor y = Tensor::deviceEmpty(input.getShape());
  vit::check(vitk_gelu_fwd(input.deviceData(), (int64_t)input.getSize(), y.deviceDataOverwrite(), S()), "vitk_gelu_fwd");
  return y;
}

Tensor GELU::backward(const Tensor &outputGradient) {
  if (!inputTensor.defined() || !outputGradient.isContiguous())
    throw std::runtime_error("GELU::backward solo implementado para tensores contiguos.");
  Tensor dx = Tensor::deviceEmpty(inputTensor.getShape());
  vit::check(vitk_gelu_bwd(inputTensor.deviceData(), outputGradient.deviceData(), (int64_t)inputTensor.getSize(),
                           dx.deviceDataOverwrite(), S()),
             "vitk_gelu_bwd");
  return dx;
}

// ---------------------------------------------------------------------------------- Dropout
Dropout::Dropout(float rate) : rate(rate) {
  if (rate < 0.0f || rate >= 1.0f) throw std::invalid_argument("La tasa de Dropout debe estar en el rango [0, 1).");
  scale = rate > 0.0f ? 1.0f / (1.0f - rate) : 1.0f;
}

Tensor Dropout::forward(const Tensor &input, bool isTraining) {
  if (!isTraining || rate == 0.0f) {
    count = 0;
    return input;
  }
  if (!input.isContiguous()) throw std::runtime_error("Dropout::forward actualmente solo soporta tensores contiguos.");
  Tensor y = Tensor::deviceEmpty(input.getShape());
  count = input.getSize();
  rowStride = 0;
  if ((replayed = vit::dropout_replay_take(count)) != nullptr) {
    vit::check(vitk_dropout_bwd(input.deviceData(), replayed, (int64_t)count, y.deviceDataOverwrite(), S()),
               "vitk_dropout_bwd");
    return y;
  }
  vit::dropout_stream(count, &seed, &offset);
  vit::check(vitk_dropout_residual_fwd(input.deviceData(), nullptr, (int64_t)count, rate, seed, offset,
                                       y.deviceDataOverwrite(), nullptr, S()),
             "vitk_dropout_residual_fwd");
  return y;
}

Tensor Dropout::forwardAdd(const Tensor &input, const Tensor &residual, bool isTraining) {
  if (!isTraining || rate == 0.0f) {
    count = 0;
    return residual + input;
  }
  if (!input.isContiguous() || !residual.isContiguous() || input.getSize() != residual.getSize())
    throw std::runtime_error("Dropout::forwardAdd espera dos tensores contiguos del mismo tamano.");
  Tensor y = Tensor::deviceEmpty(input.getShape());
  count = input.getSize();
  rowStride = 0;
  if ((replayed = vit::dropout_replay_take(count)) != nullptr) {
    vit::check(vitk_dropout_bwd(input.deviceData(), replayed, (int64_t)count, y.deviceDataOverwrite(), S()),
               "vitk_dropout_bwd");
    vit::check(vitk_add_f32(residual.deviceData(), y.deviceData(), (int64_t)count, y.deviceDataOverwrite(), S()),
               "vitk_add_f32");
    return y;
  }
  vit::dropout_stream(count, &seed, &offset);
  vit::check(vitk_dropout_residual_fwd(input.deviceData(), residual.deviceData(), (int64_t)count, rate, seed, offset,
                                       y.deviceDataOverwrite(), nullptr, S()),
             "vitk_dropout_residual_fwd");
  return y;
}

Tensor Dropout::forwardAddFirstToken(const Tensor &input, const Tensor *residual, size_t tokens, bool isTraining) {
  if (!isTraining || rate == 0.0f) {
    count = 0;
    return residual ? *residual + input : input;
  }
  const auto &sh = input.getShape();
  if (!input.isContiguous() || sh.size() != 3 || sh[1] != 1 || (residual && (!residual->isContiguous() || residual->getSize() != input.getSize())))
    throw std::runtime_error("Dropout::forwardAddFirstToken espera tensores contiguos {B,1,D}.");
  rows = sh[0], cols = sh[2], rowStride = tokens * cols;
  count = input.getSize();
  Tensor y = Tensor::deviceEmpty(sh);
  // the stream (or the recorded masks) moves on by the whole {B,tokens,D} tensor
  replayed = vit::dropout_replay_take(rows * rowStride);
  if (!replayed) vit::dropout_stream(rows * rowStride, &seed, &offset);
  vit::check(vitk_dropout_rows(input.deviceData(), residual ? residual->deviceData() : nullptr, replayed, (int64_t)rows,
                               (int64_t)cols, (int64_t)rowStride, rate, seed, offset, y.deviceDataOverwrite(), S()),
             "vitk_dropout_rows");
  return y;
}

Tensor Dropout::backward(const Tensor &outputGradient) {
  if (count == 0) return outputGradient;  // forward ran in inference mode or with rate 0
  if (rowStride != 0) {
    if (!outputGradient.isContiguous() || outputGradient.getSize() != count)
      throw std::runtime_error("Dropout::backward espera un gradiente contiguo con la forma de la entrada.");
    Tensor dx = Tensor::deviceEmpty(outputGradient.getShape());
    vit::check(vitk_dropout_rows(outputGradient.deviceData(), nullptr, replayed, (int64_t)rows, (int64_t)cols,
                                 (int64_t)rowStride, rate, seed, offset, dx.deviceDataOverwrite(), S()),
               "vitk_dropout_rows");
    return dx;
  }
  if (!outputGradient.isContiguous() || outputGradient.getSize() != count)
    throw std::runtime_error("Dropout::backward espera un gradiente contiguo con la forma de la entrada.");
  Tensor dx = Tensor::deviceEmpty(outputGradient.getShape());
  if (replayed) {
    vit::check(vitk_dropout_bwd(outputGradient.deviceData(), replayed, (int64_t)count, dx.deviceDataOverwrite(), S()),
               "vitk_dropout_bwd");
    return dx;
  }
  vit::check(vitk_dropout_stream_bwd(outputGradient.deviceData(), (int64_t)count, rate, seed, offset,
                                     dx.deviceDataOverwrite(), S()),
             "vitk_dropout_stream_bwd");
  return dx;
}

// ------------------------------------------------------------------------------ FeedForward
FeedForward::FeedForward(size_t embedding_dim, size_t hidden_dim, float dropout_rate)
    : FeedForward(embedding_dim, hidden_dim, dropout_rate,
                  *std::make_unique<vit::ParamArena>(2 * embedding_dim * hidden_dim + hidden_dim + embedding_dim)) {}

FeedForward::FeedForward(size_t embedding_dim, size_t hidden_dim, float dropout_rate, vit::ParamArena &arena)
    : dense1(embedding_dim, hidden_dim, arena), activation(), dense2(hidden_dim, embedding_dim, arena),
      dropout(dropout_rate) {}

Tensor FeedForward::forward(const Tensor &input, bool isTraining) { return forwardFused(input, isTraining, nullptr); }

Tensor FeedForward::forwardFused(const Tensor &input, bool isTraining, const Tensor *residual, size_t firstTokenOf) {
  // dense1 + GELU in one kernel; the pre-activation is kept for backward
  Tensor h = dense1.forwardFused(input, isTraining, VITK_ACT_GELU_FWD, isTraining ? &preActivation : nullptr, nullptr);
  const bool drop = dropout.active(isTraining);
  Tensor y = dense2.forwardFused(h, isTraining, VITK_ACT_NONE, nullptr, drop ? nullptr : residual);
  if (drop && firstTokenOf != 0) return dropout.forwardAddFirstToken(y, residual, firstTokenOf, isTraining);
  if (drop) y = residual ? dropout.forwardAdd(y, *residual, isTraining) : dropout.forward(y, isTraining);
  return y;
}

Tensor FeedForward::backward(const Tensor &outputGradient) {
  // The reference feeds dense2 the ORIGINAL output gradient: the dropout-masked one is computed
  // and discarded (src/layers/FeedForward.cpp:30-31), so the mask never reaches the FFN gradient.
  Tensor grad = dense2.backwardFused(outputGradient, VITK_ACT_GELU_BWD, &preActivation, true);
  return dense1.backward(grad);
}

std::vector<Tensor *> FeedForward::getParameters() {
  auto p = dense1.getParameters();
  auto q = dense2.getParameters();
  p.insert(p.end(), q.begin(), q.end());
  return p;
}

std::vector<Tensor *> FeedForward::getGradients() {
  auto p = dense1.getGradients();
  auto q = dense2.getGradients();
  p.insert(p.end(), q.begin(), q.end());
  return p;
}
