// Dense: Y = X W + b (reference: include/layers/Dense.hpp, src/layers/Dense.cpp).  W is {in,out},
// b is {1,out}; 2-D {rows,in} or 3-D {B,N,in} inputs.  The three contractions of a training step
// (forward, dW = X^T dY, dX = dY W^T) run on the matrix cores with the bias add, GELU and residual
// add fused into the epilogue where the caller asks for it.
#ifndef DENSE_HPP
#define DENSE_HPP

#include <cmath>

#include "core/Ops.hpp"
#include "layers/Layer.hpp"

class Dense : public Layer {
public:
  Dense(size_t inputSize, size_t outputSize);                          // own storage, N(0, 0.02) / 0 init
  Dense(size_t inputSize, size_t outputSize, vit::ParamArena &arena);  // storage carved from a model arena
  // explicit (possibly strided) views, e.g. one third of a packed QKV weight; no initialisation
  Dense(size_t inputSize, size_t outputSize, Tensor W, Tensor b, Tensor dW, Tensor db);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "Dense"; }

  // Fused forms used by the composite layers.
  //  act = VITK_ACT_GELU_FWD writes the pre-activation to *preact (if given) and returns GELU(.);
  //  residual (same shape as the output) is added in the epilogue.
  Tensor forwardFused(const Tensor &input, bool isTraining, int act, Tensor *preact, const Tensor *residual);
  //  act = VITK_ACT_GELU_BWD multiplies the input gradient by GELU'(*preact) in the epilogue;
  //  needInputGrad = false skips the dX contraction (returns an undefined Tensor).
  Tensor backwardFused(const Tensor &outputGradient, int act, const Tensor *preact, bool needInputGrad);
  void initialise();  // N(0, 0.02) weights, zero bias (src/layers/Dense.cpp:5-19)

private:
  size_t in, out;
  Tensor weights, bias;
  Tensor weightGradients, biasGradients;
  Tensor inputTensor;  // dense copy/alias of the last training input
};

#endif  // DENSE_HPP
