// GELU, tanh approximation (reference: include/activations/GELU.hpp, src/activations/GELU.cpp).
#ifndef GELU_HPP
#define GELU_HPP

#include "layers/Layer.hpp"

class GELU : public Layer {
public:
  GELU();
  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::string getName() const override { return "GELU"; }

private:
  Tensor inputTensor;
};

#endif  // GELU_HPP
