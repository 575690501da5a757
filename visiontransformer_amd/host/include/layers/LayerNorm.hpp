// LayerNorm over the last axis (reference: include/layers/LayerNorm.hpp, src/layers/LayerNorm.cpp):
// two-pass mean / biased variance, eps inside the sqrt, gamma/beta {1,D}.  Saves the input plus
// per-row mean and 1/std; x-hat is recomputed in backward.
#ifndef LAYERNORM_HPP
#define LAYERNORM_HPP

#include "core/Ops.hpp"
#include "layers/Layer.hpp"

class LayerNorm : public Layer {
public:
  LayerNorm(size_t featureSize, float epsilon = 1e-5f);
  LayerNorm(size_t featureSize, vit::ParamArena &arena, float epsilon = 1e-5f);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "LayerNorm"; }

  // backward with the skip-branch gradient added to dx in the same pass
  Tensor backwardFused(const Tensor &outputGradient, const Tensor *residual);

private:
  void initialise();
  size_t featureSize;
  float epsilon;
  Tensor gamma, beta, gammaGradient, betaGradient;
  Tensor inputTensor, mean, variance;  // `variance` holds 1/sqrt(var+eps), as in the reference
};

#endif  // LAYERNORM_HPP
