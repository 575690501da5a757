// Position-wise MLP: dense1 -> GELU -> dense2 -> dropout (reference: include/layers/FeedForward.hpp,
// src/layers/FeedForward.cpp).  GELU runs in dense1's epilogue (pre-activation saved for backward)
// and GELU' in the epilogue of dense2's input-gradient contraction.
#ifndef FEEDFORWARD_HPP
#define FEEDFORWARD_HPP

#include <vector>

#include "activations/GELU.hpp"
#include "layers/Dense.hpp"
#include "layers/Dropout.hpp"
#include "layers/Layer.hpp"

class FeedForward : public Layer {
public:
  FeedForward(size_t embedding_dim, size_t hidden_dim, float dropout_rate);
  FeedForward(size_t embedding_dim, size_t hidden_dim, float dropout_rate, vit::ParamArena &arena);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "FeedForward"; }

  // forward with `residual + .` fused into dense2's epilogue when dropout is inactive
  // firstTokenOf != 0: input is {B,1,D}, the first token of sequences of that many tokens; the
  // dropout then applies the factors of those rows of the whole {B,tokens,D} tensor
  Tensor forwardFused(const Tensor &input, bool isTraining, const Tensor *residual, size_t firstTokenOf = 0);

private:
  Dense dense1;
  GELU activation;
  Dense dense2;
  Dropout dropout;
  Tensor preActivation;
};

#endif  // FEEDFORWARD_HPP
