// Pre-norm transformer encoder block (reference: include/model/TransformerEncoderBlock.hpp,
// src/model/TransformerEncoderBlock.cpp).  16 parameter tensors in the reference's order:
// LN1(gamma,beta), q(W,b), k, v, out, LN2(gamma,beta), dense1(W,b), dense2(W,b).  Both residual
// adds ride in GEMM epilogues and both skip-gradient adds in the LayerNorm backward kernels.
#ifndef TRANSFORMERENCODERBLOCK_HPP
#define TRANSFORMERENCODERBLOCK_HPP

#include <vector>

#include "layers/Dropout.hpp"
#include "layers/FeedForward.hpp"
#include "layers/Layer.hpp"
#include "layers/LayerNorm.hpp"
#include "layers/MultiHeadAttention.hpp"

class TransformerEncoderBlock : public Layer {
public:
  TransformerEncoderBlock(size_t embedding_dim, size_t num_heads, size_t mlp_hidden_dim, float dropout_rate);
  TransformerEncoderBlock(size_t embedding_dim, size_t num_heads, size_t mlp_hidden_dim, float dropout_rate,
                          vit::ParamArena &arena);
  TransformerEncoderBlock(TransformerEncoderBlock &&) = default;

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "TransformerEncoderBlock"; }
  MultiHeadAttention &getAttention() { return attention; }

private:
  LayerNorm norm1;
  MultiHeadAttention attention;
  Dropout attention_dropout;
  LayerNorm norm2;
  FeedForward ffn;
  Tensor input_skip1, input_skip2;
};

#endif  // TRANSFORMERENCODERBLOCK_HPP
