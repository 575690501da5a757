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
  // The block when only the FIRST token of its output is consumed (the last block of the model:
  // final norm -> CLS slice -> head, src/model/VisionTransformer.cpp:27-37).  Every per-token
  // stage after the attention core -- output projection, residuals, second norm, feed-forward --
  // then runs on one row per sequence, and so does the backward up to the attention core, whose
  // incoming gradient is zero for all other tokens anyway (src/model/VisionTransformer.cpp:49-56).
  // The values produced for that token, and all parameter gradients, are those of forward() /
  // backward().  input {B,N,D} -> {B,1,D}; backwardFirstToken({B,1,D}) -> dense {B,N,D}.
  bool firstTokenSupported(size_t tokens, bool isTraining) const;
  Tensor forwardFirstToken(const Tensor &input, bool isTraining);
  Tensor backwardFirstToken(const Tensor &outputGradient);

private:
  LayerNorm norm1;
  MultiHeadAttention attention;
  Dropout attention_dropout;
  LayerNorm norm2;
  FeedForward ffn;
  Tensor input_skip1, input_skip2;
};

#endif  // TRANSFORMERENCODERBLOCK_HPP
