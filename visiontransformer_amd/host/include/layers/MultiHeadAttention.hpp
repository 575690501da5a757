// Multi-head self-attention (reference: include/layers/MultiHeadAttention.hpp,
// src/layers/MultiHeadAttention.cpp).  The four projections keep the reference's parameter order
// (q W,b, k W,b, v W,b, out W,b), but q/k/v weights live side by side in one packed [D][3D] block
// (the Dense members hold strided views of it), so the projection is a single contraction, head
// split/merge is address arithmetic on the packed [B*N][3D] activations, and the sum of the three
// input gradients falls out of one contraction over 3D.
#ifndef MULTIHEADATTENTION_HPP
#define MULTIHEADATTENTION_HPP

#include <memory>
#include <vector>

#include "layers/Dense.hpp"
#include "layers/Layer.hpp"

class MultiHeadAttention : public Layer {
public:
  MultiHeadAttention(size_t embedding_dim, size_t num_heads);
  MultiHeadAttention(size_t embedding_dim, size_t num_heads, vit::ParamArena &arena);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "MultiHeadAttention"; }

  Tensor forwardFused(const Tensor &input, bool isTraining, const Tensor *residual);
  // Attention whose output is wanted for the FIRST token only (the last encoder block: only its
  // CLS token reaches the classification head, src/model/VisionTransformer.cpp:27-37).  Keys and
  // values are projected for all tokens, the query, the attention core and the output projection
  // for that one token: input {B,N,D} -> {B,1,D} (+ residual {B,1,D}).  backwardFirstToken takes the
  // gradient of that output and returns the dense {B,N,D} input gradient.
  bool firstTokenSupported(size_t tokens) const;
  Tensor forwardFirstToken(const Tensor &input, bool isTraining, const Tensor *residual);
  Tensor backwardFirstToken(const Tensor &outputGradient);
  // VITK_DK_REFERENCE (default) reproduces the reference's key-gradient layout; VITK_DK_EXACT is
  // the mathematically correct gradient.  Process-wide default for newly built layers.
  static void setDefaultKeyGradMode(int mode);
  void setKeyGradMode(int mode) { dk_mode = mode; }
  // Fused attention kernels (default, where the shape is supported) or the chain of batched
  // contractions + row softmax that materialises the probabilities.
  static void setDefaultFused(bool on);
  void setFused(bool on) { use_fused = on; }

private:
  void build(vit::ParamArena &arena, bool initialise);
  Tensor projectionBackward(const Tensor &dqkv);
  size_t embedding_dim, num_heads, head_dim;
  int dk_mode;
  bool use_fused;
  Tensor wqkv, bqkv, dwqkv, dbqkv;       // packed {D,3D} / {1,3D}
  std::unique_ptr<Dense> q_proj, k_proj, v_proj, out_proj;
  Tensor inputTensor, qkv, attention_weights, softmax_stats, context;
  Tensor firstInput, firstQuery;  // first-token form: {B,D} input rows and their query projection
};

#endif  // MULTIHEADATTENTION_HPP
