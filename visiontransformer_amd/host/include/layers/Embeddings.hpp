// Patch embeddings + CLS token + positional encoding (reference: include/layers/Embeddings.hpp,
// src/layers/Embeddings.cpp).  Parameter order: W_pe, b_pe, cls {1,1,D}, pos {1,N,D}.
#ifndef EMBEDDINGS_HPP
#define EMBEDDINGS_HPP

#include <memory>

#include "layers/Layer.hpp"
#include "layers/PatchEmbedding.hpp"

class Embeddings : public Layer {
public:
  Embeddings(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels, size_t embedding_dim);
  Embeddings(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels, size_t embedding_dim,
             vit::ParamArena &arena);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "Embeddings"; }
  PatchEmbedding &getPatcher() { return *patcher; }

private:
  void build(vit::ParamArena &arena, size_t image_height, size_t image_width, size_t patch_size, size_t in_channels);
  std::unique_ptr<PatchEmbedding> patcher;
  Tensor clsToken, positionalEncoding, clsTokenGradient, positionalEncodingGradient;
  size_t num_patches, embedding_dim;
};

#endif  // EMBEDDINGS_HPP
