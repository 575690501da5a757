// Patch extraction + linear projection (reference: include/layers/PatchEmbedding.hpp,
// src/layers/PatchEmbedding.cpp).  Two extraction modes: VITK_PATCH_REFERENCE reproduces the
// reference CPU path (a patch is P*P*C contiguous floats from the patch's top-left offset -- what
// its shipped checkpoints were trained with); VITK_PATCH_IM2COL is the true (c,i,j) patch.
// backward produces the projection's parameter gradients; the image gradient, which the reference
// computes and its Trainer discards, is not produced.
#ifndef PATCHEMBEDDING_HPP
#define PATCHEMBEDDING_HPP

#include <memory>

#include "layers/Dense.hpp"
#include "layers/Layer.hpp"

class PatchEmbedding : public Layer {
public:
  PatchEmbedding(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels, size_t embedding_dim);
  PatchEmbedding(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels, size_t embedding_dim,
                 vit::ParamArena &arena);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "PatchEmbedding"; }
  size_t getNumPatches() const { return num_patches; }
  static void setDefaultPatchMode(int mode);
  void setPatchMode(int mode) { patch_mode = mode; }
  // backward() returns the image gradient {B,C,H,W} like the reference's; a caller that drops it
  // (the training step: src/model/Trainer.cpp:212) turns its computation off and gets an undefined
  // tensor back.
  void setNeedInputGradient(bool need) { need_input_grad = need; }

private:
  size_t image_height, image_width, patch_size, in_channels, embedding_dim;
  size_t patch_dim, num_patches, num_patches_h, num_patches_w;
  int patch_mode;
  bool need_input_grad = true;
  std::unique_ptr<Dense> projectionLayer;
};

#endif  // PATCHEMBEDDING_HPP
