// Checkpoint and configuration IO (reference: include/utils/ModelUtils.hpp,
// src/utils/ModelUtils.cpp).  The .weights format is the reference's, byte for byte: for each
// tensor of getParameters(): u64 rank, u64 dims[rank], f32 data[prod(dims)], native endian, no
// header.  The .json holds the nine ViTConfig fields (keys sorted, 4-space indent).
#ifndef MODELUTILS_HPP
#define MODELUTILS_HPP

#include <string>
#include <vector>

#include "model/Trainer.hpp"
#include "model/VisionTransformer.hpp"

namespace ModelUtils {

void save_weights(const VisionTransformer &model, const std::string &filePath);
// Also reads checkpoints of the reference's CUDA tree (cuda_Vit/models/*.weights): their patch
// projection is a Conv2d weight {D, C, P, P} + bias {1, D, 1, 1}; it is transposed into this
// model's {C*P*P, D} matrix and the model is switched to im2col patch extraction, which is what
// that convolution computes.
void load_weights(VisionTransformer &model, const std::string &filePath);
void save_config(const ViTConfig &config, const std::string &filePath);
ViTConfig load_config(const std::string &filePath);
void print_hyperparameters_box(const ViTConfig &model_config, const TrainerConfig &train_config);

// Flat copies in getParameters() order with every tensor contiguous (the payload order of a
// .weights file without its per-tensor headers).
std::vector<float> flatten(const std::vector<Tensor *> &tensors);
void unflatten(const std::vector<Tensor *> &tensors, const float *flat, size_t count);
// silence the progress messages the reference prints from save/load
void set_verbose(bool on);

}  // namespace ModelUtils

#endif  // MODELUTILS_HPP
