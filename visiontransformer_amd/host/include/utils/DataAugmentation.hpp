// DataAugmentation (reference: include/utils/DataAugmentation.hpp, src/utils/DataAugmentation.cpp):
// per image, with independent probabilities, a random rotation, translation and zoom, each a
// nearest-neighbour resampling with zero fill.  Same Config, same constructor and apply(); the
// random numbers are drawn on the host from a std::mt19937 in exactly the reference's order and
// the resampling runs on the device as one gather per output pixel (vitk_augment_f32).  The
// reference seeds the engine from std::random_device; here the one-argument constructor takes
// its seed from the process seed (vit::seed), and a second constructor takes it explicitly.
#ifndef DATA_AUGMENTATION_HPP
#define DATA_AUGMENTATION_HPP

#include <cstdint>
#include <random>
#include <vector>

#include "core/Tensor.hpp"
#include "vitk.h"

class DataAugmentation {
public:
  struct Config {
    float rotation_prob = 0.5f;
    float translate_prob = 0.5f;
    float zoom_prob = 0.5f;
    float rotation_factor = 10.0f;  // degrees, +/-
    float translate_factor = 0.1f;  // fraction of width / height, +/-
    float zoom_min = 0.9f;
    float zoom_max = 1.1f;
  };

  explicit DataAugmentation(const Config &cfg);
  DataAugmentation(const Config &cfg, uint32_t seed) : config(cfg), rng(seed) {}

  // Augmented copy of a {B,C,H,W} batch.
  Tensor apply(const Tensor &batch);

  // The random draws of `count` consecutive images of height x width pixels, in the order
  // apply() makes them (src/utils/DataAugmentation.cpp:12-22, 38-42, 83-88, 115-116).
  void draw(size_t count, size_t height, size_t width, std::vector<vitk_augment_params> &out);

private:
  Config config;
  std::mt19937 rng;
};

#endif  // DATA_AUGMENTATION_HPP
