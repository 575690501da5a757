// DataAugmentation: host-side random draws in the reference's order, device-side resampling.
#include "utils/DataAugmentation.hpp"

#include <cmath>

#include "core/Device.hpp"

DataAugmentation::DataAugmentation(const Config &cfg) : config(cfg), rng(static_cast<uint32_t>(vit::next_seed())) {}

void DataAugmentation::draw(size_t count, size_t height, size_t width, std::vector<vitk_augment_params> &out) {
  out.reserve(out.size() + count);
  for (size_t i = 0; i < count; ++i) {
    vitk_augment_params p{};
    p.zoom = 1.0f;
    std::uniform_real_distribution<float> dist(0.0f, 1.0f);
    if (dist(rng) < config.rotation_prob) {
      std::uniform_real_distribution<float> angle_dist(-config.rotation_factor, config.rotation_factor);
      const float angle = angle_dist(rng);
      const float rad = angle * M_PI / 180.0f;  // evaluated in double, as in the reference (:45)
      p.cos_theta = std::cos(rad);
      p.sin_theta = std::sin(rad);
      p.flags |= VITK_AUG_ROTATE;
    }
    if (dist(rng) < config.translate_prob) {
      std::uniform_real_distribution<float> shift(-config.translate_factor, config.translate_factor);
      const float tx = shift(rng) * width;  // x first, then y (:87-88)
      const float ty = shift(rng) * height;
      p.tx = static_cast<int>(tx);
      p.ty = static_cast<int>(ty);
      p.flags |= VITK_AUG_TRANSLATE;
    }
    if (dist(rng) < config.zoom_prob) {
      std::uniform_real_distribution<float> zoom_dist(config.zoom_min, config.zoom_max);
      p.zoom = zoom_dist(rng);
      p.flags |= VITK_AUG_ZOOM;
    }
    out.push_back(p);
  }
}

Tensor DataAugmentation::apply(const Tensor &batch) {
  const auto &shape = batch.getShape();
  if (shape.size() != 4) throw std::invalid_argument("DataAugmentation::apply espera un tensor {B,C,H,W}");
  const size_t B = shape[0];
  Tensor out = Tensor::deviceEmpty(shape);
  if (B == 0) return out;
  std::vector<vitk_augment_params> params;
  draw(B, shape[2], shape[3], params);
  constexpr size_t kWords = sizeof(vitk_augment_params) / sizeof(float);
  Tensor params_dev = Tensor::deviceEmpty({B * kWords});
  vit::copy_to_device(params_dev.deviceDataOverwrite(), reinterpret_cast<const float *>(params.data()), B * kWords);
  Tensor src = batch.dense();
  vit::check(vitk_augment_f32(src.deviceData(), (int64_t)B, nullptr,
                              reinterpret_cast<const vitk_augment_params *>(params_dev.deviceData()), (int64_t)B,
                              (int32_t)shape[1], (int32_t)shape[2], (int32_t)shape[3], out.deviceDataOverwrite(),
                              vit::stream()),
             "vitk_augment_f32");
  return out;
}
