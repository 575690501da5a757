// VisionTransformer + ViTConfig (reference: include/model/VisionTransformer.hpp,
// src/model/VisionTransformer.cpp): embeddings -> L encoder blocks -> final LayerNorm over all
// tokens -> CLS row -> classification head.  Parameter order (getParameters): embeddings (4),
// blocks (16 each), final norm (2), head (2).  All parameters and all gradients live in two flat
// device arenas owned by the model.
#ifndef VISIONTRANSFORMER_HPP
#define VISIONTRANSFORMER_HPP

#include <memory>
#include <vector>

#include "layers/Dense.hpp"
#include "layers/Embeddings.hpp"
#include "layers/Layer.hpp"
#include "layers/LayerNorm.hpp"
#include "model/TransformerEncoderBlock.hpp"

struct ViTConfig {
  size_t image_size = 28;
  size_t patch_size = 7;
  size_t in_channels = 1;
  size_t num_classes = 10;
  size_t embedding_dim = 128;
  size_t num_heads = 8;
  size_t num_layers = 4;
  size_t mlp_hidden_dim = 512;
  float dropout_rate = 0.1;
};

// Conversion hooks for nlohmann::json, as include/model/VisionTransformer.hpp:32-57 of the reference
// has them (same nine keys): found by argument-dependent lookup, so `nlohmann::json j = cfg;` and
// `j.get<ViTConfig>()` work for an application that uses that library.  Written as templates so
// this header does not depend on it; when the reference's vendored utils/json.hpp is on the include
// path it is pulled in and the reference's `json` alias is provided too.
#if defined(__has_include)
#if __has_include("utils/json.hpp")
#include "utils/json.hpp"
using json = nlohmann::json;
#endif
#endif

template <typename Json>
void to_json(Json &j, const ViTConfig &config) {
  j = Json{{"image_size", config.image_size},       {"patch_size", config.patch_size},
           {"in_channels", config.in_channels},     {"num_classes", config.num_classes},
           {"embedding_dim", config.embedding_dim}, {"num_heads", config.num_heads},
           {"num_layers", config.num_layers},       {"mlp_hidden_dim", config.mlp_hidden_dim},
           {"dropout_rate", config.dropout_rate}};
}

template <typename Json>
void from_json(const Json &j, ViTConfig &config) {
  j.at("image_size").get_to(config.image_size);
  j.at("patch_size").get_to(config.patch_size);
  j.at("in_channels").get_to(config.in_channels);
  j.at("num_classes").get_to(config.num_classes);
  j.at("embedding_dim").get_to(config.embedding_dim);
  j.at("num_heads").get_to(config.num_heads);
  j.at("num_layers").get_to(config.num_layers);
  j.at("mlp_hidden_dim").get_to(config.mlp_hidden_dim);
  j.at("dropout_rate").get_to(config.dropout_rate);
}

class VisionTransformer : public Layer {
public:
  explicit VisionTransformer(const ViTConfig &config);

  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::vector<Tensor *> getParameters() override;
  std::vector<Tensor *> getGradients() override;
  std::string getName() const override { return "VisionTransformer"; }

  const ViTConfig &getConfig() const { return config; }
  static size_t parameterCount(const ViTConfig &config);

  // ---- extensions used by the data-parallel driver and the parity tests -----------------------
  // Flat arenas (device layout; q/k/v weights packed as one {D,3D} block per layer).
  vit::ParamArena &arena() { return *params; }
  // Backward in stages, in the order gradients become final:
  //   stage 0: head + final norm; stage 1..L: block L-1..0; stage L+1: embeddings.
  size_t numBackwardStages() const { return encoder_blocks.size() + 2; }
  void backwardBegin(const Tensor &outputGradient);
  void backwardStage(size_t stage);
  // backward() returns the image gradient {B,C,H,W} (src/model/VisionTransformer.cpp:68); the staged
  // form used by training steps (which drop it, src/model/Trainer.cpp:212) skips it unless asked
  // with setStageInputGradient(true).
  // Process-wide switch (default on): let the last encoder block and the final norm work on the CLS
  // token only -- the only token the classification head reads (src/model/VisionTransformer.cpp:
  // 27-37) and the only one whose gradient is non-zero (:49-56).  Logits, loss, every parameter
  // gradient and every update are those of the dense computation (to rounding: other reduction
  // lengths); what is skipped is arithmetic whose results nothing reads, and products with zeros.
  static void setFirstTokenTail(bool on);
  static bool firstTokenTail();
  void setStageInputGradient(bool want) { wantInputGradient = want; }
  // [offset, offset+count) of the gradient arena finalised by `stage`
  void stageGradientRange(size_t stage, size_t *offset, size_t *count) const;
  // token stream after embeddings / each block / final norm of the last forward ((L+2) tensors),
  // kept only when tracing is on
  void setTrace(bool on) { trace = on; }
  const std::vector<Tensor> &tracedActivations() const { return traceActs; }
  const std::vector<Tensor> &tracedGradients() const { return traceGrads; }
  std::vector<TransformerEncoderBlock> &blocks() { return encoder_blocks; }
  Embeddings &getEmbeddings() { return embeddings; }

private:
  static std::vector<TransformerEncoderBlock> makeBlocks(const ViTConfig &c, vit::ParamArena &arena,
                                                         std::vector<size_t> &offsets);

private:
  ViTConfig config;
  std::unique_ptr<vit::ParamArena> params;
  std::vector<size_t> stageOffsets;  // arena offsets: [0, end of embeddings, end of block 0, ..., end]
  Embeddings embeddings;
  std::vector<TransformerEncoderBlock> encoder_blocks;
  LayerNorm final_norm;
  Dense mlp_head;
  size_t num_tokens;
  Tensor final_norm_output;
  Tensor stageGrad;
  bool trace = false;
  bool tailForward = false;  // the last training forward used the first-token form of the last block
  bool normFirstOnly = false;  // ... or the dense last block with the final norm on the CLS rows only
  bool wantInputGradient = false;
  std::vector<Tensor> traceActs, traceGrads;
};

#endif  // VISIONTRANSFORMER_HPP
