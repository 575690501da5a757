// TransformerEncoderBlock and VisionTransformer.
#include "model/TransformerEncoderBlock.hpp"
#include "model/VisionTransformer.hpp"

namespace {
void *S() { return vit::stream(); }
size_t block_count(size_t D, size_t H) { return 2 * D + 4 * (D * D + D) + 2 * D + (D * H + H) + (H * D + D); }
}  // namespace

// ------------------------------------------------------------------ TransformerEncoderBlock
TransformerEncoderBlock::TransformerEncoderBlock(size_t embedding_dim, size_t num_heads, size_t mlp_hidden_dim,
                                                 float dropout_rate)
    : TransformerEncoderBlock(embedding_dim, num_heads, mlp_hidden_dim, dropout_rate,
                              *std::make_unique<vit::ParamArena>(block_count(embedding_dim, mlp_hidden_dim))) {}

TransformerEncoderBlock::TransformerEncoderBlock(size_t embedding_dim, size_t num_heads, size_t mlp_hidden_dim,
                                                 float dropout_rate, vit::ParamArena &arena)
    : norm1(embedding_dim, arena), attention(embedding_dim, num_heads, arena), attention_dropout(dropout_rate),
      norm2(embedding_dim, arena), ffn(embedding_dim, mlp_hidden_dim, dropout_rate, arena) {}

Tensor TransformerEncoderBlock::forward(const Tensor &input, bool isTraining) {
  if (isTraining) input_skip1 = input;
  // residual1 = input + dropout(MHA(LN1(input)))
  Tensor x = norm1.forward(input, isTraining);
  Tensor residual1;
  if (attention_dropout.active(isTraining)) {
    x = attention.forward(x, isTraining);
    residual1 = attention_dropout.forwardAdd(x, input, isTraining);
  } else {
    residual1 = attention.forwardFused(x, isTraining, &input);  // add fused into out_proj's epilogue
  }
  if (isTraining) input_skip2 = residual1;
  // out = residual1 + FFN(LN2(residual1))
  Tensor y = norm2.forward(residual1, isTraining);
  return ffn.forwardFused(y, isTraining, &residual1);
}

Tensor TransformerEncoderBlock::backward(const Tensor &outputGradient) {
  // out = r1 + FFN(LN2(r1)):  d r1 = dOut + LN2'(FFN'(dOut))
  Tensor g = ffn.backward(outputGradient);
  Tensor grad1 = norm2.backwardFused(g, &outputGradient);
  // r1 = in + drop(MHA(LN1(in))):  d in = d r1 + LN1'(MHA'(drop'(d r1)))
  Tensor gm = attention_dropout.backward(grad1);
  gm = attention.backward(gm);
  return norm1.backwardFused(gm, &grad1);
}

bool TransformerEncoderBlock::firstTokenSupported(size_t tokens, bool isTraining) const {
  (void)isTraining;  // (active dropout: the first token takes the factors of its rows of the full {B,N,D} masks)
  return attention.firstTokenSupported(tokens);
}

Tensor TransformerEncoderBlock::forwardFirstToken(const Tensor &input, bool isTraining) {
  const auto &s = input.getShape();
  const size_t B = s[0], N = s[1], D = s[2];
  Tensor in = input.dense();
  Tensor x = norm1.forward(in, isTraining);  // all tokens: keys and values need them
  Tensor in0 = Tensor::deviceEmpty({B, 1, D});
  vit::check(vitk_copy2d_f32(in.deviceData(), (int64_t)(N * D), in0.deviceDataOverwrite(), (int64_t)D, (int64_t)B,
                             (int64_t)D, S()),
             "vitk_copy2d_f32");
  Tensor residual1;  // {B,1,D}
  if (attention_dropout.active(isTraining)) {
    Tensor a = attention.forwardFirstToken(x, isTraining, nullptr);
    residual1 = attention_dropout.forwardAddFirstToken(a, &in0, N, isTraining);
  } else {
    residual1 = attention.forwardFirstToken(x, isTraining, &in0);  // add fused into out_proj's epilogue
  }
  Tensor y = norm2.forward(residual1, isTraining);
  return ffn.forwardFused(y, isTraining, &residual1, N);
}

Tensor TransformerEncoderBlock::backwardFirstToken(const Tensor &outputGradient) {
  Tensor g = ffn.backward(outputGradient);                   // {B,1,D}
  Tensor grad1 = norm2.backwardFused(g, &outputGradient);    // d residual1 of the first token
  Tensor gm = attention.backwardFirstToken(attention_dropout.backward(grad1));  // dense {B,N,D}: keys / values of every token
  Tensor dx = norm1.backwardFused(gm, nullptr);
  const auto &s = dx.getShape();
  // skip connection of the first token: d input[:,0,:] += d residual1
  vit::check(vitk_add2d_f32(grad1.deviceData(), (int64_t)s[2], dx.deviceData(), (int64_t)(s[1] * s[2]), (int64_t)s[0],
                            (int64_t)s[2], S()),
             "vitk_add2d_f32");
  return dx;
}

std::vector<Tensor *> TransformerEncoderBlock::getParameters() {
  std::vector<Tensor *> params;
  for (Layer *l : std::initializer_list<Layer *>{&norm1, &attention, &norm2, &ffn}) {
    auto p = l->getParameters();
    params.insert(params.end(), p.begin(), p.end());
  }
  return params;
}

std::vector<Tensor *> TransformerEncoderBlock::getGradients() {
  std::vector<Tensor *> grads;
  for (Layer *l : std::initializer_list<Layer *>{&norm1, &attention, &norm2, &ffn}) {
    auto g = l->getGradients();
    grads.insert(grads.end(), g.begin(), g.end());
  }
  return grads;
}

// ------------------------------------------------------------------------ VisionTransformer
size_t VisionTransformer::parameterCount(const ViTConfig &c) {
  const size_t np = (c.image_size / c.patch_size) * (c.image_size / c.patch_size), D = c.embedding_dim;
  return c.patch_size * c.patch_size * c.in_channels * D + D + D + (np + 1) * D +
         c.num_layers * block_count(D, c.mlp_hidden_dim) + 2 * D + D * c.num_classes + c.num_classes;
}

std::vector<TransformerEncoderBlock> VisionTransformer::makeBlocks(const ViTConfig &c, vit::ParamArena &arena,
                                                                   std::vector<size_t> &offsets) {
  std::vector<TransformerEncoderBlock> blocks;
  blocks.reserve(c.num_layers);
  offsets.assign({0, arena.used()});  // the embeddings were carved first
  for (size_t i = 0; i < c.num_layers; ++i) {
    blocks.emplace_back(c.embedding_dim, c.num_heads, c.mlp_hidden_dim, c.dropout_rate, arena);
    offsets.push_back(arena.used());
  }
  return blocks;
}

// Members are constructed in declaration order (embeddings, blocks, final norm, head), which is
// also the order their tensors are carved from the arena = the reference's getParameters() order.
VisionTransformer::VisionTransformer(const ViTConfig &config)
    : config(config), params(std::make_unique<vit::ParamArena>(parameterCount(config))),
      embeddings(config.image_size, config.image_size, config.patch_size, config.in_channels, config.embedding_dim,
                 *params),
      encoder_blocks(makeBlocks(config, *params, stageOffsets)), final_norm(config.embedding_dim, *params),
      mlp_head(config.embedding_dim, config.num_classes, *params) {
  num_tokens = 1 + (config.image_size / config.patch_size) * (config.image_size / config.patch_size);
  stageOffsets.push_back(params->used());
  if (params->used() != params->size()) throw std::runtime_error("VisionTransformer: arena size mismatch");
}

namespace {
bool g_first_token_tail = true;
}  // namespace
void VisionTransformer::setFirstTokenTail(bool on) { g_first_token_tail = on; }
bool VisionTransformer::firstTokenTail() { return g_first_token_tail; }

Tensor VisionTransformer::forward(const Tensor &input, bool isTraining) {
  if (trace) traceActs.clear();
  Tensor x = embeddings.forward(input, isTraining);
  if (trace) traceActs.push_back(x);
  const size_t L = encoder_blocks.size(), B = input.getShape()[0];
  // Only the CLS token of the last block's output is read below, so that block (and the final
  // norm) can work on one row per sequence.  Tracing wants every token of every layer: dense then.
  const bool tail = g_first_token_tail && !trace && L > 0 && encoder_blocks[L - 1].firstTokenSupported(num_tokens, isTraining);
  for (size_t i = 0; i + (tail ? 1 : 0) < L; ++i) {
    x = encoder_blocks[i].forward(x, isTraining);
    if (trace) traceActs.push_back(x);
  }
  if (isTraining) tailForward = tail;
  if (tail) {
    x = encoder_blocks[L - 1].forwardFirstToken(x, isTraining);  // {B,1,D}
    x = final_norm.forward(x, isTraining);
    if (isTraining) final_norm_output = x;
    return mlp_head.forward(x.reshape({B, config.embedding_dim}), isTraining);
  }
  if (isTraining) normFirstOnly = !trace;
  if (!trace) {
    // The head reads the CLS row of the normalised sequence and LayerNorm works row by row: pick
    // the row first and normalise B rows instead of B*N (src/model/VisionTransformer.cpp:34-40).
    Tensor cls = Tensor::deviceEmpty({B, 1, config.embedding_dim});
    vit::check(vitk_cls_gather(x.deviceData(), (int32_t)B, (int32_t)num_tokens, (int32_t)config.embedding_dim,
                               cls.deviceDataOverwrite(), S()),
               "vitk_cls_gather");
    cls = final_norm.forward(cls, isTraining);
    if (isTraining) final_norm_output = cls;
    return mlp_head.forward(cls.reshape({B, config.embedding_dim}), isTraining);
  }
  x = final_norm.forward(x, isTraining);
  traceActs.push_back(x);
  if (isTraining) final_norm_output = x;
  Tensor cls = Tensor::deviceEmpty({B, config.embedding_dim});
  vit::check(vitk_cls_gather(x.deviceData(), (int32_t)B, (int32_t)num_tokens, (int32_t)config.embedding_dim,
                             cls.deviceDataOverwrite(), S()),
             "vitk_cls_gather");
  return mlp_head.forward(cls, isTraining);
}

void VisionTransformer::backwardBegin(const Tensor &outputGradient) {
  stageGrad = outputGradient;
  if (trace) traceGrads.clear();
}

void VisionTransformer::backwardStage(size_t stage) {
  const size_t L = encoder_blocks.size();
  if (stage == 0 && tailForward) {
    Tensor g = mlp_head.backward(stageGrad);  // {B,D}: the gradient of the CLS rows, all others are zero
    const size_t B = g.getShape()[0];
    stageGrad = final_norm.backward(g.reshape({B, 1, config.embedding_dim}));
    if (trace) {  // traced gradients are dense {B,N,D}: zero except the CLS rows
      Tensor seq = Tensor::deviceEmpty({B, num_tokens, config.embedding_dim});
      vit::check(vitk_cls_scatter(stageGrad.deviceData(), (int32_t)B, (int32_t)num_tokens, (int32_t)config.embedding_dim,
                                  seq.deviceDataOverwrite(), S()),
                 "vitk_cls_scatter");
      traceGrads.push_back(seq);
    }
    return;
  } else if (stage == 1 && tailForward) {
    stageGrad = encoder_blocks[L - 1].backwardFirstToken(stageGrad);
  } else if (stage == 0) {
    Tensor g = mlp_head.backward(stageGrad);  // {B,D}
    const size_t B = g.getShape()[0];
    // re-inject the CLS gradient into a zero {B,N,D} sequence (src/model/VisionTransformer.cpp:49-56)
    Tensor seq = Tensor::deviceEmpty({B, num_tokens, config.embedding_dim});
    if (normFirstOnly) {  // rows with a zero gradient add exactly nothing to dgamma / dbeta and get dx = 0
      g = final_norm.backward(g.reshape({B, 1, config.embedding_dim}));
      vit::check(vitk_cls_scatter(g.deviceData(), (int32_t)B, (int32_t)num_tokens, (int32_t)config.embedding_dim,
                                  seq.deviceDataOverwrite(), S()),
                 "vitk_cls_scatter");
      stageGrad = seq;
    } else {
      vit::check(vitk_cls_scatter(g.deviceData(), (int32_t)B, (int32_t)num_tokens, (int32_t)config.embedding_dim,
                                  seq.deviceDataOverwrite(), S()),
                 "vitk_cls_scatter");
      stageGrad = final_norm.backward(seq);
    }
  } else if (stage <= L) {
    stageGrad = encoder_blocks[L - stage].backward(stageGrad);
  } else if (stage == L + 1) {
    embeddings.getPatcher().setNeedInputGradient(wantInputGradient);
    stageGrad = embeddings.backward(stageGrad);
    return;
  } else {
    throw std::out_of_range("VisionTransformer::backwardStage: stage fuera de rango.");
  }
  if (trace) traceGrads.push_back(stageGrad);
}

void VisionTransformer::stageGradientRange(size_t stage, size_t *offset, size_t *count) const {
  const size_t L = encoder_blocks.size();
  size_t lo, hi;
  if (stage == 0) {
    lo = stageOffsets[L + 1]; hi = stageOffsets[L + 2];
  } else if (stage <= L) {
    lo = stageOffsets[L - stage + 1]; hi = stageOffsets[L - stage + 2];
  } else if (stage == L + 1) {
    lo = stageOffsets[0]; hi = stageOffsets[1];
  } else {
    throw std::out_of_range("VisionTransformer::stageGradientRange: stage fuera de rango.");
  }
  *offset = lo;
  *count = hi - lo;
}

Tensor VisionTransformer::backward(const Tensor &outputGradient) {
  backwardBegin(outputGradient);
  const bool staged_default = wantInputGradient;
  wantInputGradient = true;  // Layer::backward returns the input gradient, as in the reference
  for (size_t s = 0; s < numBackwardStages(); ++s) backwardStage(s);
  wantInputGradient = staged_default;
  Tensor out = stageGrad;
  stageGrad = Tensor();
  return out;
}

std::vector<Tensor *> VisionTransformer::getParameters() {
  std::vector<Tensor *> params_;
  auto e = embeddings.getParameters();
  params_.insert(params_.end(), e.begin(), e.end());
  for (auto &block : encoder_blocks) {
    auto b = block.getParameters();
    params_.insert(params_.end(), b.begin(), b.end());
  }
  auto n = final_norm.getParameters();
  params_.insert(params_.end(), n.begin(), n.end());
  auto h = mlp_head.getParameters();
  params_.insert(params_.end(), h.begin(), h.end());
  return params_;
}

std::vector<Tensor *> VisionTransformer::getGradients() {
  std::vector<Tensor *> grads;
  auto e = embeddings.getGradients();
  grads.insert(grads.end(), e.begin(), e.end());
  for (auto &block : encoder_blocks) {
    auto b = block.getGradients();
    grads.insert(grads.end(), b.begin(), b.end());
  }
  auto n = final_norm.getGradients();
  grads.insert(grads.end(), n.begin(), n.end());
  auto h = mlp_head.getGradients();
  grads.insert(grads.end(), h.begin(), h.end());
  return grads;
}
