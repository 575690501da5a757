// PatchEmbedding and Embeddings on the libvitk kernels.
#include "layers/Embeddings.hpp"
#include "layers/PatchEmbedding.hpp"

namespace {
void *S() { return vit::stream(); }
int g_default_patch_mode = VITK_PATCH_REFERENCE;
}  // namespace

void PatchEmbedding::setDefaultPatchMode(int mode) { g_default_patch_mode = mode; }

PatchEmbedding::PatchEmbedding(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels,
                               size_t embedding_dim)
    : PatchEmbedding(image_height, image_width, patch_size, in_channels, embedding_dim,
                     *std::make_unique<vit::ParamArena>(patch_size * patch_size * in_channels * embedding_dim +
                                                        embedding_dim)) {}

PatchEmbedding::PatchEmbedding(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels,
                               size_t embedding_dim, vit::ParamArena &arena)
    : image_height(image_height), image_width(image_width), patch_size(patch_size), in_channels(in_channels),
      embedding_dim(embedding_dim), patch_mode(g_default_patch_mode) {
  if (patch_size == 0 || image_height % patch_size != 0 || image_width % patch_size != 0)
    throw std::invalid_argument("Las dimensiones de la imagen deben ser divisibles por el tamaño del parche.");
  if (image_height != image_width) throw std::invalid_argument("PatchEmbedding: solo imágenes cuadradas.");
  num_patches_h = image_height / patch_size;
  num_patches_w = image_width / patch_size;
  num_patches = num_patches_h * num_patches_w;
  patch_dim = patch_size * patch_size * in_channels;
  projectionLayer = std::make_unique<Dense>(patch_dim, embedding_dim, arena);
}

Tensor PatchEmbedding::forward(const Tensor &input, bool isTraining) {
  const auto &s = input.getShape();
  if (s.size() != 4 || s[1] != in_channels || s[2] != image_height || s[3] != image_width)
    throw std::runtime_error("PatchEmbedding::forward espera {B,C,H,W} acorde a la configuración.");
  const size_t B = s[0];
  Tensor img = input.dense();
  Tensor patches = Tensor::deviceEmpty({B * num_patches, patch_dim});
  vit::check(vitk_patch_gather(img.deviceData(), (int32_t)B, (int32_t)in_channels, (int32_t)image_height,
                               (int32_t)patch_size, patch_mode, patches.deviceDataOverwrite(), S()),
             "vitk_patch_gather");
  Tensor proj = projectionLayer->forward(patches, isTraining);
  return proj.reshape({B, num_patches, embedding_dim});
}

Tensor PatchEmbedding::backward(const Tensor &outputGradient) {
  const size_t B = outputGradient.getShape()[0];
  Tensor g2 = outputGradient.dense().reshape({B * num_patches, embedding_dim});
  if (!need_input_grad) {  // parameter gradients only
    projectionLayer->backwardFused(g2, VITK_ACT_NONE, nullptr, false);
    return Tensor();
  }
  Tensor dpatches = projectionLayer->backwardFused(g2, VITK_ACT_NONE, nullptr, true);
  Tensor dimg = Tensor::deviceEmpty({B, in_channels, image_height, image_width});
  vit::check(vitk_patch_scatter_bwd(dpatches.deviceData(), (int32_t)B, (int32_t)in_channels, (int32_t)image_height,
                                    (int32_t)patch_size, patch_mode, dimg.deviceDataOverwrite(), S()),
             "vitk_patch_scatter_bwd");
  return dimg;
}

std::vector<Tensor *> PatchEmbedding::getParameters() { return projectionLayer->getParameters(); }
std::vector<Tensor *> PatchEmbedding::getGradients() { return projectionLayer->getGradients(); }

// ------------------------------------------------------------------------------- Embeddings
static size_t emb_count(size_t img, size_t P, size_t C, size_t D) {
  const size_t np = (img / P) * (img / P);
  return P * P * C * D + D + D + (np + 1) * D;
}

Embeddings::Embeddings(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels,
                       size_t embedding_dim)
    : Embeddings(image_height, image_width, patch_size, in_channels, embedding_dim,
                 *std::make_unique<vit::ParamArena>(emb_count(image_height, patch_size, in_channels, embedding_dim))) {}

Embeddings::Embeddings(size_t image_height, size_t image_width, size_t patch_size, size_t in_channels,
                       size_t embedding_dim, vit::ParamArena &arena)
    : embedding_dim(embedding_dim) {
  build(arena, image_height, image_width, patch_size, in_channels);
}

void Embeddings::build(vit::ParamArena &arena, size_t image_height, size_t image_width, size_t patch_size,
                       size_t in_channels) {
  patcher = std::make_unique<PatchEmbedding>(image_height, image_width, patch_size, in_channels, embedding_dim, arena);
  num_patches = patcher->getNumPatches();
  arena.take({1, 1, embedding_dim}, &clsToken, &clsTokenGradient);
  arena.take({1, 1 + num_patches, embedding_dim}, &positionalEncoding, &positionalEncodingGradient);
  Tensor c({1, 1, embedding_dim});
  c.randomizeNormal(0.0f, 0.02f);
  clsToken.copyFrom(c);
  Tensor p({1, 1 + num_patches, embedding_dim});
  p.randomizeNormal(0.0f, 0.02f);
  positionalEncoding.copyFrom(p);
}

Tensor Embeddings::forward(const Tensor &input, bool isTraining) {
  const size_t B = input.getShape()[0];
  Tensor proj = patcher->forward(input, isTraining);  // {B,Np,D}
  Tensor out = Tensor::deviceEmpty({B, num_patches + 1, embedding_dim});
  vit::check(vitk_embed_assemble(proj.deviceData(), clsToken.deviceData(), positionalEncoding.deviceData(), (int32_t)B,
                                 (int32_t)num_patches, (int32_t)embedding_dim, out.deviceDataOverwrite(), S()),
             "vitk_embed_assemble");
  return out;
}

Tensor Embeddings::backward(const Tensor &outputGradient) {
  const auto &s = outputGradient.getShape();
  const size_t B = s[0], N = num_patches + 1, D = embedding_dim;
  Tensor dy = outputGradient.dense();
  // dpos = dY.sum(0) over the batch; dcls = dY[:,0,:].sum(0) = the first row of dpos
  vit::colsum(dy.deviceData(), (int64_t)B, (int64_t)(N * D), (int64_t)(N * D), positionalEncodingGradient.deviceData());
  vit::copy_on_device(clsTokenGradient.deviceData(), positionalEncodingGradient.deviceData(), D);
  Tensor dproj = Tensor::deviceEmpty({B, num_patches, D});
  vit::check(vitk_embed_split_bwd(dy.deviceData(), (int32_t)B, (int32_t)num_patches, (int32_t)D,
                                  dproj.deviceDataOverwrite(), S()),
             "vitk_embed_split_bwd");
  return patcher->backward(dproj);
}

std::vector<Tensor *> Embeddings::getParameters() {
  auto p = patcher->getParameters();
  p.push_back(&clsToken);
  p.push_back(&positionalEncoding);
  return p;
}

std::vector<Tensor *> Embeddings::getGradients() {
  auto g = patcher->getGradients();
  g.push_back(&clsTokenGradient);
  g.push_back(&positionalEncodingGradient);
  return g;
}
