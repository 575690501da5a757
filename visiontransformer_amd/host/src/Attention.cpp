// MultiHeadAttention on the libvitk kernels: packed QKV projection, per-head batched contractions
// addressed straight inside the packed [B*N][3D] activations, row softmax, and the full backward.
#include "layers/MultiHeadAttention.hpp"

#include <cmath>

using vit::GemmOpts;

namespace {
void *S() { return vit::stream(); }
int g_default_dk_mode = VITK_DK_REFERENCE;
bool g_default_fused = true;
}  // namespace

void MultiHeadAttention::setDefaultKeyGradMode(int mode) { g_default_dk_mode = mode; }
void MultiHeadAttention::setDefaultFused(bool on) { g_default_fused = on; }

MultiHeadAttention::MultiHeadAttention(size_t embedding_dim, size_t num_heads)
    : MultiHeadAttention(embedding_dim, num_heads,
                         *std::make_unique<vit::ParamArena>(4 * (embedding_dim * embedding_dim + embedding_dim))) {}

MultiHeadAttention::MultiHeadAttention(size_t embedding_dim, size_t num_heads, vit::ParamArena &arena)
    : embedding_dim(embedding_dim), num_heads(num_heads), dk_mode(g_default_dk_mode), use_fused(g_default_fused) {
  if (num_heads == 0 || embedding_dim % num_heads != 0)
    throw std::invalid_argument("embedding_dim debe ser divisible por num_heads.");
  head_dim = embedding_dim / num_heads;
  build(arena, true);
}

void MultiHeadAttention::build(vit::ParamArena &arena, bool initialise) {
  const size_t D = embedding_dim;
  // packed block: W_qkv {D,3D} then b_qkv {1,3D}; q/k/v Dense objects get strided views of it
  arena.take({D, 3 * D}, &wqkv, &dwqkv);
  arena.take({1, 3 * D}, &bqkv, &dbqkv);
  auto third = [&](const Tensor &t, size_t i, bool weight) {
    const size_t off = t.getDataOffset() + i * D;
    return weight ? Tensor(t.storage(), {D, D}, {3 * D, 1}, off) : Tensor(t.storage(), {1, D}, {D, 1}, off);
  };
  q_proj = std::make_unique<Dense>(D, D, third(wqkv, 0, true), third(bqkv, 0, false), third(dwqkv, 0, true),
                                   third(dbqkv, 0, false));
  k_proj = std::make_unique<Dense>(D, D, third(wqkv, 1, true), third(bqkv, 1, false), third(dwqkv, 1, true),
                                   third(dbqkv, 1, false));
  v_proj = std::make_unique<Dense>(D, D, third(wqkv, 2, true), third(bqkv, 2, false), third(dwqkv, 2, true),
                                   third(dbqkv, 2, false));
  out_proj = std::make_unique<Dense>(D, D, arena);
  if (initialise) {
    Tensor w({D, 3 * D});
    w.randomizeNormal(0.0f, 0.02f);
    wqkv.copyFrom(w);
    bqkv.fill(0.0f);
  }
}

Tensor MultiHeadAttention::forward(const Tensor &input, bool isTraining) {
  return forwardFused(input, isTraining, nullptr);
}

Tensor MultiHeadAttention::forwardFused(const Tensor &input, bool isTraining, const Tensor *residual) {
  const auto &s = input.getShape();
  if (s.size() != 3 || s[2] != embedding_dim) throw std::runtime_error("MultiHeadAttention::forward espera {B,N,D}.");
  const int64_t B = (int64_t)s[0], N = (int64_t)s[1], D = (int64_t)embedding_dim, h = (int64_t)num_heads,
                dh = (int64_t)head_dim, M = B * N, BH = B * h;
  Tensor x = input.dense();
  // 1. q|k|v = x W_qkv + b_qkv   -> packed {M,3D}
  Tensor qkvT = Tensor::deviceEmpty({(size_t)M, (size_t)(3 * D)});
  GemmOpts o1;
  o1.bias = bqkv.deviceData();
  vit::gemm(false, false, M, 3 * D, D, x.deviceData(), D, wqkv.deviceData(), 3 * D, qkvT.deviceDataOverwrite(), 3 * D, o1);
  const float *q = qkvT.deviceData(), *k = q + D, *v = q + 2 * D;
  const float scale = 1.0f / std::sqrt(static_cast<float>(head_dim));
  if (use_fused && vitk_sdpa_supported((int32_t)N, (int32_t)dh)) {
    // 2+3 in one launch: scores, softmax and P V stay in registers; only the row statistics of the
    // softmax are kept for the backward.
    Tensor ctx = Tensor::deviceEmpty({(size_t)B, (size_t)N, (size_t)D});
    Tensor stats = isTraining ? Tensor::deviceEmpty({(size_t)BH, (size_t)N, 2}) : Tensor();
    vit::check(vitk_sdpa_fwd(q, k, v, 3 * D, (int32_t)B, (int32_t)h, (int32_t)N, (int32_t)dh, scale,
                             ctx.deviceDataOverwrite(), D, isTraining ? stats.deviceDataOverwrite() : nullptr, S()),
               "vitk_sdpa_fwd");
    if (isTraining) {
      inputTensor = x;
      qkv = qkvT;
      attention_weights = Tensor();
      softmax_stats = stats;
      context = ctx;
    }
    return out_proj->forwardFused(ctx, isTraining, VITK_ACT_NONE, nullptr, residual);
  }
  // 2. scores = (Q K^T) / sqrt(dh), softmax over keys (rows padded to a multiple of 4 floats)
  const int64_t ld = (N + 3) / 4 * 4;
  Tensor P = Tensor::deviceEmpty({(size_t)BH, (size_t)N, (size_t)ld});
  GemmOpts o2;
  o2.alpha = scale;
  o2.batch_outer = (int)B; o2.batch_inner = (int)h;
  o2.sa_o = N * 3 * D; o2.sa_i = dh; o2.sb_o = N * 3 * D; o2.sb_i = dh; o2.sc_o = h * N * ld; o2.sc_i = N * ld;
  float *Pd = P.deviceDataOverwrite();
  vit::gemm(false, true, N, N, dh, q, 3 * D, k, 3 * D, Pd, ld, o2);
  vit::check(vitk_softmax_rows(Pd, BH * N, N, ld, Pd, S()), "vitk_softmax_rows");
  // 3. context = P V, written head-merged into {M,D}
  Tensor ctx = Tensor::deviceEmpty({(size_t)B, (size_t)N, (size_t)D});
  GemmOpts o3;
  o3.batch_outer = (int)B; o3.batch_inner = (int)h;
  o3.sa_o = h * N * ld; o3.sa_i = N * ld; o3.sb_o = N * 3 * D; o3.sb_i = dh; o3.sc_o = N * D; o3.sc_i = dh;
  vit::gemm(false, false, N, dh, N, Pd, ld, v, 3 * D, ctx.deviceDataOverwrite(), D, o3);
  if (isTraining) {
    inputTensor = x;
    qkv = qkvT;
    attention_weights = P;
    softmax_stats = Tensor();
    context = ctx;
  }
  // 4. output projection (+ residual of the enclosing block)
  return out_proj->forwardFused(ctx, isTraining, VITK_ACT_NONE, nullptr, residual);
}

bool MultiHeadAttention::firstTokenSupported(size_t tokens) const {
  return use_fused && vitk_sdpa_row_supported((int32_t)tokens, (int32_t)head_dim) && embedding_dim % 4 == 0;
}

Tensor MultiHeadAttention::forwardFirstToken(const Tensor &input, bool isTraining, const Tensor *residual) {
  const auto &s = input.getShape();
  if (s.size() != 3 || s[2] != embedding_dim) throw std::runtime_error("MultiHeadAttention::forward espera {B,N,D}.");
  const int64_t B = (int64_t)s[0], N = (int64_t)s[1], D = (int64_t)embedding_dim, h = (int64_t)num_heads,
                dh = (int64_t)head_dim, M = B * N;
  Tensor x = input.dense();
  // k|v = x W_kv + b_kv for every token: the last two thirds of the packed projection
  Tensor qkvT = Tensor::deviceEmpty({(size_t)M, (size_t)(3 * D)});
  GemmOpts okv;
  okv.bias = bqkv.deviceData() + D;
  vit::gemm(false, false, M, 2 * D, D, x.deviceData(), D, wqkv.deviceData() + D, 3 * D, qkvT.deviceDataOverwrite() + D,
            3 * D, okv);
  // q = x[:,0,:] W_q + b_q for the first token of every sequence
  Tensor x0 = Tensor::deviceEmpty({(size_t)B, (size_t)D});
  vit::check(vitk_copy2d_f32(x.deviceData(), N * D, x0.deviceDataOverwrite(), D, B, D, S()), "vitk_copy2d_f32");
  Tensor q0 = Tensor::deviceEmpty({(size_t)B, (size_t)D});
  GemmOpts oq;
  oq.bias = bqkv.deviceData();
  vit::gemm(false, false, B, D, D, x0.deviceData(), D, wqkv.deviceData(), 3 * D, q0.deviceDataOverwrite(), D, oq);
  const float *k = qkvT.deviceData() + D, *v = k + D;
  Tensor ctx = Tensor::deviceEmpty({(size_t)B, 1, (size_t)D});
  Tensor stats = isTraining ? Tensor::deviceEmpty({(size_t)(B * h), 2}) : Tensor();
  vit::check(vitk_sdpa_row_fwd(q0.deviceData(), D, k, v, 3 * D, (int32_t)B, (int32_t)h, (int32_t)N, (int32_t)dh,
                               1.0f / std::sqrt(static_cast<float>(head_dim)), ctx.deviceDataOverwrite(), D,
                               isTraining ? stats.deviceDataOverwrite() : nullptr, S()),
             "vitk_sdpa_row_fwd");
  if (isTraining) {
    inputTensor = x;
    qkv = qkvT;
    attention_weights = Tensor();
    softmax_stats = stats;
    context = ctx;
    firstInput = x0;
    firstQuery = q0;
  }
  return out_proj->forwardFused(ctx, isTraining, VITK_ACT_NONE, nullptr, residual);
}

Tensor MultiHeadAttention::backwardFirstToken(const Tensor &outputGradient) {
  if (!firstQuery.defined()) throw std::runtime_error("MultiHeadAttention::backward sin forward de entrenamiento.");
  const auto &s = inputTensor.getShape();
  const int64_t B = (int64_t)s[0], N = (int64_t)s[1], D = (int64_t)embedding_dim, h = (int64_t)num_heads,
                dh = (int64_t)head_dim, M = B * N;
  Tensor dctx = out_proj->backward(outputGradient);  // {B,1,D}
  Tensor dqkv = Tensor::deviceEmpty({(size_t)M, (size_t)(3 * D)});  // only the k|v columns are produced
  Tensor dq0 = Tensor::deviceEmpty({(size_t)B, (size_t)D});
  const float *k = qkv.deviceData() + D, *v = k + D;
  float *dk = dqkv.deviceDataOverwrite() + D, *dv = dk + D;
  vit::check(vitk_sdpa_row_bwd(firstQuery.deviceData(), D, k, v, 3 * D, context.deviceData(), dctx.deviceData(), D,
                               softmax_stats.deviceData(), (int32_t)B, (int32_t)h, (int32_t)N, (int32_t)dh,
                               1.0f / std::sqrt(static_cast<float>(head_dim)), dk_mode, dq0.deviceDataOverwrite(), D, dk,
                               dv, 3 * D, S()),
             "vitk_sdpa_row_bwd");
  const float *x = inputTensor.deviceData();
  {  // dW_kv = X^T dKV over all tokens (+ bias gradients from the same launch)
    GemmOpts o;
    o.split_k = 0;
    o.colsum_b = dbqkv.deviceData() + D;
    vit::gemm(true, false, D, 2 * D, M, x, D, dk, 3 * D, dwqkv.deviceData() + D, 3 * D, o);
  }
  {  // dW_q = X0^T dQ0 over the first tokens
    GemmOpts o;
    o.colsum_b = dbqkv.deviceData();
    vit::gemm(true, false, D, D, B, firstInput.deviceData(), D, dq0.deviceData(), D, dwqkv.deviceData(), 3 * D, o);
  }
  // dX = dKV W_kv^T for every token, plus dQ0 W_q^T added into the first token's rows
  Tensor dx = Tensor::deviceEmpty(inputTensor.getShape());
  float *dxp = dx.deviceDataOverwrite();
  vit::gemm(false, true, M, D, 2 * D, dk, 3 * D, wqkv.deviceData() + D, 3 * D, dxp, D);
  {
    GemmOpts o;
    o.residual = dxp;
    o.ldr = N * D;
    vit::gemm(false, true, B, D, D, dq0.deviceData(), D, wqkv.deviceData(), 3 * D, dxp, N * D, o);
  }
  return dx;
}

Tensor MultiHeadAttention::backward(const Tensor &outputGradient) {
  if (!inputTensor.defined()) throw std::runtime_error("MultiHeadAttention::backward sin forward de entrenamiento.");
  const auto &s = inputTensor.getShape();
  const int64_t B = (int64_t)s[0], N = (int64_t)s[1], D = (int64_t)embedding_dim, h = (int64_t)num_heads,
                dh = (int64_t)head_dim, M = B * N, BH = B * h;
  // 1. output projection
  Tensor dctx = out_proj->backward(outputGradient);  // {B,N,D}
  const float *dc = dctx.deviceData();
  const float *q = qkv.deviceData(), *k = q + D, *v = q + 2 * D;
  Tensor dqkv = Tensor::deviceEmpty({(size_t)M, (size_t)(3 * D)});
  float *dq = dqkv.deviceDataOverwrite(), *dk = dq + D, *dv = dq + 2 * D;
  if (softmax_stats.defined()) {
    // 2-5 fused: probabilities are recomputed from the saved row statistics
    const size_t wsb = vitk_sdpa_bwd_workspace_bytes((int32_t)B, (int32_t)h, (int32_t)N);
    vit::check(vitk_sdpa_bwd(q, k, v, 3 * D, context.deviceData(), dc, D, softmax_stats.deviceData(), (int32_t)B,
                             (int32_t)h, (int32_t)N, (int32_t)dh, 1.0f / std::sqrt(static_cast<float>(head_dim)),
                             dk_mode, dq, dk, dv, 3 * D, vit::workspace(wsb), wsb, S()),
               "vitk_sdpa_bwd");
    return projectionBackward(dqkv);
  }
  const int64_t ld = (int64_t)attention_weights.getShape()[2];
  const float *P = attention_weights.deviceData();
  const int64_t tok = N * 3 * D, sP_o = h * N * ld, sP_i = N * ld;
  auto heads = [&](GemmOpts &o) { o.batch_outer = (int)B; o.batch_inner = (int)h; };
  // 2. dP = dctx V^T ; dV = P^T dctx
  Tensor dP = Tensor::deviceEmpty({(size_t)BH, (size_t)N, (size_t)ld});
  float *dPd = dP.deviceDataOverwrite();
  {
    GemmOpts o; heads(o);
    o.sa_o = N * D; o.sa_i = dh; o.sb_o = tok; o.sb_i = dh; o.sc_o = sP_o; o.sc_i = sP_i;
    vit::gemm(false, true, N, N, dh, dc, D, v, 3 * D, dPd, ld, o);
  }
  {
    GemmOpts o; heads(o);
    o.sa_o = sP_o; o.sa_i = sP_i; o.sb_o = N * D; o.sb_i = dh; o.sc_o = tok; o.sc_i = dh;
    vit::gemm(true, false, N, dh, N, P, ld, dc, D, dv, 3 * D, o);
  }
  // 3. dS = softmax'(dP) / sqrt(dh)   (in place)
  const float scale = 1.0f / std::sqrt(static_cast<float>(head_dim));
  vit::check(vitk_softmax_rows_bwd(dPd, P, BH * N, N, ld, scale, dPd, S()), "vitk_softmax_rows_bwd");
  // 4. dQ = dS K
  {
    GemmOpts o; heads(o);
    o.sa_o = sP_o; o.sa_i = sP_i; o.sb_o = tok; o.sb_i = dh; o.sc_o = tok; o.sc_i = dh;
    vit::gemm(false, false, N, dh, N, dPd, ld, k, 3 * D, dq, 3 * D, o);
  }
  // 5. key gradient
  if (dk_mode == VITK_DK_EXACT) {
    GemmOpts o; heads(o);  // dK = dS^T Q
    o.sa_o = sP_o; o.sa_i = sP_i; o.sb_o = tok; o.sb_i = dh; o.sc_o = tok; o.sc_i = dh;
    vit::gemm(true, false, N, dh, N, dPd, ld, q, 3 * D, dk, 3 * D, o);
  } else {
    // the reference forms Q^T dS as {B*h, dh, N} and re-reads that buffer as {B,h,N,dh}
    // (src/layers/MultiHeadAttention.cpp:207-222)
    Tensor T = Tensor::deviceEmpty({(size_t)BH, (size_t)dh, (size_t)N});
    GemmOpts o; heads(o);
    o.sa_o = tok; o.sa_i = dh; o.sb_o = sP_o; o.sb_i = sP_i; o.sc_o = h * dh * N; o.sc_i = dh * N;
    vit::gemm(true, false, dh, N, N, q, 3 * D, dPd, ld, T.deviceDataOverwrite(), N, o);
    vit::check(vitk_dk_reference_scatter(T.deviceData(), (int32_t)B, (int32_t)h, (int32_t)N, (int32_t)dh, dk, 3 * D, S()),
               "vitk_dk_reference_scatter");
  }
  return projectionBackward(dqkv);
}

// packed projection backward: dW_qkv = X^T dQKV (+ db_qkv = column sums, same launch), dX = dQKV W_qkv^T
Tensor MultiHeadAttention::projectionBackward(const Tensor &dqkv) {
  const auto &s = inputTensor.getShape();
  const int64_t D = (int64_t)embedding_dim, M = (int64_t)(s[0] * s[1]);
  Tensor dx = Tensor::deviceEmpty(inputTensor.getShape());
  vit::dense_backward_gemms(M, D, 3 * D, inputTensor.deviceData(), dqkv.deviceData(), wqkv.deviceData(), 3 * D,
                            dwqkv.deviceData(), 3 * D, dbqkv.deviceData(), dx.deviceDataOverwrite(), GemmOpts());
  return dx;
}

std::vector<Tensor *> MultiHeadAttention::getParameters() {
  std::vector<Tensor *> all;
  for (Dense *d : {q_proj.get(), k_proj.get(), v_proj.get(), out_proj.get()}) {
    auto p = d->getParameters();
    all.insert(all.end(), p.begin(), p.end());
  }
  return all;
}

std::vector<Tensor *> MultiHeadAttention::getGradients() {
  std::vector<Tensor *> all;
  for (Dense *d : {q_proj.get(), k_proj.get(), v_proj.get(), out_proj.get()}) {
    auto p = d->getGradients();
    all.insert(all.end(), p.begin(), p.end());
  }
  return all;
}
