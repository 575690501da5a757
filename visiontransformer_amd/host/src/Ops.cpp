#include "core/Ops.hpp"

#include <algorithm>

namespace vit {

ParamArena::ParamArena(size_t count) : p(std::make_shared<Storage>(count)), g(std::make_shared<Storage>(count)) {
  // both arenas live on the device from the start (zero-filled)
  p->dev_read();
  g->dev_read();
}

size_t ParamArena::takeBlock(size_t count) {
  if (cursor + count > p->count) throw std::runtime_error("ParamArena: out of space");
  const size_t off = cursor;
  cursor += count;
  return off;
}

void ParamArena::take(const std::vector<size_t> &shape, Tensor *param, Tensor *grad) {
  size_t n = 1;
  for (size_t s : shape) n *= s;
  const size_t off = takeBlock(n);
  std::vector<size_t> strides(shape.size());
  size_t st = 1;
  for (int i = (int)shape.size() - 1; i >= 0; --i) {
    strides[i] = st;
    st *= shape[i];
  }
  *param = Tensor(p, shape, strides, off);
  *grad = Tensor(g, shape, strides, off);
}

static vitk_gemm_args gemm_args(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                                const float *B, int64_t ldb, float *C, int64_t ldc, const GemmOpts &o) {
  vitk_gemm_args g{};
  g.trans_a = trans_a; g.trans_b = trans_b;
  g.M = (int32_t)M; g.N = (int32_t)N; g.K = (int32_t)K;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.alpha = o.alpha; g.act = o.act; g.bias = o.bias; g.residual = o.residual; g.ldr = o.ldr;
  g.aux = o.aux; g.ldaux = o.ldaux;
  g.batch_outer = o.batch_outer; g.batch_inner = o.batch_inner;
  g.stride_a_outer = o.sa_o; g.stride_a_inner = o.sa_i;
  g.stride_b_outer = o.sb_o; g.stride_b_inner = o.sb_i;
  g.stride_c_outer = o.sc_o; g.stride_c_inner = o.sc_i;
  g.split_k = o.split_k;
  g.colsum_b = o.colsum_b;
  return g;
}

void gemm(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
          int64_t ldb, float *C, int64_t ldc, const GemmOpts &o) {
  vitk_gemm_args g = gemm_args(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, o);
  const size_t wsb = vitk_gemm_workspace_bytes(&g);
  if (wsb) {
    g.workspace = workspace(wsb);
    g.workspace_bytes = wsb;
  }
  check(vitk_gemm_f32(&g, stream()), "vitk_gemm_f32");
}

void dense_backward_gemms(int64_t rows, int64_t in, int64_t out, const float *x, const float *dy, const float *W, int64_t ldw,
                          float *dW, int64_t lddw, float *db, float *dx, const GemmOpts &dx_opts) {
  GemmOpts ow;
  ow.split_k = 0;
  ow.colsum_b = db;
  vitk_gemm_args gw = gemm_args(true, false, in, out, rows, x, in, dy, out, dW, lddw, ow);
  vitk_gemm_args gx = gemm_args(false, true, rows, in, out, dy, out, W, ldw, dx, in, dx_opts);
  // the two launches (if it comes to two) run one after the other on one stream: one workspace serves both
  const size_t wsb = std::max(vitk_gemm_workspace_bytes(&gw), vitk_gemm_workspace_bytes(&gx));
  if (wsb) {
    gw.workspace = gx.workspace = workspace(wsb);
    gw.workspace_bytes = gx.workspace_bytes = wsb;
  }
  check(vitk_gemm_pair_f32(&gx, &gw, stream()), "vitk_gemm_pair_f32");
}

void colsum(const float *x, int64_t rows, int64_t cols, int64_t ldx, float *out) {
  const size_t wsb = vitk_colsum_workspace_bytes(rows, cols);
  check(vitk_colsum_f32(x, rows, cols, ldx, out, workspace(wsb), wsb, stream()), "vitk_colsum_f32");
}

}  // namespace vit
