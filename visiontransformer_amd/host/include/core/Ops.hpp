// Thin C++ wrappers over the libvitk C-ABI used by the layer classes: they fill the argument
// structs, provide scratch workspace and turn error codes into exceptions.
#ifndef VIT_OPS_HPP
#define VIT_OPS_HPP

#include <cstdint>

#include "core/Device.hpp"
#include "core/Tensor.hpp"
#include "vitk.h"

namespace vit {

// Parameter and gradient arenas: two equally laid out flat device buffers.  Layers carve their
// tensors out of them in construction order, so the whole model's gradients can be all-reduced
// and its parameters updated with one kernel launch each.
class ParamArena {
public:
  explicit ParamArena(size_t count);
  // next `count` elements as (parameter, gradient) views of the given shape/strides
  void take(const std::vector<size_t> &shape, Tensor *param, Tensor *grad);
  // raw block of `count` elements; returns its offset (views are built by the caller)
  size_t takeBlock(size_t count);
  const std::shared_ptr<Storage> &params() const { return p; }
  const std::shared_ptr<Storage> &grads() const { return g; }
  size_t size() const { return p->count; }
  size_t used() const { return cursor; }

private:
  std::shared_ptr<Storage> p, g;
  size_t cursor = 0;
};

struct GemmOpts {
  float alpha = 1.0f;
  int act = VITK_ACT_NONE;
  const float *bias = nullptr;
  const float *residual = nullptr;
  int64_t ldr = 0;
  float *aux = nullptr;
  int64_t ldaux = 0;
  int batch_outer = 1, batch_inner = 1;
  int64_t sa_o = 0, sa_i = 0, sb_o = 0, sb_i = 0, sc_o = 0, sc_i = 0;
  int split_k = 1;  // 0 = let the library split the contraction (weight gradients)
  float *colsum_b = nullptr;  // also produce the column sums of B (bias gradient next to dW = X^T dY)
};

void gemm(bool trans_a, bool trans_b, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
          int64_t ldb, float *C, int64_t ldc, const GemmOpts &o = GemmOpts());
// Both products of Dense::backward (src/layers/Dense.cpp:90-98): dW = X^T dY with db = dY.sum(0) from
// the same launch, and dX = dY W^T with dx_opts' epilogue (vitk_gemm_pair_f32: one launch on the
// reference's small models, two otherwise).  x {rows,in}, dy {rows,out}, W {in,out} with pitch ldw.
void dense_backward_gemms(int64_t rows, int64_t in, int64_t out, const float *x, const float *dy, const float *W, int64_t ldw,
                          float *dW, int64_t lddw, float *db, float *dx, const GemmOpts &dx_opts);
void colsum(const float *x, int64_t rows, int64_t cols, int64_t ldx, float *out);

}  // namespace vit

#endif
