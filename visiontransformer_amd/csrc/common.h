// Shared helpers for the gfx950 kernels behind include/vitk.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>

#include "vitk.h"

namespace vitk {

// Thread-local message returned by vitk_last_error().
void set_error(const char *fmt, ...);

inline hipStream_t as_stream(vitk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that has just been enqueued; returns 0 or records the error.
int check_launch(const char *what);

// Caches the current device's address of the scratch array behind vitk_layernorm_bwd_static
// (norm.hip); called by vitk_init.
int norm_static_setup();

// Small-problem GEMM (gemm_small.hip): whether vitk_gemm_f32 routes these arguments to it, and the launch.
bool gemm_small_applies(const vitk_gemm_args *a);
int gemm_small_launch(const vitk_gemm_args *a, hipStream_t st);
// The two products of a Dense backward (dX = dY W^T, dW = X^T dY) as one launch of that kernel.
bool gemm_small_pair_applies(const vitk_gemm_args *dx, const vitk_gemm_args *dw);
int gemm_small_pair_launch(const vitk_gemm_args *dx, const vitk_gemm_args *dw, hipStream_t st);

// Forward attention on 16-row tiles (attention16.hip): whether vitk_sdpa_fwd routes a call to it, and the launch.
bool sdpa_fwd16_applies(int N, int dh, bool vec, float scale, long long ld);
int sdpa_fwd16_launch(const float *q, const float *k, const float *v, long long ld, int B, int H, int N, int dh,
                      float scale, float *out, long long ldo, float *stats, hipStream_t st);

// Backward attention on 16-row tiles (attention16_bwd.hip), same convention.
bool sdpa_bwd16_applies(int B, int N, int dh, bool vec, long long ld, long long ldo, long long ldg);
int sdpa_bwd16_launch(const float *q, const float *k, const float *v, long long ld, const float *o, const float *dout,
                      long long ldo, const float *stats, int B, int H, int N, int dh, float scale, int dk_mode,
                      float *dq, float *dk, float *dv, long long ldg, hipStream_t st);

// Compute units of the current device (hipDeviceProp_t::multiProcessorCount), cached per device;
// 256 if the query fails.
int cu_count();
// first call with this flag word on the current device?
bool first_use_on_device(std::atomic<unsigned long long> *done);

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;
constexpr int kMaxGridBlocks = 2048;  // 256 CUs x 8 resident blocks; larger problems grid-stride

inline int grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kMaxGridBlocks) b = kMaxGridBlocks;
  return static_cast<int>(b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// GELU, tanh approximation (src/activations/GELU.cpp:9,30-33,61-74 of the reference):
//   y = 0.5 x (1 + tanh u),  u = k (x + 0.044715 x^3),  k = sqrt(2/pi).
// Written through the logistic function, 0.5 (1 + tanh u) = s = 1 / (1 + e^(-2u)), which needs one
// v_exp_f32 and one v_rcp_f32 instead of a library tanh (these run in GEMM epilogues, where every
// vector instruction takes matrix-pipe time): y = x s, and with 1 - tanh^2 u = 4 s (1 - s)
//   dy/dx = s + 2 x s (1 - s) k (1 + 3 * 0.044715 x^2).
// Within 4e-7 of the reference's expression relative to |x| (checked against float64 in the tests).
__device__ __forceinline__ float gelu_sigmoid(float x, float x_squared) {
  const float a = -2.0f * 1.4426950408889634f * 0.7978845608028654f;  // -2 log2(e) k
  const float e = __builtin_amdgcn_exp2f(x * (a + (a * 0.044715f) * x_squared));
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_fwd(float x) { return x * gelu_sigmoid(x, x * x); }
__device__ __forceinline__ float gelu_grad(float x) {
  const float k = 0.7978845608028654f;
  const float x_squared = x * x;
  const float s = gelu_sigmoid(x, x_squared);
  const float d_inner = k * (1.0f + 3.0f * 0.044715f * x_squared);
  return s + 2.0f * x * (s * (1.0f - s)) * d_inner;
}

}  // namespace vitk

#define VITK_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      ::vitk::set_error(__VA_ARGS__);  \
      return 1;                        \
    }                                  \
  } while (0)
