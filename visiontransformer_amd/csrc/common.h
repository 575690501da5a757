// Shared helpers for the gfx950 kernels behind include/vitk.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "vitk.h"

namespace vitk {

// Thread-local message returned by vitk_last_error().
void set_error(const char *fmt, ...);

inline hipStream_t as_stream(vitk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that has just been enqueued; returns 0 or records the error.
int check_launch(const char *what);

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

// GELU (tanh approximation) exactly as the reference writes it (src/activations/GELU.cpp:9,30-33,61-74).
__device__ __forceinline__ float gelu_fwd(float x) {
  const float k = 0.7978845608028654f;
  float x_cubed = x * x * x;
  float inner = k * (x + 0.044715f * x_cubed);
  return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float gelu_grad(float x) {
  const float k = 0.7978845608028654f;
  float x_squared = x * x;
  float x_cubed = x_squared * x;
  float inner = k * (x + 0.044715f * x_cubed);
  float t = tanhf(inner);
  float d_inner = k * (1.0f + 3.0f * 0.044715f * x_squared);
  float sech2 = 1.0f - t * t;
  return 0.5f * (1.0f + t) + 0.5f * x * sech2 * d_inner;
}

}  // namespace vitk

#define VITK_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      ::vitk::set_error(__VA_ARGS__);  \
      return 1;                        \
    }                                  \
  } while (0)
