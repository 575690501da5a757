// Single-pass bandwidth kernels: GELU, dropout, adds, scale, fill, strided copy.
// All are grid-stride, 16 bytes per lane when the buffers allow it (scalar tail / fallback otherwise).
// Floating-point contraction is off in this file so results match the reference's separately
// rounded multiply/add sequences.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int TPB = 256;

template <typename F>
__global__ void map1_kernel(const float *__restrict__ x, float *__restrict__ y, long long n, bool vec, F f) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = tid; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4 *>(x)[i];
      v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
      reinterpret_cast<float4 *>(y)[i] = v;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) y[i] = f(x[i]);
  } else {
    for (long long i = tid; i < n; i += stride) y[i] = f(x[i]);
  }
}

template <typename F>
__global__ void map2_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y,
                            long long n, bool vec, F f) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = tid; i < n4; i += stride) {
      const float4 u = reinterpret_cast<const float4 *>(a)[i];
      const float4 w = reinterpret_cast<const float4 *>(b)[i];
      float4 v;
      v.x = f(u.x, w.x); v.y = f(u.y, w.y); v.z = f(u.z, w.z); v.w = f(u.w, w.w);
      reinterpret_cast<float4 *>(y)[i] = v;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) y[i] = f(a[i], b[i]);
  } else {
    for (long long i = tid; i < n; i += stride) y[i] = f(a[i], b[i]);
  }
}

__global__ void add3_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c,
                            float *__restrict__ y, long long n, bool vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = tid; i < n4; i += stride) {
      const float4 u = reinterpret_cast<const float4 *>(a)[i];
      const float4 w = reinterpret_cast<const float4 *>(b)[i];
      const float4 t = reinterpret_cast<const float4 *>(c)[i];
      float4 v;
      v.x = (u.x + w.x) + t.x; v.y = (u.y + w.y) + t.y; v.z = (u.z + w.z) + t.z; v.w = (u.w + w.w) + t.w;
      reinterpret_cast<float4 *>(y)[i] = v;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) y[i] = (a[i] + b[i]) + c[i];
  } else {
    for (long long i = tid; i < n; i += stride) y[i] = (a[i] + b[i]) + c[i];
  }
}

__global__ void fill_kernel(float *__restrict__ x, float value, long long n, bool vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = n >> 2;
    const float4 v = make_float4(value, value, value, value);
    for (long long i = tid; i < n4; i += stride) reinterpret_cast<float4 *>(x)[i] = v;
    for (long long i = (n4 << 2) + tid; i < n; i += stride) x[i] = value;
  } else {
    for (long long i = tid; i < n; i += stride) x[i] = value;
  }
}

__global__ void copy2d_kernel(const float *__restrict__ src, long long lds, float *__restrict__ dst, long long ldd,
                              long long rows, long long cols, bool vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const long long c4 = cols >> 2, total = rows * c4;
    for (long long i = tid; i < total; i += stride) {
      const long long r = i / c4, c = (i % c4) << 2;
      *reinterpret_cast<float4 *>(dst + r * ldd + c) = *reinterpret_cast<const float4 *>(src + r * lds + c);
    }
  } else {
    const long long total = rows * cols;
    for (long long i = tid; i < total; i += stride) {
      const long long r = i / cols, c = i % cols;
      dst[r * ldd + c] = src[r * lds + c];
    }
  }
}

// Generic strided gather of a view with up to 4 dimensions into a contiguous buffer.
struct Dims4 { long long shape[4]; long long stride[4]; };
__global__ void copy4d_kernel(const float *__restrict__ src, Dims4 d, float *__restrict__ dst, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += stride) {
    long long r = i;
    const long long i3 = r % d.shape[3]; r /= d.shape[3];
    const long long i2 = r % d.shape[2]; r /= d.shape[2];
    const long long i1 = r % d.shape[1]; r /= d.shape[1];
    dst[i] = src[r * d.stride[0] + i1 * d.stride[1] + i2 * d.stride[2] + i3 * d.stride[3]];
  }
}

__global__ void add_rowvec_kernel(float *__restrict__ x, long long rows, long long cols, const float *__restrict__ v) {
  const long long total = rows * cols, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += stride) x[i] += v[i % cols];
}

// Philox4x32-10 (Salmon et al.), counter = (ctr_lo, ctr_hi, 0, 0), key = seed.
__device__ __forceinline__ uint4 philox4x32(unsigned long long ctr, unsigned long long seed) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// One Philox block per group of 4 consecutive elements: element i uses word i%4 of block
// offset + i/4, so the mask does not depend on the launch geometry.
__device__ __forceinline__ void dropout_factors(unsigned long long block, unsigned long long seed, float rate,
                                                float scale, float (&m)[4]) {
  const uint4 r = philox4x32(block, seed);
  m[0] = u01(r.x) < rate ? 0.f : scale;
  m[1] = u01(r.y) < rate ? 0.f : scale;
  m[2] = u01(r.z) < rate ? 0.f : scale;
  m[3] = u01(r.w) < rate ? 0.f : scale;
}

// y = x * m (+ residual); mask (optional) receives m.
__global__ void dropout_fwd_kernel(const float *__restrict__ x, const float *__restrict__ residual, long long n,
                                   float rate, float scale, unsigned long long seed, unsigned long long offset,
                                   float *__restrict__ y, float *__restrict__ mask, bool vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long n4 = (n + 3) >> 2;
  for (long long i = tid; i < n4; i += stride) {
    float m[4];
    dropout_factors(offset + (unsigned long long)i, seed, rate, scale, m);
    const long long e = i << 2;
    if (vec && e + 3 < n) {
      const float4 v = reinterpret_cast<const float4 *>(x)[i];
      float4 o = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
      if (residual != nullptr) {
        const float4 r = reinterpret_cast<const float4 *>(residual)[i];
        o = make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w);
      }
      if (mask != nullptr) reinterpret_cast<float4 *>(mask)[i] = make_float4(m[0], m[1], m[2], m[3]);
      reinterpret_cast<float4 *>(y)[i] = o;
    } else {
      for (int j = 0; j < 4 && e + j < n; ++j) {
        if (mask != nullptr) mask[e + j] = m[j];
        const float o = x[e + j] * m[j];
        y[e + j] = residual != nullptr ? residual[e + j] + o : o;
      }
    }
  }
}

// dx = dy * m with m regenerated from the stream position of the forward call.
__global__ void dropout_bwd_stream_kernel(const float *__restrict__ dy, long long n, float rate, float scale,
                                          unsigned long long seed, unsigned long long offset, float *__restrict__ dx,
                                          bool vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long n4 = (n + 3) >> 2;
  for (long long i = tid; i < n4; i += stride) {
    float m[4];
    dropout_factors(offset + (unsigned long long)i, seed, rate, scale, m);
    const long long e = i << 2;
    if (vec && e + 3 < n) {
      const float4 g = reinterpret_cast<const float4 *>(dy)[i];
      reinterpret_cast<float4 *>(dx)[i] = make_float4(g.x * m[0], g.y * m[1], g.z * m[2], g.w * m[3]);
    } else {
      for (int j = 0; j < 4 && e + j < n; ++j) dx[e + j] = dy[e + j] * m[j];
    }
  }
}

// The same factors for a compact [rows][cols] slice of a larger tensor: element (r, c) takes the
// factor of flat position r * idx_stride + c of that tensor (from `mask` when given, else from the
// stream).  y = x * m (+ residual).  Small by construction (one token per sequence): one element
// per thread.
__global__ void dropout_rows_kernel(const float *__restrict__ x, const float *__restrict__ residual,
                                    const float *__restrict__ mask, long long rows, long long cols, long long idx_stride,
                                    float rate, float scale, unsigned long long seed, unsigned long long offset,
                                    float *__restrict__ y) {
  const long long total = rows * cols, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long flat = (i / cols) * idx_stride + i % cols;
    float f;
    if (mask != nullptr) {
      f = mask[flat];
    } else {
      float m[4];
      dropout_factors(offset + (unsigned long long)(flat >> 2), seed, rate, scale, m);
      const int w = (int)(flat & 3);
      f = w == 0 ? m[0] : w == 1 ? m[1] : w == 2 ? m[2] : m[3];
    }
    const float o = x[i] * f;
    y[i] = residual != nullptr ? residual[i] + o : o;
  }
}

struct GeluF { __device__ float operator()(float v) const { return vitk::gelu_fwd(v); } };
struct GeluB { __device__ float operator()(float x, float dy) const { return vitk::gelu_grad(x) * dy; } };
struct AddF { __device__ float operator()(float a, float b) const { return a + b; } };
struct MulF { __device__ float operator()(float a, float b) const { return a * b; } };
struct ScaleF { float s; __device__ float operator()(float v) const { return v * s; } };

}  // namespace

extern "C" {

int vitk_gelu_fwd(const float *x, int64_t n, float *y, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (x && y)), "vitk_gelu_fwd: bad arguments");
  if (n == 0) return 0;
  const bool vec = vitk::aligned16(x) && vitk::aligned16(y);
  hipLaunchKernelGGL((map1_kernel<GeluF>), dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     x, y, (long long)n, vec, GeluF{});
  return vitk::check_launch("vitk_gelu_fwd");
}

int vitk_gelu_bwd(const float *x, const float *dy, int64_t n, float *dx, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (x && dy && dx)), "vitk_gelu_bwd: bad arguments");
  if (n == 0) return 0;
  const bool vec = vitk::aligned16(x) && vitk::aligned16(dy) && vitk::aligned16(dx);
  hipLaunchKernelGGL((map2_kernel<GeluB>), dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     x, dy, dx, (long long)n, vec, GeluB{});
  return vitk::check_launch("vitk_gelu_bwd");
}

int vitk_dropout_fwd(const float *x, int64_t n, float rate, uint64_t seed, uint64_t offset, float *y, float *mask,
                     vitk_stream_t stream) {
  VITK_REQUIRE(mask != nullptr || n == 0, "vitk_dropout_fwd: null mask");
  return vitk_dropout_residual_fwd(x, nullptr, n, rate, seed, offset, y, mask, stream);
}

int vitk_dropout_residual_fwd(const float *x, const float *residual, int64_t n, float rate, uint64_t seed,
                              uint64_t offset, float *y, float *mask, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (x && y)), "vitk_dropout_fwd: bad arguments");
  VITK_REQUIRE(rate >= 0.0f && rate < 1.0f, "vitk_dropout_fwd: rate %g outside [0,1)", (double)rate);
  if (n == 0) return 0;
  const float scale = rate > 0.0f ? 1.0f / (1.0f - rate) : 1.0f;
  const bool vec = vitk::aligned16(x) && vitk::aligned16(y) && vitk::aligned16(mask) && vitk::aligned16(residual);
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream), x,
                     residual, (long long)n, rate, scale, (unsigned long long)seed, (unsigned long long)offset, y, mask,
                     vec);
  return vitk::check_launch("vitk_dropout_fwd");
}

int vitk_dropout_stream_bwd(const float *dy, int64_t n, float rate, uint64_t seed, uint64_t offset, float *dx,
                            vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (dy && dx)), "vitk_dropout_stream_bwd: bad arguments");
  VITK_REQUIRE(rate >= 0.0f && rate < 1.0f, "vitk_dropout_stream_bwd: rate %g outside [0,1)", (double)rate);
  if (n == 0) return 0;
  const float scale = rate > 0.0f ? 1.0f / (1.0f - rate) : 1.0f;
  const bool vec = vitk::aligned16(dy) && vitk::aligned16(dx);
  hipLaunchKernelGGL(dropout_bwd_stream_kernel, dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0,
                     vitk::as_stream(stream), dy, (long long)n, rate, scale, (unsigned long long)seed,
                     (unsigned long long)offset, dx, vec);
  return vitk::check_launch("vitk_dropout_stream_bwd");
}

int vitk_dropout_rows(const float *x, const float *residual, const float *mask, int64_t rows, int64_t cols,
                      int64_t idx_stride, float rate, uint64_t seed, uint64_t offset, float *y, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && cols >= 0 && idx_stride >= cols, "vitk_dropout_rows: bad shape");
  if (rows == 0 || cols == 0) return 0;
  VITK_REQUIRE(x && y, "vitk_dropout_rows: null pointer");
  VITK_REQUIRE(rate >= 0.0f && rate < 1.0f, "vitk_dropout_rows: rate %g outside [0,1)", (double)rate);
  hipLaunchKernelGGL(dropout_rows_kernel, dim3(vitk::grid_for(rows * cols, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     x, residual, mask, (long long)rows, (long long)cols, (long long)idx_stride, rate,
                     1.0f / (1.0f - rate), (unsigned long long)seed, (unsigned long long)offset, y);
  return vitk::check_launch("vitk_dropout_rows");
}

int vitk_dropout_bwd(const float *dy, const float *mask, int64_t n, float *dx, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (dy && mask && dx)), "vitk_dropout_bwd: bad arguments");
  if (n == 0) return 0;
  const bool vec = vitk::aligned16(dy) && vitk::aligned16(mask) && vitk::aligned16(dx);
  hipLaunchKernelGGL((map2_kernel<MulF>), dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     dy, mask, dx, (long long)n, vec, MulF{});
  return vitk::check_launch("vitk_dropout_bwd");
}

int vitk_add_f32(const float *a, const float *b, int64_t n, float *out, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (a && b && out)), "vitk_add_f32: bad arguments");
  if (n == 0) return 0;
  const bool vec = vitk::aligned16(a) && vitk::aligned16(b) && vitk::aligned16(out);
  hipLaunchKernelGGL((map2_kernel<AddF>), dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     a, b, out, (long long)n, vec, AddF{});
  return vitk::check_launch("vitk_add_f32");
}

int vitk_add3_f32(const float *a, const float *b, const float *c, int64_t n, float *out, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (a && b && c && out)), "vitk_add3_f32: bad arguments");
  if (n == 0) return 0;
  const bool vec = vitk::aligned16(a) && vitk::aligned16(b) && vitk::aligned16(c) && vitk::aligned16(out);
  hipLaunchKernelGGL(add3_kernel, dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream), a, b, c,
                     out, (long long)n, vec);
  return vitk::check_launch("vitk_add3_f32");
}

int vitk_scale_f32(const float *x, float s, int64_t n, float *out, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || (x && out)), "vitk_scale_f32: bad arguments");
  if (n == 0) return 0;
  const bool vec = vitk::aligned16(x) && vitk::aligned16(out);
  hipLaunchKernelGGL((map1_kernel<ScaleF>), dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     x, out, (long long)n, vec, ScaleF{s});
  return vitk::check_launch("vitk_scale_f32");
}

int vitk_fill_f32(float *x, float value, int64_t n, vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0 && (n == 0 || x), "vitk_fill_f32: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(vitk::grid_for(n / 4 + 1, TPB)), dim3(TPB), 0, vitk::as_stream(stream), x, value,
                     (long long)n, vitk::aligned16(x));
  return vitk::check_launch("vitk_fill_f32");
}

int vitk_copy2d_f32(const float *src, int64_t ld_src, float *dst, int64_t ld_dst, int64_t rows, int64_t cols,
                    vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && cols >= 0, "vitk_copy2d_f32: negative extent");
  if (rows == 0 || cols == 0) return 0;
  VITK_REQUIRE(src && dst && ld_src >= cols && ld_dst >= cols, "vitk_copy2d_f32: bad pointer or pitch");
  const bool vec = vitk::aligned16(src) && vitk::aligned16(dst) && ld_src % 4 == 0 && ld_dst % 4 == 0 && cols % 4 == 0;
  hipLaunchKernelGGL(copy2d_kernel, dim3(vitk::grid_for(rows * cols / 4 + 1, TPB)), dim3(TPB), 0,
                     vitk::as_stream(stream), src, (long long)ld_src, dst, (long long)ld_dst, (long long)rows,
                     (long long)cols, vec);
  return vitk::check_launch("vitk_copy2d_f32");
}

int vitk_copy4d_f32(const float *src, const int64_t *shape4, const int64_t *strides4, float *dst,
                    vitk_stream_t stream) {
  VITK_REQUIRE(shape4 && strides4, "vitk_copy4d_f32: null shape/strides");
  Dims4 d;
  long long total = 1;
  for (int i = 0; i < 4; ++i) {
    VITK_REQUIRE(shape4[i] >= 0 && strides4[i] >= 0, "vitk_copy4d_f32: negative extent or stride");
    d.shape[i] = shape4[i];
    d.stride[i] = strides4[i];
    total *= shape4[i];
  }
  if (total == 0) return 0;
  VITK_REQUIRE(src && dst, "vitk_copy4d_f32: null pointer");
  hipLaunchKernelGGL(copy4d_kernel, dim3(vitk::grid_for(total, TPB)), dim3(TPB), 0, vitk::as_stream(stream), src, d, dst,
                     total);
  return vitk::check_launch("vitk_copy4d_f32");
}

int vitk_add_rowvec_f32(float *x, int64_t rows, int64_t cols, const float *v, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && cols >= 0, "vitk_add_rowvec_f32: negative extent");
  if (rows == 0 || cols == 0) return 0;
  VITK_REQUIRE(x && v, "vitk_add_rowvec_f32: null pointer");
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(vitk::grid_for(rows * cols, TPB)), dim3(TPB), 0, vitk::as_stream(stream), x,
                     (long long)rows, (long long)cols, v);
  return vitk::check_launch("vitk_add_rowvec_f32");
}

}  // extern "C"
