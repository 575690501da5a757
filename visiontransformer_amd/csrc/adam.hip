// Adam with coupled L2 regularisation over a flat parameter arena: one launch for the whole model,
// 16-byte accesses, 28 bytes of HBM traffic per parameter (read p,g,m,v; write p,m,v).
#pragma clang fp contract(off)
#include "common.h"

namespace {

struct AdamC {
  float lr, b1, b2, omb1, omb2, c1, c2, eps, wd, gs;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamC &c) {
  g *= c.gs;
  if (c.wd > 0.0f) g += c.wd * p;
  m = c.b1 * m + c.omb1 * g;
  v = c.b2 * v + c.omb2 * (g * g);
  const float m_hat = m / c.c1;
  const float v_hat = v / c.c2;
  p -= c.lr * m_hat / (sqrtf(v_hat) + c.eps);
}

// Streaming kernel, 28 bytes per parameter and nothing re-read.  Measured on MI355X (86.6 M
// parameters, tools/bench_adam.py): one wave per workgroup, each thread two float4 per array a
// wave apart, every access non-temporal: 0.406 ms = 5.97 TB/s.  The same with 256-thread
// workgroups 5.7, with temporal accesses 4.8-5.3, four or more float4 per thread in flight or a
// grid capped at a few workgroups per CU 4.0-4.8.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kAdamThreads = 64, kAdamUnroll = 2;

__device__ __forceinline__ void adam_quad(f32x4 &p, const f32x4 &g, f32x4 &m, f32x4 &v, const AdamC &c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float pe = p[e], me = m[e], ve = v[e];
    adam_one(pe, g[e], me, ve, c);
    p[e] = pe, m[e] = me, v[e] = ve;
  }
}

__global__ __launch_bounds__(kAdamThreads) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                            float *__restrict__ m, float *__restrict__ v, long long n,
                                                            AdamC c, bool vec) {
  constexpr int U = kAdamUnroll;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = n >> 2;
    f32x4 *p4 = reinterpret_cast<f32x4 *>(p), *m4 = reinterpret_cast<f32x4 *>(m), *v4 = reinterpret_cast<f32x4 *>(v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    // a workgroup owns U * blockDim consecutive float4 (the grid covers the array once)
    const long long base = (long long)blockIdx.x * U * blockDim.x + threadIdx.x;
    if (base + (long long)(U - 1) * blockDim.x < n4) {
      f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + (long long)u * blockDim.x;
        pp[u] = __builtin_nontemporal_load(p4 + i);
        gg[u] = __builtin_nontemporal_load(g4 + i);
        mm[u] = __builtin_nontemporal_load(m4 + i);
        vv[u] = __builtin_nontemporal_load(v4 + i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) adam_quad(pp[u], gg[u], mm[u], vv[u], c);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + (long long)u * blockDim.x;
        __builtin_nontemporal_store(pp[u], p4 + i);
        __builtin_nontemporal_store(mm[u], m4 + i);
        __builtin_nontemporal_store(vv[u], v4 + i);
      }
    } else {  // the last workgroup(s): whatever float4 are left, then the last n % 4 elements
      for (int u = 0; u < U; ++u) {
        const long long i = base + (long long)u * blockDim.x;
        if (i < n4) {
          f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
          adam_quad(pp, g4[i], mm, vv, c);
          p4[i] = pp, m4[i] = mm, v4[i] = vv;
        }
      }
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], c);
  } else {
    for (long long i = tid; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], c);
  }
}

}  // namespace

extern "C" int vitk_adam_l2(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                            float beta2, float beta1_t, float beta2_t, float eps, float wd, float grad_scale,
                            vitk_stream_t stream) {
  VITK_REQUIRE(n >= 0, "vitk_adam_l2: negative length");
  if (n == 0) return 0;
  VITK_REQUIRE(p && g && m && v, "vitk_adam_l2: null pointer");
  VITK_REQUIRE(beta1_t != 1.0f && beta2_t != 1.0f, "vitk_adam_l2: beta^t == 1 (step counter must start at 1)");
  AdamC c;
  c.lr = lr; c.b1 = beta1; c.b2 = beta2; c.omb1 = 1.0f - beta1; c.omb2 = 1.0f - beta2;
  c.c1 = 1.0f - beta1_t; c.c2 = 1.0f - beta2_t; c.eps = eps; c.wd = wd; c.gs = grad_scale;
  const bool vec = vitk::aligned16(p) && vitk::aligned16(g) && vitk::aligned16(m) && vitk::aligned16(v);
  // one pass over the array: ceil(n/4 / (U * threads)) workgroups (the scalar fallback strides)
  long long blocks = (n / 4 + (long long)kAdamThreads * kAdamUnroll) / ((long long)kAdamThreads * kAdamUnroll);
  if (!vec && blocks > vitk::kMaxGridBlocks) blocks = vitk::kMaxGridBlocks;
  VITK_REQUIRE(blocks < (1ll << 31), "vitk_adam_l2: %lld parameters exceed one launch", (long long)n);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0, vitk::as_stream(stream), p, g, m, v,
                     (long long)n, c, vec);
  return vitk::check_launch("vitk_adam_l2");
}
