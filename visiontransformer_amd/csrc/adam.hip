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

// U float4 per array and thread are loaded before any is used: 4*U 16-byte loads in flight per lane
// keep the memory system busy across the dependent divide / square root chain.
constexpr int kAdamUnroll = 4;

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long long n,
                                                   AdamC c, bool vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    constexpr int U = kAdamUnroll;
    const long long n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    // block-contiguous chunks of U*256 float4: consecutive lanes touch consecutive 16 bytes
    const long long chunk = (long long)U * blockDim.x, nchunks = n4 / chunk;
    for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
      const long long base = ch * chunk + threadIdx.x;
      float4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + (long long)u * blockDim.x;
        pp[u] = p4[i];
        gg[u] = g4[i];
        mm[u] = m4[i];
        vv[u] = v4[i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        adam_one(pp[u].x, gg[u].x, mm[u].x, vv[u].x, c);
        adam_one(pp[u].y, gg[u].y, mm[u].y, vv[u].y, c);
        adam_one(pp[u].z, gg[u].z, mm[u].z, vv[u].z, c);
        adam_one(pp[u].w, gg[u].w, mm[u].w, vv[u].w, c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + (long long)u * blockDim.x;
        p4[i] = pp[u];
        m4[i] = mm[u];
        v4[i] = vv[u];
      }
    }
    for (long long i = nchunks * chunk + tid; i < n4; i += stride) {
      float4 pp = p4[i];
      const float4 gg = g4[i];
      float4 mm = m4[i];
      float4 vv = v4[i];
      adam_one(pp.x, gg.x, mm.x, vv.x, c);
      adam_one(pp.y, gg.y, mm.y, vv.y, c);
      adam_one(pp.z, gg.z, mm.z, vv.z, c);
      adam_one(pp.w, gg.w, mm.w, vv.w, c);
      p4[i] = pp;
      m4[i] = mm;
      v4[i] = vv;
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
  hipLaunchKernelGGL(adam_kernel, dim3(vitk::grid_for(n / (4 * kAdamUnroll) + 1, 256)), dim3(256), 0, vitk::as_stream(stream), p, g, m,
                     v, (long long)n, c, vec);
  return vitk::check_launch("vitk_adam_l2");
}
