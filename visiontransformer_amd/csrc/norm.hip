// LayerNorm forward/backward and the two-stage column sum.  Bandwidth kernels: one wave per row,
// 16-byte loads, wave-level (shuffle) reductions; cross-row reductions (dgamma, dbeta, column
// sums) go through per-workgroup partials that a second kernel adds in a fixed order.
#pragma clang fp contract(off)
#include "common.h"

#include <mutex>

namespace {

constexpr int TPB = 256;
constexpr int WPB = TPB / 64;  // waves (= rows in flight) per workgroup

// ---- forward --------------------------------------------------------------------------------
// NV = float4 per lane (D <= 256*NV, D % 4 == 0).  NV == 0: generic scalar path for any D.
template <int NV>
__global__ __launch_bounds__(TPB) void ln_fwd_kernel(const float *__restrict__ x, long long rows, int D,
                                                     long long stride, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float eps,
                                                     float *__restrict__ y, float *__restrict__ mean_out,
                                                     float *__restrict__ inv_out) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = blockIdx.x * (long long)WPB + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * WPB;
  for (long long row = wave0; row < rows; row += nwaves) {
    const float *xr = x + row * stride;
    float *yr = y + row * stride;
    if constexpr (NV > 0) {
      float4 v[NV];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        v[j] = c < D ? *reinterpret_cast<const float4 *>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      }
      const float mean = vitk::wave_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
          const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
          q += (a * a + b * b) + (cc * cc + d * d);
        }
      }
      const float var = vitk::wave_sum(q) / (float)D;
      const float inv = 1.0f / sqrtf(var + eps);
      if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (inv_out) inv_out[row] = inv;
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
          const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
          const float4 b = *reinterpret_cast<const float4 *>(beta + c);
          float4 o;
          o.x = g.x * ((v[j].x - mean) * inv) + b.x;
          o.y = g.y * ((v[j].y - mean) * inv) + b.y;
          o.z = g.z * ((v[j].z - mean) * inv) + b.z;
          o.w = g.w * ((v[j].w - mean) * inv) + b.w;
          *reinterpret_cast<float4 *>(yr + c) = o;
        }
      }
    } else {
      float s = 0.f;
      for (int c = lane; c < D; c += 64) s += xr[c];
      const float mean = vitk::wave_sum(s) / (float)D;
      float q = 0.f;
      for (int c = lane; c < D; c += 64) {
        const float a = xr[c] - mean;
        q += a * a;
      }
      const float var = vitk::wave_sum(q) / (float)D;
      const float inv = 1.0f / sqrtf(var + eps);
      if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (inv_out) inv_out[row] = inv;
      }
      for (int c = lane; c < D; c += 64) yr[c] = gamma[c] * ((xr[c] - mean) * inv) + beta[c];
    }
  }
}

// ---- backward -------------------------------------------------------------------------------
// Per row: dxhat = dy*gamma, s1 = sum dxhat, s2 = sum dxhat*xhat,
// dx = (1/D)*inv*(D*dxhat - s1 - xhat*s2) (+ residual).  Each lane keeps dgamma/dbeta partials for
// its own columns over all rows its wave visits; the four waves are combined through LDS and the
// workgroup writes one [2][D] partial.
template <int NV>
__global__ __launch_bounds__(TPB) void ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                     const float *__restrict__ mean, const float *__restrict__ inv_std,
                                                     const float *__restrict__ gamma, long long rows, int D,
                                                     long long stride, const float *__restrict__ residual,
                                                     float *__restrict__ dx, float *__restrict__ partial) {
  constexpr int NC = NV > 0 ? NV * 4 : 1;
  __shared__ float red[2][WPB][NV > 0 ? NV * 256 : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wave0 = blockIdx.x * (long long)WPB + wv, nwaves = (long long)gridDim.x * WPB;
  const float fD = (float)D, invD = 1.0f / (float)D;
  if constexpr (NV > 0) {
    float dg[NC], db[NC];
    float4 gm[NV];
#pragma unroll
    for (int j = 0; j < NC; ++j) dg[j] = 0.f, db[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      gm[j] = c < D ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long row = wave0; row < rows; row += nwaves) {
      const float mu = mean[row], inv = inv_std[row];
      float4 g[NV], xh[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
          g[j] = *reinterpret_cast<const float4 *>(dy + row * stride + c);
          const float4 xv = *reinterpret_cast<const float4 *>(x + row * stride + c);
          xh[j] = make_float4((xv.x - mu) * inv, (xv.y - mu) * inv, (xv.z - mu) * inv, (xv.w - mu) * inv);
        } else {
          g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float a = g[j].x * gm[j].x, b = g[j].y * gm[j].y, c2 = g[j].z * gm[j].z, d = g[j].w * gm[j].w;
        s1 += (a + b) + (c2 + d);
        s2 += (a * xh[j].x + b * xh[j].y) + (c2 * xh[j].z + d * xh[j].w);
        dg[4 * j + 0] += g[j].x * xh[j].x; dg[4 * j + 1] += g[j].y * xh[j].y;
        dg[4 * j + 2] += g[j].z * xh[j].z; dg[4 * j + 3] += g[j].w * xh[j].w;
        db[4 * j + 0] += g[j].x; db[4 * j + 1] += g[j].y; db[4 * j + 2] += g[j].z; db[4 * j + 3] += g[j].w;
      }
      s1 = vitk::wave_sum(s1);
      s2 = vitk::wave_sum(s2);
      const float f = invD * inv;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
          float4 o;
          o.x = f * (fD * (g[j].x * gm[j].x) - s1 - xh[j].x * s2);
          o.y = f * (fD * (g[j].y * gm[j].y) - s1 - xh[j].y * s2);
          o.z = f * (fD * (g[j].z * gm[j].z) - s1 - xh[j].z * s2);
          o.w = f * (fD * (g[j].w * gm[j].w) - s1 - xh[j].w * s2);
          if (residual) {
            const float4 r = *reinterpret_cast<const float4 *>(residual + row * stride + c);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
          }
          *reinterpret_cast<float4 *>(dx + row * stride + c) = o;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[0][wv][(lane + 64 * j) * 4 + e] = dg[4 * j + e];
        red[1][wv][(lane + 64 * j) * 4 + e] = db[4 * j + e];
      }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += TPB) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WPB; ++w) a += red[0][w][c], b += red[1][w][c];
      partial[((long long)blockIdx.x * 2 + 0) * D + c] = a;
      partial[((long long)blockIdx.x * 2 + 1) * D + c] = b;
    }
  } else {
    // generic path: dx only; dgamma/dbeta come from ln_param_grad_generic below
    for (long long row = wave0; row < rows; row += nwaves) {
      const float mu = mean[row], inv = inv_std[row];
      float s1 = 0.f, s2 = 0.f;
      for (int c = lane; c < D; c += 64) {
        const float dxh = dy[row * stride + c] * gamma[c], xh = (x[row * stride + c] - mu) * inv;
        s1 += dxh;
        s2 += dxh * xh;
      }
      s1 = vitk::wave_sum(s1);
      s2 = vitk::wave_sum(s2);
      const float f = invD * inv;
      for (int c = lane; c < D; c += 64) {
        const float dxh = dy[row * stride + c] * gamma[c], xh = (x[row * stride + c] - mu) * inv;
        float o = f * (fD * dxh - s1 - xh * s2);
        if (residual) o += residual[row * stride + c];
        dx[row * stride + c] = o;
      }
    }
  }
}

// Few rows (at most kSmallRows: the 17- and 5-token models, the CLS rows of the final norm): dx and
// both parameter gradients in ONE launch, no partials.  Sixteen-wave workgroups; the first
// `col_blocks` of them own 64 columns each and sum them over all rows (64 row slots per
// workgroup, combined in slot order through LDS), the others give every wave one row of dx.
// The column workgroups read dy and x a second time -- from L2 at these sizes -- which costs
// less than a second launch (640 x 96: 6.8 against 8.2 us per call, tools/bench_ln_small.py).
constexpr int SM_WAVES = 16;
constexpr long long kSmallRows = 1024;
template <int NV>
__global__ __launch_bounds__(64 * SM_WAVES) void ln_bwd_small_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
    const float *__restrict__ inv_std, const float *__restrict__ gamma, long long rows, int D, long long stride,
    const float *__restrict__ residual, float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
    int col_blocks) {
  __shared__ __attribute__((aligned(16))) float sm[2][4 * SM_WAVES][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if ((int)blockIdx.x >= col_blocks) {  // (the column workgroups, the longer ones, come first in the grid)
    const long long row = ((int)blockIdx.x - col_blocks) * (long long)SM_WAVES + wv;
    if (row >= rows) return;
    const float fD = (float)D, invD = 1.0f / (float)D;
    const float mu = mean[row], inv = inv_std[row];
    float4 g[NV], xh[NV], gm[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < D) {
        gm[j] = *reinterpret_cast<const float4 *>(gamma + c);
        g[j] = *reinterpret_cast<const float4 *>(dy + row * stride + c);
        const float4 xv = *reinterpret_cast<const float4 *>(x + row * stride + c);
        xh[j] = make_float4((xv.x - mu) * inv, (xv.y - mu) * inv, (xv.z - mu) * inv, (xv.w - mu) * inv);
      } else {
        gm[j] = g[j] = xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float a = g[j].x * gm[j].x, b = g[j].y * gm[j].y, c2 = g[j].z * gm[j].z, d = g[j].w * gm[j].w;
      s1 += (a + b) + (c2 + d);
      s2 += (a * xh[j].x + b * xh[j].y) + (c2 * xh[j].z + d * xh[j].w);
    }
    s1 = vitk::wave_sum(s1);
    s2 = vitk::wave_sum(s2);
    const float f = invD * inv;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = (lane + 64 * j) * 4;
      if (c < D) {
        float4 o;
        o.x = f * (fD * (g[j].x * gm[j].x) - s1 - xh[j].x * s2);
        o.y = f * (fD * (g[j].y * gm[j].y) - s1 - xh[j].y * s2);
        o.z = f * (fD * (g[j].z * gm[j].z) - s1 - xh[j].z * s2);
        o.w = f * (fD * (g[j].w * gm[j].w) - s1 - xh[j].w * s2);
        if (residual) {
          const float4 r = *reinterpret_cast<const float4 *>(residual + row * stride + c);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        *reinterpret_cast<float4 *>(dx + row * stride + c) = o;
      }
    }
    return;
  }
  // column workgroup: 64 columns; lane (q, rg) takes the column quad 4q .. 4q+3 of the rows
  // 4 * wave + rg + 64 i -- 16-byte loads, eight rows in flight per lane (640 rows: two round
  // trips to memory), 64 partial sums per column combined in row-slot order through LDS
  const int q = lane & 15, slot = wv * 4 + (lane >> 4);
  const int c = (int)blockIdx.x * 64 + 4 * q;
  constexpr int U = 8;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (c < D) {
    for (long long row = slot; row < rows; row += 64 * U) {
      float4 gv[U], xv[U];
      float mu[U], iv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long r = row + 64 * u < rows ? row + 64 * u : rows - 1;  // (clamped: the loads stay unconditional)
        gv[u] = *reinterpret_cast<const float4 *>(dy + r * stride + c);
        xv[u] = *reinterpret_cast<const float4 *>(x + r * stride + c);
        mu[u] = mean[r], iv[u] = inv_std[r];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (row + 64 * u < rows) {
          a.x += gv[u].x * ((xv[u].x - mu[u]) * iv[u]); a.y += gv[u].y * ((xv[u].y - mu[u]) * iv[u]);
          a.z += gv[u].z * ((xv[u].z - mu[u]) * iv[u]); a.w += gv[u].w * ((xv[u].w - mu[u]) * iv[u]);
          b.x += gv[u].x; b.y += gv[u].y; b.z += gv[u].z; b.w += gv[u].w;
        }
    }
  }
  *reinterpret_cast<float4 *>(&sm[0][slot][4 * q]) = a;
  *reinterpret_cast<float4 *>(&sm[1][slot][4 * q]) = b;
  __syncthreads();
  if (threadIdx.x < 128) {  // threads 0..63: dgamma, 64..127: dbeta
    const int which = threadIdx.x >> 6, col = (int)blockIdx.x * 64 + (threadIdx.x & 63);
    if (col < D) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int k = 0; k < 64; k += 4) {
        t0 += sm[which][k][threadIdx.x & 63];
        t1 += sm[which][k + 1][threadIdx.x & 63];
        t2 += sm[which][k + 2][threadIdx.x & 63];
        t3 += sm[which][k + 3][threadIdx.x & 63];
      }
      (which == 0 ? dgamma : dbeta)[col] = (t0 + t1) + (t2 + t3);
    }
  }
}

// generic dgamma/dbeta partials: workgroup b handles rows b, b+grid, ...; thread per column.
__global__ void ln_param_grad_generic(const float *__restrict__ dy, const float *__restrict__ x,
                                      const float *__restrict__ mean, const float *__restrict__ inv_std,
                                      long long rows, int D, long long stride, float *__restrict__ partial) {
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
      const float g = dy[row * stride + c];
      a += g * ((x[row * stride + c] - mean[row]) * inv_std[row]);
      b += g;
    }
    partial[((long long)blockIdx.x * 2 + 0) * D + c] = a;
    partial[((long long)blockIdx.x * 2 + 1) * D + c] = b;
  }
}

// Second stage of every cross-row reduction in this file: out[c] = sum_k partial[k*pitch + c] for
// k in [0, n), in a fixed order; columns at and beyond `split` go to out2[c - split] (layer norm
// reduces its gamma and beta partials, which share a row, in one launch).  A workgroup owns 64
// columns; its 16 waves each sum every sixteenth partial row (coalesced 256-byte reads, four
// independent chains) and the sixteen sums are combined through LDS.
constexpr int RED_COLS = 64, RED_WAVES = 16;
__global__ __launch_bounds__(RED_COLS * RED_WAVES) void reduce_partials_kernel(
    const float *__restrict__ partial, int n, long long pitch, long long cols, long long split,
    float *__restrict__ out, float *__restrict__ out2) {
  __shared__ float sm[RED_WAVES][RED_COLS];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long c = (long long)blockIdx.x * RED_COLS + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    const float *__restrict__ col = partial + c;
    int k = g;
    for (; k + 3 * RED_WAVES < n; k += 4 * RED_WAVES) {
      s0 += col[(long long)k * pitch];
      s1 += col[(long long)(k + RED_WAVES) * pitch];
      s2 += col[(long long)(k + 2 * RED_WAVES) * pitch];
      s3 += col[(long long)(k + 3 * RED_WAVES) * pitch];
    }
    for (; k < n; k += RED_WAVES) s0 += col[(long long)k * pitch];
  }
  sm[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < RED_WAVES; ++w) t += sm[w][lane];
    if (c < split) out[c] = t;
    else out2[c - split] = t;
  }
}

// ---- column sum -------------------------------------------------------------------------------
__global__ void colsum_stage1(const float *__restrict__ x, long long rows, long long cols, long long ldx,
                              long long rows_per_chunk, float *__restrict__ partial) {
  const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const long long r0 = blockIdx.y * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long long r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += x[r * ldx + c];
    s1 += x[(r + 1) * ldx + c];
    s2 += x[(r + 2) * ldx + c];
    s3 += x[(r + 3) * ldx + c];
  }
  for (; r < r1; ++r) s0 += x[r * ldx + c];
  partial[blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}

bool g_ln_small_on = true;

int ln_blocks(long long rows) {
  long long b = (rows + WPB - 1) / WPB;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

void colsum_plan(long long rows, long long cols, int *nchunks, long long *rpc) {
  const long long colblocks = (cols + 255) / 256;
  long long n = 2048 / colblocks;
  const long long maxn = (rows + 31) / 32;
  if (n > maxn) n = maxn;
  if (n > 256) n = 256;
  if (n < 1) n = 1;
  *rpc = (rows + n - 1) / n;
  *nchunks = (int)((rows + *rpc - 1) / *rpc);
  if (*nchunks < 1) *nchunks = 1;
}

bool ln_vec_ok(const void *a, const void *b, const void *c, const void *d, long long D, long long stride) {
  return D % 4 == 0 && D <= 1024 && stride % 4 == 0 && vitk::aligned16(a) && vitk::aligned16(b) &&
         vitk::aligned16(c) && vitk::aligned16(d);
}

// ---- scratch of the entry point that allocates nothing -----------------------------------------
// vitk_layernorm_bwd_static keeps the per-workgroup [dgamma | dbeta] partials in this array instead
// of a caller-provided workspace: ln_blocks() <= 1024 workgroups x 2 x D floats, D <= kStaticMaxD.
// Every device that loads this code object holds its own 8 MiB copy until the process ends.
constexpr long long kStaticMaxD = 1024;
constexpr long long kStaticFloats = 1024ll * 2 * kStaticMaxD;
constexpr int kMaxDevices = 64;

// Device address of `scratch` on each device, filled by static_scratch() with that device current.
// The array's name itself is only a host-side handle for hipGetSymbolAddress.
float *g_static_ptr[kMaxDevices] = {};
std::mutex g_static_mutex;

}  // namespace

static __device__ __align__(256) float scratch[kStaticFloats];

namespace {

// Looks the array up once per device (the calling thread's current device) and caches the address.
int static_scratch(float **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  VITK_REQUIRE(e == hipSuccess, "vitk_layernorm_bwd_static: hipGetDevice: %s", hipGetErrorString(e));
  VITK_REQUIRE(dev >= 0 && dev < kMaxDevices, "vitk_layernorm_bwd_static: device %d beyond the %d supported", dev,
               kMaxDevices);
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (!g_static_ptr[dev]) {
    float *p = nullptr;
    e = hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch));
    VITK_REQUIRE(e == hipSuccess && p != nullptr, "vitk_layernorm_bwd_static: hipGetSymbolAddress on device %d: %s",
                 dev, hipGetErrorString(e));
    g_static_ptr[dev] = p;
  }
  *out = g_static_ptr[dev];
  return 0;
}

}  // namespace

namespace vitk {
int norm_static_setup() {
  float *p;
  return static_scratch(&p);
}
}  // namespace vitk

extern "C" {

size_t vitk_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  int n;
  long long rpc;
  colsum_plan(rows, cols, &n, &rpc);
  return (size_t)n * cols * sizeof(float);
}

int vitk_colsum_f32(const float *x, int64_t rows, int64_t cols, int64_t ldx, float *out, float *workspace,
                    size_t workspace_bytes, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && cols >= 0, "vitk_colsum_f32: negative extent");
  if (cols == 0) return 0;
  VITK_REQUIRE(out != nullptr, "vitk_colsum_f32: null output");
  hipStream_t st = vitk::as_stream(stream);
  if (rows == 0) return vitk_fill_f32(out, 0.f, cols, stream);
  VITK_REQUIRE(x != nullptr && ldx >= cols, "vitk_colsum_f32: bad input pointer or pitch");
  int n;
  long long rpc;
  colsum_plan(rows, cols, &n, &rpc);
  VITK_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)n * cols * sizeof(float),
               "vitk_colsum_f32: workspace too small (%zu < %zu)", workspace_bytes, (size_t)n * cols * sizeof(float));
  VITK_REQUIRE(n <= 65535, "vitk_colsum_f32: too many row chunks");
  const unsigned cb = (unsigned)((cols + 255) / 256);
  hipLaunchKernelGGL(colsum_stage1, dim3(cb, (unsigned)n), dim3(256), 0, st, x, (long long)rows, (long long)cols,
                     (long long)ldx, rpc, workspace);
  if (vitk::check_launch("vitk_colsum_f32/stage1")) return 1;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((cols + RED_COLS - 1) / RED_COLS)),
                     dim3(RED_COLS * RED_WAVES), 0, st, workspace, n, (long long)cols, (long long)cols,
                     (long long)cols, out, (float *)nullptr);
  return vitk::check_launch("vitk_colsum_f32/stage2");
}

int vitk_layernorm_fwd(const float *x, int64_t rows, int64_t D, int64_t row_stride, const float *gamma,
                       const float *beta, float eps, float *y, float *mean, float *inv_std, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && D > 0 && row_stride >= D, "vitk_layernorm_fwd: bad shape (rows=%lld D=%lld stride=%lld)",
               (long long)rows, (long long)D, (long long)row_stride);
  if (rows == 0) return 0;
  VITK_REQUIRE(x && y && gamma && beta, "vitk_layernorm_fwd: null pointer");
  VITK_REQUIRE(D < (1ll << 30), "vitk_layernorm_fwd: D too large");
  hipStream_t st = vitk::as_stream(stream);
  const dim3 g2((unsigned)vitk::grid_for(rows, WPB)), blk(TPB);
  const bool vec = ln_vec_ok(x, y, gamma, beta, D, row_stride);
  const int nv = vec ? (int)((D + 255) / 256) : 0;
#define LNF(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV>), g2, blk, 0, st, x, (long long)rows, (int)D, \
                                   (long long)row_stride, gamma, beta, eps, y, mean, inv_std)
  switch (nv) {
    case 1: LNF(1); break;
    case 2: LNF(2); break;
    case 3: LNF(3); break;
    case 4: LNF(4); break;
    default: LNF(0); break;
  }
#undef LNF
  return vitk::check_launch("vitk_layernorm_fwd");
}

size_t vitk_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D) {
  if (rows <= 0 || D <= 0) return 0;
  return (size_t)ln_blocks(rows) * 2 * D * sizeof(float);
}

int vitk_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *inv_std, const float *gamma,
                       int64_t rows, int64_t D, int64_t row_stride, const float *residual, float *dx, float *dgamma,
                       float *dbeta, float *workspace, size_t workspace_bytes, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && D > 0 && row_stride >= D, "vitk_layernorm_bwd: bad shape");
  VITK_REQUIRE(dgamma && dbeta, "vitk_layernorm_bwd: null parameter-gradient pointer");
  if (rows == 0) {
    if (vitk_fill_f32(dgamma, 0.f, D, stream)) return 1;
    return vitk_fill_f32(dbeta, 0.f, D, stream);
  }
  VITK_REQUIRE(dy && x && mean && inv_std && gamma && dx, "vitk_layernorm_bwd: null pointer");
  const int nblk = ln_blocks(rows);
  const size_t need = (size_t)nblk * 2 * D * sizeof(float);
  VITK_REQUIRE(workspace && workspace_bytes >= need, "vitk_layernorm_bwd: workspace too small (%zu < %zu)",
               workspace_bytes, need);
  hipStream_t st = vitk::as_stream(stream);
  const dim3 grid(nblk), blk(TPB);
  const bool vec = ln_vec_ok(dy, x, gamma, dx, D, row_stride) && (residual == nullptr || vitk::aligned16(residual));
  const int nv = vec ? (int)((D + 255) / 256) : 0;
  if (g_ln_small_on && nv >= 1 && nv <= 4 && rows <= kSmallRows) {
    const int row_blocks = (int)((rows + SM_WAVES - 1) / SM_WAVES), col_blocks = (int)((D + 63) / 64);
    const dim3 sgrid((unsigned)(row_blocks + col_blocks)), sblk(64 * SM_WAVES);
#define LNS(NV) hipLaunchKernelGGL((ln_bwd_small_kernel<NV>), sgrid, sblk, 0, st, dy, x, mean, inv_std, gamma, \
                                   (long long)rows, (int)D, (long long)row_stride, residual, dx, dgamma, dbeta, col_blocks)
    switch (nv) {
      case 1: LNS(1); break;
      case 2: LNS(2); break;
      case 3: LNS(3); break;
      default: LNS(4); break;
    }
#undef LNS
    return vitk::check_launch("vitk_layernorm_bwd");
  }
#define LNB(NV) hipLaunchKernelGGL((ln_bwd_kernel<NV>), grid, blk, 0, st, dy, x, mean, inv_std, gamma, \
                                   (long long)rows, (int)D, (long long)row_stride, residual, dx, workspace)
  switch (nv) {
    case 1: LNB(1); break;
    case 2: LNB(2); break;
    case 3: LNB(3); break;
    case 4: LNB(4); break;
    default:
      LNB(0);
      if (vitk::check_launch("vitk_layernorm_bwd/dx")) return 1;
      hipLaunchKernelGGL(ln_param_grad_generic, grid, blk, 0, st, dy, x, mean, inv_std, (long long)rows, (int)D,
                         (long long)row_stride, workspace);
      break;
  }
#undef LNB
  if (vitk::check_launch("vitk_layernorm_bwd")) return 1;
  // a partial row is [dgamma(D) | dbeta(D)]: one launch reduces both
  const dim3 rg((unsigned)((2 * D + RED_COLS - 1) / RED_COLS));
  hipLaunchKernelGGL(reduce_partials_kernel, rg, dim3(RED_COLS * RED_WAVES), 0, st, workspace, nblk,
                     (long long)(2 * D), (long long)(2 * D), (long long)D, dgamma, dbeta);
  return vitk::check_launch("vitk_layernorm_bwd/final");
}

int vitk_layernorm_small_enable(int on) {
  g_ln_small_on = on != 0;
  return 0;
}

size_t vitk_layernorm_bwd_static_max_d(void) { return (size_t)kStaticMaxD; }

int vitk_layernorm_bwd_static(const float *dy, const float *x, const float *mean, const float *inv_std,
                              const float *gamma, int64_t rows, int64_t D, int64_t row_stride, const float *residual,
                              float *dx, float *dgamma, float *dbeta, vitk_stream_t stream) {
  VITK_REQUIRE(D > 0 && D <= kStaticMaxD, "vitk_layernorm_bwd_static: D = %lld is beyond the built-in scratch (D <= %lld)",
               (long long)D, kStaticMaxD);
  VITK_REQUIRE(rows >= 0, "vitk_layernorm_bwd_static: bad shape");
  const size_t need = vitk_layernorm_bwd_workspace_bytes(rows, D);
  VITK_REQUIRE(need <= (size_t)kStaticFloats * sizeof(float),
               "vitk_layernorm_bwd_static: %zu bytes of partials exceed the built-in scratch", need);
  float *ws = nullptr;
  if (static_scratch(&ws)) return 1;
  // Every partial the second stage reads is written by the first stage of this same call, so the
  // previous call's contents need no clearing.
  return vitk_layernorm_bwd(dy, x, mean, inv_std, gamma, rows, D, row_stride, residual, dx, dgamma, dbeta, ws,
                            (size_t)kStaticFloats * sizeof(float), stream);
}

}  // extern "C"
