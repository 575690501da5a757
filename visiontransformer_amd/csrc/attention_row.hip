// Scaled-dot-product attention for ONE query token per (batch, head): forward and backward.
//
// The classification head of the reference's ViT reads only the CLS token of the last encoder
// block's output (src/model/VisionTransformer.cpp:27-37), and its backward re-injects the gradient
// into a sequence that is zero everywhere else (:49-56).  In the last block the attention core is
// therefore needed for a single query row: scores = q K^T * scale (1 x N), softmax, context = p V
// (src/layers/MultiHeadAttention.cpp:96-156 restricted to one row), and in the backward
// (:160-249) dV = p^T dO and dK = dS^T q are outer products and dQ = dS K one row.  That is
// O(N * dh) per head instead of O(N^2 * dh): bandwidth work (K and V are read once), so these are
// plain coalesced kernels with wave-level reductions, not matrix-core kernels.
//
// One workgroup of 256 threads per (batch, head); thread t owns key t (N <= 256).
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAX_DH = 64;

struct RowParams {
  const float *q;          // query token: element (b, h, d) at q[b*ldq + h*dh + d]
  long long ldq;
  const float *k, *v;      // element (b, n, h, d) at ptr[(b*N + n)*ld + h*dh + d]
  long long ld;
  float *out;              // forward: context of the query token, element (b, h, d) at out[b*ldo + h*dh + d]
  const float *o, *dout;   // backward: that context and its gradient, same addressing
  long long ldo;
  float *stats;            // [B*H][2]: maximum of the scaled scores, sum of exp(score - maximum)
  float *dq;               // [B][ldq-like compact]: element (b, h, d) at dq[b*lddq + h*dh + d]
  long long lddq;
  float *dk, *dv;          // all keys, addressing as k / v with ldg
  long long ldg;
  int B, H, N, dh;
  float scale;
  int dk_mode;
};

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *red) {
  v = is_max ? vitk::wave_max(v) : vitk::wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // red may still be read from a previous reduction
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < TPB / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// score of this thread's key: scale * sum_d q[d] K[key][d]  (q in LDS, broadcast reads)
__device__ __forceinline__ float key_dot(const float *__restrict__ krow, const float *qs, int dh, bool vec) {
  float s = 0.f;
  if (vec) {
    for (int d = 0; d < dh; d += 4) {
      const float4 kv = *reinterpret_cast<const float4 *>(krow + d);
      s += kv.x * qs[d];
      s += kv.y * qs[d + 1];
      s += kv.z * qs[d + 2];
      s += kv.w * qs[d + 3];
    }
  } else {
    for (int d = 0; d < dh; ++d) s += krow[d] * qs[d];
  }
  return s;
}

__global__ __launch_bounds__(TPB) void sdpa_row_fwd_kernel(const RowParams p, int vec) {
  __shared__ float qs[MAX_DH], prob[TPB], red[TPB / 64], part[TPB / 64][MAX_DH];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, t = threadIdx.x;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  if (t < p.dh) qs[t] = p.q[(long long)b * p.ldq + (long long)h * p.dh + t];
  __syncthreads();
  const bool key_ok = t < p.N;
  float s = -INFINITY;
  if (key_ok) s = key_dot(p.k + head + (long long)t * p.ld, qs, p.dh, vec) * p.scale;
  const float m = block_reduce(s, true, red);
  const float e = key_ok ? __expf(s - m) : 0.f;
  const float l = block_reduce(e, false, red);
  prob[t] = e * (1.0f / l);
  if (t == 0 && p.stats) {
    p.stats[2 * (long long)bh] = m;
    p.stats[2 * (long long)bh + 1] = l;
  }
  __syncthreads();
  // context[d] = sum_key prob[key] V[key][d]: wave w takes keys w, w+4, ...; lane = d (coalesced rows)
  const int lane = t & 63, wave = t >> 6;
  float acc = 0.f;
  if (lane < p.dh)
    for (int key = wave; key < p.N; key += TPB / 64) acc += prob[key] * p.v[head + (long long)key * p.ld + lane];
  part[wave][lane] = acc;
  __syncthreads();
  if (t < p.dh) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) r += part[w][t];
    p.out[(long long)b * p.ldo + (long long)h * p.dh + t] = r;
  }
}

// Backward of the same row: with p = softmax row, dP = dO V^T, dsum = dO . O,
//   dS = p (dP - dsum) * scale;  dV[key] = p[key] dO;  dK[key] = dS[key] q;  dQ = sum_key dS[key] K[key].
// dk_mode VITK_DK_REFERENCE stores the key gradient the way the reference re-reads its [dh][N]
// buffer as [N][dh] (src/layers/MultiHeadAttention.cpp:207-222): element (d, key) at flat d*N + key.
// Sixteen lanes share a key row (lane j holds columns 4j .. 4j+3), so every K / V read and every
// dK / dV write of a wave covers whole 256-byte rows; the sixteen lane groups of the workgroup
// take keys g, g+16, ...
__global__ __launch_bounds__(TPB) void sdpa_row_bwd_kernel(const RowParams p, int vec) {
  __shared__ float qs[MAX_DH], gs[MAX_DH], dsv[TPB], red[TPB / 64], part[TPB / 16][MAX_DH];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, t = threadIdx.x;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  const long long ghead = (long long)b * p.N * p.ldg + (long long)h * p.dh;
  float od = 0.f;
  if (t < MAX_DH) qs[t] = 0.f, gs[t] = 0.f;  // (columns past dh stay zero: the quads below read them)
  __syncthreads();
  if (t < p.dh) {
    qs[t] = p.q[(long long)b * p.ldq + (long long)h * p.dh + t];
    gs[t] = p.dout[(long long)b * p.ldo + (long long)h * p.dh + t];
    od = gs[t] * p.o[(long long)b * p.ldo + (long long)h * p.dh + t];
  }
  const float dsum = block_reduce(od, false, red);  // (its barriers also publish qs / gs)
  const float m = p.stats[2 * (long long)bh], inv = 1.0f / p.stats[2 * (long long)bh + 1];
  const int grp = t >> 4, j = t & 15, d0 = 4 * j;
  const bool col_ok = d0 < p.dh;
  float q4[4], g4[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) q4[e] = qs[d0 + e], g4[e] = gs[d0 + e];
  for (int key0 = 0; key0 < p.N; key0 += TPB / 16) {
    const int key = key0 + grp;
    const bool ok = key < p.N && col_ok;
    float k4[4] = {0.f, 0.f, 0.f, 0.f}, v4[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const float *krow = p.k + head + (long long)key * p.ld + d0, *vrow = p.v + head + (long long)key * p.ld + d0;
      if (vec) {
        const float4 a = *reinterpret_cast<const float4 *>(krow), c = *reinterpret_cast<const float4 *>(vrow);
        k4[0] = a.x, k4[1] = a.y, k4[2] = a.z, k4[3] = a.w;
        v4[0] = c.x, v4[1] = c.y, v4[2] = c.z, v4[3] = c.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (d0 + e < p.dh) k4[e] = krow[e], v4[e] = vrow[e];
      }
    }
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += k4[e] * q4[e], dp += v4[e] * g4[e];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1) s += __shfl_xor(s, w, 16), dp += __shfl_xor(dp, w, 16);
    const float pr = key < p.N ? __expf(s * p.scale - m) * inv : 0.f;
    const float ds = (pr * (dp - dsum)) * p.scale;
    if (j == 0 && key < TPB) dsv[key] = key < p.N ? ds : 0.f;
    if (ok) {
      float *dvrow = p.dv + ghead + (long long)key * p.ldg + d0;
      if (vec) {
        *reinterpret_cast<float4 *>(dvrow) = make_float4(pr * g4[0], pr * g4[1], pr * g4[2], pr * g4[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (d0 + e < p.dh) dvrow[e] = pr * g4[e];
      }
      if (p.dk_mode == VITK_DK_EXACT) {
        float *dkrow = p.dk + ghead + (long long)key * p.ldg + d0;
        if (vec) {
          *reinterpret_cast<float4 *>(dkrow) = make_float4(ds * q4[0], ds * q4[1], ds * q4[2], ds * q4[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (d0 + e < p.dh) dkrow[e] = ds * q4[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += ds * k4[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[grp][d0 + e] = acc[e];
  __syncthreads();  // dsv and the per-group dQ partials are complete
  if (p.dk_mode != VITK_DK_EXACT) {
    // T[d][key] = q[d] dS[key] lands at flat index f = d*N + key of the head's [N][dh] block; walk
    // the block in memory order (row f / dh, column f % dh) so that the writes are coalesced
    const int total = p.N * p.dh;
    const float inv_n = 1.0f / (float)p.N;
    for (int f = t; f < total; f += TPB) {
      int d = (int)(((float)f + 0.5f) * inv_n);  // f / N, corrected below (f < 2^24)
      if (d * p.N > f) --d;
      else if ((d + 1) * p.N <= f) ++d;
      const int key = f - d * p.N;
      const int row = p.dh == 64 ? f >> 6 : f / p.dh, col = p.dh == 64 ? f & 63 : f % p.dh;
      p.dk[ghead + (long long)row * p.ldg + col] = qs[d] * dsv[key];
    }
  }
  if (t < p.dh) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 16; ++w) r += part[w][t];
    p.dq[(long long)b * p.lddq + (long long)h * p.dh + t] = r;
  }
}

__global__ void add2d_kernel(const float *__restrict__ src, long long lds, float *__restrict__ dst, long long ldd,
                             long long rows, long long cols) {
  const long long total = rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols, c = i % cols;
    dst[r * ldd + c] += src[r * lds + c];
  }
}

bool row_vec_ok(const RowParams &p) {
  return p.dh % 4 == 0 && p.ld % 4 == 0 && p.ldg % 4 == 0 && vitk::aligned16(p.k) && vitk::aligned16(p.v) &&
         (p.dk == nullptr || (vitk::aligned16(p.dk) && vitk::aligned16(p.dv)));
}

}  // namespace

extern "C" {

int vitk_sdpa_row_supported(int32_t N, int32_t dh) { return N >= 1 && N <= TPB && dh >= 1 && dh <= MAX_DH; }

int vitk_sdpa_row_fwd(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, int32_t B, int32_t H,
                      int32_t N, int32_t dh, float scale, float *out, int64_t ld_out, float *stats,
                      vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && H >= 1 && N >= 1 && dh >= 1, "vitk_sdpa_row_fwd: bad shape (B=%d H=%d N=%d dh=%d)", B, H, N, dh);
  if (B == 0) return 0;
  VITK_REQUIRE(vitk_sdpa_row_supported(N, dh), "vitk_sdpa_row_fwd: needs N <= 256 and dh <= 64 (N=%d dh=%d)", N, dh);
  VITK_REQUIRE(q && k && v && out, "vitk_sdpa_row_fwd: null pointer");
  VITK_REQUIRE(ld_q >= (int64_t)H * dh && ld_kv >= (int64_t)H * dh && ld_out >= (int64_t)H * dh,
               "vitk_sdpa_row_fwd: pitch smaller than H*dh");
  RowParams p{};
  p.q = q; p.ldq = ld_q; p.k = k; p.v = v; p.ld = ld_kv; p.out = out; p.ldo = ld_out; p.stats = stats;
  p.B = B; p.H = H; p.N = N; p.dh = dh; p.scale = scale;
  hipLaunchKernelGGL(sdpa_row_fwd_kernel, dim3((unsigned)(B * H)), dim3(TPB), 0, vitk::as_stream(stream), p,
                     (int)row_vec_ok(p));
  return vitk::check_launch("vitk_sdpa_row_fwd");
}

int vitk_sdpa_row_bwd(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *out,
                      const float *dout, int64_t ld_out, const float *stats, int32_t B, int32_t H, int32_t N, int32_t dh,
                      float scale, int32_t dk_mode, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                      vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && H >= 1 && N >= 1 && dh >= 1, "vitk_sdpa_row_bwd: bad shape (B=%d H=%d N=%d dh=%d)", B, H, N, dh);
  if (B == 0) return 0;
  VITK_REQUIRE(vitk_sdpa_row_supported(N, dh), "vitk_sdpa_row_bwd: needs N <= 256 and dh <= 64 (N=%d dh=%d)", N, dh);
  VITK_REQUIRE(q && k && v && out && dout && stats && dq && dk && dv, "vitk_sdpa_row_bwd: null pointer");
  VITK_REQUIRE(dk_mode == VITK_DK_REFERENCE || dk_mode == VITK_DK_EXACT, "vitk_sdpa_row_bwd: unknown dk_mode %d", dk_mode);
  VITK_REQUIRE(ld_q >= (int64_t)H * dh && ld_kv >= (int64_t)H * dh && ld_out >= (int64_t)H * dh &&
                   ld_dq >= (int64_t)H * dh && ld_dkv >= (int64_t)H * dh,
               "vitk_sdpa_row_bwd: pitch smaller than H*dh");
  RowParams p{};
  p.q = q; p.ldq = ld_q; p.k = k; p.v = v; p.ld = ld_kv; p.o = out; p.dout = dout; p.ldo = ld_out;
  p.stats = const_cast<float *>(stats); p.dq = dq; p.lddq = ld_dq; p.dk = dk; p.dv = dv; p.ldg = ld_dkv;
  p.B = B; p.H = H; p.N = N; p.dh = dh; p.scale = scale; p.dk_mode = dk_mode;
  hipLaunchKernelGGL(sdpa_row_bwd_kernel, dim3((unsigned)(B * H)), dim3(TPB), 0, vitk::as_stream(stream), p,
                     (int)row_vec_ok(p));
  return vitk::check_launch("vitk_sdpa_row_bwd");
}

int vitk_add2d_f32(const float *src, int64_t ld_src, float *dst, int64_t ld_dst, int64_t rows, int64_t cols,
                   vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && cols >= 0, "vitk_add2d_f32: negative extent");
  if (rows == 0 || cols == 0) return 0;
  VITK_REQUIRE(src && dst && ld_src >= cols && ld_dst >= cols, "vitk_add2d_f32: bad pointer or pitch");
  hipLaunchKernelGGL(add2d_kernel, dim3(vitk::grid_for(rows * cols, 256)), dim3(256), 0, vitk::as_stream(stream), src,
                     (long long)ld_src, dst, (long long)ld_dst, (long long)rows, (long long)cols);
  return vitk::check_launch("vitk_add2d_f32");
}

}  // extern "C"
