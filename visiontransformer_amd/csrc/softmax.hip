// Row softmax (forward/backward), fused softmax cross-entropy, first-maximum argmax.
// One wave per row; reductions are wave shuffles.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int WPB = TPB / 64;

__global__ __launch_bounds__(TPB) void softmax_rows_kernel(const float *__restrict__ x, long long rows, int n,
                                                           long long ld, float *__restrict__ p) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = blockIdx.x * (long long)WPB + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * WPB;
  for (long long row = wave0; row < rows; row += nwaves) {
    const float *xr = x + row * ld;
    float *pr = p + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, xr[c]);
    mx = vitk::wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += expf(xr[c] - mx);
    s = vitk::wave_sum(s);
    for (int c = lane; c < ld; c += 64) pr[c] = c < n ? expf(xr[c] - mx) / s : 0.f;
  }
}

__global__ __launch_bounds__(TPB) void softmax_rows_bwd_kernel(const float *__restrict__ g, const float *__restrict__ p,
                                                               long long rows, int n, long long ld, float scale,
                                                               float *__restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = blockIdx.x * (long long)WPB + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * WPB;
  for (long long row = wave0; row < rows; row += nwaves) {
    const float *gr = g + row * ld, *pr = p + row * ld;
    float dot = 0.f;
    for (int c = lane; c < n; c += 64) dot += gr[c] * pr[c];
    dot = vitk::wave_sum(dot);
    for (int c = lane; c < ld; c += 64) dx[row * ld + c] = c < n ? (pr[c] * (gr[c] - dot)) * scale : 0.f;
  }
}

// Per row: probabilities, dlogits = (p - y)/grad_batch, weighted NLL of the labelled class(es),
// and whether the first-maximum logit is a labelled class.
__global__ __launch_bounds__(TPB) void softmax_ce_kernel(const float *__restrict__ logits, const float *__restrict__ y,
                                                         int B, int K, const float *__restrict__ w, float grad_batch,
                                                         float *__restrict__ dlogits, float *__restrict__ probs,
                                                         float *__restrict__ row_loss, float *__restrict__ row_ok) {
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = gridDim.x * WPB;
  for (int row = wave0; row < B; row += nwaves) {
    const float *xr = logits + (long long)row * K, *yr = y + (long long)row * K;
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < K; c += 64) {
      const float v = xr[c];
      if (v > mx) { mx = v; arg = c; }
    }
    // wave arg-max with ties resolved to the lowest index
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float om = __shfl_xor(mx, off, 64);
      const int oa = __shfl_xor(arg, off, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float s = 0.f;
    for (int c = lane; c < K; c += 64) s += expf(xr[c] - mx);
    s = vitk::wave_sum(s);
    float loss = 0.f;
    for (int c = lane; c < K; c += 64) {
      const float pc = expf(xr[c] - mx) / s;
      const float yc = yr[c];
      if (probs) probs[(long long)row * K + c] = pc;
      if (dlogits) dlogits[(long long)row * K + c] = (pc - yc) / grad_batch;
      if (yc == 1.0f) loss += -(w ? w[c] : 1.0f) * logf(pc + 1e-12f);
    }
    loss = vitk::wave_sum(loss);
    if (lane == 0) {
      row_loss[row] = loss;
      row_ok[row] = (arg < K && yr[arg] == 1.0f) ? 1.0f : 0.f;
    }
  }
}

// out2[0] = mean row loss, out2[1] = number of correct rows.  Single workgroup, fixed order.
__global__ __launch_bounds__(TPB) void ce_finalize_kernel(const float *__restrict__ row_loss,
                                                          const float *__restrict__ row_ok, int B,
                                                          float *__restrict__ out2) {
  __shared__ float sl[TPB], sk[TPB];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < B; i += TPB) a += row_loss[i], b += row_ok[i];
  sl[threadIdx.x] = a;
  sk[threadIdx.x] = b;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sl[threadIdx.x] += sl[threadIdx.x + s], sk[threadIdx.x] += sk[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = sl[0] / (float)B;
    out2[1] = sk[0];
  }
}

__global__ __launch_bounds__(TPB) void argmax_rows_kernel(const float *__restrict__ x, int B, int K,
                                                          int *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = gridDim.x * WPB;
  for (int row = wave0; row < B; row += nwaves) {
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < K; c += 64) {
      const float v = x[(long long)row * K + c];
      if (v > mx) { mx = v; arg = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float om = __shfl_xor(mx, off, 64);
      const int oa = __shfl_xor(arg, off, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    if (lane == 0) out[row] = arg < K ? arg : -1;
  }
}

}  // namespace

extern "C" {

int vitk_softmax_rows(const float *x, int64_t rows, int64_t n, int64_t ld, float *p, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && n >= 0 && ld >= n && n < (1ll << 30), "vitk_softmax_rows: bad shape");
  if (rows == 0 || ld == 0) return 0;
  VITK_REQUIRE(x && p, "vitk_softmax_rows: null pointer");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(vitk::grid_for(rows, WPB)), dim3(TPB), 0, vitk::as_stream(stream), x,
                     (long long)rows, (int)n, (long long)ld, p);
  return vitk::check_launch("vitk_softmax_rows");
}

int vitk_softmax_rows_bwd(const float *g, const float *p, int64_t rows, int64_t n, int64_t ld, float scale, float *dx,
                          vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && n >= 0 && ld >= n && n < (1ll << 30), "vitk_softmax_rows_bwd: bad shape");
  if (rows == 0 || ld == 0) return 0;
  VITK_REQUIRE(g && p && dx, "vitk_softmax_rows_bwd: null pointer");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(vitk::grid_for(rows, WPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     g, p, (long long)rows, (int)n, (long long)ld, scale, dx);
  return vitk::check_launch("vitk_softmax_rows_bwd");
}

int vitk_softmax_ce(const float *logits, const float *y_onehot, int32_t B, int32_t K, const float *class_w,
                    float grad_batch, float *dlogits, float *probs, float *out2, float *workspace,
                    size_t workspace_bytes, vitk_stream_t stream) {
  VITK_REQUIRE(B > 0 && K > 0, "vitk_softmax_ce: empty batch or class set (B=%d K=%d)", B, K);
  VITK_REQUIRE(logits && y_onehot && out2, "vitk_softmax_ce: null pointer");
  VITK_REQUIRE(workspace && workspace_bytes >= (size_t)2 * B * sizeof(float), "vitk_softmax_ce: workspace < 2*B floats");
  VITK_REQUIRE(grad_batch > 0.f, "vitk_softmax_ce: grad_batch must be positive");
  hipStream_t st = vitk::as_stream(stream);
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(vitk::grid_for(B, WPB)), dim3(TPB), 0, st, logits, y_onehot, B, K, class_w,
                     grad_batch, dlogits, probs, workspace, workspace + B);
  if (vitk::check_launch("vitk_softmax_ce")) return 1;
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(TPB), 0, st, workspace, workspace + B, B, out2);
  return vitk::check_launch("vitk_softmax_ce/finalize");
}

int vitk_argmax_rows(const float *x, int32_t B, int32_t K, int32_t *out, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && K > 0, "vitk_argmax_rows: bad shape");
  if (B == 0) return 0;
  VITK_REQUIRE(x && out, "vitk_argmax_rows: null pointer");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(vitk::grid_for(B, WPB)), dim3(TPB), 0, vitk::as_stream(stream), x, B, K,
                     out);
  return vitk::check_launch("vitk_argmax_rows");
}

}  // extern "C"
