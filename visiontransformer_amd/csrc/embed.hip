// Index-shuffling kernels around the embeddings: patch extraction, CLS/positional assembly and
// its backward split, CLS gather/scatter, and the reference's key-gradient re-read.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int TPB = 256;

// V = 4: four consecutive elements of a patch row per thread (they are consecutive in the image
// too: P % 4 == 0), 16-byte accesses; V = 1: any shape.
template <int V>
__global__ void patch_gather_kernel(const float *__restrict__ img, int B, int C, int HW, int P, int mode,
                                    float *__restrict__ out) {
  const int np = HW / P, pd = P * P * C;
  const long long total = (long long)B * np * np * pd / V, img_sz = (long long)C * HW * HW;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const long long i = q * V;
    const int j = (int)(i % pd);
    const long long pidx = i / pd;
    const int pw = (int)(pidx % np), ph = (int)((pidx / np) % np);
    const long long b = pidx / ((long long)np * np);
    long long src;
    if (mode == VITK_PATCH_REFERENCE) {
      src = b * img_sz + (long long)ph * P * HW + (long long)pw * P + j;  // contiguous run
    } else {
      const int c = j / (P * P), ii = (j / P) % P, jj = j % P;
      src = b * img_sz + ((long long)c * HW + ph * P + ii) * HW + pw * P + jj;
    }
    if constexpr (V == 4) *reinterpret_cast<float4 *>(out + i) = *reinterpret_cast<const float4 *>(img + src);
    else out[i] = img[src];
  }
}

// Backward of the patch extraction ("un-patch"): the image gradient from the patch-row gradients
// dpatches [B*Np][P*P*C].  Every image pixel belongs to exactly one patch (stride = patch size),
// so this is a pure gather per pixel.  VITK_PATCH_IM2COL: pixel (b, c, ph*P+i, pw*P+j) takes
// element (c, i, j) of its own patch's row.  VITK_PATCH_REFERENCE: what the reference does
// (src/layers/PatchEmbedding.cpp:86-94) -- it takes the row pointer through getData(), which ignores
// the slice offset, so EVERY patch of EVERY image receives row 0 (patch 0 of image 0), laid out
// with the (c, i, j) geometry.
__global__ void patch_scatter_kernel(const float *__restrict__ dpatches, int B, int C, int HW, int P, int mode,
                                     float *__restrict__ dimg) {
  const int np = HW / P;
  const long long img_sz = (long long)C * HW * HW, total = (long long)B * img_sz;
  const long long pd = (long long)P * P * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / img_sz;
    const int c = (int)((i % img_sz) / ((long long)HW * HW));
    const int y = (int)((i / HW) % HW), x = (int)(i % HW);
    const int ph = y / P, ii = y % P, pw = x / P, jj = x % P;
    const long long row = mode == VITK_PATCH_REFERENCE ? 0 : (b * np + ph) * np + pw;
    dimg[i] = dpatches[row * pd + ((long long)c * P + ii) * P + jj];
  }
}

// dst[r][:] = src[index[r]][:]
__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ index, long long rows,
                                   long long row_len, float *__restrict__ dst) {
  const long long total = rows * row_len;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / row_len, c = i % row_len;
    dst[i] = src[(long long)index[r] * row_len + c];
  }
}

// Rotation, translation and zoom of one image are three successive nearest-neighbour resamplings
// with zero fill (src/utils/DataAugmentation.cpp:37-143 of the reference).  Each one maps an
// output pixel to at most one source pixel, so the chain is evaluated backwards per output pixel
// -- zoom, then translation, then rotation -- and the result is one gather from the source image
// (or zero as soon as any stage falls outside).  The coordinate arithmetic repeats the
// reference's float operations one by one (no contraction into fused multiply-adds), so the
// selected source pixel is the same.
__global__ void augment_kernel(const float *__restrict__ src, const int32_t *__restrict__ index,
                               const vitk_augment_params *__restrict__ prm, long long B, int C, int H, int W,
                               float *__restrict__ dst) {
#pragma clang fp contract(off)
  const long long plane = (long long)H * W, total = B * plane;
  const float fw = (float)W, fh = (float)H, cx = fw / 2.0f, cy = fh / 2.0f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / plane;
    const int h = (int)((i % plane) / W), w = (int)(i % W);
    const vitk_augment_params P = prm[b];
    bool inside = true;
    int ch = h, cw = w;
    if (P.flags & VITK_AUG_ZOOM) {  // :114-143
      const float x = ((float)cw - cx) / P.zoom + cx;
      const float y = ((float)ch - cy) / P.zoom + cy;
      inside = x >= 0.f && x < fw && y >= 0.f && y < fh;
      cw = (int)x;
      ch = (int)y;
    }
    if (inside && (P.flags & VITK_AUG_TRANSLATE)) {  // :82-112
      ch -= P.ty;
      cw -= P.tx;
      inside = ch >= 0 && ch < H && cw >= 0 && cw < W;
    }
    if (inside && (P.flags & VITK_AUG_ROTATE)) {  // :37-80
      const float x = (float)cw - cx, y = (float)ch - cy;
      const float sx = P.cos_theta * x + P.sin_theta * y + cx;
      const float sy = -P.sin_theta * x + P.cos_theta * y + cy;
      inside = sx >= 0.f && sx < fw && sy >= 0.f && sy < fh;
      cw = (int)sx;
      ch = (int)sy;
    }
    const long long sb = index ? (long long)index[b] : b;
    const float *s = src + sb * C * plane + (long long)ch * W + cw;
    float *d = dst + b * C * plane + (long long)h * W + w;
    for (int c = 0; c < C; ++c) d[c * plane] = inside ? s[c * plane] : 0.f;
  }
}

template <int V>
__global__ void embed_assemble_kernel(const float *__restrict__ proj, const float *__restrict__ cls,
                                      const float *__restrict__ pos, int B, int Np, int D, float *__restrict__ out) {
  const int N = Np + 1;
  const long long total = (long long)B * N * D / V;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const long long i = q * V;
    const int d = (int)(i % D);
    const int n = (int)((i / D) % N);
    const long long b = i / ((long long)D * N);
    const float *base = n == 0 ? cls + d : proj + (b * Np + (n - 1)) * D + d;
    const float *pp = pos + (long long)n * D + d;
    if constexpr (V == 4) {
      const float4 a = *reinterpret_cast<const float4 *>(base), c = *reinterpret_cast<const float4 *>(pp);
      *reinterpret_cast<float4 *>(out + i) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    } else {
      out[i] = *base + *pp;
    }
  }
}

template <int V>
__global__ void embed_split_kernel(const float *__restrict__ dy, int B, int Np, int D, float *__restrict__ dproj) {
  const int N = Np + 1;
  const long long total = (long long)B * Np * D / V;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const long long i = q * V;
    const int d = (int)(i % D);
    const int p = (int)((i / D) % Np);
    const long long b = i / ((long long)D * Np);
    const float *src = dy + (b * N + 1 + p) * D + d;
    if constexpr (V == 4) *reinterpret_cast<float4 *>(dproj + i) = *reinterpret_cast<const float4 *>(src);
    else dproj[i] = *src;
  }
}

__global__ void cls_gather_kernel(const float *__restrict__ x, int B, int N, int D, float *__restrict__ out) {
  const long long total = (long long)B * D;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = x[(i / D) * N * D + (i % D)];
}

__global__ void cls_scatter_kernel(const float *__restrict__ dcls, int B, int N, int D, float *__restrict__ dx) {
  const long long total = (long long)B * N * D;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int n = (int)((i / D) % N);
    dx[i] = n == 0 ? dcls[(i / ((long long)D * N)) * D + d] : 0.f;
  }
}

__global__ void dk_scatter_kernel(const float *__restrict__ t, int B, int heads, int N, int dh, float *__restrict__ dk,
                                  long long ld) {
  const long long per = (long long)N * dh, total = (long long)B * heads * per;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long bh = i / per, f = i % per;  // f = flat index into the [dh][N] buffer, re-read as [N][dh]
    const int n = (int)(f / dh), d = (int)(f % dh);
    const long long b = bh / heads, h = bh % heads;
    dk[(b * N + n) * ld + h * dh + d] = t[i];
  }
}

}  // namespace

extern "C" {

int vitk_patch_gather(const float *img, int32_t B, int32_t C, int32_t HW, int32_t P, int32_t mode, float *patches,
                      vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && C > 0 && HW > 0 && P > 0 && HW % P == 0, "vitk_patch_gather: bad geometry (HW=%d P=%d C=%d)",
               HW, P, C);
  VITK_REQUIRE(mode == VITK_PATCH_REFERENCE || mode == VITK_PATCH_IM2COL, "vitk_patch_gather: unknown mode %d", mode);
  if (B == 0) return 0;
  VITK_REQUIRE(img && patches, "vitk_patch_gather: null pointer");
  const int np = HW / P;
  // the contiguous run of the last patch must stay inside its own image
  VITK_REQUIRE(mode != VITK_PATCH_REFERENCE ||
                   (long long)(np - 1) * P * HW + (long long)(np - 1) * P + (long long)P * P * C <=
                       (long long)C * HW * HW,
               "vitk_patch_gather: reference-mode run of the last patch leaves the image");
  const long long total = (long long)B * np * np * P * P * C;
  if (P % 4 == 0 && vitk::aligned16(img) && vitk::aligned16(patches))
    hipLaunchKernelGGL(patch_gather_kernel<4>, dim3(vitk::grid_for(total / 4, TPB)), dim3(TPB), 0,
                       vitk::as_stream(stream), img, B, C, HW, P, mode, patches);
  else
    hipLaunchKernelGGL(patch_gather_kernel<1>, dim3(vitk::grid_for(total, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                       img, B, C, HW, P, mode, patches);
  return vitk::check_launch("vitk_patch_gather");
}

int vitk_patch_scatter_bwd(const float *dpatches, int32_t B, int32_t C, int32_t HW, int32_t P, int32_t mode,
                           float *dimg, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && C > 0 && HW > 0 && P > 0 && HW % P == 0, "vitk_patch_scatter_bwd: bad shape");
  VITK_REQUIRE(mode == VITK_PATCH_REFERENCE || mode == VITK_PATCH_IM2COL, "vitk_patch_scatter_bwd: unknown mode %d", mode);
  if (B == 0) return 0;
  VITK_REQUIRE(dpatches && dimg, "vitk_patch_scatter_bwd: null pointer");
  const long long total = (long long)B * C * HW * HW;
  hipLaunchKernelGGL(patch_scatter_kernel, dim3(vitk::grid_for(total, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                     dpatches, B, C, HW, P, mode, dimg);
  return vitk::check_launch("vitk_patch_scatter_bwd");
}

int vitk_embed_assemble(const float *proj, const float *cls, const float *pos, int32_t B, int32_t Np, int32_t D,
                        float *out, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && Np >= 0 && D > 0, "vitk_embed_assemble: bad shape");
  if (B == 0) return 0;
  VITK_REQUIRE((proj || Np == 0) && cls && pos && out, "vitk_embed_assemble: null pointer");
  const long long total = (long long)B * (Np + 1) * D;
  if (D % 4 == 0 && vitk::aligned16(proj) && vitk::aligned16(cls) && vitk::aligned16(pos) && vitk::aligned16(out))
    hipLaunchKernelGGL(embed_assemble_kernel<4>, dim3(vitk::grid_for(total / 4, TPB)), dim3(TPB), 0,
                       vitk::as_stream(stream), proj, cls, pos, B, Np, D, out);
  else
    hipLaunchKernelGGL(embed_assemble_kernel<1>, dim3(vitk::grid_for(total, TPB)), dim3(TPB), 0,
                       vitk::as_stream(stream), proj, cls, pos, B, Np, D, out);
  return vitk::check_launch("vitk_embed_assemble");
}

int vitk_embed_split_bwd(const float *dy, int32_t B, int32_t Np, int32_t D, float *dproj, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && Np >= 0 && D > 0, "vitk_embed_split_bwd: bad shape");
  if (B == 0 || Np == 0) return 0;
  VITK_REQUIRE(dy && dproj, "vitk_embed_split_bwd: null pointer");
  const long long total = (long long)B * Np * D;
  if (D % 4 == 0 && vitk::aligned16(dy) && vitk::aligned16(dproj))
    hipLaunchKernelGGL(embed_split_kernel<4>, dim3(vitk::grid_for(total / 4, TPB)), dim3(TPB), 0,
                       vitk::as_stream(stream), dy, B, Np, D, dproj);
  else
    hipLaunchKernelGGL(embed_split_kernel<1>, dim3(vitk::grid_for(total, TPB)), dim3(TPB), 0, vitk::as_stream(stream),
                       dy, B, Np, D, dproj);
  return vitk::check_launch("vitk_embed_split_bwd");
}

int vitk_gather_rows_f32(const float *src, int64_t src_rows, const int32_t *index, int64_t rows, int64_t row_len,
                         float *dst, vitk_stream_t stream) {
  VITK_REQUIRE(rows >= 0 && row_len >= 0 && src_rows >= 0, "vitk_gather_rows_f32: bad shape");
  if (rows == 0 || row_len == 0) return 0;
  VITK_REQUIRE(src && index && dst && src_rows > 0, "vitk_gather_rows_f32: null pointer or empty source");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(vitk::grid_for(rows * row_len, TPB)), dim3(TPB), 0,
                     vitk::as_stream(stream), src, index, (long long)rows, (long long)row_len, dst);
  return vitk::check_launch("vitk_gather_rows_f32");
}

int vitk_augment_f32(const float *src, int64_t src_images, const int32_t *index, const vitk_augment_params *params,
                     int64_t B, int32_t C, int32_t H, int32_t W, float *dst, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && src_images >= 0, "vitk_augment_f32: bad shape");
  if (B == 0) return 0;
  VITK_REQUIRE(src && params && dst && src != dst, "vitk_augment_f32: null pointer or in-place call");
  VITK_REQUIRE(index != nullptr || src_images >= B, "vitk_augment_f32: source holds fewer images than the batch");
  hipLaunchKernelGGL(augment_kernel, dim3(vitk::grid_for(B * H * W, TPB)), dim3(TPB), 0, vitk::as_stream(stream), src,
                     index, params, (long long)B, C, H, W, dst);
  return vitk::check_launch("vitk_augment_f32");
}

int vitk_cls_gather(const float *x, int32_t B, int32_t N, int32_t D, float *cls_rows, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && N > 0 && D > 0, "vitk_cls_gather: bad shape");
  if (B == 0) return 0;
  VITK_REQUIRE(x && cls_rows, "vitk_cls_gather: null pointer");
  hipLaunchKernelGGL(cls_gather_kernel, dim3(vitk::grid_for((long long)B * D, TPB)), dim3(TPB), 0,
                     vitk::as_stream(stream), x, B, N, D, cls_rows);
  return vitk::check_launch("vitk_cls_gather");
}

int vitk_cls_scatter(const float *dcls, int32_t B, int32_t N, int32_t D, float *dx, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && N > 0 && D > 0, "vitk_cls_scatter: bad shape");
  if (B == 0) return 0;
  VITK_REQUIRE(dcls && dx, "vitk_cls_scatter: null pointer");
  hipLaunchKernelGGL(cls_scatter_kernel, dim3(vitk::grid_for((long long)B * N * D, TPB)), dim3(TPB), 0,
                     vitk::as_stream(stream), dcls, B, N, D, dx);
  return vitk::check_launch("vitk_cls_scatter");
}

int vitk_dk_reference_scatter(const float *t, int32_t B, int32_t heads, int32_t N, int32_t dh, float *dk,
                              int64_t ld_dk, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && heads > 0 && N > 0 && dh > 0 && ld_dk >= (int64_t)heads * dh, "vitk_dk_reference_scatter: bad shape");
  if (B == 0) return 0;
  VITK_REQUIRE(t && dk, "vitk_dk_reference_scatter: null pointer");
  hipLaunchKernelGGL(dk_scatter_kernel, dim3(vitk::grid_for((long long)B * heads * N * dh, TPB)), dim3(TPB), 0,
                     vitk::as_stream(stream), t, B, heads, N, dh, dk, (long long)ld_dk);
  return vitk::check_launch("vitk_dk_reference_scatter");
}

}  // extern "C"
