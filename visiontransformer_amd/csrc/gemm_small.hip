// fp32 GEMM for problems too small to occupy the chip: one wave per 16 x 16 output tile.
//
// The tiled kernel of gemm.hip is built for throughput: a workgroup walks the contraction in
// 16-deep slabs through an LDS ring, and with a handful of workgroups on an otherwise idle chip
// every slab costs a full global-memory round trip (measured: 8 us at K = 96, 15 us at K = 384 for
// a 320-row problem whose arithmetic is worth well under a microsecond).  The small models of the
// reference (vit_mnist_4_8: 5 tokens, 96 features) consist of nothing but such contractions.
// Here nothing is staged or shared: each wave reads its 16 rows of A and 16 columns of B straight
// from global memory (they come from L2 after the first touch) in the operand layout of
// v_mfma_f32_16x16x4_f32, eight k-chunks of loads in flight ahead of the multiplies, and the whole
// problem is spread over as many SIMDs as it has 16 x 16 tiles -- times up to eight where the
// contraction is longer than one batch of loads and the tiles are few (the weight gradients of
// the 5-token model: 36 .. 144 tiles, K = 640): every wave of a workgroup then takes one batch
// (128 k, a single round trip to memory) and the accumulators are added in wave order through
// LDS (7.5 -> 5.2 us for those gradients; a four-way split with 1.25 batches per wave ran 10 us).  Same epilogue and the same column-sum side output as
// vitk_gemm_f32.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SmallParams {
  int M, N, K;
  const float *A;  // element (m, k) at A[m * sam + k * sak]
  long long sam, sak;
  const float *B;  // element (k, n) at B[k * sbk + n * sbn]
  long long sbk, sbn;
  float *C;
  long long ldc;
  float alpha;
  int act;
  const float *bias;
  const float *residual;
  long long ldr;
  float *aux;
  long long ldaux;
  float *colsum;
  int tiles_n;
  int k_per_wave;  // SPLIT > 1: the share of k of each wave (one batch of loads)
};

constexpr int U = 8;  // k-chunks (of 16) per batch of loads (4: +5 % step time on the 5-token model; 16 spills)

// Four consecutive k (k0 .. k0+3) of one row of A or one column of B; `base` points at k = 0 of it.
// VEC: k is contiguous in memory and the 16 bytes are aligned.  TAIL: k may run past K -- the
// address is clamped into the operand and the value replaced by zero, so the loads stay
// unconditional (a batch of them issues back to back, with no branch in between).
template <bool VEC, bool TAIL>
__device__ __forceinline__ f32x4 load4(const float *__restrict__ base, long long sk, int k0, int K) {
  f32x4 v;
  if constexpr (VEC) {  // K % 4 == 0 and k0 % 4 == 0: a quad is inside or outside as a whole
    const int kc = TAIL ? min(k0, K - 4) : k0;
    v = *reinterpret_cast<const f32x4 *>(base + kc);
    if (TAIL && k0 >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    const float *__restrict__ q = base + (long long)(TAIL ? min(k0, K - 1) : k0) * sk;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (TAIL) {
        const float x = base[(long long)min(k0 + e, K - 1) * sk];
        v[e] = k0 + e < K ? x : 0.f;
      } else {
        v[e] = q[e * sk];
      }
    }
  }
  return v;
}

// One 16 x 16 output tile (index `tile`, row-major over the tile grid) by the calling workgroup:
// one wave, or SPLIT waves that share the contraction.
template <bool AV, bool BV, int SPLIT>
__device__ __forceinline__ void small_tile(const SmallParams &p, const int tile) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = SPLIT > 1 ? (int)(threadIdx.x >> 6) : 0;
  const int kb = SPLIT > 1 ? min(wave * p.k_per_wave, p.K) : 0;            // this wave's k: [kb, ke)
  const int ke = SPLIT > 1 ? min(kb + p.k_per_wave, p.K) : p.K;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  // lane (c, g) feeds row m of A and column n of B at k = 16*chunk + 4*g + e into step e of the
  // chunk: v_mfma_f32_16x16x4_f32 takes A[i = lane % 16][k = lane / 16] and B[k = lane / 16][j = lane % 16].
  // Rows and columns beyond the matrix read the last valid one instead: row i of A only reaches
  // row i of the product and column j of B only column j, and those are never stored.
  const int m = tm * 16 + c, n = tn * 16 + c;
  const bool n_ok = n < p.N;
  const float *__restrict__ ap = p.A + (long long)min(m, p.M - 1) * p.sam;
  const float *__restrict__ bp = p.B + (long long)min(n, p.N - 1) * p.sbn;
  const bool do_colsum = p.colsum != nullptr && tm == 0;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, csum = acc0;
  f32x4 a0[U], b0[U], a1[U], b1[U];
  const int nfull = (ke - kb) / (16 * U);                   // batches with every k inside the contraction
  const int nbatch = (ke - kb + 16 * U - 1) / (16 * U);     // nfull, or one more (ragged)

#define VITK_LOAD(BATCH, RA, RB)                                              \
  if ((BATCH) < nfull) {                                                      \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                           \
      const int k0 = kb + ((BATCH)*U + u) * 16 + 4 * g;                       \
      RA[u] = load4<AV, false>(ap, p.sak, k0, ke);                            \
      RB[u] = load4<BV, false>(bp, p.sbk, k0, ke);                            \
    }                                                                         \
  } else {                                                                    \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                           \
      const int k0 = kb + ((BATCH)*U + u) * 16 + 4 * g;                       \
      RA[u] = load4<AV, true>(ap, p.sak, k0, ke);                             \
      RB[u] = load4<BV, true>(bp, p.sbk, k0, ke);                             \
    }                                                                         \
  }
  // two accumulators (even / odd chunks): a chain of dependent 16x16x4 MFMAs issues every 40
  // cycles, two interleaved chains every 32
#define VITK_MULTIPLY(RA, RB)                                                               \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(RA[u][e], RB[u][e], acc1, 0, 0, 0); \
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(RA[u][e], RB[u][e], acc0, 0, 0, 0);  \
    }                                                                                       \
    csum += RB[u];                                                                          \
  }

  if (nbatch > 0) { VITK_LOAD(0, a0, b0) }
  for (int bt = 0; bt < nbatch; bt += 2) {
    if (bt + 1 < nbatch) { VITK_LOAD(bt + 1, a1, b1) }
    VITK_MULTIPLY(a0, b0)
    if (bt + 2 < nbatch) { VITK_LOAD(bt + 2, a0, b0) }
    if (bt + 1 < nbatch) { VITK_MULTIPLY(a1, b1) }
  }
#undef VITK_LOAD
#undef VITK_MULTIPLY

  f32x4 acc = acc0 + acc1;
  if constexpr (SPLIT > 1) {  // add the waves' shares in wave order; wave 0 carries on alone
    __shared__ f32x4 part[2][SPLIT][64];
    part[0][wave][lane] = acc;
    part[1][wave][lane] = csum;
    __syncthreads();
    if (wave != 0) return;
    const int nw = (int)(blockDim.x >> 6);
    for (int w = 1; w < nw; ++w) acc += part[0][w][lane], csum += part[1][w][lane];
  }
  if (do_colsum) {  // wave-uniform
    // this lane added B[k][n] over its k (4g + e of every chunk); the other three k-groups of the
    // column sit 16, 32 and 48 lanes away.  Fixed order.
    float t = (csum[0] + csum[1]) + (csum[2] + csum[3]);
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    if (g == 0 && n_ok) p.colsum[n] = t;
  }

  // accumulator register r of lane (c, g) is C[4g + r][c] of the tile
  if (!n_ok) return;
  const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = tm * 16 + 4 * g + r;
    if (row >= p.M) continue;
    float v = p.alpha * acc[r];
    if (p.act == VITK_ACT_GELU_BWD) v *= vitk::gelu_grad(p.aux[(long long)row * p.ldaux + n]);
    v += bv;
    if (p.act == VITK_ACT_GELU_FWD) {
      if (p.aux != nullptr) p.aux[(long long)row * p.ldaux + n] = v;
      v = vitk::gelu_fwd(v);
    }
    if (p.residual != nullptr) v += p.residual[(long long)row * p.ldr + n];
    p.C[(long long)row * p.ldc + n] = v;
  }
}

template <bool AV, bool BV, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void gemm_small_kernel(const SmallParams p) {
  small_tile<AV, BV, SPLIT>(p, (int)blockIdx.x);
}

// Two independent products in one launch: the two halves of a Dense layer's backward pass,
// dX = dY W^T (both operands contiguous along k: one wave per tile) and dW = X^T dY (strided along
// k; SPLIT waves per tile where the contraction is long).  The first tiles0 workgroups take the
// first problem -- one tile per WAVE, so that its workgroups are as full as the second problem's,
// whose SPLIT-wave workgroups take one tile each (a workgroup of eight waves of this kernel fills
// a CU: with seven of them idle the launch would need two passes over the chip).  On models this
// small a launch costs more than its arithmetic, so one launch instead of two is the gain.
struct SmallPair {
  SmallParams first, second;
  int tiles0, groups0;  // tiles of the first problem, and the workgroups that take them
};
template <int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void gemm_small_pair_kernel(const SmallPair pp) {
  if ((int)blockIdx.x < pp.groups0) {  // workgroup-uniform
    const int tile = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    if (tile < pp.tiles0) small_tile<true, true, 1>(pp.first, tile);  // (wave-uniform; no barrier inside)
  } else {
    small_tile<false, false, SPLIT>(pp.second, (int)blockIdx.x - pp.groups0);
  }
}

bool g_small_on = true;
bool g_small_split_on = true;

}  // namespace

namespace vitk {

// Whether vitk_gemm_f32 hands these arguments to the small-problem kernel: one contraction (no
// batch, no forced split), short enough that a single wave's chain of MFMAs stays in the
// microseconds, and few enough tiles that every wave gets a SIMD of its own (or nearly).
bool gemm_small_applies(const vitk_gemm_args *a) {
  if (!g_small_on || a->reserved != 0 || a->split_k > 1) return false;
  if ((a->batch_outer > 1) || (a->batch_inner > 1)) return false;
  if (a->K < 1 || a->K > 1024) return false;
  const long long tiles = (long long)((a->M + 15) / 16) * ((a->N + 15) / 16);
  return tiles <= 8LL * cu_count();
}

namespace {
// kernel arguments of one problem; *av / *bv: whether A / B can be read 16 bytes at a time along k;
// *nw: the number of waves that share a tile's contraction (1 = no split)
SmallParams small_params(const vitk_gemm_args *a, bool *av, bool *bv, long long *tiles, int *nw_out) {
  SmallParams p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.sam = a->trans_a ? 1 : a->lda; p.sak = a->trans_a ? a->lda : 1;
  p.B = a->B; p.sbk = a->trans_b ? 1 : a->ldb; p.sbn = a->trans_b ? a->ldb : 1;
  p.C = a->C; p.ldc = a->ldc; p.alpha = a->alpha; p.act = a->act; p.bias = a->bias;
  p.residual = a->residual; p.ldr = a->ldr; p.aux = a->aux; p.ldaux = a->ldaux;
  p.colsum = a->colsum_b;
  p.tiles_n = (a->N + 15) / 16;
  *av = !a->trans_a && a->lda % 4 == 0 && a->K % 4 == 0 && aligned16(a->A);
  *bv = a->trans_b && a->ldb % 4 == 0 && a->K % 4 == 0 && aligned16(a->B);
  *tiles = (long long)((a->M + 15) / 16) * p.tiles_n;
  // a long contraction over few tiles: four waves per tile, a quarter of k each, while that still
  // leaves (nearly) every wave a SIMD of its own
  // (one batch of loads = 16 * U k per wave: every wave makes a single round trip to memory)
  // Only for the weight-gradient form (A^T B: k runs over the token rows, up to 1024 of them
  // here).  Those results go straight into the optimizer; activations and activation gradients
  // keep the single chain in k order, whose rounding stays closest to the reference's sequential
  // sum -- a different association there is amplified by every layer the value still passes
  // through (measured: twelve layers take it to the edge of the 1e-4 parity bound).
  const int nw = (a->K + 16 * U - 1) / (16 * U);
  const bool split = g_small_split_on && a->trans_a && !a->trans_b && nw >= 2 && nw <= 8 && *tiles * nw <= 4LL * cu_count();
  p.k_per_wave = split ? 16 * U : a->K;
  *nw_out = split ? nw : 1;
  return p;
}
}  // namespace

// dX = dY W^T and dW = X^T dY of one Dense layer in one launch: both small, the first with both
// operands readable 16 bytes at a time along k, the second the weight-gradient form.
bool gemm_small_pair_applies(const vitk_gemm_args *x, const vitk_gemm_args *w) {
  if (!gemm_small_applies(x) || !gemm_small_applies(w)) return false;
  if (x->trans_a || !x->trans_b || !w->trans_a || w->trans_b || x->colsum_b != nullptr) return false;
  bool av, bv;
  long long tiles;
  int nw;
  (void)small_params(x, &av, &bv, &tiles, &nw);
  return av && bv;
}

int gemm_small_pair_launch(const vitk_gemm_args *x, const vitk_gemm_args *w, hipStream_t st) {
  SmallPair pp;
  bool av, bv;
  long long t0, t1;
  int nw0, nw1;
  pp.first = small_params(x, &av, &bv, &t0, &nw0);
  pp.second = small_params(w, &av, &bv, &t1, &nw1);
  pp.tiles0 = (int)t0;
  pp.groups0 = (int)((t0 + nw1 - 1) / nw1);
  const dim3 grid((unsigned)(pp.groups0 + t1));
  if (nw1 > 1) hipLaunchKernelGGL((gemm_small_pair_kernel<8>), grid, dim3(64 * nw1), 0, st, pp);
  else hipLaunchKernelGGL((gemm_small_pair_kernel<1>), grid, dim3(64), 0, st, pp);
  return check_launch("vitk_gemm_pair_f32/small");
}

int gemm_small_launch(const vitk_gemm_args *a, hipStream_t st) {
  bool av, bv;
  long long tiles;
  int nw;
  SmallParams p = small_params(a, &av, &bv, &tiles, &nw);
  const bool split = nw > 1;
  const dim3 grid((unsigned)tiles);
#define VITK_SMALL(AV_, BV_)                                                                                  \
  if (split) hipLaunchKernelGGL((gemm_small_kernel<AV_, BV_, 8>), grid, dim3(64 * nw), 0, st, p);             \
  else hipLaunchKernelGGL((gemm_small_kernel<AV_, BV_, 1>), grid, dim3(64), 0, st, p);
  if (av && bv) { VITK_SMALL(true, true) }
  else if (av) { VITK_SMALL(true, false) }
  else if (bv) { VITK_SMALL(false, true) }
  else { VITK_SMALL(false, false) }
#undef VITK_SMALL
  return check_launch("vitk_gemm_f32/small");
}

}  // namespace vitk

extern "C" int vitk_gemm_small_enable(int on) {
  g_small_on = on != 0;      // 0: tiled kernel only; 1: the default; 2: small kernel, always one wave per tile
  g_small_split_on = on != 2;
  return 0;
}
