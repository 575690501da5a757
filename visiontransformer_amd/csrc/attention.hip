// Fused multi-head scaled-dot-product attention on the CDNA4 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32): forward, and the backward either as one kernel per head (head
// dimensions 33..64, the ViT case) or as two kernels (query side, key side) for smaller heads and
// unaligned operands.
// Replaces the chain   scores = Q K^T * scale -> softmax -> P V   and its backward
// (reference: src/layers/MultiHeadAttention.cpp:96-156 forward, :160-308 backward) without ever
// writing the [B*h][N][N] score / probability / gradient matrices to memory: the forward keeps
// one row block of probabilities in accumulator registers and saves only the row maximum and row
// sum; the backward recomputes probabilities from those.
//
// Work split: one wave owns 32 rows (queries in the forward and the query-side backward, keys in
// the key-side backward) of one head; a workgroup is 4 such waves of the same head, which share
// the tiles of the other side streamed through a two-stage LDS ring.
//
// Register-level layout (lane l = li + 32*lh, li = l&31, lh = l>>5):
//   * a 32x32 accumulator tile D = A B holds D[(r&3) + 8*(r>>2) + 4*lh][li] in register r;
//   * an MFMA k-step contracts two indices, one supplied by lanes lh=0 and one by lanes lh=1:
//     operand A gives A[li][k(lh)], operand B gives B[k(lh)][li].  Contractions over the head
//     dimension pair d = (DHP/2)*lh + j, so a lane's operand fragment is one contiguous half row;
//   * an accumulator tile whose ROW index is the next contraction index can be fed straight back
//     as operand B, register by register (register r pairs rows (r&3)+8*(r>>2) and that +4).
//     Scores are therefore accumulated transposed ([key][query]) where the next product contracts
//     over keys (P V, dS K) and plain ([query][key]) where it contracts over queries (P^T dO,
//     dS^T Q): probabilities never leave registers.
#include "common.h"

#include <initializer_list>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int PT = 33;  // row pitch of a transposed tile [d][32 rows]: conflict-free transposing stores

struct SdpaParams {
  const float *q, *k, *v;  // element (b, n, h, d) at ptr[(b*N + n)*ld + h*dh + d]
  long long ld;
  float *out;              // forward output, same addressing with ldo
  const float *o, *dout;   // backward: forward output and its gradient (ldo)
  long long ldo;
  float *stats;            // [B*H][N][2]: row max of the scaled scores, row sum of exp
  float *dsum;             // [B*H][N]: sum_d dO*O (written by the query-side kernel)
  float *dq, *dk, *dv;     // gradients, addressing as q/k/v with ldg
  long long ldg;
  int B, H, N, dh;
  float scale;
  int dk_mode;
  int vec;                 // all pointers 16-byte aligned, all pitches and dh multiples of 4
#ifdef VITK_SDPA_STAMPS
  unsigned long long *dbg;  // diagnostic build only: 16 clock stamps per wave
#endif
};
// Diagnostic build (-DVITK_SDPA_STAMPS, tools/sdpa_stamps.py): lane 0 of every wave of the fused
// backward records the shader clock at the phase boundaries of query-tile iteration 3.
#ifdef VITK_SDPA_STAMPS
unsigned long long *g_sdpa_dbg = nullptr;
#define SD_STAMP(k) if (p.dbg && lane == 0 && i == 3) p.dbg[((long long)blockIdx.x * 8 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define SD_STAMP(k)
#endif


__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// Cooperative 32-row x DHP tile of a [N][ld] matrix (head columns), zero-filled outside
// (rows >= N, columns >= dh).  Thread t holds PER consecutive columns of row t/8.  Loads are
// branch-free (clamped address + select) so that they stay in flight until their LDS store.
// VEC: all pointers 16-byte aligned, all pitches and dh multiples of 4.
template <int DHP, bool VEC>
struct Tile {
  static constexpr int PER = DHP / 8;
  static constexpr int PW = DHP >= 32 ? DHP : 32;  // pitch of a plain tile [32 rows][d]
  static constexpr int PLAIN = 32 * PW, TRANS = DHP * PT;
  __device__ static void load(const float *__restrict__ base, long long ld, int row0, int N, int dh, float (&r)[PER]) {
    const int row = row0 + (threadIdx.x >> 3), c0 = (threadIdx.x & 7) * PER;
    const bool row_ok = row < N;
    const float *p = base + (long long)min(row, N - 1) * ld;
    if constexpr (VEC && PER >= 4) {
#pragma unroll
      for (int v = 0; v < PER / 4; ++v) {
        const int c = c0 + 4 * v;
        const bool ok = row_ok && c < dh;
        const f32x4 t = *reinterpret_cast<const f32x4 *>(p + (c < dh ? c : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) r[4 * v + e] = ok ? t[e] : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const int c = c0 + e;
        const float t = p[c < dh ? c : 0];
        r[e] = (row_ok && c < dh) ? t : 0.f;
      }
    }
  }
  __device__ static void store_plain(float *__restrict__ dst, const float (&r)[PER]) {
    float *d = dst + (threadIdx.x >> 3) * PW + (threadIdx.x & 7) * PER;
    if constexpr (PER >= 4) {
#pragma unroll
      for (int v = 0; v < PER / 4; ++v) {
        const f32x4 t = {r[4 * v], r[4 * v + 1], r[4 * v + 2], r[4 * v + 3]};
        *reinterpret_cast<f32x4 *>(d + 4 * v) = t;
      }
    } else {
#pragma unroll
      for (int e = 0; e < PER; ++e) d[e] = r[e];
    }
  }
  __device__ static void store_trans(float *__restrict__ dst, const float (&r)[PER]) {
    float *d = dst + ((threadIdx.x & 7) * PER) * PT + (threadIdx.x >> 3);
#pragma unroll
    for (int e = 0; e < PER; ++e) d[e * PT] = r[e];
  }
};

// One lane's operand fragment of its own row: X[row][(DHP/2)*lh + j], j < DHP/2 (zero outside).
template <int DHP, bool VEC>
__device__ __forceinline__ void load_fragment(const float *__restrict__ base, long long ld, int row, int N, int dh,
                                              int lh, float (&f)[DHP / 2]) {
  constexpr int HALF = DHP / 2;
  const bool row_ok = row < N;
  const float *p = base + (long long)min(row, N - 1) * ld;
  if constexpr (VEC && HALF % 4 == 0) {
#pragma unroll
    for (int v = 0; v < HALF / 4; ++v) {
      const int c = HALF * lh + 4 * v;
      const bool ok = row_ok && c < dh;
      const f32x4 t = *reinterpret_cast<const f32x4 *>(p + (c < dh ? c : 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) f[4 * v + e] = ok ? t[e] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
      const int c = HALF * lh + j;
      const float t = p[c < dh ? c : 0];
      f[j] = (row_ok && c < dh) ? t : 0.f;
    }
  }
}

// Store a transposed result tile set acc[t][4g+e] = X^T[d = 32t + 8g + 4lh + e][row li] as row `row`.
template <int DT, bool VEC>
__device__ __forceinline__ void store_rows(float *__restrict__ rowp, const f32x16 (&acc)[DT], int dh, int lh) {
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 32 * t + 8 * g + 4 * lh;
      if constexpr (VEC) {
        if (d0 < dh) {
          const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
          *reinterpret_cast<f32x4 *>(rowp + d0) = v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (d0 + e < dh) rowp[d0 + e] = acc[t][4 * g + e];
      }
    }
}

#define VITK_ROW_OF(r) (((r)&3) + 8 * ((r) >> 2))

// ---- forward --------------------------------------------------------------------------------------
// Pass 1 streams the key tiles and accumulates the whole score row block S^T[key][query] of the
// wave (NKT tiles x 16 registers); softmax runs on registers (a row lives in one lane pair);
// pass 2 streams the value tiles and accumulates O^T[d][query] with the probabilities as operand B.
template <int DHP, int NKT, bool VEC>
__global__ __launch_bounds__(THREADS, 2) void sdpa_fwd_kernel(const SdpaParams p) {
  using T = Tile<DHP, VEC>;
  constexpr int HALF = DHP / 2, DT = DHP >= 32 ? DHP / 32 : 1, PW = T::PW;
  constexpr int STAGE = T::TRANS > T::PLAIN ? T::TRANS : T::PLAIN;
  __shared__ __attribute__((aligned(16))) float ring[2 * STAGE];
  const int nqb = (p.N + 127) / 128;
  const int bh = blockIdx.x / nqb, qb = blockIdx.x % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int q0 = qb * 128 + wave * 32;
  const bool active = q0 < p.N;  // wave-uniform; idle waves still help to stage tiles
  const int nkt = (p.N + 31) / 32;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  const float *Q = p.q + head, *K = p.k + head, *V = p.v + head;
  if constexpr (DHP < 32) {  // padding columns of the plain tiles stay zero
    for (int i = threadIdx.x; i < 2 * STAGE; i += THREADS) ring[i] = 0.f;
    __syncthreads();
  }

  float qf[HALF];
  load_fragment<DHP, VEC>(Q, p.ld, q0 + li, p.N, p.dh, lh, qf);

  f32x16 s[NKT];
#pragma unroll
  for (int j = 0; j < NKT; ++j) s[j] = zero16();
  float lr[T::PER];
  T::load(K, p.ld, 0, p.N, p.dh, lr);
  T::store_trans(ring, lr);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    if (j < nkt) {
      const bool more_keys = j + 1 < nkt;  // next in the stream: key tile j+1, or value tile 0
      T::load(more_keys ? K : V, p.ld, more_keys ? 32 * (j + 1) : 0, p.N, p.dh, lr);
      if (active) {
        const float *ks = ring + (j & 1) * STAGE + (HALF * lh) * PT + li;
#pragma unroll
        for (int jj = 0; jj < HALF; ++jj)
          s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[jj * PT], qf[jj], s[j], 0, 0, 0);
      }
      float *nxt = ring + ((j + 1) & 1) * STAGE;
      if (more_keys) T::store_trans(nxt, lr);
      else T::store_plain(nxt, lr);
      __syncthreads();
    }
  }

  if (active) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * j + VITK_ROW_OF(r) + 4 * lh;
        const float x = key < p.N ? s[j][r] * p.scale : -INFINITY;
        s[j][r] = x;
        m = fmaxf(m, x);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(s[j][r] - m);
        s[j][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[j][r] *= inv;
    if (p.stats != nullptr && lh == 0 && q0 + li < p.N) {
      float *st = p.stats + ((long long)bh * p.N + q0 + li) * 2;
      st[0] = m;
      st[1] = l;
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] = zero16();
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    if (j < nkt) {
      if (j + 1 < nkt) T::load(V, p.ld, 32 * (j + 1), p.N, p.dh, lr);
      if (active) {
        const float *vs = ring + ((nkt + j) & 1) * STAGE + (4 * lh) * PW + li;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int t = 0; t < DT; ++t)
            o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[VITK_ROW_OF(r) * PW + 32 * t], s[j][r], o[t], 0, 0, 0);
      }
      if (j + 1 < nkt) T::store_plain(ring + ((nkt + j + 1) & 1) * STAGE, lr);
      __syncthreads();
    }
  }
  if (active && q0 + li < p.N)
    store_rows<DT, VEC>(p.out + ((long long)b * p.N + q0 + li) * p.ldo + (long long)h * p.dh, o, p.dh, lh);
}

// Stores of one key row's dV and dK (transposed accumulators), dK in the layout dk_mode asks for.
template <int DT, bool VEC>
__device__ __forceinline__ void store_key_grads(const SdpaParams &p, int b, int h, int key, int lh,
                                                const f32x16 (&dk)[DT], const f32x16 (&dv)[DT]) {
  store_rows<DT, VEC>(p.dv + ((long long)b * p.N + key) * p.ldg + (long long)h * p.dh, dv, p.dh, lh);
  if (p.dk_mode == VITK_DK_EXACT) {
    store_rows<DT, VEC>(p.dk + ((long long)b * p.N + key) * p.ldg + (long long)h * p.dh, dk, p.dh, lh);
  } else {
    // The reference forms Q^T dS as [dh][N] per head and re-reads that buffer as [N][dh]
    // (src/layers/MultiHeadAttention.cpp:207-222): element (d, key) lands at flat index d*N + key.
    float *base = p.dk + (long long)b * p.N * p.ldg + (long long)h * p.dh;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * t + VITK_ROW_OF(r) + 4 * lh;
        if (d < p.dh) {
          const int f = d * p.N + key;
          base[(long long)(f / p.dh) * p.ldg + f % p.dh] = dk[t][r];
        }
      }
  }
}

// ---- backward, query side: dQ (and dsum = rowsum(dO*O) for the key side) --------------------------
// Per key tile: S^T = K Q^T and dP^T = V dO^T (both [key][query]), P from the saved row statistics,
// dS^T = P^T (dP^T - dsum) * scale, dQ^T[d][query] += K^T dS^T with dS^T as operand B.
template <int DHP, bool VEC>
__global__ __launch_bounds__(THREADS, 2) void sdpa_bwd_dq_kernel(const SdpaParams p) {
  using T = Tile<DHP, VEC>;
  constexpr int HALF = DHP / 2, DT = DHP >= 32 ? DHP / 32 : 1, PW = T::PW;
  constexpr int STAGE = 2 * T::TRANS + T::PLAIN;  // K^T, V^T, K
  __shared__ __attribute__((aligned(16))) float ring[2 * STAGE];
  const int nqb = (p.N + 127) / 128;
  const int bh = blockIdx.x / nqb, qb = blockIdx.x % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int q0 = qb * 128 + wave * 32, q = q0 + li;
  const bool active = q0 < p.N;
  const int nkt = (p.N + 31) / 32;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  const long long ohead = (long long)b * p.N * p.ldo + (long long)h * p.dh;
  const float *Q = p.q + head, *K = p.k + head, *V = p.v + head;
  if constexpr (DHP < 32) {
    for (int i = threadIdx.x; i < 2 * STAGE; i += THREADS) ring[i] = 0.f;
    __syncthreads();
  }

  float qf[HALF], dof[HALF];
  load_fragment<DHP, VEC>(Q, p.ld, q, p.N, p.dh, lh, qf);
  load_fragment<DHP, VEC>(p.dout + ohead, p.ldo, q, p.N, p.dh, lh, dof);
  float dsum = 0.f, m = 0.f, inv = 0.f;
  if (active) {
    float of[HALF];
    load_fragment<DHP, VEC>(p.o + ohead, p.ldo, q, p.N, p.dh, lh, of);
#pragma unroll
    for (int j = 0; j < HALF; ++j) dsum += dof[j] * of[j];
    dsum += __shfl_xor(dsum, 32, 64);
    if (q < p.N) {
      const float *st = p.stats + ((long long)bh * p.N + q) * 2;
      m = st[0];
      inv = 1.0f / st[1];
      if (lh == 0) p.dsum[(long long)bh * p.N + q] = dsum;
    }
  }

  f32x16 dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) dq[t] = zero16();
  float kr[T::PER], vr[T::PER];
  T::load(K, p.ld, 0, p.N, p.dh, kr);
  T::load(V, p.ld, 0, p.N, p.dh, vr);
  T::store_trans(ring, kr);
  T::store_trans(ring + T::TRANS, vr);
  T::store_plain(ring + 2 * T::TRANS, kr);
  __syncthreads();
  for (int j = 0; j < nkt; ++j) {
    if (j + 1 < nkt) {
      T::load(K, p.ld, 32 * (j + 1), p.N, p.dh, kr);
      T::load(V, p.ld, 32 * (j + 1), p.N, p.dh, vr);
    }
    if (active) {
      const float *st = ring + (j & 1) * STAGE;
      const float *kt = st + (HALF * lh) * PT + li, *vt = kt + T::TRANS;
      const float *kp = st + 2 * T::TRANS + (4 * lh) * PW + li;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int jj = 0; jj < HALF; ++jj) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[jj * PT], qf[jj], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[jj * PT], dof[jj], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * j + VITK_ROW_OF(r) + 4 * lh;
        const float pr = key < p.N ? __expf(s[r] * p.scale - m) * inv : 0.f;
        s[r] = (pr * (dp[r] - dsum)) * p.scale;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < DT; ++t)
          dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[VITK_ROW_OF(r) * PW + 32 * t], s[r], dq[t], 0, 0, 0);
    }
    if (j + 1 < nkt) {
      float *nx = ring + ((j + 1) & 1) * STAGE;
      T::store_trans(nx, kr);
      T::store_trans(nx + T::TRANS, vr);
      T::store_plain(nx + 2 * T::TRANS, kr);
    }
    __syncthreads();
  }
  if (active && q < p.N)
    store_rows<DT, VEC>(p.dq + ((long long)b * p.N + q) * p.ldg + (long long)h * p.dh, dq, p.dh, lh);
}

template <int DHP>
struct DkvStage {
  static constexpr int value = 2 * Tile<DHP, true>::TRANS + 2 * Tile<DHP, true>::PLAIN + 96;
};

// ---- backward, key side: dK and dV ----------------------------------------------------------------
// Per query tile: S = Q K^T and dP = dO V^T (both [query][key]), P, dS as above (row statistics
// come from LDS: the query index is the register index here), then
// dV^T[d][key] += dO^T P and dK^T[d][key] += Q^T dS with P / dS as operand B.
template <int DHP, bool VEC>
__global__ __launch_bounds__(THREADS, 2) void sdpa_bwd_dkv_kernel(const SdpaParams p) {
  using T = Tile<DHP, VEC>;
  constexpr int HALF = DHP / 2, DT = DHP >= 32 ? DHP / 32 : 1, PW = T::PW;
  constexpr int STAGE = DkvStage<DHP>::value;  // Q^T, dO^T, Q, dO, {max, 1/sum, dsum}[32]
  extern __shared__ __attribute__((aligned(16))) float ring[];  // 2 * STAGE floats (above 64 KB for dh = 64)
  const int nkb = (p.N + 127) / 128;
  const int bh = blockIdx.x / nkb, kb = blockIdx.x % nkb;
  const int b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int k0 = kb * 128 + wave * 32, key = k0 + li;
  const bool active = k0 < p.N;
  const int nqt = (p.N + 31) / 32;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  const long long ohead = (long long)b * p.N * p.ldo + (long long)h * p.dh;
  const float *Q = p.q + head, *K = p.k + head, *V = p.v + head, *DO = p.dout + ohead;
  if constexpr (DHP < 32) {
    for (int i = threadIdx.x; i < 2 * STAGE; i += THREADS) ring[i] = 0.f;
    __syncthreads();
  }

  float kf[HALF], vf[HALF];
  load_fragment<DHP, VEC>(K, p.ld, key, p.N, p.dh, lh, kf);
  load_fragment<DHP, VEC>(V, p.ld, key, p.N, p.dh, lh, vf);
  const bool key_ok = key < p.N;

  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) dk[t] = zero16(), dv[t] = zero16();
  float qr[T::PER], gr[T::PER];
  // row statistics {max, sum, dsum} of query 32*i + (t & 31): fetched branch-free by every thread
  // (clamped index), finished and written to LDS by threads 0..31 when the stage is stored
  float st_max = 0.f, st_sum = 1.f, st_dsum = 0.f;
  bool st_ok = false;
#define VITK_LOAD_STATS(I)                                                    \
  {                                                                           \
    const int qq = 32 * (I) + (threadIdx.x & 31);                             \
    const long long at = (long long)bh * p.N + min(qq, p.N - 1);              \
    st_ok = qq < p.N;                                                         \
    st_max = p.stats[at * 2];                                                 \
    st_sum = p.stats[at * 2 + 1];                                             \
    st_dsum = p.dsum[at];                                                     \
  }
#define VITK_STORE_STAGE(DST)                                                 \
  {                                                                           \
    float *nx = (DST);                                                        \
    T::store_trans(nx, qr);                                                   \
    T::store_trans(nx + T::TRANS, gr);                                        \
    T::store_plain(nx + 2 * T::TRANS, qr);                                    \
    T::store_plain(nx + 2 * T::TRANS + T::PLAIN, gr);                         \
    if (threadIdx.x < 32) {                                                   \
      float *ss = nx + 2 * T::TRANS + 2 * T::PLAIN;                           \
      ss[threadIdx.x] = st_ok ? st_max : 0.f;                                 \
      ss[32 + threadIdx.x] = st_ok ? 1.0f / st_sum : 0.f;                     \
      ss[64 + threadIdx.x] = st_ok ? st_dsum : 0.f;                           \
    }                                                                         \
  }
  T::load(Q, p.ld, 0, p.N, p.dh, qr);
  T::load(DO, p.ldo, 0, p.N, p.dh, gr);
  VITK_LOAD_STATS(0)
  VITK_STORE_STAGE(ring)
  __syncthreads();
  for (int i = 0; i < nqt; ++i) {
    if (i + 1 < nqt) {
      T::load(Q, p.ld, 32 * (i + 1), p.N, p.dh, qr);
      T::load(DO, p.ldo, 32 * (i + 1), p.N, p.dh, gr);
      VITK_LOAD_STATS(i + 1)
    }
    if (active) {
      const float *st = ring + (i & 1) * STAGE;
      const float *qt = st + (HALF * lh) * PT + li, *gt = qt + T::TRANS;
      const float *qp = st + 2 * T::TRANS + (4 * lh) * PW + li, *gp = qp + T::PLAIN;
      const float *ss = st + 2 * T::TRANS + 2 * T::PLAIN + 4 * lh;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int jj = 0; jj < HALF; ++jj) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qt[jj * PT], kf[jj], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[jj * PT], vf[jj], dp, 0, 0, 0);
      }
      SD_STAMP(1)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 m4 = *reinterpret_cast<const f32x4 *>(ss + 8 * g);
        const f32x4 i4 = *reinterpret_cast<const f32x4 *>(ss + 32 + 8 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(ss + 64 + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = key_ok ? __expf(s[r] * p.scale - m4[e]) * i4[e] : 0.f;
          s[r] = pr;
          dp[r] = (pr * (dp[r] - d4[e])) * p.scale;
        }
      }
      SD_STAMP(2)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          dv[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[VITK_ROW_OF(r) * PW + 32 * t], s[r], dv[t], 0, 0, 0);
          dk[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[VITK_ROW_OF(r) * PW + 32 * t], dp[r], dk[t], 0, 0, 0);
        }
    }
    if (i + 1 < nqt) VITK_STORE_STAGE(ring + ((i + 1) & 1) * STAGE)
    __syncthreads();
  }
#undef VITK_LOAD_STATS
#undef VITK_STORE_STAGE
  if (!active || !key_ok) return;
  store_key_grads<DT, VEC>(p, b, h, key, lh, dk, dv);
}

// ---- backward, one workgroup per head (32 < dh <= 64, 16-byte accesses, N <= 256) ------------------
// Eight waves, wave w owns keys 32w .. 32w+31 (dK, dV accumulators and its K / V fragments live in
// registers for the whole kernel); the query tiles stream through LDS.  Per query tile every wave
// forms S and dP of its (query tile x key tile) block ONCE (plain, [query][key]) and uses them
// three ways: dV += dO^T P and dK += Q^T dS straight from registers (operand B), and -- after a
// 4 KB transpose of dS through a wave-private LDS buffer -- its share of dQ^T = K^T dS^T.  The
// eight partial dQ tiles are summed in a fixed order through LDS.  10 N^2 dh FLOP per head instead
// of the 14 of the two-kernel form, no row-sum pre-pass (sum_d dO*O is formed while staging).
constexpr int FT = 512;
constexpr int F_TILE = 2 * (64 * PT) + 2 * (32 * 64) + 96;  // Q^T, dO^T, Q, dO, {max, 1/sum, dsum}[32]
constexpr int F_TB = 32 * PT;                               // dS transpose buffer of one wave
constexpr int F_RBP = 68;                                   // pitch of a partial dQ tile [query][d]
constexpr int F_RB = 32 * F_RBP;
constexpr int F_LDS_FLOATS = F_TILE + 8 * F_TB + 8 * F_RB;

__global__ __launch_bounds__(FT, 1) void sdpa_bwd_fused_kernel(const SdpaParams p) {
  constexpr int HALF = 32, DT = 2, PW = 64, TRANS = 64 * PT, PLAIN = 32 * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int k0 = wave * 32, key = k0 + li;
  const bool active = k0 < p.N, key_ok = key < p.N;
  const int nt = (p.N + 31) / 32;  // query tiles = key tiles = active waves
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  const long long ohead = (long long)b * p.N * p.ldo + (long long)h * p.dh;
  const float *Q = p.q + head, *K = p.k + head, *V = p.v + head;
  const float *DO = p.dout + ohead, *O = p.o + ohead;
  float *tile = lds, *tb = lds + F_TILE + wave * F_TB, *rb_all = lds + F_TILE + 8 * F_TB;

  // this wave's key tile: fragments for S = Q K^T / dP = dO V^T (operand B) and K rows for
  // dQ^T = K^T dS^T (operand A: K[key(r, lh)][li + 32t])
  float kf[HALF], vf[HALF], kpl[32];
  load_fragment<64, true>(K, p.ld, key, p.N, p.dh, lh, kf);
  load_fragment<64, true>(V, p.ld, key, p.N, p.dh, lh, vf);
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int row = k0 + VITK_ROW_OF(r) + 4 * lh, col = li + 32 * t;
      const float x = K[(long long)min(row, p.N - 1) * p.ld + min(col, p.dh - 1)];
      kpl[2 * r + t] = (row < p.N && col < p.dh) ? x : 0.f;
    }
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) dk[t] = zero16(), dv[t] = zero16();

  // staging: thread t moves columns 4*(t%16) .. +3 of row t/16 of the Q, dO and O tiles
  const int lr = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 4;
  const bool col_ok = c0 < p.dh;
  const int cc = col_ok ? c0 : 0;
  f32x4 rq, rg, ro;
  float st_max = 0.f, st_sum = 1.f;
  bool row_ok = false;
#define VITK_LOAD_TILE(I)                                                                  \
  {                                                                                        \
    const int row = 32 * (I) + lr, rc = min(row, p.N - 1);                                 \
    row_ok = row < p.N;                                                                    \
    rq = *reinterpret_cast<const f32x4 *>(Q + (long long)rc * p.ld + cc);                  \
    rg = *reinterpret_cast<const f32x4 *>(DO + (long long)rc * p.ldo + cc);                \
    ro = *reinterpret_cast<const f32x4 *>(O + (long long)rc * p.ldo + cc);                 \
    st_max = p.stats[((long long)bh * p.N + rc) * 2];                                      \
    st_sum = p.stats[((long long)bh * p.N + rc) * 2 + 1];                                  \
  }
#define VITK_STORE_TILE()                                                                  \
  {                                                                                        \
    const bool ok = row_ok && col_ok;                                                      \
    float part = 0.f;                                                                      \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
      rq[e] = ok ? rq[e] : 0.f;                                                            \
      rg[e] = ok ? rg[e] : 0.f;                                                            \
      part += ok ? rg[e] * ro[e] : 0.f;                                                    \
      tile[(c0 + e) * PT + lr] = rq[e];                                                    \
      tile[TRANS + (c0 + e) * PT + lr] = rg[e];                                            \
    }                                                                                      \
    *reinterpret_cast<f32x4 *>(tile + 2 * TRANS + lr * PW + c0) = rq;                      \
    *reinterpret_cast<f32x4 *>(tile + 2 * TRANS + PLAIN + lr * PW + c0) = rg;              \
    _Pragma("unroll") for (int m = 8; m > 0; m >>= 1) part += __shfl_xor(part, m, 16);     \
    if ((threadIdx.x & 15) == 0) {                                                         \
      float *ss = tile + 2 * TRANS + 2 * PLAIN;                                            \
      ss[lr] = row_ok ? st_max : 0.f;                                                      \
      ss[32 + lr] = row_ok ? 1.0f / st_sum : 0.f;                                          \
      ss[64 + lr] = row_ok ? part : 0.f;                                                   \
    }                                                                                      \
  }
  VITK_LOAD_TILE(0)
  VITK_STORE_TILE()
  __syncthreads();
  for (int i = 0; i < nt; ++i) {
    SD_STAMP(0)
    if (i + 1 < nt) VITK_LOAD_TILE(i + 1)
    if (active) {
      const float *qt = tile + (HALF * lh) * PT + li, *gt = qt + TRANS;
      const float *qp = tile + 2 * TRANS + (4 * lh) * PW + li, *gp = qp + PLAIN;
      const float *ss = tile + 2 * TRANS + 2 * PLAIN + 4 * lh;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int jj = 0; jj < HALF; ++jj) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qt[jj * PT], kf[jj], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[jj * PT], vf[jj], dp, 0, 0, 0);
      }
      SD_STAMP(1)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 m4 = *reinterpret_cast<const f32x4 *>(ss + 8 * g);
        const f32x4 i4 = *reinterpret_cast<const f32x4 *>(ss + 32 + 8 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(ss + 64 + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = key_ok ? __expf(s[r] * p.scale - m4[e]) * i4[e] : 0.f;
          s[r] = pr;
          dp[r] = (pr * (dp[r] - d4[e])) * p.scale;
          tb[li * PT + VITK_ROW_OF(r) + 4 * lh] = dp[r];  // dS[query][key] -> tb[key][query]
        }
      }
      SD_STAMP(2)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          dv[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[VITK_ROW_OF(r) * PW + 32 * t], s[r], dv[t], 0, 0, 0);
          dk[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[VITK_ROW_OF(r) * PW + 32 * t], dp[r], dk[t], 0, 0, 0);
        }
      SD_STAMP(3)
      // this wave's share of dQ^T[d][query] = sum over its keys of K[key][d] dS^T[key][query]
      f32x16 dq[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) dq[t] = zero16();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dst = tb[(VITK_ROW_OF(r) + 4 * lh) * PT + li];
#pragma unroll
        for (int t = 0; t < DT; ++t) dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kpl[2 * r + t], dst, dq[t], 0, 0, 0);
      }
      SD_STAMP(4)
      float *rb = rb_all + wave * F_RB + li * F_RBP + 4 * lh;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {dq[t][4 * g], dq[t][4 * g + 1], dq[t][4 * g + 2], dq[t][4 * g + 3]};
          *reinterpret_cast<f32x4 *>(rb + 32 * t + 8 * g) = v;
        }
    }
    SD_STAMP(5)
    __syncthreads();  // every wave is done with this query tile; all partial dQ tiles are in LDS
    SD_STAMP(6)
    if (i + 1 < nt) VITK_STORE_TILE()
    {
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < nt; ++w) sum += *reinterpret_cast<const f32x4 *>(rb_all + w * F_RB + lr * F_RBP + c0);
      const int row = 32 * i + lr;
      if (row < p.N && col_ok)
        *reinterpret_cast<f32x4 *>(p.dq + ((long long)b * p.N + row) * p.ldg + (long long)h * p.dh + c0) = sum;
    }
    SD_STAMP(7)
    __syncthreads();  // next query tile staged; partial buffers free again
    SD_STAMP(8)
  }
#undef VITK_LOAD_TILE
#undef VITK_STORE_TILE
  if (!active || !key_ok) return;
  store_key_grads<DT, true>(p, b, h, key, lh, dk, dv);
}

// ---- short sequences: one wave per head ---------------------------------------------------------------
// N <= 32 tokens and dh <= 32 (the 5- and 17-token models): a head is ONE 32 x 32 tile, so a wave
// does all of it from registers -- row-operand fragments and column operands are read straight
// from global memory, nothing is staged, no barrier.  Four heads (waves) per workgroup.  The
// backward forms the score tile in both orientations (transposed for dQ, plain for dK / dV) and
// needs neither the row-sum pre-pass nor a second launch.
constexpr int HEAD_WAVES = 4;

// Column operand of a tile: X[row = VITK_ROW_OF(r) + 4 lh][d = li] for r = 0 .. 15 (zero outside).
__device__ __forceinline__ void load_columns(const float *__restrict__ base, long long ld, int N, int dh, int li, int lh,
                                             float (&c)[16]) {
  const bool d_ok = li < dh;
  const float *col = base + (d_ok ? li : 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = VITK_ROW_OF(r) + 4 * lh;
    const float t = col[(long long)min(row, N - 1) * ld];
    c[r] = (d_ok && row < N) ? t : 0.f;
  }
}

template <int DHP>
__global__ __launch_bounds__(64 * HEAD_WAVES) void sdpa_head_fwd_kernel(const SdpaParams p) {
  constexpr int HALF = DHP / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int bh = blockIdx.x * HEAD_WAVES + wave;
  if (bh >= p.B * p.H) return;  // (wave-uniform; the kernel has no barrier)
  const int b = bh / p.H, h = bh % p.H;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  float qf[HALF], kf[HALF], vc[16];
  load_fragment<DHP, true>(p.q + head, p.ld, li, p.N, p.dh, lh, qf);
  load_fragment<DHP, true>(p.k + head, p.ld, li, p.N, p.dh, lh, kf);
  load_columns(p.v + head, p.ld, p.N, p.dh, li, lh, vc);
  f32x16 s = zero16();  // S^T[key][query = li]
#pragma unroll
  for (int jj = 0; jj < HALF; ++jj) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[jj], qf[jj], s, 0, 0, 0);
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float x = VITK_ROW_OF(r) + 4 * lh < p.N ? s[r] * p.scale : -INFINITY;
    s[r] = x;
    m = fmaxf(m, x);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float e = __expf(s[r] - m);
    s[r] = e;
    l += e;
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] *= inv;
  if (p.stats != nullptr && lh == 0 && li < p.N) {
    float *st = p.stats + ((long long)bh * p.N + li) * 2;
    st[0] = m;
    st[1] = l;
  }
  f32x16 o[1] = {zero16()};  // O^T[d][query = li]
#pragma unroll
  for (int r = 0; r < 16; ++r)  // (a step contracts rows VITK_ROW_OF(r) and that + 4: past the sequence both are zero)
    if (VITK_ROW_OF(r) < p.N) o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[r], s[r], o[0], 0, 0, 0);
  if (li < p.N) store_rows<1, true>(p.out + ((long long)b * p.N + li) * p.ldo + (long long)h * p.dh, o, p.dh, lh);
}

template <int DHP>
__global__ __launch_bounds__(64 * HEAD_WAVES) void sdpa_head_bwd_kernel(const SdpaParams p) {
  constexpr int HALF = DHP / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  // two waves per head, each with half of the work (and of the registers): the even one the query
  // side, the odd one the key side
  const int bh = blockIdx.x * (HEAD_WAVES / 2) + (wave >> 1);
  const bool key_side = wave & 1;
  if (bh >= p.B * p.H) return;
  const int b = bh / p.H, h = bh % p.H;
  const long long head = (long long)b * p.N * p.ld + (long long)h * p.dh;
  const long long ohead = (long long)b * p.N * p.ldo + (long long)h * p.dh;
  const bool row_ok = li < p.N;  // li is a query where the lane stands for a column of the transposed tiles, a key otherwise
  float qf[HALF], kf[HALF], vf[HALF], gf[HALF];
  load_fragment<DHP, true>(p.q + head, p.ld, li, p.N, p.dh, lh, qf);
  load_fragment<DHP, true>(p.k + head, p.ld, li, p.N, p.dh, lh, kf);
  load_fragment<DHP, true>(p.v + head, p.ld, li, p.N, p.dh, lh, vf);
  load_fragment<DHP, true>(p.dout + ohead, p.ldo, li, p.N, p.dh, lh, gf);
  float dsum = 0.f, m = 0.f, inv = 0.f;  // of query li
  {
    float of[HALF];
    load_fragment<DHP, true>(p.o + ohead, p.ldo, li, p.N, p.dh, lh, of);
#pragma unroll
    for (int j = 0; j < HALF; ++j) dsum += gf[j] * of[j];
    dsum += __shfl_xor(dsum, 32, 64);
    if (row_ok) {
      const float *st = p.stats + ((long long)bh * p.N + li) * 2;
      m = st[0];
      inv = 1.0f / st[1];
    }
  }
  // query side: S^T = K Q^T, dP^T = V dO^T ([key][query = li]); dQ^T[d][query] = K^T dS^T
  if (!key_side) {
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int jj = 0; jj < HALF; ++jj) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[jj], qf[jj], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[jj], gf[jj], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = VITK_ROW_OF(r) + 4 * lh < p.N ? __expf(s[r] * p.scale - m) * inv : 0.f;
      s[r] = (pr * (dp[r] - dsum)) * p.scale;
    }
    float kc[16];
    load_columns(p.k + head, p.ld, p.N, p.dh, li, lh, kc);
    f32x16 dq[1] = {zero16()};
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (VITK_ROW_OF(r) < p.N) dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[r], s[r], dq[0], 0, 0, 0);
    if (row_ok) store_rows<1, true>(p.dq + ((long long)b * p.N + li) * p.ldg + (long long)h * p.dh, dq, p.dh, lh);
    return;
  }
  // key side: S = Q K^T, dP = dO V^T ([query][key = li]); dV^T[d][key] = dO^T P, dK^T[d][key] = Q^T dS
  f32x16 s = zero16(), dp = zero16();
#pragma unroll
  for (int jj = 0; jj < HALF; ++jj) {
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[jj], kf[jj], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[jj], vf[jj], dp, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {  // the statistics of query VITK_ROW_OF(r) + 4 lh sit in that lane
    const int query = VITK_ROW_OF(r) + 4 * lh;
    const float mq = __shfl(m, query, 64), iq = __shfl(inv, query, 64), dq_ = __shfl(dsum, query, 64);
    const float pr = row_ok ? __expf(s[r] * p.scale - mq) * iq : 0.f;  // (iq = 0 for queries past the sequence)
    s[r] = pr;
    dp[r] = (pr * (dp[r] - dq_)) * p.scale;
  }
  float gc[16], qc[16];
  load_columns(p.dout + ohead, p.ldo, p.N, p.dh, li, lh, gc);
  load_columns(p.q + head, p.ld, p.N, p.dh, li, lh, qc);
  f32x16 dk[1] = {zero16()}, dv[1] = {zero16()};
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (VITK_ROW_OF(r) < p.N) {
      dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[r], s[r], dv[0], 0, 0, 0);
      dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[r], dp[r], dk[0], 0, 0, 0);
    }
  if (row_ok) store_key_grads<1, true>(p, b, h, li, lh, dk, dv);
}

bool g_head_kernels_on = true;
bool head_kernels_apply(const SdpaParams &p) { return g_head_kernels_on && p.vec && p.N <= 32 && p.dh <= 32; }

template <int DHP, bool VEC>
int launch_fwd(const SdpaParams &p, int grid, hipStream_t st) {
  const int nkt = (p.N + 31) / 32;
  const dim3 g(grid), b(THREADS);
  if (nkt <= 1) hipLaunchKernelGGL((sdpa_fwd_kernel<DHP, 1, VEC>), g, b, 0, st, p);
  else if (nkt <= 2) hipLaunchKernelGGL((sdpa_fwd_kernel<DHP, 2, VEC>), g, b, 0, st, p);
  else if (nkt <= 4) hipLaunchKernelGGL((sdpa_fwd_kernel<DHP, 4, VEC>), g, b, 0, st, p);
  else if (nkt <= 7) hipLaunchKernelGGL((sdpa_fwd_kernel<DHP, 7, VEC>), g, b, 0, st, p);  // 224 / 16 patches + CLS
  else hipLaunchKernelGGL((sdpa_fwd_kernel<DHP, 8, VEC>), g, b, 0, st, p);
  return vitk::check_launch("vitk_sdpa_fwd");
}

template <int DHP, bool VEC>
int launch_bwd(const SdpaParams &p, dim3 grid, hipStream_t st) {
  const size_t lds = 2 * DkvStage<DHP>::value * sizeof(float);
  static std::atomic<unsigned long long> configured{0};  // raise the dynamic LDS limit once per instantiation and device
  if (vitk::first_use_on_device(&configured)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sdpa_bwd_dkv_kernel<DHP, VEC>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    VITK_REQUIRE(e == hipSuccess, "vitk_sdpa_bwd: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL((sdpa_bwd_dq_kernel<DHP, VEC>), grid, dim3(THREADS), 0, st, p);
  hipLaunchKernelGGL((sdpa_bwd_dkv_kernel<DHP, VEC>), grid, dim3(THREADS), lds, st, p);
  return vitk::check_launch("vitk_sdpa_bwd");
}

int launch_bwd_fused(const SdpaParams &p, hipStream_t st) {
  const size_t lds = F_LDS_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> configured{0};
  if (vitk::first_use_on_device(&configured)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sdpa_bwd_fused_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    VITK_REQUIRE(e == hipSuccess, "vitk_sdpa_bwd: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(sdpa_bwd_fused_kernel, dim3((unsigned)(p.B * p.H)), dim3(FT), lds, st, p);
  return vitk::check_launch("vitk_sdpa_bwd");
}

bool all_vec(std::initializer_list<const void *> ptrs, std::initializer_list<long long> pitches) {
  for (const void *q : ptrs)
    if (q != nullptr && !vitk::aligned16(q)) return false;
  for (long long v : pitches)
    if (v % 4 != 0) return false;
  return true;
}

}  // namespace

extern "C" {

int vitk_sdpa_head_kernels_enable(int on) {
  g_head_kernels_on = on != 0;
  return 0;
}

int vitk_sdpa_supported(int32_t N, int32_t dh) { return N >= 1 && N <= 256 && dh >= 1 && dh <= 64; }

int vitk_sdpa_fwd(const float *q, const float *k, const float *v, int64_t ld_qkv, int32_t B, int32_t H, int32_t N,
                  int32_t dh, float scale, float *out, int64_t ld_out, float *stats, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && H >= 1 && N >= 0 && dh >= 1, "vitk_sdpa_fwd: bad shape (B=%d H=%d N=%d dh=%d)", B, H, N, dh);
  if (B == 0 || N == 0) return 0;
  VITK_REQUIRE(vitk_sdpa_supported(N, dh), "vitk_sdpa_fwd: needs N <= 256 and dh <= 64 (N=%d dh=%d)", N, dh);
  VITK_REQUIRE(q && k && v && out, "vitk_sdpa_fwd: null pointer");
  VITK_REQUIRE(ld_qkv >= (int64_t)H * dh && ld_out >= (int64_t)H * dh, "vitk_sdpa_fwd: pitch smaller than H*dh");
  const long long blocks = (long long)B * H * ((N + 127) / 128);
  VITK_REQUIRE(blocks < (1ll << 31), "vitk_sdpa_fwd: too many workgroups");
  SdpaParams p{};
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.out = out; p.ldo = ld_out; p.stats = stats;
  p.B = B; p.H = H; p.N = N; p.dh = dh; p.scale = scale;
  p.vec = all_vec({q, k, v, out}, {ld_qkv, ld_out, dh});
  hipStream_t st = vitk::as_stream(stream);
  // 16-byte accesses where every pointer, pitch and dh allow them (all ViT configurations); the
  // scalar variant exists once, at the full 64-wide head tile
  if (vitk::sdpa_fwd16_applies(N, dh, p.vec != 0, scale, ld_qkv))  // head dimensions 33..64: 16-row tiles, four heads per workgroup
    return vitk::sdpa_fwd16_launch(q, k, v, ld_qkv, B, H, N, dh, scale, out, ld_out, stats, st);
  if (head_kernels_apply(p)) {  // a head is one tile: one wave per head
    const dim3 g((unsigned)((B * H + HEAD_WAVES - 1) / HEAD_WAVES)), blk(64 * HEAD_WAVES);
    if (dh <= 16) hipLaunchKernelGGL(sdpa_head_fwd_kernel<16>, g, blk, 0, st, p);
    else hipLaunchKernelGGL(sdpa_head_fwd_kernel<32>, g, blk, 0, st, p);
    return vitk::check_launch("vitk_sdpa_fwd");
  }
  if (!p.vec) return launch_fwd<64, false>(p, (int)blocks, st);
  if (dh <= 16) return launch_fwd<16, true>(p, (int)blocks, st);
  if (dh <= 32) return launch_fwd<32, true>(p, (int)blocks, st);
  return launch_fwd<64, true>(p, (int)blocks, st);
}

size_t vitk_sdpa_bwd_workspace_bytes(int32_t B, int32_t H, int32_t N) {
  return (B > 0 && H > 0 && N > 0) ? (size_t)B * H * N * sizeof(float) : 0;
}

int vitk_sdpa_bwd(const float *q, const float *k, const float *v, int64_t ld_qkv, const float *out, const float *dout,
                  int64_t ld_out, const float *stats, int32_t B, int32_t H, int32_t N, int32_t dh, float scale,
                  int32_t dk_mode, float *dq, float *dk, float *dv, int64_t ld_dqkv, float *workspace,
                  size_t workspace_bytes, vitk_stream_t stream) {
  VITK_REQUIRE(B >= 0 && H >= 1 && N >= 0 && dh >= 1, "vitk_sdpa_bwd: bad shape (B=%d H=%d N=%d dh=%d)", B, H, N, dh);
  if (B == 0 || N == 0) return 0;
  VITK_REQUIRE(vitk_sdpa_supported(N, dh), "vitk_sdpa_bwd: needs N <= 256 and dh <= 64 (N=%d dh=%d)", N, dh);
  VITK_REQUIRE(q && k && v && out && dout && stats && dq && dk && dv, "vitk_sdpa_bwd: null pointer");
  VITK_REQUIRE(dk_mode == VITK_DK_REFERENCE || dk_mode == VITK_DK_EXACT, "vitk_sdpa_bwd: unknown dk_mode %d", dk_mode);
  VITK_REQUIRE(ld_qkv >= (int64_t)H * dh && ld_out >= (int64_t)H * dh && ld_dqkv >= (int64_t)H * dh,
               "vitk_sdpa_bwd: pitch smaller than H*dh");
  VITK_REQUIRE(workspace != nullptr && workspace_bytes >= vitk_sdpa_bwd_workspace_bytes(B, H, N),
               "vitk_sdpa_bwd: needs %zu workspace bytes, got %zu", vitk_sdpa_bwd_workspace_bytes(B, H, N),
               workspace_bytes);
  const long long blocks = (long long)B * H * ((N + 127) / 128);
  VITK_REQUIRE(blocks < (1ll << 31), "vitk_sdpa_bwd: too many workgroups");
  SdpaParams p{};
  p.q = q; p.k = k; p.v = v; p.ld = ld_qkv; p.o = out; p.dout = dout; p.ldo = ld_out;
  p.stats = const_cast<float *>(stats); p.dsum = workspace;
  p.dq = dq; p.dk = dk; p.dv = dv; p.ldg = ld_dqkv;
  p.B = B; p.H = H; p.N = N; p.dh = dh; p.scale = scale; p.dk_mode = dk_mode;
#ifdef VITK_SDPA_STAMPS
  p.dbg = g_sdpa_dbg;
#endif
  p.vec = all_vec({q, k, v, out, dout, dq, dk, dv}, {ld_qkv, ld_out, ld_dqkv, dh});
  hipStream_t st = vitk::as_stream(stream);
  const dim3 grid((unsigned)blocks);
  if (vitk::sdpa_bwd16_applies(B, N, dh, p.vec != 0, ld_qkv, ld_out, ld_dqkv))  // 16-row tiles, several heads per workgroup
    return vitk::sdpa_bwd16_launch(q, k, v, ld_qkv, out, dout, ld_out, stats, B, H, N, dh, scale, dk_mode, dq, dk, dv,
                                   ld_dqkv, st);
  if (head_kernels_apply(p)) {
    const dim3 g((unsigned)((B * H + HEAD_WAVES / 2 - 1) / (HEAD_WAVES / 2))), blk(64 * HEAD_WAVES);
    if (dh <= 16) hipLaunchKernelGGL(sdpa_head_bwd_kernel<16>, g, blk, 0, st, p);
    else hipLaunchKernelGGL(sdpa_head_bwd_kernel<32>, g, blk, 0, st, p);
    return vitk::check_launch("vitk_sdpa_bwd");
  }
  if (!p.vec) return launch_bwd<64, false>(p, grid, st);
  if (dh <= 16) return launch_bwd<16, true>(p, grid, st);
  if (dh <= 32) return launch_bwd<32, true>(p, grid, st);
  return launch_bwd_fused(p, st);  // 32 < dh <= 64: one workgroup per head
}

#ifdef VITK_SDPA_STAMPS
void vitk_sdpa_set_stamps(unsigned long long *b) { g_sdpa_dbg = b; }
#endif
}  // extern "C"
