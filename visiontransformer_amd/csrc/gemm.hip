// fp32 GEMM family on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32): NN / NT / TN / TT, batched,
// fused epilogues, deterministic split-K.  Entry points: vitk_gemm_f32, vitk_gemm_workspace_bytes.
//
// Structure of one workgroup (256 threads = 4 waves, one per SIMD, arranged 2 x 2):
//   * output tile TILE x TILE (128 or 64); each wave owns a (TILE/2)^2 sub-tile = NT x NT MFMA tiles
//     of 32x32, accumulators live in registers for the whole K loop;
//   * the K loop walks BK = 16 deep slabs.  Slab t+1 is fetched global->registers (16-byte loads)
//     while slab t is consumed from LDS; it is written to the other LDS buffer after the MFMAs, so
//     there is one barrier per slab and global latency hides behind 8 x NT^2 MFMAs per wave;
//   * both operands sit in LDS as [k][m|n] so that an MFMA operand read is one conflict-free
//     ds_read_b32 per lane (lane l reads element [k0 + l/32][i0 + l%32]).  Operands whose
//     contraction index is contiguous in memory are transposed on the LDS write; the row pitch
//     (TILE+2 for those, TILE+4 for the others) keeps both kinds of write conflict-free.
// The fp32 MFMA issues at 64 FLOP/clk/SIMD (a 32x32x2 takes 64 cycles), so per wave and k-step of
// 2 the LDS traffic is only 2*NT operand reads against NT^2 MFMAs: the loop is MFMA-paced.
#include "common.h"

#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector (not HIP's f32x4 struct): keeps the staging registers out of scratch memory
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr int THREADS = 256;
// Workgroups resident per CU (= waves per SIMD) the kernels are compiled for; the split-K plan
// sizes its grid to one full round of these.
constexpr int kBlocksPerCU = 3;

struct Params {
  int M, N, K;
  const float *A;
  long long lda;
  const float *B;
  long long ldb;
  float *C;
  long long ldc;
  float alpha;
  int act;
  const float *bias;
  const float *residual;
  long long ldr;
  float *aux;
  long long ldaux;
  int batch_inner;
  long long sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;
  int split_k, k_chunk;
  float *ws;
  int tiles_m, tiles_n;
  int vec_a, vec_b;
};

// Operand whose contraction index k is contiguous in memory: element (row, k) at base[row*ld + k].
// Thread q handles row r = q / (BK/4), k-quad kq = q % (BK/4).
template <int ROWS>
struct KContig {
  static constexpr int NL = ROWS * BK / 4 / THREADS;
  static constexpr int LD = ROWS + 2;
  __device__ static void load(const float *__restrict__ base, long long ld, int row0, int rows, int k0, int kend,
                              bool vec, f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      const int r = q / (BK / 4), kq = q % (BK / 4);
      const int row = row0 + r, k = k0 + kq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < rows && k < kend) {
        const float *p = base + (long long)row * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const f32x4 *>(p);
        } else {
          v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
      reg[it] = v;
    }
  }
  // FULL path: per-lane byte offsets inside the tile (constant over the K loop); the tile origin
  // is a uniform pointer that advances by slab_bytes() per slab, so the loop needs no vector
  // address arithmetic.
  __device__ static void lane_offsets(long long ld, unsigned (&off)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      off[it] = (unsigned)(((q / (BK / 4)) * ld + (q % (BK / 4)) * 4) * 4);
    }
  }
  __device__ static long long origin(long long ld, int row0, int k0) { return (long long)row0 * ld + k0; }
  __device__ static long long slab_bytes(long long) { return BK * 4; }
  __device__ static void store(float *__restrict__ s, const f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      const int r = q / (BK / 4), kq = q % (BK / 4);
      float *d = s + (kq * 4) * LD + r;
      d[0] = reg[it].x;
      d[LD] = reg[it].y;
      d[2 * LD] = reg[it].z;
      d[3 * LD] = reg[it].w;
    }
  }
};

// Operand whose m/n index is contiguous in memory: element (k, row) at base[k*ld + row].
// Thread q handles k-row kk = q / (ROWS/4), row-quad rq = q % (ROWS/4).
template <int ROWS>
struct MnContig {
  static constexpr int NL = ROWS * BK / 4 / THREADS;
  static constexpr int LD = ROWS + 4;
  __device__ static void load(const float *__restrict__ base, long long ld, int row0, int rows, int k0, int kend,
                              bool vec, f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      const int kk = q / (ROWS / 4), rq = q % (ROWS / 4);
      const int k = k0 + kk, row = row0 + rq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k < kend && row < rows) {
        const float *p = base + (long long)k * ld + row;
        if (vec && row + 3 < rows) {
          v = *reinterpret_cast<const f32x4 *>(p);
        } else {
          v.x = p[0];
          if (row + 1 < rows) v.y = p[1];
          if (row + 2 < rows) v.z = p[2];
          if (row + 3 < rows) v.w = p[3];
        }
      }
      reg[it] = v;
    }
  }
  __device__ static void lane_offsets(long long ld, unsigned (&off)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      off[it] = (unsigned)(((q / (ROWS / 4)) * ld + (q % (ROWS / 4)) * 4) * 4);
    }
  }
  __device__ static long long origin(long long ld, int row0, int k0) { return (long long)k0 * ld + row0; }
  __device__ static long long slab_bytes(long long ld) { return BK * ld * 4; }
  __device__ static void store(float *__restrict__ s, const f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      const int kk = q / (ROWS / 4), rq = q % (ROWS / 4);
      *reinterpret_cast<f32x4 *>(s + kk * LD + rq * 4) = reg[it];
    }
  }
};

template <bool KC, int ROWS>
struct Operand;
template <int ROWS>
struct Operand<true, ROWS> : KContig<ROWS> {};
template <int ROWS>
struct Operand<false, ROWS> : MnContig<ROWS> {};

// FULL: every tile and every K slab is complete and both operands allow 16-byte loads, so the
// loop carries no bounds checks at all (true for every large contraction of ViT-Ti/S/B).
template <int TILE, bool TA, bool TB, bool FULL>
__global__ __launch_bounds__(THREADS, kBlocksPerCU) void gemm_kernel(const Params p) {
  using OpA = Operand<!TA, TILE>;  // A is k-contiguous when not transposed
  using OpB = Operand<TB, TILE>;   // B is k-contiguous when transposed
  constexpr int WT = TILE / 2;     // per-wave tile
  constexpr int NT = WT / 32;      // MFMA tiles per wave per dimension
  __shared__ __attribute__((aligned(16))) float As[2][BK * OpA::LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * OpB::LD];

  // XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) get a contiguous run of
  // tiles, and consecutive tiles share the A row panel, so panel re-reads hit that XCD's L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
  const int tm = swz / p.tiles_n, tn = swz % p.tiles_n;
  const int m0 = tm * TILE, n0 = tn * TILE;

  const int z = blockIdx.y;
  const int batch = z / p.split_k, ks = z % p.split_k;
  const int bo = batch / p.batch_inner, bi = batch % p.batch_inner;
  const float *__restrict__ A = p.A + bo * p.sa_o + bi * p.sa_i;
  const float *__restrict__ B = p.B + bo * p.sb_o + bi * p.sb_i;
  const int kbeg = ks * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nslab = (kend - kbeg + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[OpA::NL], rb[OpB::NL];
  const bool va = p.vec_a, vb = p.vec_b;
  // FULL: uniform tile-origin pointers (scalar registers) + constant per-lane byte offsets
  unsigned offa[OpA::NL], offb[OpB::NL];
  const char *pa = reinterpret_cast<const char *>(A + OpA::origin(p.lda, m0, kbeg));
  const char *pb = reinterpret_cast<const char *>(B + OpB::origin(p.ldb, n0, kbeg));
  const long long step_a = OpA::slab_bytes(p.lda), step_b = OpB::slab_bytes(p.ldb);
  if constexpr (FULL) {
    OpA::lane_offsets(p.lda, offa);
    OpB::lane_offsets(p.ldb, offb);
  }
// (a macro rather than a lambda: arrays captured by reference end up in scratch memory)
#define VITK_FETCH(T)                                                                                       \
  if constexpr (FULL) {                                                                                     \
    _Pragma("unroll") for (int it = 0; it < OpA::NL; ++it) ra[it] =                                         \
        *reinterpret_cast<const f32x4 *>(pa + offa[it]);                                                   \
    _Pragma("unroll") for (int it = 0; it < OpB::NL; ++it) rb[it] =                                         \
        *reinterpret_cast<const f32x4 *>(pb + offb[it]);                                                   \
    pa += step_a;                                                                                           \
    pb += step_b;                                                                                           \
  } else {                                                                                                  \
    OpA::load(A, p.lda, m0, p.M, kbeg + (T)*BK, kend, va, ra);                                              \
    OpB::load(B, p.ldb, n0, p.N, kbeg + (T)*BK, kend, vb, rb);                                              \
  }
  if (nslab > 0) {
    VITK_FETCH(0)
    OpA::store(As[0], ra);
    OpB::store(Bs[0], rb);
  }
  __syncthreads();

  for (int t = 0; t < nslab; ++t) {
    const int cur = t & 1;
    if (t + 1 < nslab) {
      VITK_FETCH(t + 1)
    }
    const float *__restrict__ as = As[cur] + wr * WT + li + lh * OpA::LD;
    const float *__restrict__ bs = Bs[cur] + wc * WT + li + lh * OpB::LD;
    // Operand reads run one k-step ahead of the MFMAs that consume them (two register sets), and
    // the scheduler is pinned to that order: hoisting all 32 reads of a slab would cost 32 extra
    // registers, push the global prefetch registers into scratch and serialise on their loads.
    float a[2][NT], b[2][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) a[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < NT; ++j) b[0][j] = bs[j * 32];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      if (ks + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < NT; ++i) a[n][i] = as[(2 * ks + 2) * OpA::LD + i * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[n][j] = bs[(2 * ks + 2) * OpB::LD + j * 32];
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);   // the next step's LDS reads ...
      __builtin_amdgcn_sched_group_barrier(0x008, NT * NT, 0);  // ... then this step's MFMAs
    }
    if (t + 1 < nslab) {
      OpA::store(As[cur ^ 1], ra);
      OpB::store(Bs[cur ^ 1], rb);
    }
    __syncthreads();
  }
#undef VITK_FETCH

  // Epilogue.  Accumulator register r of lane l holds C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] of its
  // 32x32 tile: per register, lanes 0-31 and 32-63 each write one 128-byte row segment.
  // Addresses are a uniform per-register base plus one per-lane byte offset per pitch, so the 64
  // stores (and the residual / pre-activation loads) cost no vector registers for addressing.
  const int row_w = m0 + wr * WT, col_w = n0 + wc * WT;  // this wave's sub-tile origin
  if (p.split_k > 1) {
    char *W = reinterpret_cast<char *>(p.ws + (long long)z * p.M * p.N + (long long)row_w * p.N + col_w);
    const unsigned lane_w = (unsigned)((4 * lh * (long long)p.N + li) * 4);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = col_w + j * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
          const int row = row_w + dr + 4 * lh;
          if (FULL || (row < p.M && col < p.N))
            *reinterpret_cast<float *>(W + ((long long)dr * p.N + j * 32) * 4 + lane_w) = acc[i][j][r];
        }
      }
    return;
  }

  char *C = reinterpret_cast<char *>(p.C + bo * p.sc_o + bi * p.sc_i + (long long)row_w * p.ldc + col_w);
  const char *R = reinterpret_cast<const char *>(p.residual ? p.residual + (long long)row_w * p.ldr + col_w : nullptr);
  char *X = reinterpret_cast<char *>(p.aux ? p.aux + (long long)row_w * p.ldaux + col_w : nullptr);
  const unsigned lane_c = (unsigned)((4 * lh * p.ldc + li) * 4);
  const unsigned lane_r = (unsigned)((4 * lh * p.ldr + li) * 4);
  const unsigned lane_x = (unsigned)((4 * lh * p.ldaux + li) * 4);
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col_w + j * 32 + li;
      const float bv = (p.bias != nullptr && (FULL || col < p.N)) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
        const int row = row_w + dr + 4 * lh;
        if (FULL || (row < p.M && col < p.N)) {
          float v = p.alpha * acc[i][j][r];
          float *xp = reinterpret_cast<float *>(X + ((long long)dr * p.ldaux + j * 32) * 4 + lane_x);
          if (p.act == VITK_ACT_GELU_BWD) v *= vitk::gelu_grad(*xp);
          v += bv;
          if (p.act == VITK_ACT_GELU_FWD) {
            if (p.aux != nullptr) *xp = v;
            v = vitk::gelu_fwd(v);
          }
          if (p.residual != nullptr)
            v += *reinterpret_cast<const float *>(R + ((long long)dr * p.ldr + j * 32) * 4 + lane_r);
          *reinterpret_cast<float *>(C + ((long long)dr * p.ldc + j * 32) * 4 + lane_c) = v;
        }
      }
      // keep the loads of one 32x32 tile from being hoisted above the stores of the previous one
      asm volatile("" ::: "memory");
    }
}

// C[b][m][n] = sum_s ws[b*S + s][m][n], s ascending (fixed order => reproducible).
__global__ void splitk_reduce_kernel(const float *__restrict__ ws, int S, int M, int N, float *__restrict__ C,
                                     long long ldc, int batch_inner, long long sc_o, long long sc_i, long long total) {
  const long long mn = (long long)M * N;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long b = idx / mn, e = idx % mn;
    const int m = (int)(e / N), n = (int)(e % N);
    const float *w = ws + b * S * mn + e;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += w[(long long)k * mn];
    const long long bo = b / batch_inner, bi = b % batch_inner;
    C[bo * sc_o + bi * sc_i + (long long)m * ldc + n] = s;
  }
}

struct Plan {
  int tile, tiles_m, tiles_n, batch, split_k, k_chunk;
};

int choose_split(long long tiles, int K) {
  // Weight-gradient GEMMs contract over all B*N token rows into a small [in][out] matrix: too few
  // output tiles to occupy 256 CUs.  Split the contraction so that tiles*splits comes just under
  // one full round of resident workgroups (256 CUs x kBlocksPerCU): every workgroup then runs
  // concurrently and they all finish together -- a second, mostly empty round would cost as much
  // as the first.
  const long long slots = 256 * kBlocksPerCU;
  if (tiles >= slots / 2) return 1;
  long long want = slots / tiles;
  long long max_by_k = K / (8 * BK);
  if (max_by_k < 1) max_by_k = 1;
  if (want > max_by_k) want = max_by_k;
  if (want > 64) want = 64;
  return (int)(want < 1 ? 1 : want);
}

int make_plan(const vitk_gemm_args *a, Plan *pl) {
  const int bo = a->batch_outer > 0 ? a->batch_outer : 1, bi = a->batch_inner > 0 ? a->batch_inner : 1;
  pl->batch = bo * bi;
  // 128-wide tiles when both extents fill at least one and either there are enough of them to
  // occupy the chip or the contraction is long enough for split-K to do so (weight gradients);
  // 64-wide for narrow outputs (attention heads, classification head, tiny models).
  const bool plain0 = a->alpha == 1.0f && a->bias == nullptr && a->residual == nullptr && a->act == VITK_ACT_NONE;
  const long long t128 = (long long)((a->M + 127) / 128) * ((a->N + 127) / 128) * pl->batch;
  const bool can_split = plain0 && a->split_k != 1 && a->K >= 4096;
  pl->tile = (a->M >= 128 && a->N >= 128 && (t128 >= 64 || can_split)) ? 128 : 64;
  pl->tiles_m = (a->M + pl->tile - 1) / pl->tile;
  pl->tiles_n = (a->N + pl->tile - 1) / pl->tile;
  const bool plain = a->alpha == 1.0f && a->bias == nullptr && a->residual == nullptr && a->act == VITK_ACT_NONE;
  int s = a->split_k;
  if (s == 0) s = plain ? choose_split((long long)pl->tiles_m * pl->tiles_n * pl->batch, a->K) : 1;
  if (s < 1) s = 1;
  int chunk = ((a->K + s - 1) / s + BK - 1) / BK * BK;
  if (chunk < BK) chunk = BK;
  s = (a->K + chunk - 1) / chunk;
  if (s < 1) s = 1;
  pl->split_k = s;
  pl->k_chunk = chunk;
  return 0;
}

template <int TILE, bool FULL>
void launch2(const Params &p, bool ta, bool tb, dim3 grid, hipStream_t st) {
  if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<TILE, false, false, FULL>), grid, dim3(THREADS), 0, st, p);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<TILE, false, true, FULL>), grid, dim3(THREADS), 0, st, p);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<TILE, true, false, FULL>), grid, dim3(THREADS), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<TILE, true, true, FULL>), grid, dim3(THREADS), 0, st, p);
}

template <int TILE>
void launch(const Params &p, bool ta, bool tb, bool full, dim3 grid, hipStream_t st) {
  if (full) launch2<TILE, true>(p, ta, tb, grid, st);
  else launch2<TILE, false>(p, ta, tb, grid, st);
}

bool strides_vec_ok(const float *p, long long ld, long long so, long long si) {
  return vitk::aligned16(p) && ld % 4 == 0 && so % 4 == 0 && si % 4 == 0;
}

// ---- optional per-launch timing (measurement only) -------------------------------------------
// When enabled, every main GEMM launch is bracketed by two HIP events on its own stream and
// tagged with its kernel variant and algorithmic FLOPs; vitk_gemm_profile_read sums them up.
struct ProfRec {
  int variant;
  double flops;
  hipEvent_t e0, e1;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t prof_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

int variant_of(int tile, bool ta, bool tb, bool full) {
  return (full ? 8 : 0) + (tile == 128 ? 0 : 4) + (ta ? 2 : 0) + (tb ? 1 : 0);
}

}  // namespace

extern "C" size_t vitk_gemm_workspace_bytes(const vitk_gemm_args *a) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->K <= 0) return 0;
  Plan pl;
  make_plan(a, &pl);
  if (pl.split_k <= 1) return 0;
  return (size_t)pl.split_k * pl.batch * a->M * a->N * sizeof(float);
}

extern "C" int vitk_gemm_f32(const vitk_gemm_args *a, vitk_stream_t stream) {
  VITK_REQUIRE(a != nullptr, "vitk_gemm_f32: null args");
  VITK_REQUIRE(a->M >= 0 && a->N >= 0 && a->K >= 0, "vitk_gemm_f32: negative extent (M=%d N=%d K=%d)", a->M, a->N,
               a->K);
  if (a->M == 0 || a->N == 0) return 0;
  VITK_REQUIRE(a->A && a->B && a->C, "vitk_gemm_f32: null operand pointer");
  VITK_REQUIRE(a->lda >= (a->trans_a ? a->M : a->K), "vitk_gemm_f32: lda %lld too small", (long long)a->lda);
  VITK_REQUIRE(a->ldb >= (a->trans_b ? a->K : a->N), "vitk_gemm_f32: ldb %lld too small", (long long)a->ldb);
  VITK_REQUIRE(a->ldc >= a->N, "vitk_gemm_f32: ldc %lld < N %d", (long long)a->ldc, a->N);
  VITK_REQUIRE(a->act >= VITK_ACT_NONE && a->act <= VITK_ACT_GELU_BWD, "vitk_gemm_f32: unknown act %d", a->act);
  VITK_REQUIRE(a->act != VITK_ACT_GELU_BWD || a->aux != nullptr, "vitk_gemm_f32: GELU_BWD needs aux (pre-activation)");
  VITK_REQUIRE(a->aux == nullptr || a->ldaux >= a->N, "vitk_gemm_f32: ldaux %lld < N", (long long)a->ldaux);
  VITK_REQUIRE(a->residual == nullptr || a->ldr >= a->N, "vitk_gemm_f32: ldr %lld < N", (long long)a->ldr);

  Plan pl;
  make_plan(a, &pl);
  const bool plain = a->alpha == 1.0f && a->bias == nullptr && a->residual == nullptr && a->act == VITK_ACT_NONE;
  VITK_REQUIRE(pl.split_k == 1 || plain, "vitk_gemm_f32: split-K needs alpha == 1 and no bias/act/residual");
  const size_t need = pl.split_k > 1 ? (size_t)pl.split_k * pl.batch * a->M * a->N * sizeof(float) : 0;
  VITK_REQUIRE(need == 0 || (a->workspace != nullptr && a->workspace_bytes >= need),
               "vitk_gemm_f32: split-K=%d needs %zu workspace bytes, got %zu", pl.split_k, need, a->workspace_bytes);
  VITK_REQUIRE((long long)pl.batch * pl.split_k <= 65535, "vitk_gemm_f32: batch*split_k = %lld exceeds grid limit",
               (long long)pl.batch * pl.split_k);

  Params p;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb; p.C = a->C; p.ldc = a->ldc;
  p.alpha = a->alpha; p.act = a->act; p.bias = a->bias; p.residual = a->residual; p.ldr = a->ldr;
  p.aux = a->aux; p.ldaux = a->ldaux;
  p.batch_inner = a->batch_inner > 0 ? a->batch_inner : 1;
  p.sa_o = a->stride_a_outer; p.sa_i = a->stride_a_inner;
  p.sb_o = a->stride_b_outer; p.sb_i = a->stride_b_inner;
  p.sc_o = a->stride_c_outer; p.sc_i = a->stride_c_inner;
  p.split_k = pl.split_k; p.k_chunk = pl.k_chunk; p.ws = a->workspace;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  p.vec_a = strides_vec_ok(a->A, a->lda, a->stride_a_outer, a->stride_a_inner);
  p.vec_b = strides_vec_ok(a->B, a->ldb, a->stride_b_outer, a->stride_b_inner);

  hipStream_t st = vitk::as_stream(stream);
  dim3 grid((unsigned)(pl.tiles_m * pl.tiles_n), (unsigned)(pl.batch * pl.split_k), 1);
  // every K chunk starts and ends on a slab boundary when K and the chunk are multiples of BK
  const bool full = p.vec_a && p.vec_b && a->M % pl.tile == 0 && a->N % pl.tile == 0 && a->K > 0 &&
                    a->K % BK == 0 && pl.k_chunk % BK == 0 && a->lda < (1 << 21) && a->ldb < (1 << 21) &&
                    a->ldc < (1 << 21) && a->ldr < (1 << 21) && a->ldaux < (1 << 21);
  ProfRec rec{};
  if (g_prof_on) {
    rec.variant = variant_of(pl.tile, a->trans_a != 0, a->trans_b != 0, full);
    rec.flops = 2.0 * a->M * a->N * a->K * pl.batch;
    rec.e0 = prof_event();
    rec.e1 = prof_event();
    (void)hipEventRecord(rec.e0, st);
  }
  if (pl.tile == 128) launch<128>(p, a->trans_a != 0, a->trans_b != 0, full, grid, st);
  else launch<64>(p, a->trans_a != 0, a->trans_b != 0, full, grid, st);
  if (g_prof_on) {
    (void)hipEventRecord(rec.e1, st);
    g_prof.push_back(rec);
  }
  if (vitk::check_launch("vitk_gemm_f32")) return 1;
  if (pl.split_k > 1) {
    const long long total = (long long)pl.batch * a->M * a->N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(vitk::grid_for(total, 256)), dim3(256), 0, st, a->workspace,
                       pl.split_k, a->M, a->N, a->C, a->ldc, p.batch_inner, p.sc_o, p.sc_i, total);
    if (vitk::check_launch("vitk_gemm_f32/splitk_reduce")) return 1;
  }
  return 0;
}

extern "C" int vitk_gemm_profile_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

extern "C" int vitk_gemm_profile_read(int variant, double *total_ms, double *total_flops, int64_t *launches,
                                      int reset) {
  VITK_REQUIRE(variant >= 0 && variant < 16, "vitk_gemm_profile_read: variant %d outside [0,16)", variant);
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
  for (const ProfRec &r : g_prof) {
    if (r.variant != variant) continue;
    hipError_t e = hipEventSynchronize(r.e1);
    VITK_REQUIRE(e == hipSuccess, "vitk_gemm_profile_read: %s", hipGetErrorString(e));
    float t = 0.f;
    e = hipEventElapsedTime(&t, r.e0, r.e1);
    VITK_REQUIRE(e == hipSuccess, "vitk_gemm_profile_read: %s", hipGetErrorString(e));
    ms += t;
    fl += r.flops;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  if (reset) {
    for (const ProfRec &r : g_prof) {
      g_event_pool.push_back(r.e0);
      g_event_pool.push_back(r.e1);
    }
    g_prof.clear();
  }
  return 0;
}

extern "C" const char *vitk_gemm_variant_name(int variant) {
  static const char *names[16] = {
      "gemm_kernel<128, false, false, false>", "gemm_kernel<128, false, true, false>",
      "gemm_kernel<128, true, false, false>",  "gemm_kernel<128, true, true, false>",
      "gemm_kernel<64, false, false, false>",  "gemm_kernel<64, false, true, false>",
      "gemm_kernel<64, true, false, false>",   "gemm_kernel<64, true, true, false>",
      "gemm_kernel<128, false, false, true>",  "gemm_kernel<128, false, true, true>",
      "gemm_kernel<128, true, false, true>",   "gemm_kernel<128, true, true, true>",
      "gemm_kernel<64, false, false, true>",   "gemm_kernel<64, false, true, true>",
      "gemm_kernel<64, true, false, true>",    "gemm_kernel<64, true, true, true>"};
  return (variant >= 0 && variant < 16) ? names[variant] : "";
}
