// fp32 GEMM family on the CDNA4 matrix cores: NN / NT / TN / TT, batched, fused epilogues,
// deterministic split-K.  Entry points: vitk_gemm_f32, vitk_gemm_workspace_bytes.
//
// Structure of one workgroup (256 threads = 4 waves, one per SIMD, arranged 2 x 2):
//   * output tile TILE x TILE (128 or 64); each wave owns a (TILE/2)^2 sub-tile whose accumulators
//     live in registers for the whole K loop;
//   * problems made of whole tiles with 16-byte-aligned operands (every large contraction of
//     ViT-Ti/S/B) run on v_mfma_f32_16x16x4_f32 with transposed accumulators (16-byte stores) and
//     read global memory through buffer descriptors; everything else runs on v_mfma_f32_32x32x2_f32
//     with plain accumulators and bounds checks.  Both instructions issue 64 FLOP/clk/SIMD, but
//     16x16x4 contracts 4 k per pass over the accumulators -- half the accumulator-file traffic
//     per FLOP -- and the chip holds a measurably higher clock under it (+5 % throughput);
//   * the K loop walks BK = 16 deep slabs through a three-slot LDS ring: while slab t is multiplied,
//     slab t+2 moves registers->LDS and slabs t+3, t+4 global->registers (16-byte loads); the one
//     barrier per slab sits in the middle of the slab's MFMAs, so operand reads never wait at a
//     slab boundary and global latency hides behind two slabs of MFMAs;
//   * both operands sit in LDS as [k][m|n] with a row pitch of TILE+4 floats, so that an MFMA
//     operand read is one conflict-free ds_read_b32 per lane.  Operands whose contraction index is
//     contiguous in memory are transposed on the LDS write (2-way, which costs nothing).
// Per wave and MFMA step the LDS traffic is only 2*T operand reads against T^2 MFMAs (T tiles per
// sub-tile edge): the loop is MFMA-paced.
#include "common.h"

#include <cstdio>
#include <type_traits>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector (not HIP's f32x4 struct): keeps the staging registers out of scratch memory
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int THREADS = 256;
constexpr int STAGES = 3;  // LDS ring depth (slabs)
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
  int tiles_m, tiles_n, strip;
  float *colsum;     // column sums of B (null = off); split-K: partial sums [split][N] in colsum_ws
  float *colsum_ws;
  int vec_a, vec_b;
#ifdef VITK_GEMM_STAMPS
  unsigned long long *stamps;  // diagnostic build only: 16 slots of clock stamps per workgroup
#endif
};
// Diagnostic build (-DVITK_GEMM_STAMPS, tools/gemm_stamps.py): thread 0 of every workgroup records
// shader-clock stamps at kernel start (0), loop start (1), loop end (2), kernel end (3), its
// HW_ID register (4) and the 100 MHz wall clock at start (5) and end (7).
#ifdef VITK_GEMM_STAMPS
#define VITK_STAMP_SLOT(i) p.stamps[(blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y) * 16 + (i)]
#define VITK_STAMP(i) \
  if (p.stamps != nullptr && threadIdx.x == 0) VITK_STAMP_SLOT(i) = __builtin_amdgcn_s_memtime();
#define VITK_STAMP_AUX()                                               \
  if (p.stamps != nullptr && threadIdx.x == 0) {                       \
    VITK_STAMP_SLOT(4) = __builtin_amdgcn_s_getreg((31 << 11) | 4);    \
    VITK_STAMP_SLOT(5) = __builtin_amdgcn_s_memrealtime();             \
  }
#define VITK_STAMP_END() \
  if (p.stamps != nullptr && threadIdx.x == 0) VITK_STAMP_SLOT(7) = __builtin_amdgcn_s_memrealtime();
#else
#define VITK_STAMP(i)
#define VITK_STAMP_AUX()
#define VITK_STAMP_END()
#endif

// Operand whose contraction index k is contiguous in memory: element (row, k) at base[row*ld + k].
// Thread q handles row r = q / (BK/4), k-quad kq = q % (BK/4).
template <int ROWS>
struct KContig {
  static constexpr int NL = ROWS * BK / 4 / THREADS;
  static constexpr int LD = ROWS + 4;  // k-rows 4 apart land 16 banks apart (see the operand reads)
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
  // is the base of a buffer descriptor and a scalar offset advances by slab_bytes() per slab, so
  // the loop needs no vector address arithmetic.
  __device__ static void lane_offsets(long long ld, unsigned (&off)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      off[it] = (unsigned)(((q / (BK / 4)) * ld + (q % (BK / 4)) * 4) * 4);
    }
  }
  __device__ static long long origin(long long ld, int row0, int k0) { return (long long)row0 * ld + k0; }
  __device__ static long long slab_bytes(long long) { return BK * 4; }
  // bytes from the tile origin to the end of what the tile may read: its own rows only (at most
  // ROWS of the `rows` that remain, so the 32-bit extent cannot wrap however large the operand
  // is), of which the last holds `ks` k
  __device__ static unsigned extent_bytes(long long ld, int rows, int ks) {
    const int r = rows < ROWS ? rows : ROWS;
    return (unsigned)(((long long)(r - 1) * ld + ks) * 4);
  }
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
  // The same store into the ring slot at literal byte offset OFF: one loop-invariant address
  // register per staged vector (lds_bases) and four ds_write_b32 whose slot and k-row offsets
  // are instruction immediates.  (Left to the compiler these become paired writes with 8-bit
  // offsets, which need a separate address register per slot and k-row pair.)
  __device__ static void lds_bases(const float *s, unsigned (&base)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      base[it] = (unsigned)(uintptr_t)(s + ((q % (BK / 4)) * 4) * LD + q / (BK / 4));
    }
  }
  template <int OFF>
  __device__ static void store_imm(const unsigned (&base)[NL], const f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base[it]), "v"(reg[it].x), "n"(OFF) : "memory");
      asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base[it]), "v"(reg[it].y), "n"(OFF + LD * 4) : "memory");
      asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base[it]), "v"(reg[it].z), "n"(OFF + 2 * LD * 4) : "memory");
      asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base[it]), "v"(reg[it].w), "n"(OFF + 3 * LD * 4) : "memory");
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
  __device__ static unsigned extent_bytes(long long ld, int rows, int ks) {
    return (unsigned)(((long long)(ks - 1) * ld + rows) * 4);
  }
  __device__ static void store(float *__restrict__ s, const f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      const int kk = q / (ROWS / 4), rq = q % (ROWS / 4);
      *reinterpret_cast<f32x4 *>(s + kk * LD + rq * 4) = reg[it];
    }
  }
  // literal ring-slot offset: the compiler folds it into the 16-byte store's immediate by itself
  __device__ static void lds_bases(const float *s, unsigned (&base)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int q = threadIdx.x + THREADS * it;
      base[it] = (unsigned)(uintptr_t)(s + (q / (ROWS / 4)) * LD + (q % (ROWS / 4)) * 4);
    }
  }
  template <int OFF>
  __device__ static void store_imm(const unsigned (&base)[NL], const f32x4 (&reg)[NL]) {
#pragma unroll
    for (int it = 0; it < NL; ++it)
      *reinterpret_cast<__attribute__((address_space(3))) f32x4 *>((uintptr_t)(base[it] + OFF)) = reg[it];
  }
};

template <bool KC, int ROWS>
struct Operand;
template <int ROWS>
struct Operand<true, ROWS> : KContig<ROWS> {};
template <int ROWS>
struct Operand<false, ROWS> : MnContig<ROWS> {};

// Workgroup barrier that orders LDS traffic only (waits for this wave's outstanding LDS operations,
// not for its global prefetch loads, which __syncthreads() would also drain).
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FULL: all operands allow 16-byte accesses and N is made of whole tiles; the loop carries no
// bounds checks at all (a ragged last tile row, and a ragged contraction of a weight-gradient
// product, are cut off by the extents of the buffer descriptors), the epilogue only a row mask.
// CS: also form the column sums of B (only with an untransposed B).
template <int TM, int TN, bool TA, bool TB, bool FULL, bool CS>
__global__ __launch_bounds__(THREADS, kBlocksPerCU) void gemm_kernel(const Params p) {
  static_assert(!(CS && TB), "column sums need an untransposed B");
  using OpA = Operand<!TA, TM>;  // A is k-contiguous when not transposed
  using OpB = Operand<TB, TN>;   // B is k-contiguous when transposed
  constexpr int WTM = TM / 2, WTN = TN / 2;  // per-wave tile
  constexpr int SA = BK * OpA::LD, SB = BK * OpB::LD;
  __shared__ __attribute__((aligned(16))) float As[STAGES * SA];
  __shared__ __attribute__((aligned(16))) float Bs[STAGES * SB];
  VITK_STAMP(0)
  VITK_STAMP_AUX()

  // XCD-aware work order.  Workgroups are handed to the 8 XCDs round-robin by linear id, and each
  // XCD has its own L2.  All (batch, K-split, tile) work items are laid out in one sequence in which
  // neighbours share operand panels -- K-split major (workgroups of one split read the same rows
  // of both operands), then column strips of `strip` tiles walked down the rows (a strip of the
  // B operand stays hot while A panels stream past) -- and XCD x takes the x-th contiguous eighth
  // of that sequence, so what is resident on an XCD at any time is a compact block of the problem.
  const int total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int q8 = total / 8, r8 = total % 8, xcd = lin % 8;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lin / 8;
  const int tiles = p.tiles_m * p.tiles_n;
  const int z = work / tiles, tix = work % tiles;
  const int strip_items = p.tiles_m * p.strip, sidx = tix / strip_items, rem = tix % strip_items;
  const int strip_w = min(p.strip, p.tiles_n - sidx * p.strip);
  const int tm = rem / strip_w, tn = sidx * p.strip + rem % strip_w;
  const int m0 = tm * TM, n0 = tn * TN;

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

  // FULL runs on v_mfma_f32_16x16x4_f32 (one instruction contracts 4 k, so the accumulator file
  // is read and written half as often per FLOP as with 32x32x2 and the chip holds a higher clock
  // under load); the guarded path keeps 32x32x2, whose plain accumulator layout stores whole
  // 128-byte row segments at any alignment.
  constexpr int TW = FULL ? 16 : 32;            // MFMA tile edge
  constexpr int NTM = WTM / TW, NTN = WTN / TW; // MFMA tiles per wave and dimension
  constexpr int KSTEP = FULL ? 4 : 2;           // k per MFMA step
  constexpr int STEPS = BK / KSTEP;             // MFMA steps per slab
  constexpr int KROW = FULL ? 1 : 2;            // LDS k-rows between the operand reads of successive steps
  using AccT = std::conditional_t<FULL, f32x4, f32x16>;
  AccT acc[NTM][NTN];
#pragma unroll
  for (int i = 0; i < NTM; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < (FULL ? 4 : 16); ++r) acc[i][j][r] = 0.f;

  f32x4 ra0[OpA::NL], rb0[OpB::NL], ra1[OpA::NL], rb1[OpB::NL];  // two slabs in flight
  const bool va = p.vec_a, vb = p.vec_b;
  // FULL: each operand is read through a buffer descriptor based at the tile origin (scalar
  // registers), a per-lane 32-bit byte offset that is constant over the K loop and a scalar offset
  // that advances by one slab -- no vector address arithmetic and no 64-bit address registers.
  unsigned offa[OpA::NL], offb[OpB::NL];
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
  int so_a = 0, so_b = 0;
  const int step_a = (int)OpA::slab_bytes(p.lda), step_b = (int)OpB::slab_bytes(p.ldb);
  if constexpr (FULL) {
    // The descriptors end exactly where the operand does (counted from the tile origin), so rows
    // of a ragged last tile and -- where k is the slow index of both operands -- the k beyond a
    // ragged contraction read as zeros without any bounds check in the loop.
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A + OpA::origin(p.lda, m0, kbeg)), 0,
                                               (int)OpA::extent_bytes(p.lda, p.M - m0, p.K - kbeg), 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B + OpB::origin(p.ldb, n0, kbeg)), 0,
                                               (int)OpB::extent_bytes(p.ldb, p.N - n0, p.K - kbeg), 0x00020000);
    OpA::lane_offsets(p.lda, offa);
    OpB::lane_offsets(p.ldb, offb);
  }
// (a macro rather than a lambda: arrays captured by reference end up in scratch memory)
#define VITK_FETCH(T, RA, RB)                                                                               \
  if constexpr (FULL) {                                                                                     \
    _Pragma("unroll") for (int it = 0; it < OpA::NL; ++it) RA[it] = __builtin_bit_cast(                     \
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, offa[it], so_a, 0));                           \
    _Pragma("unroll") for (int it = 0; it < OpB::NL; ++it) RB[it] = __builtin_bit_cast(                     \
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, offb[it], so_b, 0));                           \
    so_a += step_a;                                                                                         \
    so_b += step_b;                                                                                         \
  } else {                                                                                                  \
    OpA::load(A, p.lda, m0, p.M, kbeg + (T)*BK, kend, va, RA);                                              \
    OpB::load(B, p.ldb, n0, p.N, kbeg + (T)*BK, kend, vb, RB);                                              \
  }

  // Three-slab LDS ring fed through two register sets.  While slab t is multiplied out of ring
  // slot t%3, slab t+1 already sits in slot (t+1)%3, slab t+2 moves registers -> slot (t+2)%3 and
  // slabs t+3, t+4 are in flight global -> registers (two slabs of MFMAs cover the load latency).
  // The single barrier of a slab sits in its MIDDLE: a wave that has passed it knows every wave
  // is done with slab t-1 (so its slot may be overwritten) and that slab t+1 is completely
  // written (so operand reads run across the slab boundary without a stall).
  // Operand lanes.  32x32x2: lane (li, lh) supplies row li at k = 2*step + lh.  16x16x4: lane
  // (l16, g) supplies row l16 at the g-th of the step's four k; which four is free as long as both
  // operands agree, and k = step + 4*(g&1) + 8*(g>>1) puts the two 16-lane groups that share an
  // LDS cycle 4 k-rows = 16 banks apart (both pitches are 4 mod 32), i.e. conflict-free.
  const int l16 = lane & 15, g16 = lane >> 4, kofs = 4 * (g16 & 1) + 8 * (g16 >> 1);
  // FULL: the A-side lane takes NTM CONSECUTIVE rows, one per MFMA tile (tile i of the wave = rows
  // NTM * l + i), so that its operands of a k-step are one 8- or 16-byte LDS read instead of NTM
  // 4-byte ones; which rows make up a tile is free, the epilogue follows the same map.
  const int lane_a = FULL ? wr * WTM + NTM * l16 + kofs * OpA::LD : wr * WTM + li + lh * OpA::LD;
  const int lane_b = FULL ? wc * WTN + l16 + kofs * OpB::LD : wc * WTN + li + lh * OpB::LD;
  int cur = 0;  // ring slot of slab t
  int t = 0;
  // Operand reads run one k-step ahead of the MFMAs that consume them (two register sets).
  float a[2][NTM], b[2][NTN];
  // the A operands of one k-step at `ptr` (FULL: one vector read; else one tile edge apart)
#define VITK_READ_A(DST, PTR)                                                                       \
  if constexpr (FULL && NTM == 4) {                                                                 \
    const f32x4 v_ = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned((PTR), 16));         \
    DST[0] = v_.x, DST[1] = v_.y, DST[2] = v_.z, DST[3] = v_.w;                                     \
  } else if constexpr (FULL && NTM == 2) {                                                          \
    const f32x2 v_ = *reinterpret_cast<const f32x2 *>(__builtin_assume_aligned((PTR), 8));          \
    DST[0] = v_.x, DST[1] = v_.y;                                                                   \
  } else {                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NTM; ++i) DST[i] = (PTR)[i * TW];                         \
  }
#define VITK_FIRST_OPERANDS()                                                             \
  lds_barrier();                                                                          \
  VITK_STAMP(1)                                                                           \
  VITK_READ_A(a[0], As + lane_a)                                                          \
  _Pragma("unroll") for (int j = 0; j < NTN; ++j) b[0][j] = Bs[lane_b + j * TW];

  // One slab (index T, prefetch registers RA/RB hold slab T+2 on entry, slab T+4 on exit;
  // STEADY: slabs T+2 and T+4 are known to exist).
  // On the FULL path the second operand goes into the MFMA as the row operand: the accumulator
  // tile is C^T, i.e. each lane ends up with runs of 4 consecutive columns of one row of C
  // (16-byte stores); the guarded path keeps plain accumulators (row-segment stores, any edge).
#define VITK_SLAB(T, RA, RB, STEADY, CUR, STG)                                                                 \
  {                                                                                                         \
    const int nxt = (CUR) == STAGES - 1 ? 0 : (CUR) + 1;                                                    \
    const int fre = nxt == STAGES - 1 ? 0 : nxt + 1;                                                        \
    const float *__restrict__ as = As + (CUR)*SA + lane_a;                                                  \
    const float *__restrict__ bs = Bs + (CUR)*SB + lane_b;                                                  \
    const float *__restrict__ as_n = As + nxt * SA + lane_a;                                                \
    const float *__restrict__ bs_n = Bs + nxt * SB + lane_b;                                                \
    _Pragma("unroll") for (int ks = 0; ks < STEPS; ++ks) {                                                  \
      const int c = ks & 1, n = c ^ 1;                                                                      \
      if (ks + 1 < STEPS) {                                                                                 \
        VITK_READ_A(a[n], as + KROW * (ks + 1) * OpA::LD)                                                   \
        _Pragma("unroll") for (int j = 0; j < NTN; ++j) b[n][j] = bs[KROW * (ks + 1) * OpB::LD + j * TW];   \
      } else { /* first k-step of the next slab (after the last slab: a harmless stale read) */             \
        VITK_READ_A(a[n], as_n)                                                                             \
        _Pragma("unroll") for (int j = 0; j < NTN; ++j) b[n][j] = bs_n[j * TW];                             \
      }                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < NTM; ++i) _Pragma("unroll") for (int j = 0; j < NTN; ++j) {     \
        if constexpr (FULL) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[c][j], a[c][i], acc[i][j], 0, 0, 0); \
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);        \
      }                                                                                                     \
      __builtin_amdgcn_sched_group_barrier(0x100, (FULL ? 1 : NTM) + NTN, 0); /* the next step's LDS reads ... */ \
      __builtin_amdgcn_sched_group_barrier(0x008, NTM * NTN, 0); /* ... then this step's MFMAs */           \
      if (ks == STEPS / 2 - 1) {                                                                               \
        lds_barrier();                                                                                      \
        if (STEADY || (T) + 2 < nslab) {                                                                    \
          STG(fre, RA, RB)                                                                                  \
        }                                                                                                   \
        if (STEADY || (T) + 4 < nslab) {                                                                    \
          VITK_FETCH((T) + 4, RA, RB)                                                                       \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0); /* keep the loads here: sunk below the MFMAs they lose cover */ \
      }                                                                                                     \
    }                                                                                                       \
  }
#define VITK_ADVANCE() cur = cur == STAGES - 1 ? 0 : cur + 1;

  // Column sums of B ride along: the workgroups of the first tile row add up what they stage.
  const bool do_colsum = CS && tm == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  unsigned lbase_a[OpA::NL], lbase_b[OpB::NL];
  OpA::lds_bases(As, lbase_a);
  OpB::lds_bases(Bs, lbase_b);
#define VITK_STAGE_SUMS(RB)                                                               \
  if constexpr (CS) {                                                                     \
    /* every workgroup adds (a branch on do_colsum costs more registers than the adds cost time) */ \
    _Pragma("unroll") for (int it = 0; it < OpB::NL; ++it) csum += RB[it];                \
  }
#define VITK_STAGE_LIT(SLOT, RA, RB)                                                      \
  OpA::template store_imm<(SLOT)*SA * 4>(lbase_a, RA);                                    \
  OpB::template store_imm<(SLOT)*SB * 4>(lbase_b, RB);                                    \
  VITK_STAGE_SUMS(RB)
#define VITK_STAGE(SLOT, RA, RB)                                                          \
  OpA::store(As + (SLOT)*SA, RA);                                                         \
  OpB::store(Bs + (SLOT)*SB, RB);                                                         \
  VITK_STAGE_SUMS(RB)

  // Long contractions: a straight-line prologue and a condition-free steady state in ONE branch,
  // so the compiler counts the loads in flight exactly and a slab waits only for the older of
  // the two register sets.  Everything else (short contractions, the last slabs) takes the
  // guarded path below.
  if (nslab >= 6) {
    VITK_FETCH(0, ra0, rb0)
    VITK_STAGE(0, ra0, rb0)
    VITK_FETCH(1, ra1, rb1)
    VITK_STAGE(1, ra1, rb1)
    VITK_FETCH(2, ra0, rb0)
    VITK_FETCH(3, ra1, rb1)
    VITK_FIRST_OPERANDS()
    // six slabs per trip = both register sets x all three ring slots: every LDS address in here is
    // a loop-invariant register plus a literal, so the trip carries no address arithmetic (vector
    // instructions do not overlap with fp32 MFMAs, they take their place)
    if constexpr (FULL)  // (the guarded kernels have no registers to spare for it)
    for (; t + 9 < nslab; t += 6) {
      VITK_SLAB(t, ra0, rb0, true, 0, VITK_STAGE_LIT)
      VITK_SLAB(t + 1, ra1, rb1, true, 1, VITK_STAGE_LIT)
      VITK_SLAB(t + 2, ra0, rb0, true, 2, VITK_STAGE_LIT)
      VITK_SLAB(t + 3, ra1, rb1, true, 0, VITK_STAGE_LIT)
      VITK_SLAB(t + 4, ra0, rb0, true, 1, VITK_STAGE_LIT)
      VITK_SLAB(t + 5, ra1, rb1, true, 2, VITK_STAGE_LIT)
    }
    for (; t + 5 < nslab; t += 2) {  // (ring slot 0 again on entry)
      VITK_SLAB(t, ra0, rb0, true, cur, VITK_STAGE)
      VITK_ADVANCE()
      VITK_SLAB(t + 1, ra1, rb1, true, cur, VITK_STAGE)
      VITK_ADVANCE()
    }
  } else {
    if (nslab > 0) {
      VITK_FETCH(0, ra0, rb0)
      VITK_STAGE(0, ra0, rb0)
    }
    if (nslab > 1) {
      VITK_FETCH(1, ra1, rb1)
      VITK_STAGE(1, ra1, rb1)
    }
    if (nslab > 2) {
      VITK_FETCH(2, ra0, rb0)
    }
    if (nslab > 3) {
      VITK_FETCH(3, ra1, rb1)
    }
    VITK_FIRST_OPERANDS()
  }
  for (; t < nslab; t += 2) {
    VITK_SLAB(t, ra0, rb0, false, cur, VITK_STAGE)
    VITK_ADVANCE()
    if (t + 1 < nslab) {
      VITK_SLAB(t + 1, ra1, rb1, false, cur, VITK_STAGE)
      VITK_ADVANCE()
    }
  }
#undef VITK_ADVANCE
#undef VITK_SLAB
#undef VITK_STAGE
#undef VITK_STAGE_LIT
#undef VITK_STAGE_SUMS
#undef VITK_FIRST_OPERANDS
#undef VITK_READ_A
#undef VITK_FETCH
  VITK_STAMP(2)

  if constexpr (CS) {
    if (do_colsum) {  // workgroup-uniform
      // Thread t summed rows t / (TN/4) (+ multiples) of columns 4*(t % (TN/4)) .. +3: add the
      // THREADS / (TN/4) row groups in a fixed order through LDS (the ring is free by now).
      constexpr int QUADS = TN / 4, GROUPS = THREADS / QUADS;
      static_assert(GROUPS * TN <= STAGES * SA, "the column-sum staging must fit the A ring");
      __syncthreads();
      *reinterpret_cast<f32x4 *>(As + (threadIdx.x / QUADS) * TN + 4 * (threadIdx.x % QUADS)) = csum;
      __syncthreads();
      if (threadIdx.x < TN && n0 + (int)threadIdx.x < p.N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) t += As[g * TN + threadIdx.x];
        if (p.split_k > 1) p.colsum_ws[(long long)z * p.N + n0 + threadIdx.x] = t;
        else p.colsum[n0 + threadIdx.x] = t;
      }
    }
  }

  const int row_w = m0 + wr * WTM, col_w = n0 + wc * WTN;  // this wave's sub-tile origin
  // Epilogue.  Addresses are a uniform base plus one per-lane byte offset per pitch, so the
  // stores (and the residual / pre-activation loads) cost no vector registers for addressing.
  if constexpr (FULL) {
    // Transposed accumulators: lane l of tile (i, j) holds C[NTM * (l&15) + i][16 j + 4*(l>>4) + 0..3]
    // (rows interleaved over the tiles, see lane_a) -- one 16-byte store per lane and tile, 64
    // contiguous bytes per row and instruction.
    if (p.split_k > 1) {
      char *W = reinterpret_cast<char *>(p.ws + (long long)z * p.M * p.N + (long long)row_w * p.N + col_w);
      const unsigned lane_w = (unsigned)((NTM * l16 * (long long)p.N + 4 * g16) * 4);
#pragma unroll
      for (int i = 0; i < NTM; ++i) {
        if (row_w + NTM * l16 + i >= p.M) continue;  // ragged last tile row
#pragma unroll
        for (int j = 0; j < NTN; ++j)
          *reinterpret_cast<f32x4 *>(W + ((long long)i * p.N + j * 16) * 4 + lane_w) = acc[i][j];
      }
    } else {
      char *C = reinterpret_cast<char *>(p.C + bo * p.sc_o + bi * p.sc_i + (long long)row_w * p.ldc + col_w);
      const char *R =
          reinterpret_cast<const char *>(p.residual ? p.residual + (long long)row_w * p.ldr + col_w : nullptr);
      char *X = reinterpret_cast<char *>(p.aux ? p.aux + (long long)row_w * p.ldaux + col_w : nullptr);
      const unsigned lane_c = (unsigned)((NTM * l16 * p.ldc + 4 * g16) * 4);
      const unsigned lane_r = (unsigned)((NTM * l16 * p.ldr + 4 * g16) * 4);
      const unsigned lane_x = (unsigned)((NTM * l16 * p.ldaux + 4 * g16) * 4);
#pragma unroll
      for (int i = 0; i < NTM; ++i) {
        const bool row_ok = row_w + NTM * l16 + i < p.M;  // false only in a ragged last tile row
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
          if (!row_ok) break;
          const int col = col_w + j * 16 + 4 * g16;
          f32x4 *cp = reinterpret_cast<f32x4 *>(C + ((long long)i * p.ldc + j * 16) * 4 + lane_c);
          f32x4 *xp = reinterpret_cast<f32x4 *>(X + ((long long)i * p.ldaux + j * 16) * 4 + lane_x);
          const f32x4 *rp = reinterpret_cast<const f32x4 *>(R + ((long long)i * p.ldr + j * 16) * 4 + lane_r);
          f32x4 bv = {0.f, 0.f, 0.f, 0.f}, xv = bv, rv = bv, v, pre;
          if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4 *>(p.bias + col);
          if (p.act == VITK_ACT_GELU_BWD) xv = *xp;
          if (p.residual != nullptr) rv = *rp;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = p.alpha * acc[i][j][e];
            if (p.act == VITK_ACT_GELU_BWD) y *= vitk::gelu_grad(xv[e]);
            y += bv[e];
            pre[e] = y;
            if (p.act == VITK_ACT_GELU_FWD) y = vitk::gelu_fwd(y);
            if (p.residual != nullptr) y += rv[e];
            v[e] = y;
          }
          if (p.act == VITK_ACT_GELU_FWD && p.aux != nullptr) *xp = pre;
          *cp = v;
        }
        // keep the loads of one 16-row band from being hoisted above the stores of the previous one
        asm volatile("" ::: "memory");
      }
    }
  } else {
    // Plain accumulators: register r of lane l holds C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] of the
    // 32x32 tile -- per register, lanes 0-31 and 32-63 each write one 128-byte row segment (any
    // alignment, any edge).
    if (p.split_k > 1) {
      char *W = reinterpret_cast<char *>(p.ws + (long long)z * p.M * p.N + (long long)row_w * p.N + col_w);
      const unsigned lane_w = (unsigned)((4 * lh * (long long)p.N + li) * 4);
#pragma unroll
      for (int i = 0; i < NTM; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
          const int col = col_w + j * 32 + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
            const int row = row_w + dr + 4 * lh;
            if (row < p.M && col < p.N)
              *reinterpret_cast<float *>(W + ((long long)dr * p.N + j * 32) * 4 + lane_w) = acc[i][j][r];
          }
        }
    } else {
      char *C = reinterpret_cast<char *>(p.C + bo * p.sc_o + bi * p.sc_i + (long long)row_w * p.ldc + col_w);
      const char *R =
          reinterpret_cast<const char *>(p.residual ? p.residual + (long long)row_w * p.ldr + col_w : nullptr);
      char *X = reinterpret_cast<char *>(p.aux ? p.aux + (long long)row_w * p.ldaux + col_w : nullptr);
      const unsigned lane_c = (unsigned)((4 * lh * p.ldc + li) * 4);
      const unsigned lane_r = (unsigned)((4 * lh * p.ldr + li) * 4);
      const unsigned lane_x = (unsigned)((4 * lh * p.ldaux + li) * 4);
#pragma unroll
      for (int i = 0; i < NTM; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
          const int col = col_w + j * 32 + li;
          const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
            const int row = row_w + dr + 4 * lh;
            if (row < p.M && col < p.N) {
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
          asm volatile("" ::: "memory");
        }
    }
  }
  VITK_STAMP(3)
  VITK_STAMP_END()
}

// C[b][m][n] = sum_s ws[b*S + s][m][n], s ascending (fixed order => reproducible).
__global__ void splitk_reduce_kernel(const float *__restrict__ ws, int S, int M, int N, float *__restrict__ C,
                                     long long ldc, int batch_inner, long long sc_o, long long sc_i, long long total,
                                     const float *__restrict__ colsum_ws, float *__restrict__ colsum) {
  const long long mn = (long long)M * N;
  if (colsum != nullptr)  // column sums of B: partials [S][N], same fixed order
    for (long long n = blockIdx.x * (long long)blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
      float t = 0.f;
      for (int k = 0; k < S; ++k) t += colsum_ws[(long long)k * N + n];
      colsum[n] = t;
    }
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

// The same sum for one unbatched, unpitched, 16-byte-aligned result (every weight gradient of the
// model): a float4 per thread and no index arithmetic -- the kernel is pure streaming.
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float *__restrict__ ws, int S, long long mn4,
                                                                float *__restrict__ C, int matrix_blocks, int N,
                                                                const float *__restrict__ colsum_ws,
                                                                float *__restrict__ colsum) {
  if ((int)blockIdx.x >= matrix_blocks) {  // trailing workgroups: the column sums of B, partials [S][N]
    const int n = ((int)blockIdx.x - matrix_blocks) * 256 + threadIdx.x;
    if (n < N) {
      float t = 0.f;
      for (int k = 0; k < S; ++k) t += colsum_ws[(long long)k * N + n];
      colsum[n] = t;
    }
    return;
  }
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= mn4) return;
  const f32x4 *w = reinterpret_cast<const f32x4 *>(ws) + i;
  f32x4 s = w[0];
  int k = 1;
  for (; k + 3 < S; k += 4) {  // four loads in flight; the additions stay in split order
    const f32x4 a = w[(long long)k * mn4], b = w[(long long)(k + 1) * mn4], c = w[(long long)(k + 2) * mn4],
                d = w[(long long)(k + 3) * mn4];
    s = (((s + a) + b) + c) + d;
  }
  for (; k < S; ++k) s += w[(long long)k * mn4];
  reinterpret_cast<f32x4 *>(C)[i] = s;
}

#ifdef VITK_GEMM_STAMPS
unsigned long long *g_stamps = nullptr;
#endif

struct Plan {
  int tile;  // square tile edge the heuristics below reason with: 128 or 64
  int tile_m, tile_n;  // the launched tile (tile x tile, or 128 x 64 / 64 x 128)
  int tiles_m, tiles_n, batch, split_k, k_chunk;
};

int choose_split(long long tiles, int K) {
  // Weight-gradient GEMMs contract over all B*N token rows into a small [in][out] matrix: too few
  // output tiles to occupy the chip's CUs (256 on MI355X).  Split the contraction so that
  // tiles*splits fills whole rounds of resident workgroups (CUs x kBlocksPerCU): the workgroups of
  // a round run concurrently and finish together, and a partly empty round costs as much as a
  // full one.  Which number of rounds: the fill of the last round against the reduction launch's
  // reading of one more partial result per split -- relative to the contraction, 2 S F / (K BW)
  // with F / BW ~ 135 TFLOP/s / 5 TB/s, i.e. 54 S / K.  (ViT-B's 768 x 3072 gradients: 144 tiles;
  // 5 splits fill 94 % of one round, 16 splits fill three rounds exactly and run 5.5 % faster;
  // 108- and 36-tile outputs stay at one round of 7 and 21 splits -- measured, tools/bench_gemm_split.py.)
  const long long slots = (long long)vitk::cu_count() * kBlocksPerCU;
  if (tiles >= slots / 2) return 1;
  long long max_s = K / (8 * BK);  // (at least 8 slabs per workgroup)
  if (max_s > 96) max_s = 96;      // (a 384 x 384 gradient, 9 tiles: 85 splits run 16 % faster than 64)
  if (max_s < 1) max_s = 1;
  int best_s = 1;
  double best = -1.0;
  for (long long s = 1; s <= max_s; ++s) {
    const long long wgs = tiles * s, rounds = (wgs + slots - 1) / slots;
    const double fill = (double)wgs / (double)(rounds * slots);
    const double score = fill / (1.0 + 54.0 * (double)s / (double)K);
    if (score > best) best = score, best_s = (int)s;
  }
  // A contraction too short for that (fewer than 32 slabs per workgroup at the chosen split: the
  // 8704-row gradients of the 17-token model) is all fill, drain and reduction; one workgroup per
  // CU then beats three, with a third of the partial results to write and re-read (192 x 192:
  // 28 vs 19 us at 68 / 28 splits; 192 x 768: 53 vs 41 us at 64 / 21).
  const long long per_cu = vitk::cu_count() / tiles;
  if (K / (best_s * BK) < 32 && per_cu >= 1 && per_cu < best_s) best_s = (int)per_cu;
  return best_s;
}

// Rectangular tiles (unguarded kernels only: whole tiles along N, 16-byte-aligned operands -- the
// caller falls back to the square tile when the launch turns out not to qualify).
//   reserved == 2 / 3: 128 x 64 / 64 x 128 forced (measurement, tests).
void choose_rectangular(const vitk_gemm_args *a, Plan *pl) {
  int tm = pl->tile_m, tn = pl->tile_n;
  if (a->reserved == 2) tm = 128, tn = 64;
  else if (a->reserved == 3) tm = 64, tn = 128;
  else if (a->reserved != 0) return;
  // Short contractions into wide outputs (ViT-S: K = 384 into N = 1152 / 1536): 64 x 128 tiles run
  // 3-5 % ahead of both square tiles (twice the workgroups of the 128 x 128 tile to overlap the
  // fill and store phases that a 24-slab contraction leaves exposed, three quarters of the 64 x 64
  // tile's operand reads); no gain from K = 768 on, and none at K = 192 either shape.
  else if (!a->trans_a && a->K > 256 && a->K <= 512 && a->N % 128 == 0 && a->N >= 1024 && a->M >= 4096 && pl->split_k == 1 &&
           pl->batch == 1)
    tm = 64, tn = 128;  // (inside the ViT-S step, with its epilogues: +0.5-0.8 % on the step)
  if (tm == pl->tile_m && tn == pl->tile_n) return;
  if (a->N % tn != 0) return;
  pl->tile_m = tm;
  pl->tile_n = tn;
  pl->tiles_m = (a->M + tm - 1) / tm;
  pl->tiles_n = a->N / tn;
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
  // N a multiple of 64 but not of 128 (D = 192, 3D = 576): 64-wide tiles keep the unguarded path
  const bool n_wants_64 = a->N % 128 != 0 && a->N % 64 == 0;
  // the same for a short M (weight gradients of D = 192: 1.5 tiles of 128 rows waste a quarter of
  // the launch; 192 x 768 over 50 432 rows: 92 vs 120 TFLOP/s)
  const bool m_wants_64 = a->M % 128 != 0 && a->M % 64 == 0 && a->M < 1024;
  pl->tile = (a->reserved != 1 && !n_wants_64 && !m_wants_64 && a->M >= 128 && a->N >= 128 && (t128 >= 64 || can_split)) ? 128 : 64;
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
  // Few 128-wide tiles: workgroups are handed out per CU, so the launch takes as long as the CU
  // with the most tiles -- 300 tiles on 256 CUs cost two tile times for 1.17 tiles' worth of work
  // each.  64-wide tiles quantise four times finer; their kernel peaks ~6 % lower (twice the LDS
  // reads per FLOP).  Take whichever the fill predicts to be faster (measured on the ViT-B shapes
  // at batch 8..64: within a few percent of the better choice everywhere, up to +50 %).
  if (pl->tile == 128 && s == 1 && a->reserved != 1) {
    const long long cus = vitk::cu_count(), slots = cus * kBlocksPerCU;
    const long long t128 = (long long)pl->tiles_m * pl->tiles_n * pl->batch;
    const long long t64 = (long long)((a->M + 63) / 64) * ((a->N + 63) / 64) * pl->batch;
    auto fill = [cus](long long n) { return (double)n / ((double)cus * (double)((n + cus - 1) / cus)); };
    // a small remainder after whole rounds is cut off and run on 64-wide tiles anyway (tail round)
    double e128 = (t128 >= slots && t128 % slots <= 128) ? 0.98 : fill(t128);
    if (t128 < 2 * cus) e128 *= 0.9;  // fewer than two workgroups per CU: latencies show
    // between one and two rounds of resident workgroups the second round is what counts
    // (M = 50 432, N = 384: 1182 tiles = 1.54 rounds, 110 vs 117 TFLOP/s at K = 384, 125 vs 129 at K = 768)
    if (t128 > slots && t128 < 2 * slots && !(t128 % slots <= 128)) {
      const double two_rounds = (double)t128 / (double)(2 * slots);
      if (two_rounds < e128) e128 = two_rounds;
    }
    // a short contraction: the 128-wide tile's longer fill and store phases weigh more than its
    // better operand reuse (M = 50 432, N = 768, K = 192: 103 vs 108 TFLOP/s; no difference from K = 384 on)
    if (a->K <= 256) e128 *= 0.93;
    const double e64 = fill(t64) * 0.94;
    if (e64 > e128) {
      pl->tile = 64;
      pl->tiles_m = (a->M + 63) / 64;
      pl->tiles_n = (a->N + 63) / 64;
    }
  }
  pl->tile_m = pl->tile_n = pl->tile;
  choose_rectangular(a, pl);
  return 0;
}

template <int TM, int TN, bool FULL>
void launch2(const Params &p, bool ta, bool tb, dim3 grid, hipStream_t st) {
  const bool cs = p.colsum != nullptr;
  if (!ta && !tb && !cs) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false, FULL, false>), grid, dim3(THREADS), 0, st, p);
  else if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false, FULL, true>), grid, dim3(THREADS), 0, st, p);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true, FULL, false>), grid, dim3(THREADS), 0, st, p);
  else if (ta && !tb && !cs) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false, FULL, false>), grid, dim3(THREADS), 0, st, p);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false, FULL, true>), grid, dim3(THREADS), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true, FULL, false>), grid, dim3(THREADS), 0, st, p);
}

template <int TILE>
void launch(const Params &p, bool ta, bool tb, bool full, dim3 grid, hipStream_t st) {
  if (full) launch2<TILE, TILE, true>(p, ta, tb, grid, st);
  else launch2<TILE, TILE, false>(p, ta, tb, grid, st);
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

constexpr int kVariants = 53;  // 32 square-tile kernels, 4 small-problem kernels, 2 x 8 rectangular ones, the paired small launch
int variant_of(int tile_m, int tile_n, bool ta, bool tb, bool full, bool cs) {
  if (tile_m != tile_n) return 36 + (tile_m == 128 ? 0 : 8) + (cs ? 4 : 0) + (ta ? 2 : 0) + (tb ? 1 : 0);
  return (cs ? 16 : 0) + (full ? 8 : 0) + (tile_m == 128 ? 0 : 4) + (ta ? 2 : 0) + (tb ? 1 : 0);
}

}  // namespace

extern "C" size_t vitk_gemm_workspace_bytes(const vitk_gemm_args *a) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->K <= 0) return 0;
  Plan pl;
  make_plan(a, &pl);
  if (pl.split_k <= 1) return 0;
  return ((size_t)pl.split_k * pl.batch * a->M * a->N + (a->colsum_b ? (size_t)pl.split_k * a->N : 0)) * sizeof(float);
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

  if (vitk::gemm_small_applies(a)) {  // too small to fill the chip: one wave per 16 x 16 tile (gemm_small.hip)
    VITK_REQUIRE(a->colsum_b == nullptr || a->trans_b == 0, "vitk_gemm_f32: colsum_b needs trans_b == 0 and batch == 1");
    hipStream_t st = vitk::as_stream(stream);
    ProfRec rec{};
    if (g_prof_on) {
      rec.variant = 32 + (a->trans_a ? 2 : 0) + (a->trans_b ? 1 : 0);
      rec.flops = 2.0 * a->M * a->N * a->K;
      rec.e0 = prof_event();
      rec.e1 = prof_event();
      (void)hipEventRecord(rec.e0, st);
    }
    const int rc = vitk::gemm_small_launch(a, st);
    if (g_prof_on) {
      (void)hipEventRecord(rec.e1, st);
      g_prof.push_back(rec);
    }
    return rc;
  }

  Plan pl;
  make_plan(a, &pl);
  const bool plain = a->alpha == 1.0f && a->bias == nullptr && a->residual == nullptr && a->act == VITK_ACT_NONE;
  VITK_REQUIRE(pl.split_k == 1 || plain, "vitk_gemm_f32: split-K needs alpha == 1 and no bias/act/residual");

  // Tail round.  The chip holds CUs x kBlocksPerCU workgroups; when the tile count leaves a small
  // remainder after the last full round (N = 768: 394 x 6 tiles = 3 rounds + 60), those few 128-wide
  // tiles would each run alone on a CU while most CUs idle.  Cut the trailing tile rows off and run
  // them as a second launch, which make_plan covers with four times as many 64-wide tiles.
  if (a->reserved == 0 && pl.tile_m == 128 && pl.tile_n == 128 && pl.split_k == 1 && pl.batch == 1 &&
      a->colsum_b == nullptr) {
    const int slots = vitk::cu_count() * kBlocksPerCU;
    const int tail_max = vitk::cu_count() / 2;  // larger remainders already spread over most CUs (measured: no gain)
    const long long tiles = (long long)pl.tiles_m * pl.tiles_n;
    const int head_rows = (int)(tiles / slots * slots / pl.tiles_n);  // tile rows inside the full rounds
    const int tail_rows = pl.tiles_m - head_rows;
    if (head_rows > 0 && tail_rows > 0 && tail_rows * pl.tiles_n <= tail_max) {
      const int m_head = head_rows * 128;
      vitk_gemm_args head = *a, tail = *a;
      head.M = m_head;
      tail.M = a->M - m_head;
      tail.A = a->A + (a->trans_a ? (int64_t)m_head : (int64_t)m_head * a->lda);
      tail.C = a->C + (int64_t)m_head * a->ldc;
      if (a->residual) tail.residual = a->residual + (int64_t)m_head * a->ldr;
      if (a->aux) tail.aux = a->aux + (int64_t)m_head * a->ldaux;
      tail.reserved = 1;  // 64-wide tiles, no further splitting
      if (vitk_gemm_f32(&head, stream)) return 1;
      return vitk_gemm_f32(&tail, stream);
    }
  }

  const size_t need = vitk_gemm_workspace_bytes(a);
  VITK_REQUIRE(a->colsum_b == nullptr || (a->trans_b == 0 && pl.batch == 1),
               "vitk_gemm_f32: colsum_b needs trans_b == 0 and batch == 1");
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
  p.colsum = a->colsum_b;
  p.colsum_ws = pl.split_k > 1 && a->colsum_b ? a->workspace + (size_t)pl.split_k * pl.batch * a->M * a->N : nullptr;
  {  // column strips of at most 8 tiles, as even as possible
    constexpr int kMaxStrip = 8;
    const int nstrips = (pl.tiles_n + kMaxStrip - 1) / kMaxStrip;
    p.strip = (pl.tiles_n + nstrips - 1) / nstrips;
  }
  p.vec_a = strides_vec_ok(a->A, a->lda, a->stride_a_outer, a->stride_a_inner);
  p.vec_b = strides_vec_ok(a->B, a->ldb, a->stride_b_outer, a->stride_b_inner);
#ifdef VITK_GEMM_STAMPS
  p.stamps = g_stamps;
#endif

  hipStream_t st = vitk::as_stream(stream);
  // every K chunk starts and ends on a slab boundary when K and the chunk are multiples of BK
  // ragged M is masked in the epilogue; a ragged K only where k is the slow index of both operands
  const bool full = p.vec_a && p.vec_b && a->N % pl.tile_n == 0 && a->K > 0 &&
                    (a->K % BK == 0 || (a->trans_a && !a->trans_b)) && pl.k_chunk % BK == 0 && a->lda < (1 << 21) && a->ldb < (1 << 21) &&
                    (!a->trans_a || (long long)(a->K + BK) * a->lda * 4 < (1ll << 31)) &&
                    (a->trans_b || (long long)(a->K + BK) * a->ldb * 4 < (1ll << 31)) &&
                    a->ldc < (1 << 21) && a->ldr < (1 << 21) && a->ldaux < (1 << 21) &&
                    strides_vec_ok(a->C, a->ldc, a->stride_c_outer, a->stride_c_inner) &&
                    (a->residual == nullptr || strides_vec_ok(a->residual, a->ldr, 0, 0)) &&
                    (a->aux == nullptr || strides_vec_ok(a->aux, a->ldaux, 0, 0)) &&
                    (a->bias == nullptr || vitk::aligned16(a->bias)) &&
                    (pl.split_k == 1 || vitk::aligned16(a->workspace));
  if (!full && pl.tile_m != pl.tile_n) {  // the rectangular tiles exist unguarded only
    pl.tile_m = pl.tile_n = pl.tile;
    pl.tiles_m = (a->M + pl.tile - 1) / pl.tile;
    pl.tiles_n = (a->N + pl.tile - 1) / pl.tile;
    p.tiles_m = pl.tiles_m;
    p.tiles_n = pl.tiles_n;
    constexpr int kMaxStrip = 8;
    const int nstrips = (pl.tiles_n + kMaxStrip - 1) / kMaxStrip;
    p.strip = (pl.tiles_n + nstrips - 1) / nstrips;
  }
  dim3 grid((unsigned)(pl.tiles_m * pl.tiles_n), (unsigned)(pl.batch * pl.split_k), 1);
  ProfRec rec{};
  if (g_prof_on) {
    rec.variant = variant_of(pl.tile_m, pl.tile_n, a->trans_a != 0, a->trans_b != 0, full, a->colsum_b != nullptr);
    rec.flops = 2.0 * a->M * a->N * a->K * pl.batch;
    rec.e0 = prof_event();
    rec.e1 = prof_event();
    (void)hipEventRecord(rec.e0, st);
  }
  if (pl.tile_m == 128 && pl.tile_n == 64) launch2<128, 64, true>(p, a->trans_a != 0, a->trans_b != 0, grid, st);
  else if (pl.tile_m == 64 && pl.tile_n == 128) launch2<64, 128, true>(p, a->trans_a != 0, a->trans_b != 0, grid, st);
  else if (pl.tile == 128) launch<128>(p, a->trans_a != 0, a->trans_b != 0, full, grid, st);
  else launch<64>(p, a->trans_a != 0, a->trans_b != 0, full, grid, st);
  if (g_prof_on) {
    (void)hipEventRecord(rec.e1, st);
    g_prof.push_back(rec);
  }
  if (vitk::check_launch("vitk_gemm_f32")) return 1;
  if (pl.split_k > 1) {
    const long long total = (long long)pl.batch * a->M * a->N;
    if (pl.batch == 1 && a->ldc == a->N && total % 4 == 0 && vitk::aligned16(a->C) && vitk::aligned16(a->workspace)) {
      const int mb = (int)((total / 4 + 255) / 256), cb = p.colsum != nullptr ? (a->N + 255) / 256 : 0;
      hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)(mb + cb)), dim3(256), 0, st, a->workspace, pl.split_k,
                         total / 4, a->C, mb, a->N, p.colsum_ws, p.colsum);
    } else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(vitk::grid_for(total, 256)), dim3(256), 0, st, a->workspace,
                       pl.split_k, a->M, a->N, a->C, a->ldc, p.batch_inner, p.sc_o, p.sc_i, total, p.colsum_ws, p.colsum);
    if (vitk::check_launch("vitk_gemm_f32/splitk_reduce")) return 1;
  }
  return 0;
}

extern "C" int vitk_gemm_pair_f32(const vitk_gemm_args *x, const vitk_gemm_args *w, vitk_stream_t stream) {
  VITK_REQUIRE(x != nullptr && w != nullptr, "vitk_gemm_pair_f32: null args");
  const bool shapes_ok = x->M > 0 && x->N > 0 && x->K > 0 && w->M > 0 && w->N > 0 && w->K > 0 && x->A && x->B && x->C &&
                         w->A && w->B && w->C && x->lda >= x->K && x->ldb >= x->K && x->ldc >= x->N && w->lda >= w->M &&
                         w->ldb >= w->N && w->ldc >= w->N && x->act >= VITK_ACT_NONE && x->act <= VITK_ACT_GELU_BWD &&
                         (x->act != VITK_ACT_GELU_BWD || x->aux != nullptr) && (x->aux == nullptr || x->ldaux >= x->N) &&
                         (x->residual == nullptr || x->ldr >= x->N) && w->act == VITK_ACT_NONE && w->alpha == 1.0f &&
                         w->bias == nullptr && w->residual == nullptr;
  if (shapes_ok && vitk::gemm_small_pair_applies(x, w)) {
    hipStream_t st = vitk::as_stream(stream);
    ProfRec rec{};
    if (g_prof_on) {
      rec.variant = 52;
      rec.flops = 2.0 * x->M * x->N * x->K + 2.0 * w->M * w->N * w->K;
      rec.e0 = prof_event();
      rec.e1 = prof_event();
      (void)hipEventRecord(rec.e0, st);
    }
    const int rc = vitk::gemm_small_pair_launch(x, w, st);
    if (g_prof_on) {
      (void)hipEventRecord(rec.e1, st);
      g_prof.push_back(rec);
    }
    return rc;
  }
  if (vitk_gemm_f32(w, stream)) return 1;  // (every argument check lives there)
  return vitk_gemm_f32(x, stream);
}

#ifdef VITK_GEMM_STAMPS
// Diagnostic build only: device buffer of 16 x uint64 per workgroup of the next launches (or null).
extern "C" void vitk_gemm_set_stamps(unsigned long long *device_buffer) { g_stamps = device_buffer; }
#endif

extern "C" int vitk_gemm_profile_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

extern "C" int vitk_gemm_profile_read(int variant, double *total_ms, double *total_flops, int64_t *launches,
                                      int reset) {
  VITK_REQUIRE(variant >= 0 && variant < kVariants, "vitk_gemm_profile_read: variant %d outside [0,%d)", variant, kVariants);
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
  // the names a profiler prints: gemm_kernel<TM, TN, TA, TB, FULL, CS>
  static char names[kVariants][56];
  if (variant < 0 || variant >= kVariants) return "";
  if (variant == 52) return "gemm_small_pair_kernel (dX NT + dW TN)";
  if (variant >= 36) {
    const int v = variant - 36;
    std::snprintf(names[variant], sizeof names[variant], "gemm_kernel<%s, %s, %s, true, %s>", v & 8 ? "64, 128" : "128, 64",
                  v & 2 ? "true" : "false", v & 1 ? "true" : "false", v & 4 ? "true" : "false");
    return names[variant];
  }
  if (variant >= 32) {  // gemm_small_kernel<AV, BV>: the template flags depend on alignment; name the transposes
    std::snprintf(names[variant], sizeof names[variant], "gemm_small_kernel (%s)",
                  variant == 32 ? "NN" : variant == 33 ? "NT" : variant == 34 ? "TN" : "TT");
    return names[variant];
  }
  const bool cs = variant & 16, full = variant & 8, t64 = variant & 4, ta = variant & 2, tb = variant & 1;
  std::snprintf(names[variant], sizeof names[variant], "gemm_kernel<%d, %d, %s, %s, %s, %s>", t64 ? 64 : 128, t64 ? 64 : 128,
                ta ? "true" : "false", tb ? "true" : "false", full ? "true" : "false", cs ? "true" : "false");
  return names[variant];
}
