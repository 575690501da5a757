// Fused scaled-dot-product attention backward for head dimensions 33..64 on
// v_mfma_f32_16x16x4_f32 with 16-row granularity (reference: src/layers/MultiHeadAttention.cpp:
// 160-308): dQ, dK, dV from Q, K, V, O, dO and the row statistics of the forward, one launch,
// 10 N^2 dh FLOP per head, nothing of size N^2 in memory.
//
// Work split.  The unit of work is one block of 16 keys of one head: a wave keeps that block's K
// and V fragments and its dK^T / dV^T accumulators in registers and sweeps over the head's query
// tiles (16 queries each), which stream through a two-stage LDS ring.  197 tokens are 13 blocks;
// a workgroup of 4 waves owns several heads and deals their blocks to its waves round by round
// (4 consecutive blocks per round, which touch at most two heads), so every wave has work in
// every round but the last.  Per (query tile, key block) a wave forms S and dP ONCE, in the plain
// orientation [query][key], and uses them three ways:
//   dV^T[d][key] += dO^T P      and     dK^T[d][key] += Q^T dS      straight from the accumulator
//   registers (register r of lane (c, g) is row 4g + r = the contraction index the next product
//   wants from lane group g), and, after a 1 KB transpose of dS through a wave-private LDS
//   buffer, its share of dQ^T[d][query] = K^T dS^T.
// The (at most four) partial dQ tiles of a step are summed in wave order through LDS by all 256
// threads and written to dQ; a head's blocks are spread over several rounds, so from the head's
// second round on the sum is added to what the same thread stored before (plain load + store by
// one thread: fixed order, no atomics, bitwise reproducible).
//
// The matrix cores and the vector ALU of a SIMD do not overlap on fp32 work, so the loop carries
// no per-lane address arithmetic: global reads go through buffer descriptors (scalar base per
// head, constant per-lane offset, scalar tile offset; rows past the sequence read as zeros), the
// position in the list of blocks lives in scalar registers, and the row statistics {max * log2 e,
// 1 / sum, sum_d dO * O} of the (at most two) live heads are prepared once per head in LDS.
//
// LDS tile layout: [16 rows][68 floats].  Operand reads by row (A[i = c][k]: row c, columns
// 16g .. 16g + 15) are four 16-byte reads per lane, and a pitch of 68 floats puts the eight lanes
// one cycle serves on all 32 banks; reads by column (A[i = c][k]: row 4g + r, column 16 dt + c)
// want the two lane groups of a half-wave 16 banks apart, which 4 * 68 = 16 (mod 32) gives.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 68;                    // tile row pitch (a multiple of 4: rows are 16-byte aligned)
constexpr int TILE = 16 * TP;             // one 16 x 64 tile
constexpr int SLOT = 2 * TILE;            // Q tile + dO tile of one head
constexpr int STAGE = 2 * SLOT;           // two heads may be live in one round
constexpr int PBP = 68, PB = 16 * PBP;    // one wave's partial dQ tile [query][d]
constexpr int TBP = 20, TB = 16 * TBP;    // one wave's dS transpose buffer [query][key]
constexpr int MAXN = 256;
constexpr int F_RING = 0, F_PB = F_RING + 2 * STAGE, F_TB = F_PB + 2 * 4 * PB, F_ST = F_TB + 4 * TB;  // (partial tiles: two sets)
constexpr int F_TOTAL = F_ST + 2 * 3 * MAXN;  // statistics: [head parity][max2 | 1/sum | dsum][query]

struct Bwd16Params {
  const float *q, *k, *v;   // element (b, n, h, d) at ptr[(b*N + n)*ld + h*dh + d]
  int ld;                   // (pitches and whole-tensor element counts fit 31 bits: checked by the launcher)
  const float *o, *dout;    // forward output and its gradient (ldo)
  int ldo;
  const float *stats;       // [B*H][N][2]: row max of the scaled scores, row sum of exp
  float *dq, *dk, *dv;      // gradients (ldg)
  int ldg;
  int B, H, N, dh;
  float scale;
  int dk_mode;
  int hpw;                  // heads per workgroup
};

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ f32x4 buf_quad(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, soff, 0));
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store_quad(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)voff, soff, 0);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, soff, 0));
}

__global__ __launch_bounds__(256, 2) void sdpa_bwd16_kernel(const Bwd16Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int BH = p.B * p.H;
  const int head0 = blockIdx.x * p.hpw, nh = min(p.hpw, BH - head0);
  const int nblk = (p.N + 15) / 16, items = nh * nblk, rounds = (items + 3) / 4;
  const int nqt = nblk;  // query tiles of 16
  constexpr unsigned kOutside = 0x7fffff00u;  // beyond any descriptor: reads as zero
  const float sl2 = p.scale * 1.44269504088896340736f;

  // one descriptor per (tensor, head): it ends with the head's last row, so rows >= N read as zeros
  const int bytes_qkv = (int)(((long long)(p.N - 1) * p.ld + p.dh) * 4);
  const int bytes_o = (int)(((long long)(p.N - 1) * p.ldo + p.dh) * 4);
  const int bytes_g = (int)(((long long)(p.N - 1) * p.ldg + p.dh) * 4);
  auto rsrc = [&](const float *ptr, int off, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ptr + off), 0, bytes, 0x00020000);
  };
  // head index inside the workgroup -> element offsets of the head in the three pitch families
  auto offs = [&](int h, int *o_qkv, int *o_o, int *o_g) {
    const int hd = head0 + h, b = hd / p.H, hm = hd % p.H;
    *o_qkv = b * p.N * p.ld + hm * p.dh;
    *o_o = b * p.N * p.ldo + hm * p.dh;
    *o_g = b * p.N * p.ldg + hm * p.dh;
  };

  // ---- per-thread constants ----
  // staging / reduction role: thread t <-> row t/16, columns 4*(t%16) .. +3 of a 16 x 64 tile
  const int sr = threadIdx.x >> 4, sc = (threadIdx.x & 15) * 4;
  const bool sc_ok = sc < p.dh;
  const unsigned sv_q = sc_ok ? (unsigned)((sr * p.ld + sc) * 4) : kOutside;
  const unsigned sv_o = sc_ok ? (unsigned)((sr * p.ldo + sc) * 4) : kOutside;
  const unsigned sv_g = sc_ok ? (unsigned)((sr * p.ldg + sc) * 4) : kOutside;
  const int tile_q = (int)(16 * p.ld * 4), tile_o = (int)(16 * p.ldo * 4), tile_g = (int)(16 * p.ldg * 4);
  float *const st_dst = lds + F_RING + sr * TP + sc;
  // compute role: operand reads by row (a1) and by column (a2), see the layout note above
  const int a1 = c * TP + 16 * g, a2 = (4 * g) * TP + c;
  float *const tb = lds + F_TB + wave * TB;
  float *const stat = lds + F_ST;
  // fragments of a key block: (row c, columns 16g + i) and (rows 4g + r, columns 16 dt + c)
  unsigned fv_row[4], fv_col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fv_row[i] = 16 * g + 4 * i < p.dh ? (unsigned)((c * p.ld + 16 * g + 4 * i) * 4) : kOutside;
    fv_col[i] = 16 * i + c < p.dh ? (unsigned)(((4 * g) * p.ld + 16 * i + c) * 4) : kOutside;  // (i = dt)
  }
  const int row_bytes = (int)(p.ld * 4);

  // ---- position bookkeeping (scalar): item = (head h, key block q), 4 items per round ----
  struct Pos {
    int h, q;
  };
  auto advance = [&](Pos &x, int by) {
    x.q += by;
    while (x.q >= nblk) x.q -= nblk, ++x.h;
  };
  Pos lo{0, 0}, my{0, 0};
  advance(my, wave);

  f32x4 kf[4], vf[4], kt[4];  // this round's key block; kt[r][dt] = K[16 q + 4g + r][16 dt + c]
  auto load_block = [&](const Pos &x, f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&t)[4]) {
    int oq, oo, og;
    offs(x.h, &oq, &oo, &og);
    const __amdgpu_buffer_rsrc_t rk = rsrc(p.k, oq, bytes_qkv), rv = rsrc(p.v, oq, bytes_qkv);
    const int so = x.q * tile_q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = buf_quad(rk, fv_row[i], so);
      b[i] = buf_quad(rv, fv_row[i], so);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) t[r][dt] = buf_f32(rk, fv_col[dt], so + r * row_bytes);
  };

  // Q and dO tiles of query tile qt of the round's head(s) -> registers -> ring stage
  f32x4 lq[2], lg[2];
  auto load_tiles = [&](int oq0, int oo0, int oq1, int oo1, bool two_, int qt) {
    lq[0] = buf_quad(rsrc(p.q, oq0, bytes_qkv), sv_q, qt * tile_q);
    lg[0] = buf_quad(rsrc(p.dout, oo0, bytes_o), sv_o, qt * tile_o);
    if (two_) {  // workgroup-uniform
      lq[1] = buf_quad(rsrc(p.q, oq1, bytes_qkv), sv_q, qt * tile_q);
      lg[1] = buf_quad(rsrc(p.dout, oo1, bytes_o), sv_o, qt * tile_o);
    }
  };
  auto store_tiles = [&](int stage, bool two) {
    float *d = st_dst + stage * STAGE;
    *reinterpret_cast<f32x4 *>(d) = lq[0];
    *reinterpret_cast<f32x4 *>(d + TILE) = lg[0];
    if (two) {
      *reinterpret_cast<f32x4 *>(d + SLOT) = lq[1];
      *reinterpret_cast<f32x4 *>(d + SLOT + TILE) = lg[1];
    }
  };

  // Row statistics of head h into its LDS slot (h & 1): max * log2(e), 1 / sum, sum_d dO * O.
  // All 256 threads; 7 query tiles of loads in flight at a time.
  auto fill_stats = [&](int h) {
    int oq, oo, og;
    offs(h, &oq, &oo, &og);
    const __amdgpu_buffer_rsrc_t rg_ = rsrc(p.dout, oo, bytes_o), ro_ = rsrc(p.o, oo, bytes_o);
    float *dst = stat + (h & 1) * 3 * MAXN;
    const float *sp = p.stats + (long long)(head0 + h) * p.N * 2;
    constexpr int G = 7;
    for (int t0 = 0; t0 < nqt; t0 += G) {
      f32x4 a[G], b[G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        a[i] = buf_quad(rg_, sv_o, (t0 + i) * tile_o);  // (tiles past the sequence read as zeros)
        b[i] = buf_quad(ro_, sv_o, (t0 + i) * tile_o);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        float part = (a[i][0] * b[i][0] + a[i][1] * b[i][1]) + (a[i][2] * b[i][2] + a[i][3] * b[i][3]);
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) part += __shfl_xor(part, m, 16);
        const int row = 16 * (t0 + i) + sr;
        if ((threadIdx.x & 15) == 0 && row < MAXN && t0 + i < nqt) {
          const bool ok = row < p.N;
          const float mx = ok ? sp[2 * row] : 0.f, sm = ok ? sp[2 * row + 1] : 1.f;
          dst[row] = mx * 1.44269504088896340736f;
          dst[MAXN + row] = ok ? 1.0f / sm : 0.f;
          dst[2 * MAXN + row] = ok ? part : 0.f;
        }
      }
    }
  };

  // ---- prologue: first key block, first query tile, statistics of the first head(s) ----
  int ready = -1;  // highest head whose statistics are in LDS
  bool two = lo.q + 3 >= nblk && lo.h + 1 < nh;
  if (wave < items) load_block(my, kf, vf, kt);
  {
    int oq0, oo0, og0, oq1, oo1, og1;
    offs(0, &oq0, &oo0, &og0);
    offs(1, &oq1, &oo1, &og1);
    load_tiles(oq0, oo0, oq1, oo1, two, 0);
    store_tiles(0, two);
  }
  int stage = 0;  // ring stage of the tile about to be used = set of partial tiles about to be written

  // dQ leaves two steps behind the step that forms its partial tiles.  The partial tiles a step
  // puts in LDS are summed at the END of the next step (one barrier per step: the sets alternate),
  // together with what the head's earlier rounds stored, which that step loads at its start; the
  // sums stay in registers across the barrier and are stored FIRST thing in the step after, ahead
  // of its loads.  (The counter that waits for memory go by counts loads and stores in issue
  // order: a store issued behind a step's loads is waited for along with them, in every step; a
  // loaded value carried across the barrier costs a full wait at the top of the loop.)  All of it
  // runs in EVERY step, the first ones included -- nothing pending: no tiles to sum, and an offset
  // outside the descriptor makes the load return zeros and drops the store -- because a load that
  // some path leaves unconsumed also costs a full wait at the top of the loop.
  bool pend_acc = false;
  int pend_k0 = 0, pend_na = 0, pend_row0 = 0;
  int pend_lo = 0, pend_hi = 0;
  unsigned pend_voff = kOutside;
  f32x4 out_lo = zero4(), out_hi = zero4();  // sums on their way out (second head: only where a round spans two)
  int out_off_lo = 0, out_off_hi = 0, out_so = 0;
  unsigned out_voff = kOutside;
  bool out_two = false;
  auto load_pending_old = [&]() {
    return buf_quad(rsrc(p.dq, pend_lo, bytes_g), pend_acc ? pend_voff : kOutside, pend_row0 * (tile_g / 16));
  };
  auto write_out = [&]() {
    // thread t owns query row t/16, columns 4*(t%16) .. (rows past the sequence and columns past
    // the head fall outside the descriptor: the store is dropped)
    buf_store_quad(out_lo, rsrc(p.dq, out_off_lo, bytes_g), out_voff, out_so);
    if (out_two) buf_store_quad(out_hi, rsrc(p.dq, out_off_hi, bytes_g), out_voff, out_so);
  };
  auto sum_pending = [&](int set, const f32x4 pend_old) {  // in wave order
    const float *src = lds + F_PB + set * 4 * PB + sr * PBP + sc;
    if (pend_k0 == 4) {  // the usual round: four blocks of one head
      const f32x4 p0 = *reinterpret_cast<const f32x4 *>(src), p1 = *reinterpret_cast<const f32x4 *>(src + PB);
      const f32x4 p2 = *reinterpret_cast<const f32x4 *>(src + 2 * PB), p3 = *reinterpret_cast<const f32x4 *>(src + 3 * PB);
      out_lo = (((pend_old + p0) + p1) + p2) + p3;
      out_two = false;
    } else {
      out_lo = pend_old;
      for (int w = 0; w < pend_k0; ++w) out_lo += *reinterpret_cast<const f32x4 *>(src + w * PB);
      out_two = pend_k0 < pend_na;  // the round's second head: always its first round
      out_hi = zero4();
      for (int w = pend_k0; w < pend_na; ++w) out_hi += *reinterpret_cast<const f32x4 *>(src + w * PB);
    }
    out_off_lo = pend_lo, out_off_hi = pend_hi, out_so = pend_row0 * (tile_g / 16), out_voff = pend_voff;
  };

  for (int round = 0; round < rounds; ++round) {
    const int na = min(4, items - 4 * round);        // waves with a block in this round
    const bool active = wave < na;                   // wave-uniform
    const int k0 = min(na, nblk - lo.q);             // waves [0, k0) work on head lo.h, [k0, na) on the next
    const int slot = (active && my.h != lo.h) ? 1 : 0;
    const bool accumulate = lo.q != 0;               // head lo.h has stored dQ in an earlier round
    // statistics of heads entering this round (a head's slot is free: the head two before it is done)
    {
      const int need = two ? lo.h + 1 : lo.h;
      const bool fill = ready < need;
      while (ready < need) fill_stats(++ready);
      if (fill || round == 0) __syncthreads();  // statistics (and the first staged tile) are visible
    }
    Pos nlo = lo, nmy = my;
    advance(nlo, 4);
    advance(nmy, 4);
    const bool ntwo = nlo.q + 3 >= nblk && nlo.h + 1 < nh;
    const bool more_rounds = round + 1 < rounds;
    // the next round's key block travels during this one
    f32x4 nkf[4], nvf[4], nkt[4];
    if (more_rounds && wave + 4 * (round + 1) < items) load_block(nmy, nkf, nvf, nkt);

    int oq_lo, oo_lo, og_lo, oq_hi, oo_hi, og_hi;  // element offsets of the round's head(s)
    offs(lo.h, &oq_lo, &oo_lo, &og_lo);
    offs(lo.h + 1, &oq_hi, &oo_hi, &og_hi);
    const bool key_ok = 16 * my.q + c < p.N;
    const float *mystat = stat + (my.h & 1) * 3 * MAXN + 4 * g;

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = zero4(), dv[dt] = zero4();

    for (int qt = 0; qt < nqt; ++qt) {
      const bool last = qt + 1 == nqt;
      // next in the stream: this round's next query tile, or the next round's first
      const bool stream_on = !last || more_rounds;
      write_out();
      if (stream_on) {
        int a0 = oq_lo, b0 = oo_lo, a1_ = oq_hi, b1 = oo_hi, unused;
        if (last) {  // the tile that opens the next round
          offs(nlo.h, &a0, &b0, &unused);
          offs(nlo.h + 1, &a1_, &b1, &unused);
        }
        load_tiles(a0, b0, a1_, b1, last ? ntwo : two, last ? 0 : qt + 1);
      }
      const f32x4 old = load_pending_old();

      if (active) {
        const float *qs = lds + F_RING + stage * STAGE + slot * SLOT, *gs = qs + TILE;
        float *pb = lds + F_PB + (stage * 4 + wave) * PB;
        // S = Q K^T and dP = dO V^T, both [query][key]
        f32x4 s = zero4(), dp = zero4();
        {
          f32x4 qa[4], ga[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            qa[i] = *reinterpret_cast<const f32x4 *>(qs + a1 + 4 * i);
            ga[i] = *reinterpret_cast<const f32x4 *>(gs + a1 + 4 * i);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i / 4][i % 4], kf[i / 4][i % 4], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i / 4][i % 4], vf[i / 4][i % 4], dp, 0, 0, 0);
          }
        }
        // P from the saved statistics; dS = P (dP - dsum) scale
        const f32x4 m2 = *reinterpret_cast<const f32x4 *>(mystat + 16 * qt);
        const f32x4 li = *reinterpret_cast<const f32x4 *>(mystat + MAXN + 16 * qt);
        const f32x4 ds_ = *reinterpret_cast<const f32x4 *>(mystat + 2 * MAXN + 16 * qt);
        f32x4 pr, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -m2[r])) * li[r];
          pr[r] = key_ok ? e : 0.f;
          ds[r] = (pr[r] * (dp[r] - ds_[r])) * p.scale;
          tb[(4 * g + r) * TBP + c] = ds[r];  // dS[query][key], read back transposed below
        }
        // dV^T += dO^T P and dK^T += Q^T dS (contraction over the 16 queries)
        {
          float qa[16], ga[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) qa[i] = qs[a2 + (i / 4) * TP + 16 * (i % 4)], ga[i] = gs[a2 + (i / 4) * TP + 16 * (i % 4)];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            dv[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i], pr[i / 4], dv[i % 4], 0, 0, 0);
            dk[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i], ds[i / 4], dk[i % 4], 0, 0, 0);
          }
        }
        // this block's share of dQ^T[d][query] = K^T dS^T (contraction over the 16 keys)
        const f32x4 dst = *reinterpret_cast<const f32x4 *>(tb + c * TBP + 4 * g);  // dS[query c][keys 4g ..]
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = zero4();
#pragma unroll
        for (int i = 0; i < 16; ++i)
          dq[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[i / 4][i % 4], dst[i / 4], dq[i % 4], 0, 0, 0);
        // register r of dq[dt] is dQ[query c][16 dt + 4g + r]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4 *>(pb + c * PBP + 16 * dt + 4 * g) = dq[dt];
      }
      if (stream_on) store_tiles(stage ^ 1, last ? ntwo : two);
      sum_pending(stage ^ 1, old);
      pend_voff = sv_g, pend_acc = accumulate, pend_k0 = k0, pend_na = na, pend_row0 = 16 * qt, pend_lo = og_lo, pend_hi = og_hi;
      stage ^= 1;
      __syncthreads();  // this step's partial tiles and the next step's query tile are in LDS
    }

    // ---- this block's dK and dV: register r of d*[dt] is row (key) c, column 16 dt + 4g + r ----
    if (active && key_ok) {
      int oq, oo, og;
      offs(my.h, &oq, &oo, &og);
      const int key = 16 * my.q + c;
      float *dvrow = p.dv + og + (long long)key * p.ldg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        if (16 * dt + 4 * g < p.dh) *reinterpret_cast<f32x4 *>(dvrow + 16 * dt + 4 * g) = dv[dt];
      if (p.dk_mode == VITK_DK_EXACT) {
        float *dkrow = p.dk + og + (long long)key * p.ldg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          if (16 * dt + 4 * g < p.dh) *reinterpret_cast<f32x4 *>(dkrow + 16 * dt + 4 * g) = dk[dt];
      } else {
        // The reference forms Q^T dS as [dh][N] per head and re-reads that buffer as [N][dh]
        // (src/layers/MultiHeadAttention.cpp:207-222): element (d, key) lands at flat index d*N + key.
        float *base = p.dk + og;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = 16 * dt + 4 * g + r;
            if (d < p.dh) {
              const int f = d * p.N + key;
              const int fr = p.dh == 64 ? f >> 6 : f / p.dh, fc = p.dh == 64 ? f & 63 : f % p.dh;
              base[(long long)fr * p.ldg + fc] = dk[dt][r];
            }
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[i] = nkf[i], vf[i] = nvf[i], kt[i] = nkt[i];
    lo = nlo, my = nmy, two = ntwo;
  }
  write_out();
  sum_pending(stage ^ 1, load_pending_old());
  write_out();
}

int g_bwd16_mode = 1;  // 0 = off, 1 = on (heads per workgroup from the device size), n >= 2 = on with n heads per workgroup

}  // namespace

namespace vitk {

// (four consecutive blocks must span at most two heads: at least 4 blocks of 16 per head; the
// per-lane byte offsets into a head and the element offsets of the heads are 32-bit)
bool sdpa_bwd16_applies(int B, int N, int dh, bool vec, long long ld, long long ldo, long long ldg) {
  const long long m = ld > ldo ? (ld > ldg ? ld : ldg) : (ldo > ldg ? ldo : ldg);
  return g_bwd16_mode != 0 && vec && dh > 32 && dh <= 64 && N >= 49 && N <= MAXN && (long long)N * m * 4 < (1ll << 30) &&
         (long long)B * N * m < (1ll << 31);
}

int sdpa_bwd16_launch(const float *q, const float *k, const float *v, long long ld, const float *o, const float *dout,
                      long long ldo, const float *stats, int B, int H, int N, int dh, float scale, int dk_mode,
                      float *dq, float *dk, float *dv, long long ldg, hipStream_t st) {
  const size_t lds = F_TOTAL * sizeof(float);
  static std::atomic<unsigned long long> configured{0};  // raise the dynamic LDS limit once per device
  if (vitk::first_use_on_device(&configured)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sdpa_bwd16_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    VITK_REQUIRE(e == hipSuccess, "vitk_sdpa_bwd: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
  }
  Bwd16Params p;
  p.q = q; p.k = k; p.v = v; p.ld = (int)ld; p.o = o; p.dout = dout; p.ldo = (int)ldo; p.stats = stats;
  p.dq = dq; p.dk = dk; p.dv = dv; p.ldg = (int)ldg;
  p.B = B; p.H = H; p.N = N; p.dh = dh; p.scale = scale; p.dk_mode = dk_mode;
  // Heads per workgroup.  Two workgroups fit a CU.  A workgroup with h heads needs ceil(13 h / 4)
  // rounds (13 = blocks per head here), and a round takes ~0.63 as long when the CU holds a single
  // workgroup (measured at 768 .. 3072 heads: 0.030 vs 0.048 ms per round); workgroups beyond
  // two per CU wait for a second pass.  Take the h with the least predicted time.
  const int bh = B * H, cus = cu_count(), nblk = (N + 15) / 16;
  int hpw = 1;
  double best = 1e30;
  for (int h = 1; h <= 12; ++h) {
    const long long wgs = (bh + h - 1) / h, rounds = ((long long)h * nblk + 3) / 4;
    const long long passes = (wgs + 2LL * cus - 1) / (2LL * cus);
    const double cost = (double)rounds * (double)passes * (wgs <= cus ? 0.63 : 1.0);
    if (cost < best) best = cost, hpw = h;
  }
  if (g_bwd16_mode >= 2) hpw = g_bwd16_mode;
  p.hpw = hpw;
  hipLaunchKernelGGL(sdpa_bwd16_kernel, dim3((unsigned)((bh + hpw - 1) / hpw)), dim3(256), lds, st, p);
  return check_launch("vitk_sdpa_bwd");
}

}  // namespace vitk

extern "C" int vitk_sdpa_bwd_tile16_enable(int on) {
  g_bwd16_mode = on < 0 ? 0 : on;
  return 0;
}
