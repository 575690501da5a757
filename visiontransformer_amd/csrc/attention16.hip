// Fused scaled-dot-product attention forward for head dimensions 33..64 on v_mfma_f32_16x16x4_f32,
// with 16-row granularity (reference: src/layers/MultiHeadAttention.cpp:96-156).
//
// Why a second forward kernel.  ViT sequences are 197 tokens: in 32-row tiles that is 7 x 32 = 224
// rows (+29 % score area) and 7 query tiles for a workgroup of 4 waves (one wave in eight idle).
// In 16-row tiles it is 13 x 16 = 208 rows, and an odd tile count is evened out by dealing the
// query blocks of SEVERAL heads to the waves round by round: a team of four workgroups owns 16
// heads = 208 blocks = 13 rounds of 16 blocks, 4 consecutive ones per workgroup (one per wave),
// with every wave busy in every round (see Teams in the kernel for why four workgroups share).
//
// One round = each wave computes one block of 16 queries of "its" head against all keys:
//   pass 1  S^T[key][query] = K Q^T over all key tiles  (NKT x 4 accumulator registers per lane),
//           softmax over the registers (a query lives in the 4 lanes c, c+16, c+32, c+48),
//   pass 2  O^T[d][query] += V^T P^T with the probabilities fed back as operand B, register by
//           register (accumulator register r of lane (c, g) is row 4g + r = the contraction index
//           the next product wants from lane group g).
// The four blocks of a workgroup's round belong to at most two heads (13 >= 4); K and V tiles of
// those heads stream through a two-stage LDS ring, 32 keys per stage, loaded by all 256 threads.  The stream
// runs on across rounds, so a wave never waits for a round's first tile.
//
// Lane (c = lane & 15, g = lane >> 4) of v_mfma_f32_16x16x4_f32 supplies A[i = c][k = g] and
// B[k = g][j = c] and receives D[i = 4g + r][j = c] in register r.  Contractions over the head
// dimension use d = 16g + step (a lane's query fragment is 16 contiguous floats); the K tile sits
// in LDS as [key][d] with a pitch of 65 floats and the V tile with a pitch of 68, which makes
// both operand reads (one ds_read_b32 per MFMA) conflict-free.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TEAM = 4;           // workgroups per team; a team shares TEAM * (heads per workgroup) heads
constexpr int KP = 65, VP = 68;   // LDS row pitches of a key tile and of a value tile
constexpr int SLOT = 32 * VP;     // one head's 32-row tile (either kind)
constexpr int STAGE = 2 * SLOT;   // two heads may be live in one round
constexpr int DH = 64;            // padded head dimension

struct Fwd16Params {
  const float *q, *k, *v;  // element (b, n, h, d) at ptr[(b*N + n)*ld + h*dh + d]
  long long ld;
  float *out;
  long long ldo;
  float *stats;            // [B*H][N][2]: row max of the scaled scores, row sum of exp (may be null)
  int B, H, N, dh;
  float scale;
  int hpw;                 // heads per workgroup (1..4): enough for all workgroups to be resident at once
};

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// The matrix cores and the vector ALU of a SIMD do not overlap on fp32 work: every vector
// instruction a wave issues is time its neighbours' MFMAs do not get.  So the loop below is kept
// free of per-lane address arithmetic: all global reads go through buffer descriptors (a scalar
// base per head, a per-lane byte offset that never changes, a scalar offset per tile; rows past
// the sequence read as zeros because the descriptor ends there), the position in the stream is
// tracked in scalar registers, and the softmax touches each score three times (maximum, one
// fused multiply-add + exp2, sum) with the division applied to the 16 output registers instead
// of the probabilities.
__device__ __forceinline__ f32x4 buf_quad(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, soff, 0));
}

// NS = number of 32-key stages = ceil(N / 32), a template parameter so that every step of the
// unrolled stream knows at compile time what follows it.
template <int NS>
__global__ __launch_bounds__(256, NS > 7 ? 2 : 3) void sdpa_fwd16_kernel(const Fwd16Params p) {
  constexpr int NKT = 2 * NS;
  const bool tail2 = 16 * (2 * NS - 1) < p.N;  // the last stage holds two 16-key tiles, not one
  __shared__ __attribute__((aligned(16))) float ring[2 * STAGE];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int BH = p.B * p.H;
  // Teams.  A head's K and V are re-streamed in every round that touches the head; with every
  // workgroup on heads of its own, the 96 workgroups resident on an XCD cycle ~10 MB through its
  // 4 MB L2 between two uses and each round's tiles come from memory again (measured: 1.36 GB read
  // per launch for 0.47 GB of operands).  So TEAM workgroups that land on the same XCD (workgroups
  // go to the 8 XCDs round-robin by index) share TEAM * hpw heads and walk their block list
  // together: in round r workgroup j of the team takes blocks 16 r + 4 j .. + 3, so the team is
  // on the same one or two heads at the same time and three of its four tile fetches hit in L2.
  // Placement and pace only matter for speed: results do not depend on either.
  const int xcd = blockIdx.x % 8, slot_in_xcd = blockIdx.x / 8, tj = slot_in_xcd % TEAM;
  const int head0 = ((slot_in_xcd / TEAM) * 8 + xcd) * (p.hpw * TEAM);
  if (head0 >= BH) return;
  const int nh = min(p.hpw * TEAM, BH - head0);
  const int nblk = (p.N + 15) / 16, items = nh * nblk;
  constexpr int RSTEP = 4 * TEAM;  // blocks a team takes per round
  const int first = 4 * tj;        // this workgroup's first block in round 0
  if (first >= items) return;
  const int rounds = (items - first + RSTEP - 1) / RSTEP;
  const int srow = threadIdx.x >> 4, scol = (threadIdx.x & 15) * 4;  // this thread's part of a staged tile
  const int tile_bytes = (int)(32 * p.ld * 4), head_bytes = (int)(((long long)(p.N - 1) * p.ld + p.dh) * 4);
  constexpr unsigned kOutside = 0x7fffff00u;  // beyond any descriptor: reads as zero

  auto rsrc_at = [&](const float *ptr, long long off) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ptr + off), 0, head_bytes, 0x00020000);
  };

  // staging: thread t moves the quads (row t/16 and t/16 + 16, columns 4*(t%16) ..) of a 32-row tile
  const unsigned sv0 = scol < p.dh ? (unsigned)((srow * p.ld + scol) * 4) : kOutside;
  const unsigned sv1 = scol < p.dh ? (unsigned)(((srow + 16) * p.ld + scol) * 4) : kOutside;
  float *const lk0 = ring + srow * KP + scol, *const lv0 = ring + srow * VP + scol;
  f32x4 lr[2][2];
  // tile `t` (32 rows) of K or V of head slots lo (and hi when it differs) -> registers
  auto load_tile = [&](const float *src, long long off_lo, long long off_hi, bool two, int t) {
    const int so = t * tile_bytes;
    const __amdgpu_buffer_rsrc_t r0 = rsrc_at(src, off_lo);
    lr[0][0] = buf_quad(r0, sv0, so);
    lr[0][1] = buf_quad(r0, sv1, so);
    if (two) {  // workgroup-uniform
      const __amdgpu_buffer_rsrc_t r1 = rsrc_at(src, off_hi);
      lr[1][0] = buf_quad(r1, sv0, so);
      lr[1][1] = buf_quad(r1, sv1, so);
    }
  };
  auto store_tile = [&](int stage, bool is_v, bool two) {
    const int so = stage * STAGE;
    if (is_v) {
      *reinterpret_cast<f32x4 *>(lv0 + so) = lr[0][0];
      *reinterpret_cast<f32x4 *>(lv0 + so + 16 * VP) = lr[0][1];
      if (two) {
        *reinterpret_cast<f32x4 *>(lv0 + so + SLOT) = lr[1][0];
        *reinterpret_cast<f32x4 *>(lv0 + so + SLOT + 16 * VP) = lr[1][1];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lk0[so + e] = lr[0][0][e];
        lk0[so + 16 * KP + e] = lr[0][1][e];
      }
      if (two) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lk0[so + SLOT + e] = lr[1][0][e];
          lk0[so + SLOT + 16 * KP + e] = lr[1][1][e];
        }
      }
    }
  };
  // this lane's fragment of query row 16*qb + c: columns 16g .. 16g+15 (a per-lane constant offset)
  unsigned qv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qv[i] = 16 * g + 4 * i < p.dh ? (unsigned)((c * p.ld + 16 * g + 4 * i) * 4) : kOutside;
  auto load_query = [&](long long off, int qb, f32x4 (&q)[4]) {
    const __amdgpu_buffer_rsrc_t r = rsrc_at(p.q, off);
    const int so = qb * (tile_bytes / 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = buf_quad(r, qv[i], so);
  };

  // Position of a block in the workgroup's list of (head, block) items: head index inside the
  // workgroup, block inside the head, the head's index inside its image and its element offset.
  // Scalar registers, advanced by the team's 16 items per round (no division in the loop).
  struct Pos {
    int h, q, hm;
    long long off;
  };
  const long long next_image = (long long)p.N * p.ld - (long long)p.H * p.dh;
  auto advance = [&](Pos &x, int by) {
    x.q += by;
    while (x.q >= nblk) {
      x.q -= nblk;
      ++x.h;
      ++x.hm;
      x.off += p.dh;
      if (x.hm == p.H) x.hm = 0, x.off += next_image;
    }
  };
  Pos lo{0, 0, head0 % p.H, (long long)(head0 / p.H) * p.N * p.ld + (long long)(head0 % p.H) * p.dh};
  advance(lo, first);
  Pos my = lo;
  advance(my, wave);
  // a round's four blocks span at most two heads (nblk >= 4): the head of the first block and,
  // when the fourth block wraps, the one after it -- unless that lies beyond the workgroup's last
  // head (final round), where no wave works on it
  auto second_head = [&](const Pos &x, long long *off) {
    *off = x.off + p.dh + (x.hm + 1 == p.H ? next_image : 0);
    return x.q + 3 >= nblk && x.h + 1 < nh;
  };

  const float sl2 = p.scale * 1.44269504088896340736f;  // scores are exponentiated in base 2
  f32x4 qf[4], qn[4];
  if (first + wave < items) load_query(my.off, my.q, qf);
  {
    long long off2;
    const bool two0 = second_head(lo, &off2);
    load_tile(p.k, lo.off, off2, two0, 0);
    store_tile(0, false, two0);
  }
  __syncthreads();
  // (a round is 2 * NS steps, an even number: step t of a round always uses ring stage t & 1, so
  // every LDS address below is a per-round register plus an instruction immediate)

  for (int round = 0; round < rounds; ++round) {
    const bool active = RSTEP * round + first + wave < items;  // wave-uniform; idle waves still stage tiles
    long long hi_off, nhi_off;
    const bool two = second_head(lo, &hi_off);
    const int slot = (active && my.h != lo.h) ? 1 : 0;
    Pos nlo = lo, nmy = my;  // the next round's positions
    advance(nlo, RSTEP);
    advance(nmy, RSTEP);
    const bool ntwo = second_head(nlo, &nhi_off);

    const float *const kbase = ring + slot * SLOT + c * KP + 16 * g, *const vbase = ring + slot * SLOT + (4 * g) * VP + c;

    // ---- pass 1: scores, transposed ([key][query]) ----
    f32x4 s[NKT];  // (the last one stays zero when the last stage holds one tile)
#pragma unroll
    for (int j = 0; j < 2 * NS; ++j) s[j] = zero4();
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      {
        const bool last_k = t + 1 == NS;  // next in the stream: key tile t+1, or value tile 0
        load_tile(last_k ? p.v : p.k, lo.off, hi_off, two, last_k ? 0 : t + 1);
        if (active) {
          const float *ks = kbase + (t & 1) * STAGE;
          float ka[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) ka[i] = ks[i];
          if (!last_k || tail2) {  // both 16-key tiles of the stage
            float kb[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) kb[i] = ks[16 * KP + i];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              s[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i], qf[i / 4][i % 4], s[2 * t], 0, 0, 0);
              s[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[i], qf[i / 4][i % 4], s[2 * t + 1], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              s[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i], qf[i / 4][i % 4], s[2 * t], 0, 0, 0);
          }
        }
        store_tile((t + 1) & 1, last_k, two);
        __syncthreads();
      }
    }

    // ---- softmax over this lane's registers and the three other lanes of the query ----
    float inv = 0.f;
    if (active) {
      // keys past the sequence sit in the last stage's two tiles only (N > 32 * (NS - 1))
#pragma unroll
      for (int j = NKT - 2; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * j + 4 * g + r >= p.N) s[j][r] = -INFINITY;
      float m = s[0][0];
#pragma unroll
      for (int j = 0; j < NKT; ++j) m = fmaxf(fmaxf(m, fmaxf(s[j][0], s[j][1])), fmaxf(s[j][2], s[j][3]));
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float mb = m * sl2;
      float l = 0.f;
#pragma unroll
      for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], sl2, -mb));
          s[j][r] = e;
          l += e;
        }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      inv = 1.0f / l;
      if (p.stats != nullptr && g == 0 && 16 * my.q + c < p.N) {
        float *st = p.stats + ((long long)(head0 + my.h) * p.N + 16 * my.q + c) * 2;
        st[0] = m * p.scale;
        st[1] = l;
      }
    }
    // the next round's query fragment travels while this round's values are multiplied
    if (RSTEP * (round + 1) + first + wave < items) load_query(nmy.off, nmy.q, qn);

    // ---- pass 2: O^T[d][query] += V^T E^T (unnormalised), divided by the row sum at the end ----
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = zero4();
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      {
        const bool last_v = t + 1 == NS;
        const bool more = !last_v || round + 1 < rounds;
        if (!last_v) load_tile(p.v, lo.off, hi_off, two, t + 1);
        else if (more) load_tile(p.k, nlo.off, nhi_off, ntwo, 0);
        if (active) {
          const float *vs = vbase + ((NS + t) & 1) * STAGE;
          float va[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) va[i] = vs[(i / 4) * VP + 16 * (i % 4)];
          if (!last_v || tail2) {
            float vb[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) vb[i] = vs[(16 + i / 4) * VP + 16 * (i % 4)];
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i], s[2 * t][i / 4], o[i % 4], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[i], s[2 * t + 1][i / 4], o[i % 4], 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i], s[2 * t][i / 4], o[i % 4], 0, 0, 0);
          }
        }
        if (more) store_tile((NS + t + 1) & 1, !last_v, last_v ? ntwo : two);
        __syncthreads();
      }
    }

    // register r of o[dt] is O[query c][d = 16 dt + 4g + r]: one 16-byte store per dt
    if (active && 16 * my.q + c < p.N) {
      float *orow = p.out + (long long)((head0 + my.h) / p.H) * p.N * p.ldo + (long long)(16 * my.q + c) * p.ldo +
                    (long long)my.hm * p.dh;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        if (16 * dt + 4 * g < p.dh) *reinterpret_cast<f32x4 *>(orow + 16 * dt + 4 * g) = o[dt] * inv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) qf[i] = qn[i];
    my = nmy, lo = nlo;
  }
}

int g_tile16_mode = 1;  // 0 = off, 1 = on, n >= 2 = on with n heads per workgroup (tests, measurements)

}  // namespace

namespace vitk {

// (four consecutive blocks must span at most two heads: at least 4 blocks of 16 per head;
// the softmax takes the maximum of the raw scores, which is the maximum of the scaled ones for a
// positive scale only; the per-lane offsets into a head are 32-bit)
bool sdpa_fwd16_applies(int N, int dh, bool vec, float scale, long long ld) {
  return g_tile16_mode != 0 && vec && dh > 32 && dh <= DH && N >= 49 && N <= 256 && scale > 0.f && (long long)N * ld * 4 < (1ll << 30);
}

int sdpa_fwd16_launch(const float *q, const float *k, const float *v, long long ld, int B, int H, int N, int dh,
                      float scale, float *out, long long ldo, float *stats, hipStream_t st) {
  Fwd16Params p;
  p.q = q; p.k = k; p.v = v; p.ld = ld; p.out = out; p.ldo = ldo; p.stats = stats;
  p.B = B; p.H = H; p.N = N; p.dh = dh; p.scale = scale;
  // three workgroups fit a CU: as many heads per workgroup as keep about that many in flight
  // (ViT-B at batch 256: 3072 heads -> 4; ViT-S: 2; ViT-Ti: 1); more heads per team even out the
  // last round (13 blocks per head over 16 waves), fewer keep small problems spread over the chip
  int hpw = B * H / (3 * cu_count());
  hpw = hpw >= 4 ? 4 : hpw >= 2 ? 2 : 1;
  if (g_tile16_mode >= 2) hpw = g_tile16_mode;
  p.hpw = hpw;
  const int teams = (B * H + hpw * TEAM - 1) / (hpw * TEAM);
  const dim3 grid((unsigned)(8 * TEAM * ((teams + 7) / 8))), blk(256);  // (see Teams in the kernel)
  switch ((N + 31) / 32) {
    case 2: hipLaunchKernelGGL((sdpa_fwd16_kernel<2>), grid, blk, 0, st, p); break;
    case 3: hipLaunchKernelGGL((sdpa_fwd16_kernel<3>), grid, blk, 0, st, p); break;
    case 4: hipLaunchKernelGGL((sdpa_fwd16_kernel<4>), grid, blk, 0, st, p); break;
    case 5: hipLaunchKernelGGL((sdpa_fwd16_kernel<5>), grid, blk, 0, st, p); break;
    case 6: hipLaunchKernelGGL((sdpa_fwd16_kernel<6>), grid, blk, 0, st, p); break;
    case 7: hipLaunchKernelGGL((sdpa_fwd16_kernel<7>), grid, blk, 0, st, p); break;  // 196 patches + CLS
    default: hipLaunchKernelGGL((sdpa_fwd16_kernel<8>), grid, blk, 0, st, p); break;
  }
  return check_launch("vitk_sdpa_fwd");
}

}  // namespace vitk

extern "C" int vitk_sdpa_tile16_enable(int on) {
  g_tile16_mode = on < 0 ? 0 : on;
  return 0;
}
