// What the fp32 matrix pipes of this chip deliver when nothing else is in the way: bare
// v_mfma_f32 loops on random operands, every CU busy, launched back to back for seconds so that
// the clock has settled where the chip holds it under this load.  Answers one question for the
// GEMM in visiontransformer_amd/csrc/gemm.hip (138-143 TFLOP/s inside a training step): how far
// is that from what the hardware sustains -- 157.3 TFLOP/s is 64 FLOP/clk/SIMD at 2.4 GHz, and
// the clock under fp32 matrix load is lower.
//
//   variant  instruction             operands             waves per SIMD
//   a        v_mfma_f32_16x16x4_f32  registers            1
//   b        v_mfma_f32_16x16x4_f32  registers            3   (the GEMM's occupancy)
//   c        v_mfma_f32_16x16x4_f32  re-read from LDS     3   (8 ds_read_b32 per 16 MFMAs, the
//                                                              GEMM loop's ratio)
//   d        v_mfma_f32_32x32x2_f32  registers            3
//   e        c + 4 ds_write_b32 per 16 MFMAs (the GEMM's transposing slab writes)        3
//   f        e + 1 buffer_load_dwordx4 per 16 MFMAs from a 4 MiB (L2-resident) buffer,   3
//            its data being what the writes store (the GEMM's slab fetch)
//   g        f + a workgroup barrier per 64 MFMAs                                        3
//   h, i     c and g with the operand reads as ds_read_b128 (a quarter of the instructions)
//   j        g with one operand read as ds_read_b128, the other as four ds_read_b32
//   k, l     g with two / one workgroup per CU instead of three
//
// Per variant: TFLOP/s over the last second of >= 2 s of launches (HIP events), and the clock a
// workgroup saw inside the last launch (delta s_memtime / delta s_memrealtime x 100 MHz, median
// over workgroups).
//
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/mfma_ceiling tools/mfma_ceiling.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(1);                                                                       \
    }                                                                                     \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;   // 4 waves, one per SIMD
constexpr int kSets = 8;                // sets of 16 MFMAs per loop trip (two 16-deep slabs of a 64 x 64 wave tile)
constexpr int kLdsFloats = kSets * 8 * 64;  // eight 64-lane reads per set
constexpr int kWriteFloats = 4 * kSets * 4 * 64;  // per wave: four 64-lane writes per set
constexpr int kStreamBytes = 4 << 20;

struct Stamp {
  uint64_t c0, r0, c1, r1;
};

// Eight ds_read_b32 at literal offsets from one address register: the operands of set U.
template <int U>
__device__ __forceinline__ void read_set(uint32_t addr, float (&na)[4], float (&nb)[4]) {
  asm volatile("ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\t"
               "ds_read_b32 %3, %8 offset:%12\n\tds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\t"
               "ds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16"
               : "=v"(na[0]), "=v"(na[1]), "=v"(na[2]), "=v"(na[3]), "=v"(nb[0]), "=v"(nb[1]), "=v"(nb[2]), "=v"(nb[3])
               : "v"(addr), "n"((U * 8 + 0) * 256), "n"((U * 8 + 1) * 256), "n"((U * 8 + 2) * 256),
                 "n"((U * 8 + 3) * 256), "n"((U * 8 + 4) * 256), "n"((U * 8 + 5) * 256), "n"((U * 8 + 6) * 256),
                 "n"((U * 8 + 7) * 256));
}
// The same bytes with a quarter of the instructions: two ds_read_b128 (what a k-contiguous LDS
// layout allows: one read brings a lane's operand of four consecutive MFMAs).
template <int U>
__device__ __forceinline__ void read_set_wide(uint32_t addr, float (&na)[4], float (&nb)[4]) {
  f4 x, y;
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
               : "=v"(x), "=v"(y)
               : "v"(addr), "n"((U * 2 + 0) * 1024), "n"((U * 2 + 1) * 1024));
  for (int i = 0; i < 4; ++i) {
    na[i] = x[i];
    nb[i] = y[i];
  }
}
// One operand wide, the other narrow: one ds_read_b128 + four ds_read_b32 (five instructions).
template <int U>
__device__ __forceinline__ void read_set_mixed(uint32_t addr, float (&na)[4], float (&nb)[4]) {
  f4 x;
  asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b32 %1, %5 offset:%7\n\tds_read_b32 %2, %5 offset:%8\n\t"
               "ds_read_b32 %3, %5 offset:%9\n\tds_read_b32 %4, %5 offset:%10"
               : "=v"(x), "=v"(nb[0]), "=v"(nb[1]), "=v"(nb[2]), "=v"(nb[3])
               : "v"(addr), "n"((U * 2 + 0) * 1024), "n"((U * 8 + 4) * 256), "n"((U * 8 + 5) * 256), "n"((U * 8 + 6) * 256),
                 "n"((U * 8 + 7) * 256));
  for (int i = 0; i < 4; ++i) na[i] = x[i];
}
// the reads have landed; tied to the registers so that nothing uses them earlier
__device__ __forceinline__ void landed(float (&na)[4], float (&nb)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(na[0]), "+v"(na[1]), "+v"(na[2]), "+v"(na[3]), "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]));
}

template <int U>
__device__ __forceinline__ void write_set(uint32_t addr, float v0, float v1, float v2, float v3) {
  asm volatile("ds_write_b32 %0, %1 offset:%5\n\tds_write_b32 %0, %2 offset:%6\n\tds_write_b32 %0, %3 offset:%7\n\t"
               "ds_write_b32 %0, %4 offset:%8"
               :
               : "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"((U * 4 + 0) * 256), "n"((U * 4 + 1) * 256),
                 "n"((U * 4 + 2) * 256), "n"((U * 4 + 3) * 256)
               : "memory");
}

// MODE 0: MFMAs only; 1: + operand reads; 2: + slab writes; 3: + slab fetch; 4: + barriers
template <int MODE, int U, int WIDE>
__device__ __forceinline__ void one_set(uint32_t addr, uint32_t waddr, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff,
                                        uint32_t soff, f4 &pend, float (&a)[4], float (&b)[4], f4 (&acc)[4][4]) {
  float na[4], nb[4];
  if (MODE >= 1 && WIDE == 0) read_set<U>(addr, na, nb);
  if (MODE >= 1 && WIDE == 1) read_set_wide<U>(addr, na, nb);
  if (MODE >= 1 && WIDE == 2) read_set_mixed<U>(addr, na, nb);
  if (MODE == 2) write_set<U>(waddr, a[0], a[1], b[0], b[1]);
  if (MODE >= 3) {
    write_set<U>(waddr, pend[0], pend[1], pend[2], pend[3]);
    pend = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (soff + U * 4096) & (kStreamBytes - 1), 0));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  if (MODE >= 1) {
    landed(na, nb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = na[i];
      b[i] = nb[i];
    }
  }
  if (MODE >= 4 && (U & 3) == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MODE, int WIDE>
__global__ __launch_bounds__(kThreads, 3) void mfma16_kernel(const float *__restrict__ src, float *__restrict__ sink,
                                                             Stamp *__restrict__ stamps, int trips) {
  __shared__ float lds[kLdsFloats + (MODE >= 2 ? kWriteFloats : 0)];
  const int lane = threadIdx.x & 63;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(blockIdx.x * 8 + i) % 4096 * 64 + lane];
    b[i] = src[(blockIdx.x * 8 + 4 + i) % 4096 * 64 + lane];
  }
  if (MODE >= 1) {
    for (int i = threadIdx.x; i < kLdsFloats; i += kThreads) lds[i] = src[i];
    __syncthreads();
  }
  f4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  // LDS byte address of this lane's column: 4 bytes per lane and read, or 16
  const uint32_t addr = (uint32_t)(uintptr_t)lds + lane * (WIDE ? 16 : 4);
  const uint32_t waddr = (uint32_t)(uintptr_t)lds + lane * 4 + kLdsFloats * 4 + (threadIdx.x >> 6) * (kWriteFloats / 4) * 4;  // this wave's write area
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, kStreamBytes + 8192, 0x00020000);
  const uint32_t voff = threadIdx.x * 16;
  f4 pend = f4{a[0], a[1], b[0], b[1]};
  uint64_t c0 = 0, r0 = 0;
  if (threadIdx.x == 0) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  uint32_t soff = blockIdx.x * 4096u;
  for (int t = 0; t < trips; ++t) {
    // eight sets of 16 independent MFMAs; the operands of set u+1 are read (literal offsets from
    // one loop-invariant address, no vector ALU work) while set u multiplies
    one_set<MODE, 0, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 1, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 2, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 3, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 4, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 5, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 6, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    one_set<MODE, 7, WIDE>(addr, waddr, rsrc, voff, soff, pend, a, b, acc);
    soff += kSets * 4096u;
  }
  if (threadIdx.x == 0) {
    Stamp s{c0, r0, (uint64_t)__builtin_amdgcn_s_memtime(), (uint64_t)__builtin_amdgcn_s_memrealtime()};
    stamps[blockIdx.x] = s;
  }
  float r = pend[0];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (r == 12345.678f) sink[blockIdx.x * kThreads + threadIdx.x] = r;  // keeps the loop alive, never taken on real data
}

__global__ __launch_bounds__(kThreads, 3) void mfma32_kernel(const float *__restrict__ src, float *__restrict__ sink,
                                                             Stamp *__restrict__ stamps, int trips) {
  const int lane = threadIdx.x & 63;
  float a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    a[i] = src[(blockIdx.x * 4 + i) % 4096 * 64 + lane];
    b[i] = src[(blockIdx.x * 4 + 2 + i) % 4096 * 64 + lane];
  }
  f16v acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  uint64_t c0 = 0, r0 = 0;
  if (threadIdx.x == 0) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int t = 0; t < trips; ++t) {
    // the same FLOPs per trip as the 16x16x4 kernel: 8 x 16 x 2048 = 8 x 2 x 4 x 4096
#pragma unroll
    for (int u = 0; u < 2 * kSets; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  if (threadIdx.x == 0) {
    Stamp s{c0, r0, (uint64_t)__builtin_amdgcn_s_memtime(), (uint64_t)__builtin_amdgcn_s_memrealtime()};
    stamps[blockIdx.x] = s;
  }
  float r = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  if (r == 12345.678f) sink[blockIdx.x * kThreads + threadIdx.x] = r;
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 2.5;
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
  std::vector<float> h((kStreamBytes + 8192) / 4 + 4096 * 64);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (float &v : h) v = u(rng);
  float *src = nullptr, *sink = nullptr;
  Stamp *stamps = nullptr;
  const int max_blocks = cus * 3;
  CHECK(hipMalloc(&src, h.size() * sizeof(float)));
  CHECK(hipMalloc(&sink, (size_t)max_blocks * kThreads * sizeof(float)));
  CHECK(hipMalloc(&stamps, max_blocks * sizeof(Stamp)));
  CHECK(hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double flop_per_trip_wave = (double)kSets * 16 * 2 * 16 * 16 * 4;  // = 2 kSets x 4 x 2 x 32 x 32 x 2

  struct Variant {
    const char *name;
    int waves_per_simd;
    int kind;  // 0..4: 16x16x4 kernel in that MODE; -1: 32x32x2 kernel
  } variants[] = {{"a 16x16x4 registers, 1 wave/SIMD", 1, 0},
                  {"b 16x16x4 registers, 3 waves/SIMD", 3, 0},
                  {"c 16x16x4 + LDS operand reads, 3 waves/SIMD", 3, 1},
                  {"d 32x32x2 registers, 3 waves/SIMD", 3, -1},
                  {"e c + LDS slab writes", 3, 2},
                  {"f e + slab fetch (L2-resident)", 3, 3},
                  {"g f + barrier per 64 MFMAs", 3, 4},
                  {"h c with ds_read_b128 (2 per 16 MFMAs)", 3, 5},
                  {"i g with ds_read_b128", 3, 6},
                  {"j g with one ds_read_b128 + four ds_read_b32", 3, 7},
                  {"k g at 2 waves/SIMD (a third workgroup in its fill or store phase)", 2, 4},
                  {"l g at 1 wave/SIMD", 1, 4}};
  for (const Variant &v : variants) {
    const int blocks = cus * v.waves_per_simd;
    const int trips = 3072 / v.waves_per_simd * 3;  // about 16 ms per launch
    auto launch = [&] {
      switch (v.kind) {
        case 7: hipLaunchKernelGGL((mfma16_kernel<4, 2>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 5: hipLaunchKernelGGL((mfma16_kernel<1, 1>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 6: hipLaunchKernelGGL((mfma16_kernel<4, 1>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 0: hipLaunchKernelGGL((mfma16_kernel<0, 0>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 1: hipLaunchKernelGGL((mfma16_kernel<1, 0>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 2: hipLaunchKernelGGL((mfma16_kernel<2, 0>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 3: hipLaunchKernelGGL((mfma16_kernel<3, 0>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        case 4: hipLaunchKernelGGL((mfma16_kernel<4, 0>), dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips); break;
        default: hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(kThreads), 0, 0, src, sink, stamps, trips);
      }
    };
    // load phase: launches back to back until `seconds` minus the measured last second have passed
    const auto t0 = std::chrono::steady_clock::now();
    int warm = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds - 1.0) {
      for (int i = 0; i < 8; ++i) launch();
      CHECK(hipStreamSynchronize(0));
      warm += 8;
    }
    const double per_launch = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / std::max(warm, 1);
    const int n = std::max(8, (int)(1.0 / per_launch));
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = flop_per_trip_wave * trips * 4.0 * blocks * n;
    std::vector<Stamp> st(blocks);
    CHECK(hipMemcpy(st.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (const Stamp &s : st)
      if (s.r1 > s.r0) ghz.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    const double tf = flop / (ms * 1e-3) / 1e12;
    // cycles one SIMD spends per 16x16x4-equivalent MFMA (2048 FLOP x 4 SIMDs x CUs per cycle at 64 FLOP/clk/SIMD)
    std::printf("%-48s %7.2f TFLOP/s  %5.3f of 157.3  clock %5.3f GHz  -> %5.1f FLOP/clk/SIMD  (%d launches of %.2f ms after %.1f s of load)\n",
                v.name, tf, tf / 157.3, clk, clk > 0 ? tf * 1e12 / (clk * 1e9) / (cus * 4.0) : 0.0, n, ms / n,
                seconds - 1.0);
    std::fflush(stdout);
  }
  return 0;
}
