#include "core/Device.hpp"

#include <hip/hip_runtime_api.h>

#include <random>
#include <unordered_map>
#include <vector>

#include "vitk.h"

namespace vit {

namespace {

struct State {
  bool initialised = false;
  int device = 0;
  hipStream_t stream = nullptr;
  std::unordered_map<size_t, std::vector<void *>> free_lists;
  size_t in_use = 0, cached = 0;
  void *ws = nullptr;
  size_t ws_bytes = 0;
  std::mt19937_64 rng{std::random_device{}()};
  uint64_t dropout_seed = 0x5eedULL;
  uint64_t dropout_offset = 0;
  uint64_t dropout_salt = 0;
  const float *replay = nullptr;
  size_t replay_left = 0;
};

State &S() {
  static State s;
  return s;
}

size_t round_up(size_t bytes) { return (bytes + 255) / 256 * 256; }

}  // namespace

void check(int rc, const char *what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + vitk_last_error());
}

void hip_check(int e, const char *what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(static_cast<hipError_t>(e)));
}

void init(int device) {
  State &s = S();
  if (s.initialised && s.device == device) return;
  // Blocks, the workspace and the free lists belong to the device they were allocated on: moving
  // on while any exist would hand pointers of one device to kernels launched on another.
  if (s.initialised && (s.in_use > 0 || s.cached > 0 || s.ws != nullptr))
    throw std::runtime_error("vit::init: device " + std::to_string(s.device) + " already holds " +
                             std::to_string(s.in_use + s.cached + s.ws_bytes) + " bytes of this process's tensors; "
                             "select device " + std::to_string(device) + " (or join the communicator) before "
                             "the first tensor is created");
  check(vitk_init(device), "vitk_init");
  s.device = device;
  s.initialised = true;
}

static void ensure() {
  if (!S().initialised) init(0);
}

void set_stream(void *hip_stream) {
  ensure();
  S().stream = static_cast<hipStream_t>(hip_stream);
}

void *stream() {
  ensure();
  return S().stream;
}

void synchronize() {
  ensure();
  hip_check(hipStreamSynchronize(S().stream), "hipStreamSynchronize");
}

float *device_alloc(size_t count) {
  ensure();
  State &s = S();
  const size_t bytes = round_up((count ? count : 1) * sizeof(float));
  auto it = s.free_lists.find(bytes);
  void *p = nullptr;
  if (it != s.free_lists.end() && !it->second.empty()) {
    p = it->second.back();
    it->second.pop_back();
    s.cached -= bytes;
  } else {
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      release_cached_memory();
      hip_check(hipMalloc(&p, bytes), "hipMalloc");
    }
  }
  s.in_use += bytes;
  return static_cast<float *>(p);
}

void device_free(float *p, size_t count) {
  if (!p) return;
  State &s = S();
  const size_t bytes = round_up((count ? count : 1) * sizeof(float));
  s.free_lists[bytes].push_back(p);
  s.in_use -= bytes;
  s.cached += bytes;
}

void release_cached_memory() {
  State &s = S();
  if (s.cached == 0) return;
  (void)hipStreamSynchronize(s.stream);
  for (auto &kv : s.free_lists) {
    for (void *p : kv.second) (void)hipFree(p);
    kv.second.clear();
  }
  s.cached = 0;
}

size_t device_bytes_in_use() { return S().in_use; }

float *workspace(size_t bytes) {
  ensure();
  State &s = S();
  if (bytes > s.ws_bytes) {
    if (s.ws) {
      (void)hipStreamSynchronize(s.stream);
      (void)hipFree(s.ws);
      s.ws = nullptr;
    }
    const size_t want = round_up(bytes + bytes / 4);
    hip_check(hipMalloc(&s.ws, want), "hipMalloc(workspace)");
    s.ws_bytes = want;
  }
  return static_cast<float *>(s.ws);
}

size_t workspace_bytes() { return S().ws_bytes; }

void copy_to_device(float *dst, const float *src, size_t count) {
  ensure();
  if (!count) return;
  // pageable source: hipMemcpyAsync returns after staging, so the host buffer may be reused
  hip_check(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyHostToDevice, S().stream), "H2D copy");
  hip_check(hipStreamSynchronize(S().stream), "H2D sync");
}

void copy_to_host(float *dst, const float *src, size_t count) {
  ensure();
  if (!count) return;
  hip_check(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToHost, S().stream), "D2H copy");
  hip_check(hipStreamSynchronize(S().stream), "D2H sync");
}

void copy_on_device(float *dst, const float *src, size_t count) {
  ensure();
  if (!count) return;
  hip_check(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToDevice, S().stream), "D2D copy");
}

void zero_device(float *dst, size_t count) {
  ensure();
  if (!count) return;
  hip_check(hipMemsetAsync(dst, 0, count * sizeof(float), S().stream), "memset");
}

void seed(uint64_t s) {
  S().rng.seed(s);
  S().dropout_seed = s * 0x9E3779B97F4A7C15ULL + 0x5eedULL;
  S().dropout_offset = 0;
}

uint64_t next_seed() { return S().rng(); }

void dropout_stream(size_t count, uint64_t *seed_out, uint64_t *offset_out) {
  State &s = S();
  *seed_out = s.dropout_seed + s.dropout_salt;
  *offset_out = s.dropout_offset;
  s.dropout_offset += (count + 3) / 4;
}

void get_dropout_state(uint64_t *seed_out, uint64_t *offset_out) {
  *seed_out = S().dropout_seed + S().dropout_salt;
  *offset_out = S().dropout_offset;
}

void set_dropout_state(uint64_t seed, uint64_t offset) {
  S().dropout_seed = seed - S().dropout_salt;
  S().dropout_offset = offset;
}

void get_dropout_base_state(uint64_t *seed_out, uint64_t *offset_out) {
  *seed_out = S().dropout_seed;
  *offset_out = S().dropout_offset;
}

void set_dropout_base_state(uint64_t seed, uint64_t offset) {
  S().dropout_seed = seed;
  S().dropout_offset = offset;
}

int current_device() { return S().initialised ? S().device : -1; }

void set_dropout_rank(int rank) { S().dropout_salt = 0xD1B54A32D192ED03ULL * (uint64_t)rank; }

void set_dropout_replay(const float *masks_dev, size_t count) {
  S().replay = count ? masks_dev : nullptr;
  S().replay_left = masks_dev ? count : 0;
}

const float *dropout_replay_take(size_t count) {
  State &s = S();
  if (!s.replay) return nullptr;
  if (count > s.replay_left)
    throw std::runtime_error("dropout replay: the recorded masks are exhausted (" + std::to_string(count) +
                             " factors wanted, " + std::to_string(s.replay_left) + " left)");
  const float *p = s.replay;
  s.replay += count;
  s.replay_left -= count;
  return p;
}

}  // namespace vit
