// Library-level entry points of include/vitk.h: version, error string, device selection.
#include "common.h"

#include <cstring>

namespace vitk {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

static std::atomic<int> g_planning_cus{0};

int cu_count() {
  if (const int forced = g_planning_cus.load(std::memory_order_relaxed)) return forced;
  static int cached[64] = {0};  // benign race: every thread writes the same value
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// True the first time it is called with `done` on the current device (one bit per device), so that
// per-function attributes are set on every device a process uses, not just the first.
bool first_use_on_device(std::atomic<unsigned long long> *done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;  // unknown device: set again (idempotent)
  const unsigned long long bit = 1ull << dev;
  return (done->fetch_or(bit) & bit) == 0;
}

}  // namespace vitk

extern "C" {

int vitk_version(void) { return 100; }

const char *vitk_last_error(void) { return vitk::g_err; }

int vitk_set_planning_cu_count(int cus) {
  VITK_REQUIRE(cus >= 0 && cus <= 4096, "vitk_set_planning_cu_count: %d outside [0, 4096]", cus);
  vitk::g_planning_cus.store(cus, std::memory_order_relaxed);
  return 0;
}

int vitk_init(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  VITK_REQUIRE(e == hipSuccess && count > 0, "vitk_init: no HIP device visible (%s)", hipGetErrorString(e));
  VITK_REQUIRE(device >= 0 && device < count, "vitk_init: device %d out of range [0,%d)", device, count);
  e = hipSetDevice(device);
  VITK_REQUIRE(e == hipSuccess, "vitk_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  VITK_REQUIRE(e == hipSuccess, "vitk_init: hipGetDeviceProperties: %s", hipGetErrorString(e));
  VITK_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0,
               "vitk_init: device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
  // resolve this device's copy of the built-in scratch now, so that vitk_layernorm_bwd_static makes
  // no runtime query later (e.g. under stream capture)
  return vitk::norm_static_setup();
}

}  // extern "C"
