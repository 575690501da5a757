// Communicator and bucketed gradient reducer (see core/Comm.hpp).
#include "core/Comm.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // declarations only: the library is loaded with dlopen
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <thread>

#include "core/Device.hpp"
#include "model/VisionTransformer.hpp"

namespace vit {

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
};

struct State {
  bool active = false;
  int rank = 0, world = 1;
  Rccl rccl;
  ncclComm_t comm = nullptr;
  AllReduceFn ar = nullptr;
  BroadcastFn bc = nullptr;
  void *user = nullptr;
  bool stream_ordered = true;
  hipStream_t side = nullptr;
};

State &C() {
  static State s;
  return s;
}

template <typename F>
void load(void *lib, const char *name, F *out) {
  *out = reinterpret_cast<F>(dlsym(lib, name));
  if (!*out) throw std::runtime_error(std::string("RCCL: symbol not found: ") + name);
}

// An RCCL that the process has already mapped (for instance the one a host framework brought) is
// reused; otherwise the system one is loaded.
Rccl &rccl() {
  Rccl &r = C().rccl;
  if (r.lib) return r;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *lib = nullptr;
  for (const char *n : names)
    if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)) != nullptr) break;
  for (const char *n : names) {
    if (lib) break;
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!lib) throw std::runtime_error(std::string("RCCL: cannot load librccl.so: ") + dlerror());
  load(lib, "ncclGetUniqueId", &r.getUniqueId);
  load(lib, "ncclCommInitRank", &r.commInitRank);
  load(lib, "ncclCommDestroy", &r.commDestroy);
  load(lib, "ncclAllReduce", &r.allReduce);
  load(lib, "ncclBroadcast", &r.broadcast);
  load(lib, "ncclGetErrorString", &r.getErrorString);
  r.lib = lib;
  return r;
}

void nccl_check(ncclResult_t e, const char *what) {
  if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + C().rccl.getErrorString(e));
}

int rccl_allreduce(float *dev, uint64_t count, void *stream, void *) {
  State &s = C();
  return s.rccl.allReduce(dev, dev, count, ncclFloat, ncclSum, s.comm, static_cast<hipStream_t>(stream)) == ncclSuccess
             ? 0
             : 1;
}

int rccl_broadcast(float *dev, uint64_t count, int root, void *stream, void *) {
  State &s = C();
  return s.rccl.broadcast(dev, dev, count, ncclFloat, root, s.comm, static_cast<hipStream_t>(stream)) == ncclSuccess
             ? 0
             : 1;
}

void activate(int rank, int world) {
  State &s = C();
  if (rank < 0 || world < 1 || rank >= world) throw std::invalid_argument("comm_init: rank/world out of range");
  s.rank = rank;
  s.world = world;
  if (s.stream_ordered && !s.side)
    hip_check(hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking), "hipStreamCreate(side stream)");
  // every rank draws different dropout masks for its different samples
  set_dropout_rank(rank);
  s.active = true;
}

const char *env_or(const char *name, const char *fallback) {
  const char *v = std::getenv(name);
  return (v && *v) ? v : fallback;
}

}  // namespace

void comm_unique_id(void *id128) {
  ncclUniqueId id;
  nccl_check(rccl().getUniqueId(&id), "ncclGetUniqueId");
  static_assert(sizeof(id) == kCommIdBytes, "unexpected ncclUniqueId size");
  std::memcpy(id128, &id, sizeof id);
}

void comm_init_rccl(int rank, int world, const void *id128) {
  State &s = C();
  if (s.active) throw std::runtime_error("comm_init: a communicator already exists (call comm_destroy first)");
  if (!id128) throw std::invalid_argument("comm_init_rccl: null id");
  (void)stream();  // selects the device
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  nccl_check(rccl().commInitRank(&s.comm, world, id, rank), "ncclCommInitRank");
  s.ar = rccl_allreduce;
  s.bc = rccl_broadcast;
  s.user = nullptr;
  s.stream_ordered = true;
  activate(rank, world);
}

void comm_init_rccl_file(int rank, int world, const std::string &path, int timeout_s) {
  char id[kCommIdBytes];
  if (rank == 0) {
    comm_unique_id(id);
    const std::string tmp = path + ".tmp";
    {
      std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
      out.write(id, sizeof id);
      if (!out) throw std::runtime_error("comm_init_rccl_file: cannot write " + tmp);
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0)
      throw std::runtime_error("comm_init_rccl_file: cannot rename " + tmp + " to " + path);
  } else {
    // wait for a file that rank 0 of THIS launch wrote: one left behind by an earlier run that
    // died before joining is older than any launcher takes to start its ranks
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      struct stat st;
      if (stat(path.c_str(), &st) == 0 && st.st_size == (off_t)sizeof id && std::time(nullptr) - st.st_mtime < 600) {
        std::ifstream in(path, std::ios::binary);
        in.read(id, sizeof id);
        if (in) break;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s))
        throw std::runtime_error("comm_init_rccl_file: timed out waiting for " + path);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  comm_init_rccl(rank, world, id);
  if (rank == 0) std::remove(path.c_str());  // ncclCommInitRank returns once every rank has joined
}

bool comm_init_from_env() {
  if (C().active) return false;
  const int world = std::atoi(env_or("WORLD_SIZE", "1"));
  if (world <= 1) return false;
  const int rank = std::atoi(env_or("RANK", "0"));
  const int local = std::atoi(env_or("LOCAL_RANK", env_or("RANK", "0")));
  init(local);
  std::string path = env_or("VIT_COMM_ID_FILE", "");
  if (path.empty())
    path = std::string("/tmp/vit_comm_") + env_or("MASTER_ADDR", "local") + "_" + env_or("MASTER_PORT", "0") + ".id";
  comm_init_rccl_file(rank, world, path);
  return true;
}

void comm_init_external(int rank, int world, AllReduceFn allreduce, BroadcastFn broadcast, void *user,
                        bool stream_ordered) {
  State &s = C();
  if (s.active) throw std::runtime_error("comm_init: a communicator already exists (call comm_destroy first)");
  if (!allreduce || !broadcast) throw std::invalid_argument("comm_init_external: null transport function");
  (void)stream();
  s.ar = allreduce;
  s.bc = broadcast;
  s.user = user;
  s.stream_ordered = stream_ordered;
  activate(rank, world);
}

void comm_destroy() {
  State &s = C();
  if (!s.active) return;
  synchronize();
  if (s.side) {
    (void)hipStreamSynchronize(s.side);
    (void)hipStreamDestroy(s.side);
    s.side = nullptr;
  }
  if (s.comm) {
    (void)s.rccl.commDestroy(s.comm);
    s.comm = nullptr;
  }
  s.ar = nullptr;
  s.bc = nullptr;
  s.user = nullptr;
  s.active = false;
  s.rank = 0;
  s.world = 1;
  set_dropout_rank(0);
}

bool comm_active() { return C().active; }
int comm_rank() { return C().rank; }
int comm_world() { return C().world; }

void comm_broadcast(float *dev, size_t count, int root) {
  State &s = C();
  if (!s.active || count == 0) return;
  if (s.bc(dev, count, root, stream(), s.user)) throw std::runtime_error("comm_broadcast: transport failure");
}

void comm_allreduce(float *dev, size_t count) {
  State &s = C();
  if (!s.active || count == 0) return;
  if (s.ar(dev, count, stream(), s.user)) throw std::runtime_error("comm_allreduce: transport failure");
}

// ------------------------------------------------------------------------- GradientReducer
std::vector<GradientReducer::Bucket> GradientReducer::planBuckets(
    const std::vector<std::pair<size_t, size_t>> &stage_ranges, size_t bucket_bytes) {
  std::vector<Bucket> out;
  bool open = false;
  Bucket cur{0, 0, 0};
  for (size_t s = 0; s < stage_ranges.size(); ++s) {
    const size_t off = stage_ranges[s].first, cnt = stage_ranges[s].second;
    if (cnt == 0) continue;
    if (!open) {
      cur = Bucket{s, off, cnt};
      open = true;
    } else {
      if (off + cnt != cur.offset)
        throw std::runtime_error("GradientReducer: stage gradient slices must be adjacent and descending");
      cur = Bucket{s, off, cur.count + cnt};
    }
    if (cur.count * sizeof(float) >= bucket_bytes) {
      out.push_back(cur);
      open = false;
    }
  }
  if (open) out.push_back(cur);
  return out;
}

GradientReducer::GradientReducer(VisionTransformer &model, size_t bucket_bytes) : model(model) {
  std::vector<std::pair<size_t, size_t>> ranges(model.numBackwardStages());
  for (size_t s = 0; s < ranges.size(); ++s) model.stageGradientRange(s, &ranges[s].first, &ranges[s].second);
  plan = planBuckets(ranges, bucket_bytes);
  launched.assign(plan.size(), false);
  for (size_t i = 0; i < plan.size(); ++i) {
    hipEvent_t a = nullptr, b = nullptr;
    hip_check(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreate");
    ready.push_back(a);
    done.push_back(b);
  }
}

GradientReducer::~GradientReducer() {
  for (void *e : ready) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  for (void *e : done) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
}

void GradientReducer::afterStage(size_t stage) {
  State &s = C();
  if (!s.active) return;
  for (size_t i = 0; i < plan.size(); ++i) {
    if (plan[i].last_stage != stage) continue;
    if (launched[i]) throw std::runtime_error("GradientReducer: stage closed twice in one step (finish() missing?)");
    // dev_write(): the transport writes device memory behind the storage's back, so the host
    // mirror of the gradient arena is stale from here on
    float *g = model.arena().grads()->dev_write() + plan[i].offset;
    hipStream_t compute = static_cast<hipStream_t>(stream());
    if (s.stream_ordered) {
      hip_check(hipEventRecord(static_cast<hipEvent_t>(ready[i]), compute), "hipEventRecord(bucket ready)");
      hip_check(hipStreamWaitEvent(s.side, static_cast<hipEvent_t>(ready[i]), 0), "hipStreamWaitEvent(side)");
      if (s.ar(g, plan[i].count, s.side, s.user)) throw std::runtime_error("GradientReducer: all-reduce failed");
      hip_check(hipEventRecord(static_cast<hipEvent_t>(done[i]), s.side), "hipEventRecord(bucket done)");
    } else {
      if (s.ar(g, plan[i].count, compute, s.user)) throw std::runtime_error("GradientReducer: all-reduce failed");
    }
    launched[i] = true;
  }
}

void GradientReducer::finish() {
  State &s = C();
  if (!s.active) return;
  hipStream_t compute = static_cast<hipStream_t>(stream());
  for (size_t i = 0; i < plan.size(); ++i) {
    if (!launched[i]) throw std::runtime_error("GradientReducer::finish: a bucket was never reduced (stage skipped)");
    if (s.stream_ordered)
      hip_check(hipStreamWaitEvent(compute, static_cast<hipEvent_t>(done[i]), 0), "hipStreamWaitEvent(compute)");
    launched[i] = false;
  }
}

}  // namespace vit
