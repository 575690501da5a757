// Communicator and bucketed gradient reducer (see core/Comm.hpp).
#include "core/Comm.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // declarations only: the library is loaded with dlopen
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <random>
#include <thread>
#include <vector>

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
  // exposed-exchange timing (comm_timing): per step one event pair around the tail of the exchange
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> samples;
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
  // (A high-priority side stream was tried and dropped: with hipStreamCreateWithPriority at the
  // device's greatest priority the event chain compute -> side -> compute of a one-rank run never
  // completed on ROCm 7.0 -- the step after the first all-reduce hung.)
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

// ---- id exchange through the file system ----------------------------------------------------------
// Files are created under a temporary name with O_EXCL | O_NOFOLLOW and mode 0600, then renamed;
// readers refuse symbolic links and files of another user.
namespace {

constexpr char kIdMagic[8] = {'V', 'I', 'T', 'C', 'O', 'M', 'M', '2'};

void write_file_atomically(const std::string &path, const void *data, size_t bytes) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  (void)unlink(tmp.c_str());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) throw std::runtime_error("comm id exchange: cannot create " + tmp + ": " + std::strerror(errno));
  const ssize_t w = write(fd, data, bytes);
  const int rc = close(fd);
  if (w != (ssize_t)bytes || rc != 0 || std::rename(tmp.c_str(), path.c_str()) != 0) {
    (void)unlink(tmp.c_str());
    throw std::runtime_error("comm id exchange: cannot write " + path);
  }
}

// Reads exactly `bytes` bytes of a regular file owned by this user; false when it is not (yet) there.
bool read_own_file(const std::string &path, void *data, size_t bytes) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && st.st_size == (off_t)bytes;
  if (ok) ok = read(fd, data, bytes) == (ssize_t)bytes;
  close(fd);
  return ok;
}

struct Published {  // what rank 0 writes to `path`
  char magic[8];
  uint64_t world;
  char id[kCommIdBytes];
  // followed by world x uint64 nonces (slot 0: rank 0's own)
};

}  // namespace

// Every rank draws a fresh random nonce.  Rank r > 0 announces its own in <path>.req<r>; rank 0
// publishes {id, its nonce, the nonces it has seen} in <path> and publishes again whenever an
// announcement changes; a rank accepts a published file only if it carries ITS nonce of THIS
// process, and then confirms with {its nonce, rank 0's nonce} in <path>.ack<r>; rank 0 returns once
// every rank has confirmed the nonces it published last.  Files that an earlier launch left behind
// (any of the three kinds) carry other nonces and are ignored or overwritten, whenever they were
// written.
void comm_exchange_id_file(int rank, int world, const std::string &path, int timeout_s, void *id128) {
  if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("comm_exchange_id_file: rank/world out of range");
  if (world == 1) return;
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s); };
  auto nap = [] { std::this_thread::sleep_for(std::chrono::milliseconds(10)); };
  const size_t pub_bytes = sizeof(Published) + (size_t)world * sizeof(uint64_t);
  std::vector<char> buf(pub_bytes);
  auto *pub = reinterpret_cast<Published *>(buf.data());
  auto *nonces = reinterpret_cast<uint64_t *>(buf.data() + sizeof(Published));
  std::random_device rd;
  uint64_t nonce = 0;
  while (nonce == 0) nonce = ((uint64_t)rd() << 32) ^ rd() ^ ((uint64_t)getpid() << 17);
  if (rank != 0) {
    const std::string req = path + ".req" + std::to_string(rank), ack = path + ".ack" + std::to_string(rank);
    (void)unlink(ack.c_str());
    write_file_atomically(req, &nonce, sizeof nonce);
    for (;;) {
      if (read_own_file(path, buf.data(), pub_bytes) && std::memcmp(pub->magic, kIdMagic, 8) == 0 &&
          pub->world == (uint64_t)world && nonces[rank] == nonce)
        break;
      if (expired()) {
        (void)unlink(req.c_str());
        throw std::runtime_error("comm_exchange_id_file: timed out waiting for rank 0 to publish " + path);
      }
      nap();
    }
    std::memcpy(id128, pub->id, kCommIdBytes);
    const uint64_t confirm[2] = {nonce, nonces[0]};
    write_file_atomically(ack, confirm, sizeof confirm);
    return;
  }
  std::vector<uint64_t> seen(world, 0), published(world, 0);
  seen[0] = nonce;
  bool have_published = false;
  (void)unlink(path.c_str());
  for (;;) {
    bool all = true;
    for (int r = 1; r < world; ++r) {
      uint64_t n = 0;
      if (read_own_file(path + ".req" + std::to_string(r), &n, sizeof n) && n != 0) seen[r] = n;
      all = all && seen[r] != 0;
    }
    if (all && (!have_published || seen != published)) {
      std::memcpy(pub->magic, kIdMagic, 8);
      pub->world = (uint64_t)world;
      std::memcpy(pub->id, id128, kCommIdBytes);
      for (int r = 0; r < world; ++r) nonces[r] = seen[r];
      write_file_atomically(path, buf.data(), pub_bytes);
      published = seen;
      have_published = true;
    }
    if (have_published) {
      bool confirmed = true;
      for (int r = 1; r < world && confirmed; ++r) {
        uint64_t c[2] = {0, 0};
        confirmed = read_own_file(path + ".ack" + std::to_string(r), c, sizeof c) && c[0] == published[r] && c[1] == nonce;
      }
      if (confirmed) break;
    }
    if (expired()) {
      (void)unlink(path.c_str());
      throw std::runtime_error("comm_exchange_id_file: timed out waiting for the other ranks at " + path);
    }
    nap();
  }
  (void)unlink(path.c_str());
  for (int r = 1; r < world; ++r) {
    (void)unlink((path + ".req" + std::to_string(r)).c_str());
    (void)unlink((path + ".ack" + std::to_string(r)).c_str());
  }
}

void comm_init_rccl_file(int rank, int world, const std::string &path, int timeout_s) {
  char id[kCommIdBytes] = {0};
  if (rank == 0) comm_unique_id(id);
  comm_exchange_id_file(rank, world, path, timeout_s, id);
  comm_init_rccl(rank, world, id);
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
    path = std::string("/tmp/vit_comm_") + std::to_string((long)geteuid()) + "_" + env_or("MASTER_ADDR", "local") + "_" +
           env_or("MASTER_PORT", "0") + "_" + env_or("VIT_COMM_RUN_ID", env_or("TORCHELASTIC_RUN_ID", "run")) + ".id";
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
  for (auto &p : s.samples) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  s.samples.clear();
  s.timing = false;
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

void comm_timing(bool on) {
  State &s = C();
  if (on && !(s.active && s.stream_ordered))
    throw std::runtime_error("comm_timing: needs a stream-ordered communicator");
  s.timing = on;
}

size_t comm_exposed_ms(double *total_ms) {
  State &s = C();
  synchronize();
  if (s.side) hip_check(hipStreamSynchronize(s.side), "hipStreamSynchronize(side)");
  double sum = 0.0;
  for (auto &p : s.samples) {
    float ms = 0.0f;
    hip_check(hipEventElapsedTime(&ms, p.first, p.second), "hipEventElapsedTime(exchange tail)");
    sum += ms;
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  const size_t n = s.samples.size();
  s.samples.clear();
  if (total_ms) *total_ms = sum;
  return n;
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
      // the step's last bucket closes with the last backward kernel: from there to the end of its
      // all-reduce (the side stream runs the buckets in order) the compute stream has nothing to
      // run -- the part of the exchange that backward does not hide
      const bool sample = s.timing && i + 1 == plan.size();
      hipEvent_t t_ready = nullptr, t_done = nullptr;
      if (sample) {
        hip_check(hipEventCreate(&t_ready), "hipEventCreate");
        hip_check(hipEventCreate(&t_done), "hipEventCreate");
        hip_check(hipEventRecord(t_ready, compute), "hipEventRecord(exchange tail begin)");
      }
      hip_check(hipEventRecord(static_cast<hipEvent_t>(ready[i]), compute), "hipEventRecord(bucket ready)");
      hip_check(hipStreamWaitEvent(s.side, static_cast<hipEvent_t>(ready[i]), 0), "hipStreamWaitEvent(side)");
      if (s.ar(g, plan[i].count, s.side, s.user)) throw std::runtime_error("GradientReducer: all-reduce failed");
      hip_check(hipEventRecord(static_cast<hipEvent_t>(done[i]), s.side), "hipEventRecord(bucket done)");
      if (sample) {
        hip_check(hipEventRecord(t_done, s.side), "hipEventRecord(exchange tail end)");
        s.samples.emplace_back(t_ready, t_done);
      }
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
