// Data-parallel gradient exchange: one process per GPU, one communicator per process.
//
// The reference has no distributed code (single process, single device); this is the exchange
// step added around its per-batch training body (Trainer::train_epoch, src/model/Trainer.cpp:
// 163-236).  Samples are independent through forward/backward, so the only cross-rank data are the
// parameter gradients: every rank divides its loss gradient by the GLOBAL batch and the SUM of the
// ranks' gradients is then exactly what one device computes on the whole batch.
//
// Transport: RCCL (ncclAllReduce over xGMI), loaded with dlopen when a communicator is created so
// that single-GPU users never load it; or a caller-provided function (tests run two ranks on one
// GPU, which RCCL refuses, through host memory; other frameworks can plug their process group).
//
// Overlap: backward runs in stages (VisionTransformer::backwardStage), each of which finalises one
// contiguous slice of the flat gradient arena.  GradientReducer merges consecutive slices into
// buckets of at least bucket_bytes; when the stage that closes a bucket has been enqueued it
// records an event on the compute stream, lets a library-owned side stream wait for it and
// enqueues the bucket's all-reduce there, so the collective runs while the remaining backward
// kernels execute.  Before the optimizer the compute stream waits for every bucket's completion
// event.  Nothing here blocks the host.
#ifndef VIT_COMM_HPP
#define VIT_COMM_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

class VisionTransformer;

namespace vit {

constexpr size_t kCommIdBytes = 128;  // size of an RCCL unique id

// In-place sum over all ranks of `count` floats at `dev`.  A stream-ordered transport enqueues the
// operation on `stream` and returns; a host-synchronous one must itself wait for `stream` before it
// reads and must have the result in place (visible to work enqueued on `stream` afterwards) when it
// returns.  Non-zero return = failure.
typedef int (*AllReduceFn)(float *dev, uint64_t count, void *stream, void *user);
// In-place broadcast of `count` floats from rank `root`; same conventions.
typedef int (*BroadcastFn)(float *dev, uint64_t count, int root, void *stream, void *user);

// Creates the id that every rank of a new RCCL communicator must pass to comm_init_rccl (rank 0
// makes it and hands it to the others by whatever channel the launcher offers).
void comm_unique_id(void *id128);
// Collective over all ranks: joins the RCCL communicator on the current device.
void comm_init_rccl(int rank, int world, const void *id128);
// The same with the id exchanged through files next to `path` (see comm_exchange_id_file); every
// rank waits at most timeout_s seconds for the others.
void comm_init_rccl_file(int rank, int world, const std::string &path, int timeout_s = 120);
// The exchange on its own (pure host code, no RCCL): rank 0 passes the id in id128, the others
// receive it there.  Ranks confirm each other with per-process random nonces, so files that an
// earlier, failed launch left at the same path are never taken for this launch's; all files are
// created exclusively with mode 0600 and symbolic links or other users' files are refused.
void comm_exchange_id_file(int rank, int world, const std::string &path, int timeout_s, void *id128);
// Reads RANK / WORLD_SIZE / LOCAL_RANK (the variables torchrun and most launchers set); with
// WORLD_SIZE > 1 selects device LOCAL_RANK and joins through the file VIT_COMM_ID_FILE (default
// /tmp/vit_comm_<uid>_<MASTER_ADDR>_<MASTER_PORT>_<VIT_COMM_RUN_ID or TORCHELASTIC_RUN_ID>.id).
// Must run before this process creates its first tensor (the device cannot change afterwards:
// vit::init throws).  Returns false (and does nothing) when WORLD_SIZE is unset or 1, or a
// communicator already exists.
bool comm_init_from_env();
// Transport provided by the caller.
void comm_init_external(int rank, int world, AllReduceFn allreduce, BroadcastFn broadcast, void *user,
                        bool stream_ordered);
void comm_destroy();
bool comm_active();
int comm_rank();   // 0 without a communicator
int comm_world();  // 1 without a communicator

// Raw collectives on the compute stream (parameters at start-up, scalar metrics).
void comm_broadcast(float *dev, size_t count, int root);
void comm_allreduce(float *dev, size_t count);

// Measurement of the part of the gradient exchange that backward does not hide.  While on, every
// step records one HIP-event pair: on the compute stream when the step's last bucket closes (the
// last backward kernel has been enqueued) and on the side stream behind that bucket's all-reduce.
// comm_exposed_ms waits for both streams, adds up the elapsed times of the pairs recorded since the
// last call into *total_ms and returns their number.  Stream-ordered transports only.
void comm_timing(bool on);
size_t comm_exposed_ms(double *total_ms);

// Bucketed, overlapped all-reduce of one model's gradient arena (see the header comment).
class GradientReducer {
public:
  explicit GradientReducer(VisionTransformer &model, size_t bucket_bytes = 25u << 20);
  ~GradientReducer();
  GradientReducer(const GradientReducer &) = delete;
  GradientReducer &operator=(const GradientReducer &) = delete;

  // call after VisionTransformer::backwardStage(stage) has been enqueued
  void afterStage(size_t stage);
  // makes the compute stream wait for every bucket of this step; call before the optimizer
  void finish();

  struct Bucket {
    size_t last_stage, offset, count;
  };
  const std::vector<Bucket> &buckets() const { return plan; }
  // The merge rule on its own: stage slices in the order backward finalises them (adjacent,
  // descending in the arena) -> buckets closed once they hold at least bucket_bytes.
  static std::vector<Bucket> planBuckets(const std::vector<std::pair<size_t, size_t>> &stage_ranges,
                                         size_t bucket_bytes);

private:
  VisionTransformer &model;
  std::vector<Bucket> plan;
  std::vector<void *> ready, done;  // hipEvent_t per bucket
  std::vector<bool> launched;
};

}  // namespace vit

#endif
