// Device runtime of the host layer: one HIP stream, a caching device allocator and a grow-only
// scratch workspace.  Everything the Tensor / Layer classes enqueue goes through here, so a
// steady-state training step performs no hipMalloc/hipFree.
#ifndef VIT_DEVICE_HPP
#define VIT_DEVICE_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

namespace vit {

// Throws std::runtime_error(vitk_last_error()) when a libvitk call failed (the reference's error
// convention is C++ exceptions caught once in main, app/main.cpp:76-79).
void check(int rc, const char *what);
void hip_check(int hip_error, const char *what);

// Select the device (default 0) -- called implicitly on first use.  Selecting another device
// later is refused (std::runtime_error) once this process holds device memory: tensors, the
// workspace and cached blocks stay on the device they were allocated on.
void init(int device = 0);
int current_device();  // -1 before the first use
// Use an existing hipStream_t (e.g. the caller framework's current stream) instead of the
// library's own; nullptr selects the default (null) stream.
void set_stream(void *hip_stream);
void *stream();
void synchronize();

// Caching allocator: freed blocks are kept per size and reused; safe because every consumer is
// ordered on the one stream above.
float *device_alloc(size_t count);
void device_free(float *p, size_t count);
void release_cached_memory();
size_t device_bytes_in_use();

// Grow-only scratch for split-K partials, column-sum and LayerNorm partials.
float *workspace(size_t bytes);
size_t workspace_bytes();

void copy_to_device(float *dst, const float *src, size_t count);
void copy_to_host(float *dst, const float *src, size_t count);  // synchronises
void copy_on_device(float *dst, const float *src, size_t count);
void zero_device(float *dst, size_t count);

// Reproducible host RNG for weight init / shuffles (the reference seeds from random_device; call
// seed() to make runs repeatable).
void seed(uint64_t s);
uint64_t next_seed();
// (seed, offset) pair for the next dropout mask of `count` elements.
void dropout_stream(size_t count, uint64_t *seed_out, uint64_t *offset_out);
// Position of the dropout stream (part of a resumable training state).
void get_dropout_state(uint64_t *seed_out, uint64_t *offset_out);
void set_dropout_state(uint64_t seed, uint64_t offset);
// Data parallelism: rank r draws from the stream keyed by seed + r * (an odd constant), so that the
// ranks apply different masks to their different samples.  The seed reported by
// get_dropout_state and passed to the kernels is that effective key.
void set_dropout_rank(int rank);
// The same position with the UNSALTED seed: what a training state file holds, so that every rank
// that loads it re-applies its own salt and keeps drawing masks of its own.
void get_dropout_base_state(uint64_t *seed_out, uint64_t *offset_out);
void set_dropout_base_state(uint64_t seed, uint64_t offset);
// Replay of recorded dropout draws: while a buffer is set, every training-mode Dropout forward
// takes its `count` factors (0 or 1/(1-rate)) from it, in call order, instead of the device
// stream -- this repeats a run whose masks were recorded elsewhere (e.g. by the reference, whose
// masks come from random_device).  nullptr / 0 ends the replay.  The buffer must stay alive until
// the backward pass that uses it has run.
void set_dropout_replay(const float *masks_dev, size_t count);
// Next `count` recorded factors, or nullptr when no replay is active; throws when the buffer runs out.
const float *dropout_replay_take(size_t count);

}  // namespace vit

#endif
