/*
 * vit_host.h -- C-ABI of libvit_host.so: the C++ host layer (Tensor / Layer / Loss / Optimizer /
 * Trainer / ModelUtils classes with the reference's API, visiontransformer_amd/host/) driven
 * through opaque handles, for callers that are not C++ (tests, bench.py, the data-parallel
 * driver).  A C++ application uses the classes directly, exactly as app/main.cpp of the
 * reference does.
 *
 * Every function returns 0 on success (or a value, where documented) and non-zero on failure;
 * vith_last_error() returns the message of the calling thread's last failure (the C++
 * exception text).  Pointers named *_dev are device pointers, *_host host pointers.  All work is
 * enqueued on the stream set with vith_set_stream (default: the null stream); functions that
 * return values to the host synchronise that stream.
 */
#ifndef VIT_HOST_H
#define VIT_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ViTConfig (include/model/VisionTransformer.hpp:19-29 of the reference), field for field. */
typedef struct vith_config {
  uint64_t image_size, patch_size, in_channels, num_classes, embedding_dim, num_heads, num_layers, mlp_hidden_dim;
  float dropout_rate;
} vith_config;

typedef void *vith_model;   /* VisionTransformer */
typedef void *vith_trainer; /* CrossEntropy + Adam + schedule state around one model */

const char *vith_last_error(void);
int vith_init(int device);
int vith_set_stream(void *hip_stream);
int vith_synchronize(void);
/* seeds weight initialisation, data subsampling and dropout (the reference uses random_device) */
int vith_seed(uint64_t seed);
/* ---- data parallelism: one process per GPU, gradients all-reduced with RCCL over xGMI -----------
 * The reference has none (single process, single device); this is the exchange step around its
 * per-batch body (Trainer::train_epoch, src/model/Trainer.cpp:163-236).  While a communicator
 * exists:
 *   - vith_step_backward_stage(s) also enqueues, on a library-owned side stream gated by events,
 *     one ncclAllReduce(float, sum) for every gradient bucket that stage s closes (consecutive
 *     stage slices of the gradient arena merged until they hold >= bucket_bytes, default 25 MiB),
 *     so the exchange overlaps the remaining backward kernels;
 *   - vith_step_update first makes the compute stream wait for all buckets, then runs Adam;
 *   - vith_step_forward's grad_batch must be the GLOBAL batch, so that the sum is the single-device
 *     gradient; vith_fit shards every global batch of world x batch_size samples by rank;
 *   - every rank draws its own dropout masks.
 * vith_comm_unique_id: rank 0 creates the 128-byte id and passes it to the other ranks by any
 * channel; vith_comm_init (collective) joins.  vith_comm_init_file exchanges the id through a file
 * instead (rank 0 writes it, the others wait up to timeout_s seconds; 0 = 120).  A C++ application
 * needs none of these calls: Trainer joins by itself when the launcher has set WORLD_SIZE > 1
 * (RANK, LOCAL_RANK, optional VIT_COMM_ID_FILE).
 * vith_comm_init_external: the transport is the caller's (a test harness, another framework's
 * process group).  Both functions work in place on device memory; a stream-ordered transport
 * (stream_ordered = 1) enqueues on `stream` and returns, a host-synchronous one (0) must wait for
 * `stream` before reading and have the result in place when it returns. */
typedef int (*vith_allreduce_fn)(float *dev, uint64_t count, void *stream, void *user);
typedef int (*vith_broadcast_fn)(float *dev, uint64_t count, int root, void *stream, void *user);
int vith_comm_unique_id(void *id128);
int vith_comm_init(int rank, int world, const void *id128);
int vith_comm_init_file(int rank, int world, const char *path, int timeout_s);
int vith_comm_init_external(int rank, int world, vith_allreduce_fn allreduce, vith_broadcast_fn broadcast, void *user,
                            int stream_ordered);
/* The file exchange of vith_comm_init_file on its own (host code only, no RCCL, no GPU): rank 0
 * passes the 128-byte id in id128, the other ranks receive it there.  The ranks confirm each other
 * with per-process random nonces (files <path>, <path>.req<r>, <path>.ack<r>, all created
 * exclusively with mode 0600 and removed by rank 0 at the end), so whatever an earlier, failed
 * launch left at the same path is never taken for this launch's id. */
int vith_comm_exchange_id_file(int rank, int world, const char *path, int timeout_s, void *id128);
/* Measurement of the part of the exchange that backward does not hide: while on, every step records
 * a HIP-event pair -- compute stream, when the last bucket closes (= behind the last backward
 * kernel); side stream, behind that bucket's all-reduce.  vith_comm_exposed_ms synchronises, writes
 * the sum of the pairs' elapsed times since the last call and returns their number (-1 on error). */
int vith_comm_timing(int on);
int64_t vith_comm_exposed_ms(double *total_ms);
int vith_comm_destroy(void);
/* The bucket rule on its own (pure host code): stage s finalises [stage_offset[s], +stage_count[s])
 * of the gradient arena, slices adjacent and descending; out3 receives {last stage, offset, count}
 * per bucket.  Returns the number of buckets, -1 on error. */
int64_t vith_plan_buckets(const uint64_t *stage_offset, const uint64_t *stage_count, uint64_t stages,
                          uint64_t bucket_bytes, uint64_t *out3, uint64_t out_capacity);
int vith_comm_rank(void);
int vith_comm_world(void);

/* Dropout draws (src/layers/Dropout.cpp:42-51).  The reference seeds each mask from random_device;
 * here the factors of a Dropout forward over n elements are a pure function of (seed, offset,
 * element index) -- the mask vitk_dropout_fwd produces for the same (rate, seed, offset) -- and the
 * forward then advances offset by ceil(n/4).  vith_dropout_state reads the position the NEXT
 * dropout forward will use; vith_set_dropout_state moves it (vith_seed resets it).
 * vith_dropout_replay: while set, every training-mode Dropout forward takes its n factors
 * (0 or 1/(1-rate)) from masks_dev in call order instead -- per encoder block the attention-branch
 * mask, then the feed-forward mask (src/model/TransformerEncoderBlock.cpp:23, src/layers/
 * FeedForward.cpp:22) -- which repeats a run whose masks were recorded from the reference.
 * count = floats available; NULL / 0 ends the replay; the buffer must outlive the backward pass. */
int vith_dropout_state(uint64_t *seed, uint64_t *offset);
int vith_set_dropout_state(uint64_t seed, uint64_t offset);
int vith_dropout_replay(const float *masks_dev, uint64_t count);
/* Attention path of layers / models created AFTERWARDS: 1 (default) = the fused kernels
 * (vitk_sdpa_fwd / vitk_sdpa_bwd) where the shape is supported, 0 = batched contractions + row
 * softmax with the probabilities materialised (the reference's own sequence of operations,
 * src/layers/MultiHeadAttention.cpp:96-156). */
int vith_set_fused_attention(int on);
/* The classification head reads only the CLS token of the last encoder block's output
 * (src/model/VisionTransformer.cpp:27-37) and re-injects a gradient that is zero for every other
 * token (:49-56).  1 (default): the last block's output projection, residuals, second norm and
 * feed-forward, the final norm, and their backward run on that one token per sequence (its
 * attention core on one query row; keys / values still come from all tokens) -- same logits, loss,
 * parameter gradients and updates, to rounding.  0: every token is computed, as the reference
 * does.  Applies to forwards from then on; tracing forwards (vith_forward_trace) are always dense. */
int vith_set_first_token_tail(int on);
uint64_t vith_device_bytes_in_use(void);

/* ---- model: VisionTransformer (src/model/VisionTransformer.cpp of the reference) ------------ */
/* patch_mode: VITK_PATCH_REFERENCE (0) / VITK_PATCH_IM2COL (1); dk_mode: VITK_DK_REFERENCE (0) /
 * VITK_DK_EXACT (1) -- see include/vitk.h. */
vith_model vith_model_create(const vith_config *cfg, int patch_mode, int dk_mode);
void vith_model_destroy(vith_model m);
uint64_t vith_param_count(vith_model m);
uint64_t vith_param_tensors(vith_model m);
/* shape of parameter tensor idx in getParameters() order; dims holds 4 entries; returns the rank */
uint64_t vith_param_shape(vith_model m, uint64_t idx, uint64_t *dims);
/* flat copies in getParameters()/getGradients() order, every tensor contiguous (= the payload
 * order of a .weights file) */
int vith_get_params(vith_model m, float *flat_host);
int vith_set_params(vith_model m, const float *flat_host);
int vith_get_grads(vith_model m, float *flat_host);
/* device arenas (device layout: q/k/v weights of a layer packed as one {D,3D} block); both hold
 * vith_param_count() floats; for collectives and optimiser kernels, which are layout-agnostic */
float *vith_params_device(vith_model m);
float *vith_grads_device(vith_model m);
/* ModelUtils::{save,load}_weights / {save,load}_config (src/utils/ModelUtils.cpp) */
int vith_save_weights(vith_model m, const char *path);
int vith_load_weights(vith_model m, const char *path);
int vith_save_config(const vith_config *cfg, const char *path);
int vith_load_config(const char *path, vith_config *cfg);

/* forward: x_dev {B,C,H,W} -> logits_dev {B,K} (VisionTransformer::forward) */
int vith_forward(vith_model m, const float *x_dev, uint64_t B, int training, float *logits_dev);
/* forward in training mode that also copies the token stream after the embeddings, after every
 * block and after the final norm into acts_dev ((L+2) x {B,N,D}) */
int vith_forward_trace(vith_model m, const float *x_dev, uint64_t B, float *acts_dev, float *logits_dev);
/* backward from dlogits_dev {B,K} (VisionTransformer::backward); dacts_dev (optional) receives
 * (L+1) x {B,N,D}: gradient w.r.t. the final-norm input, then each block's input, last block first */
int vith_backward(vith_model m, const float *dlogits_dev, uint64_t B, float *dacts_dev);
/* the same backward, returning what VisionTransformer::backward returns: the image gradient
 * {B,C,H,W} (src/model/VisionTransformer.cpp:68; in the reference patch mode every patch carries
 * patch 0 of image 0, src/layers/PatchEmbedding.cpp:86-94; training steps skip this gradient) */
int vith_backward_image(vith_model m, const float *dlogits_dev, uint64_t B, float *dimage_dev);
/* top-1: first index of the row maximum of softmax(logits) (app/test.cpp:92-112) */
int vith_predict(vith_model m, const float *x_dev, uint64_t B, int32_t *pred_dev);

/* ---- whole training run: Trainer::train (src/model/Trainer.cpp:60-236) on host data -----------
 * x is [n][C][H][W], y one-hot [n][K] (x_test / y_test likewise).  Per epoch: sample order =
 * iota shuffled by mt19937(epoch), batches of batch_size (last one ragged), cosine/warm-up
 * schedule over epochs * batches steps, Adam(L2), then evaluation on the test set in batches.
 * The data set is uploaded once and batches are gathered on the device.  history (may be NULL)
 * receives {train loss, train accuracy, test loss, test accuracy} per epoch.  Writes the
 * reference's vit_results.csv log into the current directory.  No augmentation (see below). */
int vith_fit(vith_model m, const float *x, const float *y_onehot, uint64_t n, const float *x_test,
             const float *y_test_onehot, uint64_t n_test, int epochs, uint64_t batch_size, float lr_init,
             float weight_decay, float warmup_frac, float *history);

/* ---- DataAugmentation (include/utils/DataAugmentation.hpp:8-34, src/utils/DataAugmentation.cpp) --
 * Config field for field (:11-19).  vith_augment = DataAugmentation(cfg).apply(batch) with the
 * engine seeded by `seed` (the reference seeds it from random_device): x_dev, out_dev {B,C,H,W}.
 * vith_fit_augmented = vith_fit with every training batch augmented as the reference's
 * train_epoch does (src/model/Trainer.cpp:156-161, 198-201): one engine per epoch, seeded with
 * augment_seed + epoch, drawing for the epoch's images in batch order; augment == NULL = vith_fit. */
typedef struct vith_augment_config {
  float rotation_prob, translate_prob, zoom_prob, rotation_factor, translate_factor, zoom_min, zoom_max;
} vith_augment_config;
int vith_augment(const vith_augment_config *cfg, uint32_t seed, const float *x_dev, uint64_t B, uint64_t C, uint64_t H,
                 uint64_t W, float *out_dev);
int vith_fit_augmented(vith_model m, const float *x, const float *y_onehot, uint64_t n, const float *x_test,
                       const float *y_test_onehot, uint64_t n_test, int epochs, uint64_t batch_size, float lr_init,
                       float weight_decay, float warmup_frac, const vith_augment_config *augment, uint32_t augment_seed,
                       float *history);

/* Trainer::train with a TrainerConfig as app/main.cpp builds it (app/main.cpp:21-27: only epochs,
 * batch size, lr_init, weight decay and warm-up fraction set): in particular every training batch
 * is augmented with the reference's probabilities, as its train_epoch does unconditionally
 * (src/model/Trainer.cpp:156-161, 198-201), the engine seeded from the process's host generator
 * (vith_next_seed draws from it: after vith_seed(s) the sequence is repeatable). */
int vith_fit_reference_defaults(vith_model m, const float *x, const float *y_onehot, uint64_t n, const float *x_test,
                                const float *y_test_onehot, uint64_t n_test, int epochs, uint64_t batch_size,
                                float lr_init, float weight_decay, float warmup_frac, float *history);
uint64_t vith_next_seed(void);
/* wall time in milliseconds of the training part of every epoch the last vith_fit* call ran (the
 * batches of Trainer::train_epoch until the device has finished them; evaluation excluded);
 * returns the number of epochs, writes at most `capacity` values */
uint64_t vith_last_fit_epoch_ms(double *ms, uint64_t capacity);
/* vith_fit (no augmentation) with the checkpoint extensions of TrainerConfig, none of which the
 * reference has (it keeps optimizer state in memory only, src/optimizers/Adam.cpp:22-25):
 *   best_weights_path  after an epoch whose test accuracy is a new best AND above that epoch's
 *                      training accuracy, save the weights there (the policy of the reference's
 *                      CUDA tree, cuda_Vit/src/model/Trainer.cu:104-108); NULL = off
 *   state_path         after every epoch write state_path + ".weights" and the resumable training
 *                      state (Adam moments and step count, schedule position, dropout stream
 *                      position, epochs done, best accuracy) to state_path; NULL = off
 *   start_epoch        first epoch to run (epochs before it are taken as done)
 *   stop_epoch         > 0: stop before this epoch although the schedule is laid out for `epochs`
 *                      (a job that ends early and is resumed later); 0 = run to the end
 *   resume             non-zero: first load state_path + ".weights" and state_path and continue
 *                      after the last completed epoch, bit for bit like the uninterrupted run
 * history rows are written from index 0 for the epochs this call ran. */
typedef struct vith_fit_options {
  const char *best_weights_path;
  const char *state_path;
  int32_t start_epoch;
  int32_t resume;
  int32_t stop_epoch;
} vith_fit_options;
int vith_fit_with_options(vith_model m, const float *x, const float *y_onehot, uint64_t n, const float *x_test,
                          const float *y_test_onehot, uint64_t n_test, int epochs, uint64_t batch_size, float lr_init,
                          float weight_decay, float warmup_frac, const vith_fit_options *options, float *history);

/* ---- trainer: the per-batch body of Trainer::train_epoch (src/model/Trainer.cpp:201-227) ---- */
vith_trainer vith_trainer_create(vith_model m, float beta1, float beta2, float eps, float weight_decay,
                                 const float *class_weights_host, uint64_t n_weights);
void vith_trainer_destroy(vith_trainer t);
/* one whole step on one device: forward, loss, backward, Adam(lr).  loss/correct may be NULL. */
int vith_train_step(vith_trainer t, const float *x_dev, const float *y_onehot_dev, uint64_t B, float lr,
                    float *loss_host, float *correct_host);
/* the same step in pieces, for gradient all-reduce overlapped with backward:
 *   vith_step_forward  : forward + fused softmax-CE; gradient divided by grad_batch (the GLOBAL
 *                        batch under data parallelism; 0 = B).  Does not synchronise.
 *   vith_step_backward_stage(s), s = 0 .. vith_num_stages-1: head+final norm, blocks last to
 *                        first, embeddings.  After stage s the slice
 *                        [offset, offset+count) of the gradient arena given by
 *                        vith_stage_grad_range is final and may be reduced.
 *   vith_step_update   : Adam(lr) over the whole arena; gradients are multiplied by grad_scale.
 *   vith_step_result   : {mean loss of the local batch, number of correct rows} (synchronises). */
int vith_step_forward(vith_trainer t, const float *x_dev, const float *y_onehot_dev, uint64_t B, float grad_batch);
uint64_t vith_num_stages(vith_trainer t);
int vith_step_backward_stage(vith_trainer t, uint64_t stage);
int vith_stage_grad_range(vith_trainer t, uint64_t stage, uint64_t *offset, uint64_t *count);
int vith_step_update(vith_trainer t, float lr, float grad_scale);
/* data parallelism: replicas start from rank `root`'s parameters; sum of `count` device floats over
 * the ranks on the compute stream (metrics); bucket size of this trainer's gradient exchange and
 * the number of buckets it gives (0 without a communicator) */
int vith_comm_broadcast_params(vith_model m, int root);
int vith_comm_allreduce(float *dev, uint64_t count);
int vith_trainer_set_bucket_bytes(vith_trainer t, uint64_t bytes);
uint64_t vith_trainer_num_buckets(vith_trainer t);
/* Resumable optimizer state: Adam's step count and both moment buffers plus the position of the
 * dropout stream, written to / read from one file.  With vith_save_weights it continues a
 * run bit for bit (the reference keeps this state in memory only and cannot resume). */
int vith_trainer_save_state(vith_trainer t, const char *path);
int vith_trainer_load_state(vith_trainer t, const char *path);
int vith_step_result(vith_trainer t, float *loss_host, float *correct_host);
/* evaluation of one batch in inference mode: loss and number of correct rows */
int vith_eval_batch(vith_trainer t, const float *x_dev, const float *y_onehot_dev, uint64_t B, float *loss_host,
                    float *correct_host);
/* cosine_warmup_lr (include/optimizers/Scheduler.hpp:5-16) */
float vith_cosine_warmup_lr(int64_t step, int64_t warmup_steps, int64_t total_steps, float lr_max);
/* sample order of epoch `epoch` in Trainer::train_epoch: iota(n) shuffled by mt19937(epoch) */
int vith_epoch_order(uint64_t n, int32_t epoch, uint64_t *order_host);

/* ---- single layers (each a reference Layer subclass), for op-level parity tests ------------- */
/* kind: 0 Dense(a=in,b=out) 1 LayerNorm(a=D) 2 GELU 3 MultiHeadAttention(a=D,b=heads)
 *       4 FeedForward(a=D,b=H) 5 TransformerEncoderBlock(a=D,b=heads,c=H)
 *       6 Embeddings(a=image,b=patch,c=channels,d=D) */
typedef void *vith_layer;
vith_layer vith_layer_create(int kind, uint64_t a, uint64_t b, uint64_t c, uint64_t d);
/* the same with a dropout rate for the kinds that take one (4 FeedForward, 5 TransformerEncoderBlock) */
vith_layer vith_layer_create_dropout(int kind, uint64_t a, uint64_t b, uint64_t c, uint64_t d, float dropout_rate);
void vith_layer_destroy(vith_layer l);
uint64_t vith_layer_param_count(vith_layer l);
int vith_layer_set_params(vith_layer l, const float *flat_host);
int vith_layer_get_grads(vith_layer l, float *flat_host);
/* x_host has x_rank dims in x_shape; y_host receives the output (y_count floats are expected) */
int vith_layer_forward(vith_layer l, const float *x_host, const uint64_t *x_shape, uint64_t x_rank, int training,
                       float *y_host, uint64_t y_count);
int vith_layer_backward(vith_layer l, const float *dy_host, const uint64_t *dy_shape, uint64_t dy_rank,
                        float *dx_host, uint64_t dx_count);

#ifdef __cplusplus
}
#endif
#endif /* VIT_HOST_H */
