// Trainer + TrainerConfig (reference: include/model/Trainer.hpp, src/model/Trainer.cpp): epoch loop
// with the per-epoch mt19937(epoch) shuffle, cosine/warm-up schedule, Adam(L2), CSV log.  The
// reference augments every training batch with hard-wired probabilities from a random_device
// seeded engine (src/model/Trainer.cpp:156-161, 198-201); here that is TrainerConfig::augment,
// on by default like the reference, with an optional seed so that runs can be repeated.
#ifndef TRAINER_HPP
#define TRAINER_HPP

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "core/Comm.hpp"
#include "losses/CrossEntropy.hpp"
#include "model/VisionTransformer.hpp"
#include "optimizers/Adam.hpp"
#include "utils/DataAugmentation.hpp"
#include "utils/DataReader.hpp"
#include "utils/Logger.hpp"

struct TrainerConfig {
  int epochs = 10;
  size_t batch_size = 64;
  float learning_rate = 0.001f;
  float weight_decay = 0.01f;
  float lr_init = 3e-4f;
  float warmup_frac = 0.1f;
  // Extensions (the defaults are the reference's behaviour).
  // best_weights_path: after each epoch, save the weights when the test accuracy is a new best and
  // above the epoch's training accuracy (the policy of cuda_Vit/src/model/Trainer.cu:104-108).
  std::string best_weights_path;
  // state_path: write a resumable training state (optimizer moments, step counters, dropout
  // stream position; see Trainer::saveState) after every epoch, next to the weights in
  // state_path + ".weights".  start_epoch: first epoch to run when resuming from such a state.
  std::string state_path;
  int start_epoch = 0;
  // stop_epoch: when > 0, leave the epoch loop before this epoch although `epochs` (which sets the
  // length of the learning-rate schedule) is larger -- a job that ends early and is resumed later.
  int stop_epoch = 0;
  // augment: apply DataAugmentation (probabilities 0.5 / 0.5 / 0.5 and the Config defaults, as
  // src/model/Trainer.cpp:157-161) to every training batch, on the device, folded into the batch
  // gather.  On by default: the reference's train_epoch augments unconditionally.  One engine per
  // epoch, as in the reference, seeded with augment_seed + epoch; augment_seed 0 = take one from
  // the process's host generator (vit::next_seed(): random_device-seeded like the reference's
  // engine unless vit::seed() was called).
  bool augment = true;
  uint32_t augment_seed = 0;
  DataAugmentation::Config augment_config;
  // Data parallelism (one process per GPU).  When a communicator exists (vit::comm_init_*; the
  // Trainer also joins one by itself when the launcher has set WORLD_SIZE > 1, see
  // vit::comm_init_from_env) batch_size is the batch PER RANK: every step consumes
  // world x batch_size samples of the epoch's shuffled order, rank r taking the r-th slice, and
  // gradients are all-reduced in buckets of at least bucket_bytes while backward runs.  All ranks
  // start from rank 0's parameters; rank 0 alone logs and writes files.
  size_t bucket_bytes = 25u << 20;
};

class Trainer {
public:
  Trainer(VisionTransformer &model, const TrainerConfig &train_config,
          const std::vector<float> &class_weights = std::vector<float>{});
  void train(const std::pair<Tensor, Tensor> &train_data, const std::pair<Tensor, Tensor> &test_data);

  const VisionTransformer &getModel() const { return model; }
  VisionTransformer &getModel() { return model; }

  // One optimisation step on an already assembled batch (the body of train_epoch after batching):
  // forward(train) -> loss -> backward -> schedule -> Adam.  Returns the batch loss.
  float trainStep(const Tensor &X_batch, const Tensor &y_batch, float lr_now, float *accuracy = nullptr);
  std::pair<float, float> evaluate(const Tensor &X_test, const Tensor &y_test);
  void setQuiet(bool q) { quiet = q; }
  void setStartEpoch(int epoch) { config.start_epoch = epoch; }
  Adam &getOptimizer() { return optimizer; }
  // Everything besides the weights that the next step depends on: Adam's step count and moments,
  // the schedule position (global step, total steps), the dropout stream position and the number
  // of epochs completed.  saveState + ModelUtils::save_weights, later load_weights + loadState,
  // continue a run bit for bit (the reference has no way to resume: src/model/Trainer.cpp keeps
  // all of this in memory only).  loadState returns the number of completed epochs.
  void saveState(const std::string &path, int epochs_done = 0);
  int loadState(const std::string &path);
  float bestTestAccuracy() const { return best_test_acc; }
  // {train loss, train accuracy, test loss, test accuracy} of every epoch train() has run
  const std::vector<std::array<float, 4>> &history() const { return epoch_history; }
  // wall time of each epoch's training part (train_epoch until its last kernel has finished,
  // evaluation excluded), in milliseconds
  const std::vector<double> &epochTrainMs() const { return train_ms; }
  CrossEntropy &getLoss() { return loss_fn; }

private:
  std::pair<float, float> train_epoch(const Tensor &X_train, const Tensor &y_train, int epoch);
  // the data-parallel form of trainStep: local batch may be empty (X_batch undefined) on the ragged
  // last step; global_count = samples of this step over all ranks
  float trainStepParallel(const Tensor &X_batch, const Tensor &y_batch, size_t local_count, size_t global_count,
                          float lr_now, float *accuracy);

  VisionTransformer &model;
  Adam optimizer;
  CrossEntropy loss_fn;
  TrainerConfig config;
  Logger logger;
  long long global_step = 0;
  long long total_steps = 0;
  float best_test_acc = 0.0f;
  std::vector<std::array<float, 4>> epoch_history;
  std::vector<double> train_ms;
  bool quiet = false;
  std::unique_ptr<vit::GradientReducer> reducer;  // set when a communicator exists
  Tensor metrics;                                 // {loss sum, correct} exchanged per step
};

#endif  // TRAINER_HPP
