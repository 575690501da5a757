// Trainer + TrainerConfig (reference: include/model/Trainer.hpp, src/model/Trainer.cpp): epoch loop
// with the per-epoch mt19937(epoch) shuffle, cosine/warm-up schedule, Adam(L2), CSV log.  The
// reference's CPU-side random augmentation (hard-wired probabilities, random_device seeded) is not
// applied; everything else in the per-batch body is the same sequence of calls.
#ifndef TRAINER_HPP
#define TRAINER_HPP

#include <memory>
#include <vector>

#include "losses/CrossEntropy.hpp"
#include "model/VisionTransformer.hpp"
#include "optimizers/Adam.hpp"
#include "utils/DataReader.hpp"
#include "utils/Logger.hpp"

struct TrainerConfig {
  int epochs = 10;
  size_t batch_size = 64;
  float learning_rate = 0.001f;
  float weight_decay = 0.01f;
  float lr_init = 3e-4f;
  float warmup_frac = 0.1f;
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
  Adam &getOptimizer() { return optimizer; }
  CrossEntropy &getLoss() { return loss_fn; }

private:
  std::pair<float, float> train_epoch(const Tensor &X_train, const Tensor &y_train, int epoch);

  VisionTransformer &model;
  Adam optimizer;
  CrossEntropy loss_fn;
  TrainerConfig config;
  Logger logger;
  long long global_step = 0;
  long long total_steps = 0;
  bool quiet = false;
};

#endif  // TRAINER_HPP
