// Trainer (reference: src/model/Trainer.cpp): epoch loop, per-epoch deterministic shuffle,
// cosine/warm-up schedule, Adam(L2), evaluation on slices, CSV log and console summary.
#include "model/Trainer.hpp"

#include <algorithm>
#include <chrono>
#include <fstream>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <numeric>
#include <random>
#include <sstream>

#include "core/Ops.hpp"
#include "optimizers/Scheduler.hpp"
#include "utils/ModelUtils.hpp"

namespace {
// Joins the launcher's communicator if there is one (WORLD_SIZE > 1) before anything is opened:
// only rank 0 writes the epoch log.
const char *log_path() {
  vit::comm_init_from_env();
  return vit::comm_rank() == 0 ? "vit_results.csv" : "/dev/null";
}
}  // namespace

Trainer::Trainer(VisionTransformer &model, const TrainerConfig &train_config, const std::vector<float> &class_weights)
    : model(model), optimizer(train_config.learning_rate, 0.9f, 0.999f, 1e-8f, train_config.weight_decay), loss_fn(),
      config(train_config), logger(log_path()) {
  loss_fn.setClassWeights(class_weights);
  if (config.augment && config.augment_seed == 0) {
    do config.augment_seed = static_cast<uint32_t>(vit::next_seed());
    while (config.augment_seed == 0);
  }
  if (vit::comm_active()) {
    reducer = std::make_unique<vit::GradientReducer>(model, config.bucket_bytes);
    // replicas start from rank 0's parameters (each process initialised its own)
    auto &p = model.arena().params();
    vit::comm_broadcast(p->dev_write(), p->count, 0);
    if (vit::comm_rank() != 0) quiet = true;
  }
}

// forward(train) -> loss/accuracy -> loss gradient -> model backward -> LR -> Adam
// (src/model/Trainer.cpp:201-227).  Top-1 hits come out of the fused loss kernel.
float Trainer::trainStep(const Tensor &X_batch, const Tensor &y_batch, float lr_now, float *accuracy) {
  if (reducer) {
    const size_t local = X_batch.getShape()[0];
    return trainStepParallel(X_batch, y_batch, local, local * (size_t)vit::comm_world(), lr_now, accuracy);
  }
  Tensor logits = model.forward(X_batch, true);
  const float batch_loss = loss_fn.calculate(logits, y_batch);
  if (accuracy) *accuracy = loss_fn.lastCorrect() / static_cast<float>(X_batch.getShape()[0]);
  Tensor grad = loss_fn.backward(logits, y_batch);
  // The reference drops what model.backward returns (src/model/Trainer.cpp:212): the staged form
  // does not form the image gradient at all, and leaves the model's own backward() untouched.
  model.backwardBegin(grad);
  for (size_t s = 0; s < model.numBackwardStages(); ++s) model.backwardStage(s);
  auto params = model.getParameters();
  auto grads = model.getGradients();
  optimizer.setLearningRate(lr_now);
  optimizer.update(params, grads);
  return batch_loss;
}

// The same step with the batch sharded over the ranks of the communicator: the loss gradient is
// divided by the GLOBAL batch, backward runs stage by stage and each closed gradient bucket is
// all-reduced (sum) on the side stream meanwhile; Adam follows once every bucket has landed.
// Returns the loss of the global batch (the ranks' loss sums are exchanged with the hit counts).
float Trainer::trainStepParallel(const Tensor &X_batch, const Tensor &y_batch, size_t local_count, size_t global_count,
                                 float lr_now, float *accuracy) {
  void *st = vit::stream();
  if (!metrics.defined()) metrics = Tensor::deviceEmpty({2});
  float *mdev = metrics.deviceDataOverwrite();
  if (local_count > 0) {
    Tensor logits = model.forward(X_batch, true);
    loss_fn.setGradientBatch((float)global_count);
    loss_fn.enqueue(logits, y_batch);
    const float *r = loss_fn.resultDevice();  // {mean loss of the local batch, hits}
    vit::check(vitk_scale_f32(r, (float)local_count, 1, mdev, st), "vitk_scale_f32");
    vit::check(vitk_scale_f32(r + 1, 1.0f, 1, mdev + 1, st), "vitk_scale_f32");
    model.backwardBegin(loss_fn.backward(logits, y_batch));
    for (size_t s = 0; s < model.numBackwardStages(); ++s) {
      model.backwardStage(s);
      reducer->afterStage(s);
    }
  } else {
    // nothing of the (ragged) last global batch fell to this rank: it contributes zeros
    auto &g = model.arena().grads();
    vit::zero_device(g->dev_overwrite(), g->count);
    vit::zero_device(mdev, 2);
    for (size_t s = 0; s < model.numBackwardStages(); ++s) reducer->afterStage(s);
  }
  reducer->finish();
  vit::comm_allreduce(mdev, 2);
  auto params = model.getParameters();
  auto grads = model.getGradients();
  optimizer.setLearningRate(lr_now);
  optimizer.update(params, grads);
  float sums[2];
  vit::copy_to_host(sums, mdev, 2);  // synchronises, like the reference's per-batch loss read
  if (accuracy) *accuracy = sums[1] / static_cast<float>(global_count);
  return sums[0] / static_cast<float>(global_count);
}

void Trainer::train(const std::pair<Tensor, Tensor> &train_data, const std::pair<Tensor, Tensor> &test_data) {
  const Tensor &X_train = train_data.first, &y_train = train_data.second;
  const Tensor &X_test = test_data.first, &y_test = test_data.second;
  const size_t global_batch = config.batch_size * (size_t)vit::comm_world();
  const size_t batches_per_epoch = (X_train.getShape()[0] + global_batch - 1) / global_batch;
  const bool root = vit::comm_rank() == 0;
  total_steps = (long long)batches_per_epoch * config.epochs;
  const int last_epoch = config.stop_epoch > 0 ? std::min(config.stop_epoch, config.epochs) : config.epochs;
  for (int epoch = config.start_epoch; epoch < last_epoch; ++epoch) {
    const auto start = std::chrono::high_resolution_clock::now();
    auto [train_loss, train_acc] = train_epoch(X_train, y_train, epoch);
    vit::synchronize();
    train_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - start).count());
    if (!quiet) std::cout << "\r" << std::string(80, ' ') << "\r";
    auto [test_loss, test_acc] = evaluate(X_test, y_test);
    const auto ms_total =
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();
    logger.log_epoch(epoch, config.epochs, train_loss, train_acc, test_loss, test_acc);
    epoch_history.push_back({train_loss, train_acc, test_loss, test_acc});
    if (!config.best_weights_path.empty() && test_acc > best_test_acc && test_acc > train_acc) {
      if (root) ModelUtils::save_weights(model, config.best_weights_path);
      best_test_acc = test_acc;
    }
    if (!config.state_path.empty() && root) {
      ModelUtils::save_weights(model, config.state_path + ".weights");
      saveState(config.state_path, epoch + 1);
    }
    std::ostringstream t;
    t << std::setfill('0') << "[" << std::setw(2) << ms_total / 3600000 << ":" << std::setw(2)
      << (ms_total % 3600000) / 60000 << ":" << std::setw(2) << (ms_total % 60000) / 1000 << "." << std::setw(3)
      << ms_total % 1000 << "]";
    if (!quiet)
      std::cout << "\033[1m--- Epoch " << (epoch + 1) << "/" << config.epochs << "\033[0m | \033[1;34mTrain Loss: "
              << std::fixed << std::setprecision(4) << train_loss << " | Train Acc: " << train_acc
              << "\033[0m | \033[1;32mTest Loss: " << test_loss << " | Test Acc: " << test_acc << "\033[0m \033[1;35m"
              << t.str() << "\033[0m" << std::endl;
  }
}

// State file: "VITTRAIN1", i64 global step, i64 total steps, i64 epochs done, f32 best test
// accuracy, u64 dropout seed, u64 dropout offset, then Adam's record (Adam::writeState).
void Trainer::saveState(const std::string &path, int epochs_done) {
  std::ofstream out(path, std::ios::binary);
  if (!out) throw std::runtime_error("No se pudo abrir el archivo de estado para escritura: " + path);
  const int64_t gs = global_step, ts = total_steps, ep = epochs_done;
  // the unsalted seed: under data parallelism every rank that loads it re-applies its own salt
  uint64_t dseed = 0, doff = 0;
  vit::get_dropout_base_state(&dseed, &doff);
  out.write("VITTRAIN1", 9);
  out.write(reinterpret_cast<const char *>(&gs), sizeof gs);
  out.write(reinterpret_cast<const char *>(&ts), sizeof ts);
  out.write(reinterpret_cast<const char *>(&ep), sizeof ep);
  out.write(reinterpret_cast<const char *>(&best_test_acc), sizeof best_test_acc);
  out.write(reinterpret_cast<const char *>(&dseed), sizeof dseed);
  out.write(reinterpret_cast<const char *>(&doff), sizeof doff);
  optimizer.writeState(out);
  if (!out) throw std::runtime_error("Error de escritura en el archivo de estado: " + path);
}

int Trainer::loadState(const std::string &path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("No se pudo abrir el archivo de estado para lectura: " + path);
  char magic[9];
  int64_t gs = 0, ts = 0, ep = 0;
  uint64_t dseed = 0, doff = 0;
  in.read(magic, 9);
  in.read(reinterpret_cast<char *>(&gs), sizeof gs);
  in.read(reinterpret_cast<char *>(&ts), sizeof ts);
  in.read(reinterpret_cast<char *>(&ep), sizeof ep);
  in.read(reinterpret_cast<char *>(&best_test_acc), sizeof best_test_acc);
  in.read(reinterpret_cast<char *>(&dseed), sizeof dseed);
  in.read(reinterpret_cast<char *>(&doff), sizeof doff);
  if (!in || std::string(magic, 9) != "VITTRAIN1")
    throw std::runtime_error("Archivo de estado con formato desconocido: " + path);
  global_step = gs;
  total_steps = ts;
  vit::set_dropout_base_state(dseed, doff);
  optimizer.readState(in, model.getParameters());
  return (int)ep;
}

std::pair<float, float> Trainer::train_epoch(const Tensor &X_train, const Tensor &y_train, int epoch) {
  const auto &shape = X_train.getShape();
  const size_t n = shape[0], sample = shape[1] * shape[2] * shape[3], K = y_train.getShape()[1];
  const size_t world = (size_t)vit::comm_world(), rank = (size_t)vit::comm_rank();
  const size_t global_batch = config.batch_size * world;
  const size_t num_batches = (n + global_batch - 1) / global_batch;
  float total_loss = 0.0f, total_accuracy = 0.0f;
  // sample order of the epoch: iota shuffled by mt19937(epoch) (src/model/Trainer.cpp:151-154)
  std::vector<size_t> indices(n);
  std::iota(indices.begin(), indices.end(), 0);
  std::mt19937 rng(static_cast<uint32_t>(epoch));
  std::shuffle(indices.begin(), indices.end(), rng);
  // The data set stays resident in HBM (uploaded once; 288 GB hold every data set the reference
  // trains on many times over) and each batch is gathered on the device: no per-step host copy,
  // no per-step upload.  The epoch's sample order goes up as one int32 array.
  Tensor Xs = X_train.dense(), ys = y_train.dense();
  const float *xsrc = Xs.deviceData(), *ysrc = ys.deviceData();
  std::vector<int32_t> order32(indices.begin(), indices.end());
  Tensor order_dev = Tensor::deviceEmpty({n});  // n int32 values in a 4-byte-per-element block
  vit::copy_to_device(order_dev.deviceDataOverwrite(), reinterpret_cast<const float *>(order32.data()), n);
  const int32_t *order = reinterpret_cast<const int32_t *>(order_dev.deviceData());
  // Augmentation (src/model/Trainer.cpp:156-161, 198-199): one engine per epoch whose draws run
  // through the epoch's images in batch order, so all of them are made up front and go up as one
  // array; the resampling is folded into the batch gather.
  Tensor aug_dev;
  const vitk_augment_params *aug = nullptr;
  if (config.augment) {
    DataAugmentation augmentor(config.augment_config, config.augment_seed + static_cast<uint32_t>(epoch));
    std::vector<vitk_augment_params> draws;
    augmentor.draw(n, shape[2], shape[3], draws);
    constexpr size_t kWords = sizeof(vitk_augment_params) / sizeof(float);
    aug_dev = Tensor::deviceEmpty({n * kWords});
    vit::copy_to_device(aug_dev.deviceDataOverwrite(), reinterpret_cast<const float *>(draws.data()), n * kWords);
    aug = reinterpret_cast<const vitk_augment_params *>(aug_dev.deviceData());
  }
  for (size_t i = 0; i < num_batches; ++i) {
    // this step's slice of the shuffled order: [i*global_batch, ...) holds global_count samples,
    // of which this rank takes the rank-th run of batch_size (short or empty at the ragged end)
    const size_t global_count = std::min(global_batch, n - i * global_batch);
    const size_t lo = std::min(rank * config.batch_size, global_count);
    const size_t count = std::min(config.batch_size, global_count - lo);
    const size_t first = i * global_batch + lo;
    if (count == 0 && !reducer) continue;
    Tensor X_batch, y_batch;
    if (count > 0) {
      X_batch = Tensor::deviceEmpty({count, shape[1], shape[2], shape[3]});
      y_batch = Tensor::deviceEmpty({count, K});
    }
    if (count == 0) {
    } else if (aug != nullptr)
      vit::check(vitk_augment_f32(xsrc, (int64_t)n, order + first, aug + first, (int64_t)count, (int32_t)shape[1],
                                  (int32_t)shape[2], (int32_t)shape[3], X_batch.deviceDataOverwrite(), vit::stream()),
                 "vitk_augment_f32");
    else
      vit::check(vitk_gather_rows_f32(xsrc, (int64_t)n, order + first, (int64_t)count, (int64_t)sample,
                                      X_batch.deviceDataOverwrite(), vit::stream()),
                 "vitk_gather_rows_f32");
    if (count > 0)
      vit::check(vitk_gather_rows_f32(ysrc, (int64_t)n, order + first, (int64_t)count, (int64_t)K,
                                      y_batch.deviceDataOverwrite(), vit::stream()),
                 "vitk_gather_rows_f32");
    const long long warmup_steps = (long long)(config.warmup_frac * total_steps);
    const float lr_now = cosine_warmup_lr(global_step, warmup_steps, total_steps, config.lr_init);
    float acc = 0.0f;
    const float batch_loss = reducer ? trainStepParallel(X_batch, y_batch, count, global_count, lr_now, &acc)
                                     : trainStep(X_batch, y_batch, lr_now, &acc);
    total_loss += batch_loss;
    total_accuracy += acc;
    global_step++;
    if (!quiet)
      std::cout << "\rEntrenando... Batch " << (i + 1) << "/" << num_batches << " [loss: " << std::fixed
                << std::setprecision(8) << batch_loss << "]"
                << " | LR: " << std::fixed << std::setprecision(8) << optimizer.getLearningRate() << std::flush;
  }
  return {total_loss / num_batches, total_accuracy / num_batches};
}

std::pair<float, float> Trainer::evaluate(const Tensor &X_test, const Tensor &y_test) {
  const size_t n = X_test.getShape()[0];
  const size_t num_batches = (n + config.batch_size - 1) / config.batch_size;
  float total_loss = 0.0f, total_accuracy = 0.0f;
  for (size_t i = 0; i < num_batches; ++i) {
    const size_t start = i * config.batch_size;
    const size_t count = std::min(config.batch_size, n - start);
    if (count == 0) continue;
    Tensor X_batch = X_test.slice(0, start, count), y_batch = y_test.slice(0, start, count);
    Tensor logits = model.forward(X_batch, false);
    total_loss += loss_fn.calculate(logits, y_batch);
    total_accuracy += loss_fn.lastCorrect() / static_cast<float>(count);
  }
  if (num_batches == 0) return {0.0f, 0.0f};
  return {total_loss / num_batches, total_accuracy / num_batches};
}
