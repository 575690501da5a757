// Optimizer interface of the reference (include/optimizers/Optimizer.hpp:17-49), unchanged.
#ifndef OPTIMIZER_HPP
#define OPTIMIZER_HPP

#include <vector>

#include "core/Tensor.hpp"

class Optimizer {
public:
  explicit Optimizer(float learningRate) : learningRate(learningRate) {}
  virtual ~Optimizer() = default;
  virtual void update(std::vector<Tensor *> &parameters, const std::vector<Tensor *> &gradients) = 0;
  void setLearningRate(float lr) { learningRate = lr; }
  float getLearningRate() const { return learningRate; }

protected:
  float learningRate;
};

#endif  // OPTIMIZER_HPP
