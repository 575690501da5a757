// Adam with coupled L2 regularisation (reference: include/optimizers/Adam.hpp,
// src/optimizers/Adam.cpp).  When all parameters are views of one arena (a whole model), the
// update is ONE fused kernel over the flat parameter/gradient/m/v buffers; otherwise one launch
// per tensor.
#ifndef ADAM_HPP
#define ADAM_HPP

#include <iosfwd>
#include <vector>

#include "optimizers/Optimizer.hpp"

class Adam : public Optimizer {
public:
  Adam(float learningRate = 0.001f, float beta1 = 0.9f, float beta2 = 0.999f, float epsilon = 1e-8,
       float weight_decay = 0.0f);
  void update(std::vector<Tensor *> &parameters, const std::vector<Tensor *> &gradients) override;
  // gradients are multiplied by this before use (1/world after a sum all-reduce); default 1
  void setGradientScale(float s) { grad_scale = s; }
  long long stepCount() const { return t; }
  // Resumable state (the reference cannot resume: its optimizer state is never saved): step count
  // and both moment buffers, written after / read before any number of update() calls.
  // `parameters` sizes the buffers when the optimizer has not stepped yet.
  void writeState(std::ostream &out) const;
  void readState(std::istream &in, const std::vector<Tensor *> &parameters);

private:
  void allocateMoments(const std::vector<Tensor *> &parameters, bool gradients_flat);
  float beta1, beta2, epsilon, weight_decay;
  long long t;
  std::vector<Tensor> m, v;
  bool initialized;
  bool flat = false;
  float grad_scale = 1.0f;
};

#endif  // ADAM_HPP
