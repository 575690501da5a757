// Layer: the operator interface of the reference (include/layers/Layer.hpp:16-67), unchanged:
// forward / backward / getParameters / getGradients / getName.  Implementations keep the state of
// the last training-mode forward in members, exactly like the reference (one in-flight forward per
// layer object; backward must follow a training-mode forward on the same object).
#ifndef LAYER_HPP
#define LAYER_HPP

#include <string>
#include <vector>

#include "core/Tensor.hpp"

class Layer {
public:
  virtual ~Layer() = default;
  virtual Tensor forward(const Tensor &input, bool isTraining) = 0;
  virtual Tensor backward(const Tensor &outputGradient) = 0;
  virtual std::vector<Tensor *> getParameters() { return {}; }
  virtual std::vector<Tensor *> getGradients() { return {}; }
  virtual std::string getName() const = 0;
};

#endif  // LAYER_HPP
