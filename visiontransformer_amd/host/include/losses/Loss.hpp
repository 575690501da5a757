// Loss interface of the reference (include/losses/Loss.hpp:19-47), unchanged.
#ifndef LOSS_HPP
#define LOSS_HPP

#include "core/Tensor.hpp"

class Loss {
public:
  virtual ~Loss() = default;
  virtual float calculate(const Tensor &yPred, const Tensor &yTrue) = 0;
  virtual Tensor backward(const Tensor &yPred, const Tensor &yTrue) = 0;
};

#endif  // LOSS_HPP
