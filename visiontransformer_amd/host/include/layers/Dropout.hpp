// Inverted dropout (reference: include/layers/Dropout.hpp, src/layers/Dropout.cpp): identity when
// not training or rate == 0; otherwise mask in {0, 1/(1-rate)} drawn on the device (Philox).
#ifndef DROPOUT_HPP
#define DROPOUT_HPP

#include "layers/Layer.hpp"

class Dropout : public Layer {
public:
  explicit Dropout(float rate = 0.5f);
  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  std::string getName() const override { return "Dropout"; }
  bool active(bool isTraining) const { return isTraining && rate != 0.0f; }

private:
  float rate;
  float scale;
  Tensor mask;
};

#endif  // DROPOUT_HPP
