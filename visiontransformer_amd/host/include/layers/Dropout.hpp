// Inverted dropout (reference: include/layers/Dropout.hpp, src/layers/Dropout.cpp): identity when
// not training or rate == 0; otherwise mask in {0, 1/(1-rate)} drawn on the device (Philox).
// The mask is a pure function of the stream position of the forward call, so it is never stored:
// backward regenerates it.
#ifndef DROPOUT_HPP
#define DROPOUT_HPP

#include <cstdint>

#include "layers/Layer.hpp"

class Dropout : public Layer {
public:
  explicit Dropout(float rate = 0.5f);
  Tensor forward(const Tensor &input, bool isTraining) override;
  Tensor backward(const Tensor &outputGradient) override;
  // residual + dropout(input) in one pass (the skip connections of the encoder block)
  Tensor forwardAdd(const Tensor &input, const Tensor &residual, bool isTraining);
  // The same for ONE token per sequence: input / residual are {B,1,D}, and the factors are those a
  // call over the whole {B,tokens,D} tensor would have given row 0 of every sequence (the stream
  // advances as for that call), so a block that computes only the CLS token draws the masks of
  // the block that computes every token.  residual may be null.
  Tensor forwardAddFirstToken(const Tensor &input, const Tensor *residual, size_t tokens, bool isTraining);
  std::string getName() const override { return "Dropout"; }
  bool active(bool isTraining) const { return isTraining && rate != 0.0f; }

private:
  float rate;
  float scale;
  // stream position and extent of the last training-mode forward (count 0: none)
  uint64_t seed = 0, offset = 0;
  size_t count = 0;
  size_t rowStride = 0, rows = 0, cols = 0;  // rowStride != 0: the last forward was forwardAddFirstToken
  // recorded factors used by the last forward instead of the stream (vit::set_dropout_replay)
  const float *replayed = nullptr;
};

#endif  // DROPOUT_HPP
