// Softmax cross-entropy (reference: include/losses/CrossEntropy.hpp, src/losses/CrossEntropy.cpp).
// calculate() runs one fused kernel that produces the probabilities, the class-weighted mean loss,
// the (unweighted) gradient (softmax - y)/B and the top-1 hit count; backward() hands out the
// gradient computed by the preceding calculate(), like the reference reuses its softmaxOutput.
#ifndef CROSSENTROPY_HPP
#define CROSSENTROPY_HPP

#include <vector>

#include "losses/Loss.hpp"

Tensor softmax(const Tensor &logits);

class CrossEntropy : public Loss {
public:
  CrossEntropy() = default;
  float calculate(const Tensor &yPred, const Tensor &yTrue) override;
  Tensor backward(const Tensor &yPred, const Tensor &yTrue) override;
  void setClassWeights(const std::vector<float> &weights);

  // Data parallelism: divide the gradient by the GLOBAL batch instead of the local one (0 = local).
  void setGradientBatch(float global_batch) { grad_batch = global_batch; }
  // number of rows of the last calculate() whose first-maximum logit is a labelled class
  float lastCorrect() const { return last_correct; }
  // enqueue only (no host sync): results land in resultBuffer() = {loss, correct}
  void enqueue(const Tensor &yPred, const Tensor &yTrue);
  const Tensor &resultBuffer() const { return out2; }
  const float *resultDevice() const { return out2.deviceData(); }

private:
  Tensor softmaxOutput, gradient, out2, scratch, weightsDevice;
  std::vector<float> class_weights;
  float grad_batch = 0.0f;
  float last_correct = 0.0f;
};

#endif  // CROSSENTROPY_HPP
