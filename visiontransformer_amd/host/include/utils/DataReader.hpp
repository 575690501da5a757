// CSV dataset reader (reference: include/utils/DataReader.hpp, src/utils/DataReader.cpp):
// rows "label,pix0,pix1,..." -> X {n,C,H,W} normalised as (pix/255 - mean)/std, y one-hot {n,K}.
#ifndef DATAREADER_HPP
#define DATAREADER_HPP

#include <string>
#include <utility>
#include <vector>

#include "core/Tensor.hpp"

using XYPair = std::pair<Tensor, Tensor>;

std::pair<Tensor, Tensor> load_csv_data(const std::string &filePath, float sample_fraction, size_t channels,
                                        size_t height, size_t width, size_t num_classes, float mean = 0.2860f,
                                        float stddev = 0.3530f);

std::pair<XYPair, XYPair> load_csv_data_train_val(const std::string &filePath, float sample_frac, float train_frac,
                                                  float val_frac, size_t channels, size_t height, size_t width,
                                                  size_t num_classes, float mean = 0.2860f, float stddev = 0.3530f);

std::vector<float> compute_class_weights(const Tensor &y_onehot);
void print_classweights(const std::vector<float> &vec);

#endif  // DATAREADER_HPP
