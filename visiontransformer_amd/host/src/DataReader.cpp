// CSV reader (reference: src/utils/DataReader.cpp): "label,p0,p1,..." rows after a header line;
// pixels normalised as (p/255 - mean)/std; a random subset of `sample_fraction` is kept; labels are
// one-hot encoded.  The subset is drawn with the host RNG of core/Device.hpp (vit::seed() makes
// it repeatable; the reference seeds from random_device).
#include "utils/DataReader.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <numeric>
#include <random>
#include <stdexcept>

#include "core/Device.hpp"

std::pair<Tensor, Tensor> load_csv_data(const std::string &filePath, float sample_fraction, size_t channels,
                                        size_t height, size_t width, size_t num_classes, float mean, float stddev) {
  std::cout << "--- Cargando " << filePath << "  (fracción: " << sample_fraction * 100 << "%, "
            << "μ=" << mean << ", σ=" << stddev << ")" << std::endl;
  std::ifstream file(filePath);
  if (!file.is_open()) throw std::runtime_error("Error: No se pudo abrir el archivo: " + filePath);
  const size_t npix = channels * height * width;
  std::string line;
  std::getline(file, line);  // header
  std::vector<float> pixels;
  std::vector<int> labels;
  std::vector<float> row(npix);
  while (std::getline(file, line)) {
    if (line.empty()) continue;
    const char *p = line.c_str();
    char *end = nullptr;
    const long label = std::strtol(p, &end, 10);
    if (end == p) continue;
    size_t got = 0;
    p = end;
    while (*p == ',') {
      ++p;
      const float raw = std::strtof(p, &end);
      if (end == p) break;
      if (got < npix) {
        float v = raw / 255.0f;
        row[got] = (v - mean) / stddev;
      }
      ++got;
      p = end;
    }
    if (got != npix) {
      std::cerr << "Advertencia: Se encontró una fila con un número de píxeles incorrecto. Se ignora." << std::endl;
      continue;
    }
    labels.push_back((int)label);
    pixels.insert(pixels.end(), row.begin(), row.end());
  }
  const size_t total = labels.size();
  std::vector<size_t> idx(total);
  std::iota(idx.begin(), idx.end(), 0);
  std::shuffle(idx.begin(), idx.end(), std::mt19937(static_cast<uint32_t>(vit::next_seed())));
  size_t keep = static_cast<size_t>(total * sample_fraction);
  if (keep == 0 && total > 0) keep = 1;
  if (keep > total) keep = total;
  std::vector<float> X(keep * npix), Y(keep * num_classes, 0.0f);
  for (size_t i = 0; i < keep; ++i) {
    std::copy(pixels.begin() + idx[i] * npix, pixels.begin() + (idx[i] + 1) * npix, X.begin() + i * npix);
    const int l = labels[idx[i]];
    if (l >= 0 && (size_t)l < num_classes) Y[i * num_classes + l] = 1.0f;
  }
  Tensor Xt({keep, channels, height, width}, X), yt({keep, num_classes}, Y);
  std::cout << "Carga completa. " << keep << " muestras cargadas." << std::endl;
  std::cout << "  -> Forma de X (imágenes): " << Xt.shapeToString() << std::endl;
  std::cout << "  -> Forma de y (etiquetas): " << yt.shapeToString() << std::endl;
  return {std::move(Xt), std::move(yt)};
}

std::pair<XYPair, XYPair> load_csv_data_train_val(const std::string &filePath, float sample_frac, float train_frac,
                                                  float val_frac, size_t channels, size_t height, size_t width,
                                                  size_t num_classes, float mean, float stddev) {
  if (train_frac + val_frac > 1.0f + 1e-6f) throw std::invalid_argument("train_frac + val_frac no debe superar 1.0");
  XYPair all = load_csv_data(filePath, sample_frac, channels, height, width, num_classes, mean, stddev);
  const size_t N = all.first.getShape()[0];
  const size_t n_train = static_cast<size_t>(N * train_frac), n_val = static_cast<size_t>(N * val_frac);
  if (n_train + n_val > N) throw std::runtime_error("No hay suficientes muestras para el split solicitado.");
  // views into the one loaded block, as in the reference
  return {{all.first.slice(0, 0, n_train), all.second.slice(0, 0, n_train)},
          {all.first.slice(0, n_train, n_val), all.second.slice(0, n_train, n_val)}};
}

std::vector<float> compute_class_weights(const Tensor &y_onehot) {
  const size_t n = y_onehot.getShape()[0], K = y_onehot.getShape()[1];
  std::vector<size_t> counts(K, 0);
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < K; ++j)
      if (y_onehot(i, j) == 1.0f) {
        counts[j]++;
        break;
      }
  // inverse-frequency weights normalised to sum to 1 (src/utils/DataReader.cpp:178-191)
  std::vector<float> w(K, 0.0f);
  const float total = static_cast<float>(n);
  for (size_t i = 0; i < K; ++i) w[i] = total / (K * static_cast<float>(counts[i]));
  const float sum = std::accumulate(w.begin(), w.end(), 0.0f);
  for (float &x : w) x /= sum;
  return w;
}

void print_classweights(const std::vector<float> &w) {
  for (size_t i = 0; i < w.size(); i++) std::cout << "Clase " << i << ": " << w[i] << std::endl;
}
