// Checkpoint / configuration IO (reference: src/utils/ModelUtils.cpp).
#include "utils/ModelUtils.hpp"

#include <charconv>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace ModelUtils {

namespace {
bool g_verbose = true;

// number formatting of a JSON float: shortest round-trip decimal, with ".0" for integral values
std::string json_double(double v) {
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof(buf), v);
  std::string s(buf, r.ptr);
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  return s;
}

// Minimal parser for the flat {"key": number, ...} objects this format uses.
std::map<std::string, double> parse_flat_json(const std::string &text, const std::string &path) {
  std::map<std::string, double> out;
  size_t i = 0;
  auto skip = [&]() { while (i < text.size() && std::isspace((unsigned char)text[i])) ++i; };
  skip();
  if (i >= text.size() || text[i] != '{') throw std::runtime_error("JSON inválido en " + path);
  ++i;
  for (;;) {
    skip();
    if (i < text.size() && text[i] == '}') break;
    if (i >= text.size() || text[i] != '"') throw std::runtime_error("JSON inválido (clave) en " + path);
    const size_t e = text.find('"', i + 1);
    if (e == std::string::npos) throw std::runtime_error("JSON inválido (clave) en " + path);
    const std::string key = text.substr(i + 1, e - i - 1);
    i = e + 1;
    skip();
    if (i >= text.size() || text[i] != ':') throw std::runtime_error("JSON inválido (':') en " + path);
    ++i;
    skip();
    char *end = nullptr;
    const double v = std::strtod(text.c_str() + i, &end);
    if (end == text.c_str() + i) throw std::runtime_error("JSON inválido (valor de " + key + ") en " + path);
    i = end - text.c_str();
    out[key] = v;
    skip();
    if (i < text.size() && text[i] == ',') { ++i; continue; }
    if (i < text.size() && text[i] == '}') break;
    throw std::runtime_error("JSON inválido (separador) en " + path);
  }
  return out;
}
}  // namespace

void set_verbose(bool on) { g_verbose = on; }

std::vector<float> flatten(const std::vector<Tensor *> &tensors) {
  size_t n = 0;
  for (const Tensor *t : tensors) n += t->getSize();
  std::vector<float> flat(n);
  size_t off = 0;
  for (const Tensor *t : tensors) {
    Tensor d = t->dense();  // strided views (packed q/k/v) are gathered on the device
    std::memcpy(flat.data() + off, d.data(), d.getSize() * sizeof(float));
    off += d.getSize();
  }
  return flat;
}

void unflatten(const std::vector<Tensor *> &tensors, const float *flat, size_t count) {
  size_t n = 0;
  for (const Tensor *t : tensors) n += t->getSize();
  if (n != count) throw std::runtime_error("unflatten: el número de elementos no coincide.");
  size_t off = 0;
  for (Tensor *t : tensors) {
    std::vector<float> v(flat + off, flat + off + t->getSize());
    Tensor src(t->getShape(), v);
    t->copyFrom(src);
    off += t->getSize();
  }
}

void save_weights(const VisionTransformer &model, const std::string &filePath) {
  std::ofstream out(filePath, std::ios::binary);
  if (!out) throw std::runtime_error("No se pudo abrir el archivo para escritura: " + filePath);
  auto params = const_cast<VisionTransformer &>(model).getParameters();
  if (g_verbose)
    std::cout << "Guardando " << params.size() << " tensores de parámetros en " << filePath << "..." << std::endl;
  for (const Tensor *t : params) {
    const auto &shape = t->getShape();
    const size_t rank = shape.size();
    out.write(reinterpret_cast<const char *>(&rank), sizeof(size_t));
    out.write(reinterpret_cast<const char *>(shape.data()), rank * sizeof(size_t));
    Tensor d = t->dense();
    out.write(reinterpret_cast<const char *>(d.data()), d.getSize() * sizeof(float));
  }
  out.close();
  if (g_verbose) std::cout << "Pesos guardados correctamente." << std::endl;
}

void load_weights(VisionTransformer &model, const std::string &filePath) {
  std::ifstream in(filePath, std::ios::binary);
  if (!in) throw std::runtime_error("No se pudo abrir el archivo para lectura: " + filePath);
  auto params = model.getParameters();
  if (g_verbose)
    std::cout << "Cargando " << params.size() << " tensores de parámetros desde " << filePath << "..." << std::endl;
  for (size_t index = 0; index < params.size(); ++index) {
    Tensor *t = params[index];
    size_t rank = 0;
    in.read(reinterpret_cast<char *>(&rank), sizeof(size_t));
    if (!in || rank > 8) throw std::runtime_error("Error de lectura: fin de archivo inesperado o datos corruptos.");
    std::vector<size_t> shape(rank);
    in.read(reinterpret_cast<char *>(shape.data()), rank * sizeof(size_t));
    if (!in) throw std::runtime_error("Error de lectura: fin de archivo inesperado o datos corruptos.");
    // Checkpoints written by the reference's CUDA tree store the patch projection as a Conv2d:
    // weight {D, C, P, P}, bias {1, D, 1, 1} (cuda_Vit/src/layers/PatchEmbedding.cu:36,
    // cuda_Vit/src/layers/Conv2d.cu), applied to true (channel, row, column) patches.  That is
    // this model's {C*P*P, D} projection transposed, with im2col patch extraction.
    if (index == 0 && rank == 4 && t->getShape().size() == 2 && shape[0] == t->getShape()[1] &&
        shape[1] * shape[2] * shape[3] == t->getShape()[0]) {
      const size_t D = shape[0], pd = t->getShape()[0];
      std::vector<float> conv(D * pd), w(pd * D);
      in.read(reinterpret_cast<char *>(conv.data()), conv.size() * sizeof(float));
      if (static_cast<size_t>(in.gcount()) != conv.size() * sizeof(float))
        throw std::runtime_error("Error de lectura: fin de archivo inesperado o datos corruptos.");
      for (size_t d = 0; d < D; ++d)
        for (size_t j = 0; j < pd; ++j) w[j * D + d] = conv[d * pd + j];
      t->copyFrom(Tensor(t->getShape(), w));
      model.getEmbeddings().getPatcher().setPatchMode(VITK_PATCH_IM2COL);
      continue;
    }
    if (index == 1 && rank == 4 && shape[0] == 1 && shape[2] == 1 && shape[3] == 1 && shape[1] == t->getSize())
      shape = t->getShape();  // {1, D, 1, 1}: same D floats
    if (t->getShape() != shape)
      throw std::runtime_error("Incompatibilidad de formas al cargar pesos. Esperado: " + t->shapeToString() +
                               ", encontrado en archivo: " + Tensor(shape).shapeToString());
    std::vector<float> buf(t->getSize());
    in.read(reinterpret_cast<char *>(buf.data()), buf.size() * sizeof(float));
    if (static_cast<size_t>(in.gcount()) != buf.size() * sizeof(float))
      throw std::runtime_error("Error de lectura: fin de archivo inesperado o datos corruptos.");
    Tensor src(shape, buf);
    t->copyFrom(src);
  }
  if (g_verbose) std::cout << "Pesos cargados correctamente." << std::endl;
}

void save_config(const ViTConfig &c, const std::string &filePath) {
  if (g_verbose) std::cout << "Guardando configuración del modelo en: " << filePath << "..." << std::endl;
  std::ofstream out(filePath);
  if (!out) throw std::runtime_error("No se pudo abrir el archivo para escritura: " + filePath);
  // keys in sorted order, 4-space indent: the layout the reference's JSON writer produces
  out << "{\n"
      << "    \"dropout_rate\": " << json_double(static_cast<double>(c.dropout_rate)) << ",\n"
      << "    \"embedding_dim\": " << c.embedding_dim << ",\n"
      << "    \"image_size\": " << c.image_size << ",\n"
      << "    \"in_channels\": " << c.in_channels << ",\n"
      << "    \"mlp_hidden_dim\": " << c.mlp_hidden_dim << ",\n"
      << "    \"num_classes\": " << c.num_classes << ",\n"
      << "    \"num_heads\": " << c.num_heads << ",\n"
      << "    \"num_layers\": " << c.num_layers << ",\n"
      << "    \"patch_size\": " << c.patch_size << "\n"
      << "}" << std::endl;
  if (g_verbose) std::cout << "Configuración guardada correctamente." << std::endl;
}

ViTConfig load_config(const std::string &filePath) {
  if (g_verbose) std::cout << "Cargando configuración del modelo desde: " << filePath << "..." << std::endl;
  std::ifstream in(filePath);
  if (!in) throw std::runtime_error("No se pudo abrir el archivo para lectura: " + filePath);
  std::stringstream ss;
  ss << in.rdbuf();
  const auto kv = parse_flat_json(ss.str(), filePath);
  auto need = [&](const char *k) {
    auto it = kv.find(k);
    if (it == kv.end()) throw std::runtime_error(std::string("Falta la clave '") + k + "' en " + filePath);
    return it->second;
  };
  ViTConfig c;
  c.image_size = (size_t)need("image_size");
  c.patch_size = (size_t)need("patch_size");
  c.in_channels = (size_t)need("in_channels");
  c.num_classes = (size_t)need("num_classes");
  c.embedding_dim = (size_t)need("embedding_dim");
  c.num_heads = (size_t)need("num_heads");
  c.num_layers = (size_t)need("num_layers");
  c.mlp_hidden_dim = (size_t)need("mlp_hidden_dim");
  c.dropout_rate = (float)need("dropout_rate");
  if (g_verbose) std::cout << "Configuración cargada correctamente." << std::endl;
  return c;
}

void print_hyperparameters_box(const ViTConfig &m, const TrainerConfig &t) {
  const int name_width = 30, value_width = 20;
  auto row = [&](const std::string &name, const std::string &value) {
    std::cout << "║ " << std::left << std::setw(name_width) << name << " : " << std::right << std::setw(value_width)
              << value << " ║\n";
  };
  const std::string border(name_width + value_width + 5, '=');
  std::cout << "\n╔" << border << "╗\n";
  std::cout << "║" << std::setw(name_width + value_width + 5) << std::left
            << "            CONFIGURACIÓN DE HIPERPARÁMETROS           " << "║\n";
  std::cout << "╠" << border << "╣\n";
  row("Embedding dim", std::to_string(m.embedding_dim));
  row("Nro de capas", std::to_string(m.num_layers));
  row("Nro de heads", std::to_string(m.num_heads));
  row("Patch size", std::to_string(m.patch_size));
  row("Nro de clases", std::to_string(m.num_classes));
  row("Canales de entrada", std::to_string(m.in_channels));
  row("Dim MLP oculta", std::to_string(m.mlp_hidden_dim));
  row("Dropout rate", std::to_string(m.dropout_rate));
  std::cout << "╠" << border << "╣\n";
  row("Epocas   ", std::to_string(t.epochs));
  row("Batch size", std::to_string(t.batch_size));
  row("Learning rate", std::to_string(t.learning_rate));
  row("Weight decay", std::to_string(t.weight_decay));
  row("Warmup fraction", std::to_string(t.warmup_frac));
  std::cout << "╚" << border << "╝\n";
}

}  // namespace ModelUtils
