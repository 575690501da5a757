// Device-resident implementation of the reference's Tensor (src/core/Tensor.cpp of the reference).
// Arithmetic helpers run as libvitk kernels; strided views are densified on the device first
// (vitk_copy4d_f32), so no arithmetic happens on the host.
#include "core/Tensor.hpp"

#include <algorithm>
#include <cstring>
#include <iostream>
#include <random>
#include <sstream>

#include "core/Device.hpp"
#include "vitk.h"

namespace vit {

Storage::~Storage() {
  if (dev && owns_dev) device_free(dev, count);
}

float *Storage::host_sync(bool will_write) {
  if (!host) host = std::make_shared<std::vector<float>>(count, 0.0f);
  if (!host_valid) {
    if (dev_valid && count) copy_to_host(host->data(), dev, count);
    host_valid = true;
  }
  if (will_write) dev_valid = false;
  return host->data();
}

const float *Storage::dev_read() {
  if (!dev) dev = device_alloc(count);
  if (!dev_valid) {
    if (host_valid && count) copy_to_device(dev, host->data(), count);
    else if (count) zero_device(dev, count);
    dev_valid = true;
  }
  return dev;
}

float *Storage::dev_write() {
  dev_read();
  host_valid = false;
  return dev;
}

float *Storage::dev_overwrite() {
  if (!dev) dev = device_alloc(count);
  dev_valid = true;
  host_valid = false;
  return dev;
}

}  // namespace vit

namespace {

size_t product(const std::vector<size_t> &s) {
  if (s.empty()) return 0;
  size_t n = 1;
  for (size_t v : s) n *= v;
  return n;
}

std::vector<size_t> row_major(const std::vector<size_t> &shape) {
  std::vector<size_t> st(shape.size());
  size_t stride = 1;
  for (int i = (int)shape.size() - 1; i >= 0; --i) {
    st[i] = stride;
    stride *= shape[i];
  }
  return st;
}

void *S() { return vit::stream(); }

}  // namespace

void Tensor::computeStrides() { strides = row_major(shape); }

Tensor::Tensor() : dataOffset(0), totalSize(0) {}

Tensor::Tensor(const std::vector<size_t> &newShape) : shape(newShape), dataOffset(0) {
  totalSize = product(newShape);
  store = std::make_shared<vit::Storage>(totalSize);
  // zero-initialised: materialise lazily on whichever side is touched first
  computeStrides();
}

Tensor::Tensor(const std::vector<size_t> &newShape, const std::vector<float> &initialData)
    : shape(newShape), dataOffset(0) {
  totalSize = newShape.empty() ? 1 : product(newShape);
  if (totalSize != initialData.size())
    throw std::invalid_argument("El tamaño de los datos iniciales no coincide con la forma del tensor.");
  store = std::make_shared<vit::Storage>(totalSize);
  store->host = std::make_shared<std::vector<float>>(initialData);
  store->host_valid = true;
  computeStrides();
}

Tensor::Tensor(std::shared_ptr<std::vector<float>> ptr, const std::vector<size_t> &newShape,
               const std::vector<size_t> &newStrides, size_t offset)
    : shape(newShape), strides(newStrides), dataOffset(offset) {
  totalSize = product(shape);
  store = std::make_shared<vit::Storage>(ptr ? ptr->size() : 0);
  store->host = std::move(ptr);
  store->host_valid = true;
}

Tensor::Tensor(std::shared_ptr<vit::Storage> storage, const std::vector<size_t> &newShape,
               const std::vector<size_t> &newStrides, size_t offset)
    : store(std::move(storage)), shape(newShape), strides(newStrides), dataOffset(offset) {
  totalSize = product(shape);
}

Tensor Tensor::deviceEmpty(const std::vector<size_t> &shape) {
  Tensor t;
  t.shape = shape;
  t.totalSize = product(shape);
  t.computeStrides();
  t.store = std::make_shared<vit::Storage>(t.totalSize);
  t.store->dev_overwrite();
  return t;
}

Tensor Tensor::wrapDevice(float *dev, const std::vector<size_t> &shape) {
  Tensor t;
  t.shape = shape;
  t.totalSize = product(shape);
  t.computeStrides();
  t.store = std::make_shared<vit::Storage>(t.totalSize);
  t.store->dev = dev;
  t.store->owns_dev = false;
  t.store->dev_valid = true;
  return t;
}

const std::shared_ptr<std::vector<float>> &Tensor::getDataPtr() const {
  static const std::shared_ptr<std::vector<float>> null;
  if (!store) return null;
  store->host_read();
  return store->host;
}

float *Tensor::getData() { return store ? store->host_write() : nullptr; }
const float *Tensor::getData() const { return store ? store->host_read() : nullptr; }
float *Tensor::data() { return store ? store->host_write() + dataOffset : nullptr; }
const float *Tensor::data() const { return store ? store->host_read() + dataOffset : nullptr; }
const float *Tensor::deviceData() const { return store ? store->dev_read() + dataOffset : nullptr; }
float *Tensor::deviceData() { return store ? store->dev_write() + dataOffset : nullptr; }
float *Tensor::deviceDataOverwrite() {
  if (!store) return nullptr;
  // only a tensor that covers its whole storage block may skip the upload
  return (totalSize == store->count ? store->dev_overwrite() : store->dev_write()) + dataOffset;
}

bool Tensor::isContiguous() const {
  if (shape.empty()) return true;
  size_t stride = 1;
  for (int i = (int)shape.size() - 1; i >= 0; --i) {
    if (strides[i] != stride) return false;
    stride *= shape[i];
  }
  return true;
}

std::string Tensor::shapeToString() const {
  if (shape.empty()) return "()";
  std::stringstream ss;
  ss << "(";
  for (size_t i = 0; i < shape.size(); ++i) ss << shape[i] << (i == shape.size() - 1 ? "" : ", ");
  ss << ")";
  return ss.str();
}

void Tensor::fill(float value) {
  if (!isContiguous()) throw std::runtime_error("fill() solo se puede usar en tensores contiguos.");
  if (!store || totalSize == 0) return;
  vit::check(vitk_fill_f32(deviceDataOverwrite(), value, (int64_t)totalSize, S()), "fill");
}

void Tensor::randomize(float min, float max) {
  if (!isContiguous()) throw std::runtime_error("randomize() solo se puede usar en tensores contiguos.");
  if (!store) return;
  std::mt19937 gen(static_cast<uint32_t>(vit::next_seed()));
  std::uniform_real_distribution<float> dis(min, max);
  float *p = data();
  std::generate(p, p + totalSize, [&]() { return dis(gen); });
}

void Tensor::randomizeNormal(float mean, float stddev) {
  if (!isContiguous()) throw std::runtime_error("randomizeNormal() solo se puede usar en tensores contiguos.");
  if (!store) return;
  std::mt19937 gen(static_cast<uint32_t>(vit::next_seed()));
  std::normal_distribution<float> dis(mean, stddev);
  float *p = data();
  std::generate(p, p + totalSize, [&]() { return dis(gen); });
}

Tensor Tensor::slice(size_t axis, size_t start, size_t count) const {
  if (axis >= shape.size()) throw std::out_of_range("Eje de slice fuera de rango.");
  if (start + count > shape[axis])
    throw std::out_of_range("Slice fuera de los límites de la dimensión " + std::to_string(axis));
  std::vector<size_t> newShape = shape;
  newShape[axis] = count;
  return Tensor(store, newShape, strides, dataOffset + start * strides[axis]);
}

Tensor Tensor::reshape(const std::vector<size_t> &newShape) const {
  if (!isContiguous())
    throw std::runtime_error("reshape() solo se puede usar en un tensor contiguo. Use .contiguous() primero.");
  if (totalSize != product(newShape))
    throw std::runtime_error("No se puede hacer reshape: el número total de elementos debe ser el mismo.");
  return Tensor(store, newShape, row_major(newShape), dataOffset);
}

Tensor Tensor::transpose(size_t dim1, size_t dim2) const {
  if (dim1 >= shape.size() || dim2 >= shape.size()) throw std::out_of_range("Ejes para transpose fuera de rango.");
  std::vector<size_t> newShape = shape, newStrides = strides;
  std::swap(newShape[dim1], newShape[dim2]);
  std::swap(newStrides[dim1], newStrides[dim2]);
  return Tensor(store, newShape, newStrides, dataOffset);
}

Tensor Tensor::dense() const {
  if (isContiguous()) return *this;
  if (shape.size() > 4) throw std::runtime_error("contiguous() no implementado para este rank.");
  Tensor out = deviceEmpty(shape);
  int64_t sh[4] = {1, 1, 1, 1}, st[4] = {0, 0, 0, 0};
  const size_t pad = 4 - shape.size();
  for (size_t i = 0; i < shape.size(); ++i) {
    sh[pad + i] = (int64_t)shape[i];
    st[pad + i] = (int64_t)strides[i];
  }
  vit::check(vitk_copy4d_f32(deviceData(), sh, st, out.deviceDataOverwrite(), S()), "contiguous");
  return out;
}

Tensor Tensor::contiguous() const {
  // same contract as the reference (src/core/Tensor.cpp:227-254): an offset view is copied too
  if (isContiguous() && dataOffset == 0) return *this;
  if (!isContiguous()) return dense();
  Tensor out = deviceEmpty(shape);
  vit::copy_on_device(out.deviceDataOverwrite(), deviceData(), totalSize);
  return out;
}

Tensor Tensor::expand(const std::vector<size_t> &newShape) const {
  if (newShape.size() != shape.size()) throw std::invalid_argument("expand: la dimensionalidad no coincide.");
  std::vector<size_t> newStrides = strides;
  for (size_t d = 0; d < shape.size(); ++d) {
    if (newShape[d] == shape[d]) continue;
    if (shape[d] != 1) throw std::invalid_argument("expand: solo se puede expandir dimensiones de tamaño 1.");
    newStrides[d] = 0;
  }
  return Tensor(store, newShape, newStrides, dataOffset);
}

void Tensor::copyFrom(const Tensor &src) {
  if (shape != src.shape) throw std::invalid_argument("copyFrom: las shapes no coinciden.");
  if (totalSize == 0) return;
  Tensor s = src.dense();
  if (isContiguous()) {
    vit::copy_on_device(deviceData(), s.deviceData(), totalSize);
    return;
  }
  // strided destination: element-wise through the host mirror (rare; not on the training path)
  const float *sp = s.data();
  float *base = getData();
  std::vector<size_t> idx(shape.size(), 0);
  for (size_t lin = 0; lin < totalSize; ++lin) {
    size_t off = dataOffset;
    for (size_t d = 0; d < shape.size(); ++d) off += idx[d] * strides[d];
    base[off] = sp[lin];
    for (int d = (int)shape.size() - 1; d >= 0; --d) {
      if (++idx[d] < shape[d]) break;
      idx[d] = 0;
    }
  }
}

Tensor Tensor::operator+(const Tensor &other) const {
  if (shape != other.getShape())
    throw std::invalid_argument("Los tensores deben tener la misma forma para la suma. " + shapeToString() + " vs " +
                                other.shapeToString());
  Tensor a = dense(), b = other.dense();
  Tensor out = deviceEmpty(shape);
  vit::check(vitk_add_f32(a.deviceData(), b.deviceData(), (int64_t)totalSize, out.deviceDataOverwrite(), S()), "operator+");
  return out;
}

Tensor Tensor::square() const {
  Tensor a = dense();
  Tensor out = deviceEmpty(shape);
  // x*x == dy*mask form of the element-wise multiply kernel
  vit::check(vitk_dropout_bwd(a.deviceData(), a.deviceData(), (int64_t)totalSize, out.deviceDataOverwrite(), S()),
             "square");
  return out;
}

Tensor Tensor::sum(size_t axis) const {
  if (axis >= shape.size()) throw std::out_of_range("Eje para sum() fuera de rango.");
  if (shape.size() != 2 && shape.size() != 3)
    throw std::runtime_error("sum() solo está implementado para 2D y 3D por ahora.");
  std::vector<size_t> outShape = shape;
  outShape[axis] = 1;
  // View with the reduced axis first and the other axes in their original order: a column sum of
  // [R][rest] then already has the memory layout of the result.
  std::vector<size_t> pShape{shape[axis]}, pStrides{strides[axis]};
  for (size_t i = 0; i < shape.size(); ++i)
    if (i != axis) {
      pShape.push_back(shape[i]);
      pStrides.push_back(strides[i]);
    }
  Tensor src = Tensor(store, pShape, pStrides, dataOffset).dense();
  const size_t R = shape[axis], rest = R ? totalSize / R : 0;
  Tensor out = deviceEmpty(outShape);
  if (rest == 0) return out;
  const size_t wsb = vitk_colsum_workspace_bytes((int64_t)R, (int64_t)rest);
  vit::check(vitk_colsum_f32(src.deviceData(), (int64_t)R, (int64_t)rest, (int64_t)rest, out.deviceDataOverwrite(),
                             vit::workspace(wsb), wsb, S()),
             "sum");
  return out;
}

void Tensor::addBroadcast(const Tensor &other) {
  const auto &os = other.getShape();
  const bool case2d = shape.size() == 2 && os.size() == 2 && os[0] == 1 && shape[1] == os[1];
  const bool case3d = shape.size() == 3 && os.size() == 3 && os[0] == 1 && shape[1] == os[1] && shape[2] == os[2];
  if (!case2d && !case3d)
    throw std::runtime_error("Broadcasting no implementado para estas formas: " + shapeToString() + " y " +
                             other.shapeToString());
  if (!isContiguous()) throw std::runtime_error("addBroadcast: el tensor destino debe ser contiguo.");
  const size_t rows = shape[0], cols = totalSize / (rows ? rows : 1);
  Tensor o = other.dense();
  vit::check(vitk_add_rowvec_f32(deviceData(), (int64_t)rows, (int64_t)cols, o.deviceData(), S()), "addBroadcast");
}

Tensor Tensor::clone() const {
  Tensor d = dense();
  Tensor out = deviceEmpty(shape);
  vit::copy_on_device(out.deviceDataOverwrite(), d.deviceData(), totalSize);
  return out;
}

void Tensor::printDebugInfo(const std::string &name) const {
  std::cout << "--- Tensor Debug: " << name << " ---" << std::endl;
  std::cout << "  Forma: " << shapeToString() << std::endl;
  std::cout << "  Contiguo: " << (isContiguous() ? "Sí" : "NO") << std::endl;
  std::cout << "  Offset: " << dataOffset << std::endl;
  std::cout << "  Strides: ";
  for (const auto &s : strides) std::cout << s << " ";
  std::cout << std::endl << "-------------------------" << std::endl;
}

namespace {

// Describe a 2-D view as a GEMM operand without copying when one of its strides is 1.
struct Operand2D {
  Tensor keep;
  const float *ptr;
  int64_t ld;
  bool trans;
};

Operand2D operand(const Tensor &t, size_t r, size_t c) {
  // t is the 2-D matrix given by its last two axes r, c of extents shape[r] x shape[c]
  const auto &st = t.getStrides();
  const auto &sh = t.getShape();
  Operand2D o;
  if (st[c] == 1 && st[r] >= sh[c]) {
    o.keep = t; o.ld = (int64_t)st[r]; o.trans = false;
  } else if (st[r] == 1 && st[c] >= sh[r]) {
    o.keep = t; o.ld = (int64_t)st[c]; o.trans = true;
  } else {
    o.keep = t.dense(); o.ld = (int64_t)sh[c]; o.trans = false;
  }
  o.ptr = o.keep.deviceData();
  return o;
}

}  // namespace

Tensor matrixMultiply(const Tensor &a, const Tensor &b) {
  const auto &as = a.getShape();
  const auto &bs = b.getShape();
  if (as.size() != 2 || bs.size() != 2) throw std::runtime_error("matrixMultiply solo está implementada para tensores 2D.");
  if (as[1] != bs[0])
    throw std::runtime_error("Dimensiones de matriz incompatibles para la multiplicación: " + a.shapeToString() + " y " +
                             b.shapeToString());
  Operand2D A = operand(a, 0, 1), B = operand(b, 0, 1);
  Tensor out = Tensor::deviceEmpty({as[0], bs[1]});
  vitk_gemm_args g{};
  g.trans_a = A.trans; g.trans_b = B.trans;
  g.M = (int32_t)as[0]; g.N = (int32_t)bs[1]; g.K = (int32_t)as[1];
  g.A = A.ptr; g.lda = A.ld; g.B = B.ptr; g.ldb = B.ld;
  g.C = out.deviceDataOverwrite(); g.ldc = (int64_t)bs[1];
  g.alpha = 1.0f; g.split_k = 0;
  const size_t wsb = vitk_gemm_workspace_bytes(&g);
  g.workspace = wsb ? vit::workspace(wsb) : nullptr;
  g.workspace_bytes = wsb;
  vit::check(vitk_gemm_f32(&g, S()), "matrixMultiply");
  return out;
}

Tensor batchMatrixMultiply(const Tensor &a, const Tensor &b) {
  const auto &as = a.getShape();
  const auto &bs = b.getShape();
  if (as.size() != 3 || bs.size() != 3) throw std::runtime_error("BMM solo está implementado para tensores 3D.");
  if (as[0] != bs[0]) throw std::runtime_error("El tamaño del batch debe ser el mismo para ambos tensores en BMM.");
  if (as[2] != bs[1])
    throw std::runtime_error("Dimensiones de matriz incompatibles para BMM: " + a.shapeToString() + " y " +
                             b.shapeToString());
  Operand2D A = operand(a, 1, 2), B = operand(b, 1, 2);
  const int64_t sa = A.keep.getStrides()[0], sb = B.keep.getStrides()[0];
  Tensor out = Tensor::deviceEmpty({as[0], as[1], bs[2]});
  vitk_gemm_args g{};
  g.trans_a = A.trans; g.trans_b = B.trans;
  g.M = (int32_t)as[1]; g.N = (int32_t)bs[2]; g.K = (int32_t)as[2];
  g.A = A.ptr; g.lda = A.ld; g.B = B.ptr; g.ldb = B.ld;
  g.C = out.deviceDataOverwrite(); g.ldc = (int64_t)bs[2];
  g.alpha = 1.0f; g.split_k = 1;
  g.batch_outer = (int32_t)as[0]; g.batch_inner = 1;
  g.stride_a_outer = sa; g.stride_b_outer = sb; g.stride_c_outer = (int64_t)(as[1] * bs[2]);
  vit::check(vitk_gemm_f32(&g, S()), "batchMatrixMultiply");
  return out;
}

Tensor concatenate(const std::vector<Tensor> &tensors, size_t axis) {
  if (tensors.empty()) return Tensor();
  if (tensors.size() == 1) return tensors[0];
  const auto &first = tensors[0].getShape();
  size_t newDim = 0;
  for (const auto &t : tensors) {
    if (t.getShape().size() != first.size() || t.getShape().size() <= axis)
      throw std::invalid_argument("Todos los tensores deben tener el mismo rank y ser compatibles con el eje.");
    for (size_t i = 0; i < first.size(); ++i)
      if (i != axis && t.getShape()[i] != first[i])
        throw std::invalid_argument("Las dimensiones deben ser iguales excepto en el eje de concatenación.");
    newDim += t.getShape()[axis];
  }
  if (first.size() != 3) throw std::runtime_error("Concatenate solo implementado para 3D por ahora.");
  std::vector<size_t> newShape = first;
  newShape[axis] = newDim;
  Tensor result = Tensor::deviceEmpty(newShape);
  float *dst = result.deviceDataOverwrite();
  // view every piece as [outer][inner] rows and copy with a pitched 2-D copy
  size_t outer = 1, inner_per = 1;
  for (size_t i = 0; i < axis; ++i) outer *= first[i];
  for (size_t i = axis + 1; i < first.size(); ++i) inner_per *= first[i];
  const size_t row = newDim * inner_per;
  size_t off = 0;
  for (const auto &t : tensors) {
    Tensor d = t.dense();
    const size_t cols = t.getShape()[axis] * inner_per;
    vit::check(vitk_copy2d_f32(d.deviceData(), (int64_t)cols, dst + off, (int64_t)row, (int64_t)outer, (int64_t)cols,
                               S()),
               "concatenate");
    off += cols;
  }
  return result;
}

Tensor expand(const Tensor &tensor, size_t dim, size_t size) {
  if (dim > tensor.getShape().size())
    throw std::invalid_argument("La dimensión para expandir es mayor que el rank del tensor.");
  std::vector<size_t> newShape = tensor.getShape();
  newShape.insert(newShape.begin() + dim, size);
  std::vector<size_t> newStrides = tensor.getStrides();
  newStrides.insert(newStrides.begin() + dim, 0);
  return Tensor(tensor.storage(), newShape, newStrides, tensor.getDataOffset());
}
