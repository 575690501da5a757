// Tensor: the reference's N-d float tensor (include/core/Tensor.hpp:23-98 of the reference) with
// device-resident storage.  Same public surface -- constructors, views (slice / reshape /
// transpose / expand), element access, arithmetic helpers, matrixMultiply / batchMatrixMultiply /
// concatenate / expand free functions -- so code written against the reference compiles unchanged.
//
// Storage model: a ref-counted block that lives on the GPU and has a lazily synchronised host
// mirror.  Kernels read/write the device copy; operator() / getData() / getDataPtr() touch the
// host mirror and trigger a download (and, for mutable access, mark the device copy stale so the
// next kernel re-uploads).  Host reads therefore look synchronous, as they are in the reference.
#ifndef TENSOR_HPP
#define TENSOR_HPP

#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

class Tensor;
Tensor concatenate(const std::vector<Tensor> &tensors, size_t axis);
Tensor expand(const Tensor &tensor, size_t dim, size_t size);

namespace vit {

// Ref-counted device block + host mirror.
struct Storage {
  size_t count = 0;
  float *dev = nullptr;
  bool owns_dev = true;
  std::shared_ptr<std::vector<float>> host;  // sized `count` once materialised
  bool host_valid = false;
  bool dev_valid = false;

  explicit Storage(size_t n) : count(n) {}
  ~Storage();
  Storage(const Storage &) = delete;
  Storage &operator=(const Storage &) = delete;

  float *host_read() { return host_valid ? host->data() : host_sync(false); }       // host mirror valid afterwards
  float *host_write() { return (host_valid && !dev_valid) ? host->data() : host_sync(true); }  // device copy stale
  float *host_sync(bool will_write);
  const float *dev_read();    // device copy valid afterwards
  float *dev_write();         // device copy valid (uploaded first), host mirror stale
  float *dev_overwrite();     // device copy will be fully overwritten: no upload, host mirror stale
};

}  // namespace vit

class Tensor {
public:
  Tensor();
  explicit Tensor(const std::vector<size_t> &shape);  // zero-initialised
  Tensor(const std::vector<size_t> &shape, const std::vector<float> &data);
  Tensor(const Tensor &other) = default;
  Tensor(Tensor &&other) noexcept = default;
  Tensor &operator=(const Tensor &other) = default;
  Tensor &operator=(Tensor &&other) noexcept = default;
  ~Tensor() = default;

  // element access (host mirror)
  float &operator()(size_t i);
  const float &operator()(size_t i) const;
  float &operator()(size_t i, size_t j);
  const float &operator()(size_t i, size_t j) const;
  float &operator()(size_t i, size_t j, size_t k);
  const float &operator()(size_t i, size_t j, size_t k) const;
  float &operator()(size_t d0, size_t d1, size_t d2, size_t d3);
  const float &operator()(size_t d0, size_t d1, size_t d2, size_t d3) const;

  // views and operations
  Tensor slice(size_t axis, size_t start, size_t count) const;
  Tensor reshape(const std::vector<size_t> &newShape) const;
  Tensor transpose(size_t dim1, size_t dim2) const;
  Tensor square() const;
  Tensor sum(size_t axis) const;
  void addBroadcast(const Tensor &other);
  Tensor contiguous() const;
  Tensor expand(const std::vector<size_t> &newShape) const;
  void copyFrom(const Tensor &src);
  Tensor operator+(const Tensor &other) const;

  void fill(float value);
  void randomize(float min = -1.0f, float max = 1.0f);
  void randomizeNormal(float mean = 0.0f, float stddev = 1.0f);

  const std::vector<size_t> &getShape() const { return shape; }
  size_t getSize() const { return totalSize; }
  const std::vector<size_t> &getStrides() const { return strides; }
  size_t getDataOffset() const { return dataOffset; }
  // Host mirror of the whole storage block (synchronised first), as in the reference.
  const std::shared_ptr<std::vector<float>> &getDataPtr() const;
  // BASE pointer of the storage block on the host (the view offset is NOT applied, exactly like the
  // reference: src/core/Tensor.cpp:80-91).  data() below is the start of this view.
  const float *getData() const;
  float *getData();
  std::string shapeToString() const;
  bool isContiguous() const;
  Tensor clone() const;
  void printDebugInfo(const std::string &name) const;

  // view constructor over an existing host vector (public in the reference too)
  Tensor(std::shared_ptr<std::vector<float>> dataPtr, const std::vector<size_t> &shape,
         const std::vector<size_t> &strides, size_t offset);

  // ---- device-side extensions -----------------------------------------------------------------
  // Uninitialised device-resident tensor (for kernel outputs).
  static Tensor deviceEmpty(const std::vector<size_t> &shape);
  // View over storage owned elsewhere (parameter / gradient arenas).
  Tensor(std::shared_ptr<vit::Storage> storage, const std::vector<size_t> &shape, const std::vector<size_t> &strides,
         size_t offset);
  // Non-owning tensor over caller-provided device memory (count = product of shape).
  static Tensor wrapDevice(float *dev, const std::vector<size_t> &shape);
  // Start of THIS VIEW (offset applied) on the host / on the device.
  const float *data() const;
  float *data();
  const float *deviceData() const;  // read
  float *deviceData();              // read-modify-write
  float *deviceDataOverwrite();     // whole storage block will be overwritten by the caller's kernels
  const std::shared_ptr<vit::Storage> &storage() const { return store; }
  bool defined() const { return static_cast<bool>(store); }
  // same data, contiguous row-major, offset allowed (what kernels need); copies only if strided
  Tensor dense() const;

private:
  void computeStrides();
  std::shared_ptr<vit::Storage> store;
  std::vector<size_t> shape;
  std::vector<size_t> strides;
  size_t dataOffset;
  size_t totalSize;
};

Tensor matrixMultiply(const Tensor &a, const Tensor &b);
Tensor batchMatrixMultiply(const Tensor &a, const Tensor &b);

// ---- inline element access ----------------------------------------------------------------------
inline float &Tensor::operator()(size_t i) {
#ifndef NDEBUG
  if (shape.size() != 1 || i >= shape[0]) throw std::out_of_range("Acceso 1D fuera de rango.");
#endif
  return store->host_write()[dataOffset + i * strides[0]];
}
inline const float &Tensor::operator()(size_t i) const {
#ifndef NDEBUG
  if (shape.size() != 1 || i >= shape[0]) throw std::out_of_range("Acceso 1D fuera de rango.");
#endif
  return store->host_read()[dataOffset + i * strides[0]];
}
inline float &Tensor::operator()(size_t i, size_t j) {
#ifndef NDEBUG
  if (shape.size() != 2 || i >= shape[0] || j >= shape[1]) throw std::out_of_range("Acceso 2D fuera de rango.");
#endif
  return store->host_write()[dataOffset + i * strides[0] + j * strides[1]];
}
inline const float &Tensor::operator()(size_t i, size_t j) const {
#ifndef NDEBUG
  if (shape.size() != 2 || i >= shape[0] || j >= shape[1]) throw std::out_of_range("Acceso 2D fuera de rango.");
#endif
  return store->host_read()[dataOffset + i * strides[0] + j * strides[1]];
}
inline float &Tensor::operator()(size_t i, size_t j, size_t k) {
#ifndef NDEBUG
  if (shape.size() != 3 || i >= shape[0] || j >= shape[1] || k >= shape[2])
    throw std::out_of_range("Acceso 3D fuera de rango.");
#endif
  return store->host_write()[dataOffset + i * strides[0] + j * strides[1] + k * strides[2]];
}
inline const float &Tensor::operator()(size_t i, size_t j, size_t k) const {
#ifndef NDEBUG
  if (shape.size() != 3 || i >= shape[0] || j >= shape[1] || k >= shape[2])
    throw std::out_of_range("Acceso 3D fuera de rango.");
#endif
  return store->host_read()[dataOffset + i * strides[0] + j * strides[1] + k * strides[2]];
}
inline float &Tensor::operator()(size_t d0, size_t d1, size_t d2, size_t d3) {
#ifndef NDEBUG
  if (shape.size() != 4 || d0 >= shape[0] || d1 >= shape[1] || d2 >= shape[2] || d3 >= shape[3])
    throw std::out_of_range("Acceso 4D fuera de rango.");
#endif
  return store->host_write()[dataOffset + d0 * strides[0] + d1 * strides[1] + d2 * strides[2] + d3 * strides[3]];
}
inline const float &Tensor::operator()(size_t d0, size_t d1, size_t d2, size_t d3) const {
#ifndef NDEBUG
  if (shape.size() != 4 || d0 >= shape[0] || d1 >= shape[1] || d2 >= shape[2] || d3 >= shape[3])
    throw std::out_of_range("Acceso 4D fuera de rango.");
#endif
  return store->host_read()[dataOffset + d0 * strides[0] + d1 * strides[1] + d2 * strides[2] + d3 * strides[3]];
}

#endif  // TENSOR_HPP
