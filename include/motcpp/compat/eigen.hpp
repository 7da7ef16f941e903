// Minimal stand-in for the two Eigen types motcpp's public tracker API exposes
// (include/motcpp/tracker.hpp:67-69 of the reference: Eigen::MatrixXf in, Eigen::MatrixXf out).
// When a real Eigen is installed it wins; otherwise this header provides a column-major dynamic
// float matrix with the members user code of the reference touches (rows/cols/operator()/data/
// resize/setZero/Zero and the `m << a, b, c` comma initialiser). It is NOT a linear-algebra
// library: all numerics of the hot path run in the HIP kernels.
#pragma once
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#else
#include <cstddef>
#include <vector>
namespace Eigen {
using Index = std::ptrdiff_t;
class MatrixXf {
 public:
  MatrixXf() = default;
  MatrixXf(Index rows, Index cols) : r_(rows), c_(cols), a_(static_cast<size_t>(rows * cols), 0.0f) {}
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Index size() const { return r_ * c_; }
  float& operator()(Index i, Index j) { return a_[static_cast<size_t>(j * r_ + i)]; }
  float operator()(Index i, Index j) const { return a_[static_cast<size_t>(j * r_ + i)]; }
  float* data() { return a_.data(); }
  const float* data() const { return a_.data(); }
  void resize(Index rows, Index cols) { r_ = rows; c_ = cols; a_.assign(static_cast<size_t>(rows * cols), 0.0f); }
  void setZero() { a_.assign(a_.size(), 0.0f); }
  static MatrixXf Zero(Index rows, Index cols) { return MatrixXf(rows, cols); }
  // row-major fill, like Eigen's comma initialiser
  class Comma {
   public:
    Comma(MatrixXf& m, float first) : m_(m) { put(first); }
    Comma& operator,(float v) { put(v); return *this; }
   private:
    void put(float v) { const Index i = k_ / m_.cols(), j = k_ % m_.cols(); if (i < m_.rows()) m_(i, j) = v; ++k_; }
    MatrixXf& m_;
    Index k_ = 0;
  };
  Comma operator<<(float first) { return Comma(*this, first); }
 private:
  Index r_ = 0, c_ = 0;
  std::vector<float> a_;
};
}  // namespace Eigen
#endif
