// COMPILE-CHECK STAND-IN, tests only.  A minimal model of the OpenCV types that
// okvis2_amd/host/okvfe_opencv_adapters.hpp touches, so that the adapters can be syntax- and
// type-checked in a container without OpenCV (tests/test_host_adapters_compile.py).  It is NOT
// OpenCV, implements nothing beyond what the checks call, and is never used to build or stand in
// for any part of the reference.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC(n) CV_MAKETYPE(CV_32F, (n))
#define CV_32FC3 CV_32FC(3)
#define CV_Assert(expr) \
  do {                  \
    if (!(expr)) throw ::cv::Exception(#expr); \
  } while (0)

namespace cv {
typedef unsigned char uchar;
struct Exception {
  explicit Exception(const char* w) : what(w) {}
  const char* what;
};
struct Point2f {
  float x = 0, y = 0;
};
struct KeyPoint {  // same field order and size as the real one (28 bytes)
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};
template <typename T, int N>
struct Vec {
  T val[N];
  Vec(T a, T b, T c) : val{a, b, c} {}
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<float, 3> Vec3f;

class Mat {
 public:
  Mat() = default;
  Mat(int r, int c, int t) { create(r, c, t); }
  Mat(int r, int c, int t, void* d) : data(static_cast<uchar*>(d)), rows(r), cols(c), type_(t) {
    step[0] = size_t(c) * elemSize();
  }
  void create(int r, int c, int t) {
    rows = r; cols = c; type_ = t;
    store_.assign(size_t(r) * c * elemSize(), 0);
    data = store_.data();
    step[0] = size_t(c) * elemSize();
  }
  int type() const { return type_; }
  size_t elemSize() const { return ((type_ & 7) == CV_32F ? 4u : 1u) * size_t((type_ >> 3) + 1); }
  bool isContinuous() const { return true; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + size_t(r) * step[0]); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + size_t(r) * step[0]); }
  uchar* data = nullptr;
  int rows = 0, cols = 0;
  size_t step[2] = {0, 0};

 private:
  int type_ = 0;
  std::vector<uchar> store_;
};

class _InputArray {
 public:
  _InputArray() = default;
  _InputArray(const Mat& m) : m_(&m) {}  // NOLINT: implicit like the real one
  Mat getMat() const { return m_ ? Mat(m_->rows, m_->cols, m_->type(), m_->data) : Mat(); }

 private:
  const Mat* m_ = nullptr;
};
class _OutputArray {
 public:
  _OutputArray(Mat& m) : m_(&m) {}  // NOLINT
  void create(int r, int c, int t) const { m_->create(r, c, t); }
  Mat getMat() const { return Mat(m_->rows, m_->cols, m_->type(), m_->data); }

 private:
  Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline const _InputArray& noArray() {
  static _InputArray none;
  return none;
}
}  // namespace cv
