// tests/stubs/opencv2/core.hpp -- COMPILE-TEST STAND-IN, not OpenCV.
//
// Purpose: let tests/cpp/test_adapter.cpp type-check the drop-in adapter `MofFftMethod : OpticFlowCalc`
// (include/mof/processors.hpp) against the reference's real interface header
// (/root/reference/include/OpticFlowCalc.h, included read-only via -I) in an image that has no OpenCV.
// It declares exactly the few cv:: types that header and the adapter touch -- a CV_8UC1 matrix header
// (data / rows / cols / step / type()), Point_<T>, Scalar, CV_Assert -- with OpenCV's member names and
// value semantics (a Mat copy shares pixel data, as cv::Mat's ref-counted header does).
// It contains no algorithm, builds no reference translation unit and is never linked into the product
// or the oracle; on a machine with OpenCV this directory is simply not on the include path.
#pragma once

#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32FC1 5
#define CV_Assert(expr)                                                        \
  do {                                                                         \
    if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr);         \
  } while (0)

namespace cv {

typedef unsigned char uchar;

template <typename T>
struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  template <typename U>
  Point_(const Point_<U>& o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)) {}  // cv::Point -> cv::Point2d, as in OpenCV
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;

struct Scalar {
  double val[4];
  Scalar(double v0 = 0) : val{v0, 0, 0, 0} {}
};

struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
};

// CV_8UC1 only. Copies share the pixels (shared_ptr stands in for OpenCV's reference count).
class Mat {
 public:
  uchar* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;

  Mat() = default;
  Mat(int r, int c, int type) : rows(r), cols(c), step((size_t)c), type_(type) {
    CV_Assert(type == CV_8UC1);
    own_ = std::shared_ptr<uchar>(new uchar[(size_t)r * c], std::default_delete<uchar[]>());
    data = own_.get();
  }
  // header over user data (no copy), as cv::Mat(rows, cols, type, data, step)
  Mat(int r, int c, int type, void* d, size_t s = 0) : data((uchar*)d), rows(r), cols(c), step(s ? s : (size_t)c), type_(type) {}
  Mat& operator=(const Scalar& s) {
    for (int y = 0; y < rows; ++y) std::memset(data + (size_t)y * step, (int)s.val[0], (size_t)cols);
    return *this;
  }
  int type() const { return type_; }
  bool empty() const { return data == nullptr; }

 private:
  int type_ = CV_8UC1;
  std::shared_ptr<uchar> own_;
};

}  // namespace cv
