// -*- C++ -*-
// oracle/shim/opencv2/opencv.hpp -- TEST INFRASTRUCTURE ONLY.
// The handful of cv::Mat members the reference's hot-path sources touch (8-bit and float
// single-channel matrices, continuous rows).  No image processing lives here.
#ifndef ORC_SHIM_OPENCV_HPP
#define ORC_SHIM_OPENCV_HPP
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_WINDOW_AUTOSIZE 1

namespace cv {
struct Scalar { double v; Scalar(double a = 0) : v(a) {} };
struct Point2f { float x, y; Point2f(float a = 0, float b = 0) : x(a), y(b) {} };  // initialization.h members only
struct Size { int width, height; Size(int w = 0, int h = 0) : width(w), height(h) {} };

class Mat {
 public:
  struct Step { size_t p[2]; };
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  Step step{{0, 0}};
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) { create(r, c, type); fill(s.v); }
  Mat(Size sz, int type, const Scalar& s) { create(sz.height, sz.width, type); fill(s.v); }
  // wrap external memory (no copy); `stride` in bytes
  Mat(int r, int c, int type, void* ext, size_t stride = 0) : rows(r), cols(c), data((uint8_t*)ext), type_(type) {
    step.p[0] = stride ? stride : (size_t)c * elem();
    step.p[1] = elem();
  }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Size size() const { return Size(cols, rows); }
  template <typename T> T& at(int r, int c) { return *(T*)(data + (size_t)r * step.p[0] + (size_t)c * sizeof(T)); }
  template <typename T> const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * step.p[0] + (size_t)c * sizeof(T)); }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step.p[0], data + (size_t)r * step.p[0], (size_t)cols * elem());
    return m;
  }
  Mat operator*(double) const { return *this; }  // only reached with display_ == true (never)

 private:
  int type_ = CV_8U;
  std::shared_ptr<uint8_t> own_;
  size_t elem() const { return type_ == CV_32F ? 4 : 1; }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type;
    step.p[0] = (size_t)c * elem(); step.p[1] = elem();
    own_.reset(new uint8_t[(size_t)r * step.p[0] + 64], std::default_delete<uint8_t[]>());
    data = own_.get();
  }
  void fill(double v) {
    if (type_ == CV_32F) { float* f = (float*)data; for (size_t i = 0; i < (size_t)rows * cols; ++i) f[i] = (float)v; }
    else std::memset(data, (int)v, (size_t)rows * cols);
  }
};
inline void namedWindow(const std::string&, int = 0) {}
inline void imshow(const std::string&, const Mat&) {}
inline int waitKey(int = 0) { return 0; }
}  // namespace cv
#endif
