// Minimal stand-in for cv::Mat / cv::Vec3b as used by include/semantic_dsp_map.h (TEST INFRASTRUCTURE, see Eigen/Dense).
#pragma once
#include <cstdint>
#include <vector>
#ifndef CV_8UC3
#define CV_8UC3 3  /* stand-in: element size */
#endif
typedef unsigned char uchar;
namespace cv {
struct Vec3b {
  uchar v[3];
  Vec3b() : v{0, 0, 0} {}
  Vec3b(uchar a, uchar b, uchar c) : v{a, b, c} {}
  uchar operator[](int i) const { return v[i]; }
  uchar &operator[](int i) { return v[i]; }
};
enum { COLOR_RGB2HSV = 41, COLOR_HSV2RGB = 55 };
struct Mat {
  int rows = 0, cols = 0, elem = 1;
  std::vector<unsigned char> buf;
  Mat() {}
  Mat(int r, int c, int elem_size) : rows(r), cols(c), elem(elem_size), buf((size_t)r * c * elem_size) {}
  bool empty() const { return rows == 0 || cols == 0; }
  template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(&buf[((size_t)r * cols + c) * sizeof(T)]); }
  template <typename T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(&buf[((size_t)r * cols + c) * sizeof(T)]); }
  template <typename T> T *ptr(int r) { return reinterpret_cast<T *>(&buf[(size_t)r * cols * elem]); }
  template <typename T> const T *ptr(int r) const { return reinterpret_cast<const T *>(&buf[(size_t)r * cols * elem]); }
};
// stand-in: a plain copy (the real conversion is OpenCV's; only the adapter's call sequence is compile-checked here)
inline void cvtColor(const Mat &src, Mat &dst, int) { dst = src; }
}  // namespace cv
