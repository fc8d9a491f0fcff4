// oracle/shim -- TEST INFRASTRUCTURE ONLY.  The uzh-rpg/fast entry points
// svo/src/feature_detection.cpp calls, on the restatement in ../../orc_fast.h (the library is
// absent from this image and un-pinned by the reference).
#pragma once
#include <vector>
extern "C" {
#include "orc_fast.h"
}
namespace fast {
typedef unsigned char fast_byte;
struct fast_xy { short x, y; fast_xy(short x_ = 0, short y_ = 0) : x(x_), y(y_) {} };
inline void fast_corner_detect_10(const fast_byte* img, int w, int h, int stride, short b, std::vector<fast_xy>& out) {
  std::vector<short> xy((size_t)w * h * 2 / 4 + 16);
  int cap = (int)(xy.size() / 2);
  int n = orc_fast10_detect(img, w, h, stride, b, xy.data(), cap);
  if (n > cap) { xy.resize((size_t)n * 2); n = orc_fast10_detect(img, w, h, stride, b, xy.data(), n); }
  out.clear();
  for (int i = 0; i < n; ++i) out.push_back(fast_xy(xy[2 * i], xy[2 * i + 1]));
}
inline void fast_corner_detect_10_sse2(const fast_byte* img, int w, int h, int stride, short b, std::vector<fast_xy>& out) {
  fast_corner_detect_10(img, w, h, stride, b, out);  // same corner set, same raster order
}
inline void fast_corner_score_10(const fast_byte* img, const int stride, const std::vector<fast_xy>& c, const int b, std::vector<int>& scores) {
  scores.resize(c.size());
  for (size_t i = 0; i < c.size(); ++i) scores[i] = orc_fast10_score(img + c[i].y * stride + c[i].x, stride, b);
}
inline void fast_nonmax_3x3(const std::vector<fast_xy>& c, const std::vector<int>& scores, std::vector<int>& keep) {
  keep.clear();
  if (c.empty()) return;
  int w = 0, h = 0;
  std::vector<short> xy(c.size() * 2);
  for (size_t i = 0; i < c.size(); ++i) { xy[2 * i] = c[i].x; xy[2 * i + 1] = c[i].y; if (c[i].x >= w) w = c[i].x + 1; if (c[i].y >= h) h = c[i].y + 1; }
  keep.resize(c.size());
  int m = orc_fast_nonmax_3x3(xy.data(), scores.data(), (int)c.size(), w, h, keep.data());
  keep.resize((size_t)m);
}
}  // namespace fast
