// oracle/shim -- TEST INFRASTRUCTURE ONLY.  Declarations of the (absent, off-path) FAST
// corner library so svo/src/feature_detection.cpp compiles; calling them aborts.
#pragma once
#include <cstdlib>
#include <vector>
namespace fast {
typedef unsigned char fast_byte;
struct fast_xy { short x, y; fast_xy(short x_ = 0, short y_ = 0) : x(x_), y(y_) {} };
inline void fast_corner_detect_10(const fast_byte*, int, int, int, short, std::vector<fast_xy>&) { std::abort(); }
inline void fast_corner_detect_10_sse2(const fast_byte*, int, int, int, short, std::vector<fast_xy>&) { std::abort(); }
inline void fast_corner_score_10(const fast_byte*, const int, const std::vector<fast_xy>&, const int, std::vector<int>&) { std::abort(); }
inline void fast_nonmax_3x3(const std::vector<fast_xy>&, const std::vector<int>&, std::vector<int>&) { std::abort(); }
}
