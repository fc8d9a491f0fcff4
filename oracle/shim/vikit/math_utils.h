// -*- C++ -*-
// oracle/shim/vikit/math_utils.h -- TEST INFRASTRUCTURE ONLY.
// vk::project2d / unproject2d / norm_max / getMedian restated from rpg_vikit
// (vikit_common/include/vikit/math_utils.h); un-vendored and unpinned in the reference.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include <Eigen/Core>
#include <sophus/se3.h>
namespace vk {
using namespace Eigen;
using namespace std;
using namespace Sophus;
inline Vector2d project2d(const Vector3d& v) { return v.head<2>() / v[2]; }
inline Vector3d unproject2d(const Vector2d& v) { return Vector3d(v[0], v[1], 1.0); }
template <typename T, int R> inline T norm_max(const Matrix<T, R, 1>& v) {
  T max = -1;
  for (int i = 0; i < v.size(); i++) { T abs = std::fabs(v[i]); if (abs > max) max = abs; }
  return max;
}
template <class T> T getMedian(vector<T>& data_vec) {
  assert(!data_vec.empty());
  typename vector<T>::iterator it = data_vec.begin() + floor(data_vec.size() / 2);
  nth_element(data_vec.begin(), it, data_vec.end());
  return *it;
}
}  // namespace vk
