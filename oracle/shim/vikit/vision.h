// -*- C++ -*-
// oracle/shim/vikit/vision.h -- TEST INFRASTRUCTURE ONLY.  Bodies: ../../orc_vikit.h.
#pragma once
#include <opencv2/opencv.hpp>
extern "C" {
#include "orc_vikit.h"
#include "orc_fast.h"
}
namespace vk {
extern int g_halfsample_mode;  // 0 scalar, 1 SSE2 flavour, 2 x86 dispatch (defined in ref_driver.cpp)
inline void halfSample(const cv::Mat& in, cv::Mat& out) {
  orc_half_sample_impl(in.data, in.cols, in.rows, (int)in.step.p[0], out.data, (int)out.step.p[0], g_halfsample_mode);
}
inline float interpolateMat_8u(const cv::Mat& mat, float u, float v) {
  return orc_interpolate_mat_8u(mat.data, (int)mat.step.p[0], u, v);
}
inline float shiTomasiScore(const cv::Mat& img, int u, int v) {
  return orc_shi_tomasi_score(img.data, img.cols, img.rows, (int)img.step.p[0], u, v);
}
}  // namespace vk
