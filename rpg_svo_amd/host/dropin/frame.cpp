// Drop-in body for ONE function of svo/src/frame.cpp: svo::frame_utils::createImgPyramid (frame.cpp:156-165, called by
// Frame::initFrame for every image).  In the full drop-in every reader of the host pyramid's upper levels is a drop-in too --
// SparseImgAlign, the reprojector's and the depth filter's matcher trials and FastDetector take the pyramid K0 builds in
// HBM from level 0 (svo_hip_pyramid_*) -- so the three to five vk::halfSample passes per frame (25-30 us of a 0.28 ms
// frame on a 752 x 480 image) produce images nobody reads.  This body keeps level 0 (Frame::img(), the upload's source,
// the KLT initialisation) and leaves the levels above as EMPTY cv::Mat of the right count (img_pyr_.size() still tells
// the number of levels).  Every other member of frame.cpp is the reference's own (tests/dropin/Makefile strips only this
// function).  SVO_HIP_HOST_PYRAMID=1 builds the host levels as the reference does -- for a host that still calls the
// reference's CPU Matcher or FastDetector on a Frame (the stand-alone seams of INTEGRATION.md keep the reference's frame.cpp
// for that reason).
#include <cstdlib>
#include <cstring>

#include <svo/frame.h>
#include <vikit/vision.h>

namespace svo {
namespace frame_utils {

void createImgPyramid(const cv::Mat& img_level_0, int n_levels, ImgPyr& pyr) {
  static const bool host_levels = [] {
    const char* e = std::getenv("SVO_HIP_HOST_PYRAMID");
    return e && std::strcmp(e, "1") == 0;
  }();
  pyr.resize(n_levels);
  pyr[0] = img_level_0;
  if (!host_levels) {
    for (int i = 1; i < n_levels; ++i) pyr[i] = cv::Mat();
    return;
  }
  for (int i = 1; i < n_levels; ++i) {
    pyr[i] = cv::Mat(pyr[i - 1].rows / 2, pyr[i - 1].cols / 2, CV_8U);
    vk::halfSample(pyr[i - 1], pyr[i]);
  }
}

}  // namespace frame_utils
}  // namespace svo
