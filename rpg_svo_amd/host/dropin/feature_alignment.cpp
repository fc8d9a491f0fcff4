// Drop-in for svo::feature_alignment::align1D / align2D (svo/include/svo/feature_alignment.h:29-44,
// svo/src/feature_alignment.cpp:30-277), the stand-alone seam direct callers use (the reference's
// test_feature_alignment.cpp, a Matcher built from the reference's own matcher.cpp).  One trial per
// call through svo_hip_align_batch (K3): the image is placed in the scratch slot of the device context
// (it is a cv::Mat of some pyramid level's size, not necessarily level 0 of a svo::Frame), the refined
// position and the verdict come back.  align2D_SSE2 / align2D_NEON stay in the reference's own file
// (scripts/strip_members.py builds it minus align1D / align2D).
//
// The pipeline never takes this path -- Reprojector and DepthFilter batch their trials -- and one trial
// per launch is latency-bound; it exists so that every caller of the seam gets the same arithmetic.
#include <svo/feature_alignment.h>

#include "marshal.h"

namespace svo {
namespace feature_alignment {

namespace {
bool alignOne(const cv::Mat& cur_img, const float* dir, uint8_t* ref_patch_with_border, const int n_iter,
              Vector2d& cur_px_estimate, double* h_inv) {
  using namespace hip_dropin;
  svo_hip::Device& dev = svo_hip::Device::instance();
  if (!dev.configured())
    throw svo_hip::Error("feature_alignment: the device context is created by the first svo::Frame the pipeline sees; "
                         "call svo_hip::Device::instance().configure(width, height, levels) before stand-alone use");
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(L);
  int level = 0;
  const int slot = dev.scratchSlotOf(cur_img.data, cur_img.cols, cur_img.rows, (int)cur_img.step.p[0], &level, lane);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  int32_t *d_slot, *d_level; uint8_t *d_pwb, *d_use1d; float* d_dir;
  int32_t* h_slot = a.alloc<int32_t>(1, &d_slot);
  int32_t* h_level = a.alloc<int32_t>(1, &d_level);
  uint8_t* h_pwb = a.alloc<uint8_t>(100, &d_pwb);
  uint8_t* h_use1d = a.alloc<uint8_t>(1, &d_use1d);
  float* h_dir = a.alloc<float>(2, &d_dir);
  *h_slot = slot; *h_level = level;
  std::memcpy(h_pwb, ref_patch_with_border, 100);
  *h_use1d = dir ? 1 : 0;
  h_dir[0] = dir ? dir[0] : 1.f; h_dir[1] = dir ? dir[1] : 0.f;
  a.endInputs();
  double *d_px, *d_hinv; int32_t* d_ok;
  double* h_px = a.alloc<double>(2, &d_px);  // in/out
  h_px[0] = cur_px_estimate[0]; h_px[1] = cur_px_estimate[1];
  int32_t* h_ok = a.alloc<int32_t>(1, &d_ok);
  double* h_hinv = a.alloc<double>(1, &d_hinv);
  a.uploadAll(lane.stream);
  svo_hip::check(svo_hip_align_batch(&dev.layout(), dev.store(), 1, d_slot, d_level, d_pwb, d_dir, d_use1d, n_iter, d_px, d_ok, d_hinv,
                                     lane.stream), "svo_hip_align_batch");
  a.download(lane.stream);
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  cur_px_estimate = Vector2d(h_px[0], h_px[1]);
  if (h_inv) *h_inv = *h_hinv;
  return *h_ok != 0;
}
}  // namespace

bool align1D(const cv::Mat& cur_img, const Vector2f& dir, uint8_t* ref_patch_with_border, uint8_t* /*ref_patch*/,
             const int n_iter, Vector2d& cur_px_estimate, double& h_inv) {
  const float d[2] = {dir[0], dir[1]};
  return alignOne(cur_img, d, ref_patch_with_border, n_iter, cur_px_estimate, &h_inv);
}

bool align2D(const cv::Mat& cur_img, uint8_t* ref_patch_with_border, uint8_t* /*ref_patch*/, const int n_iter,
             Vector2d& cur_px_estimate, bool /*no_simd*/) {
  return alignOne(cur_img, NULL, ref_patch_with_border, n_iter, cur_px_estimate, NULL);
}

}  // namespace feature_alignment
}  // namespace svo
