// Drop-in for the two search members of svo::Matcher (svo/include/svo/matcher.h:106-123):
// findMatchDirect (svo/src/matcher.cpp:135-177) and findEpipolarMatchDirect (:179-321), for callers that
// use the class directly -- the reference's test_matcher.cpp, a Reprojector / DepthFilter built from the
// reference's own files.  One trial per call through svo_hip_find_match_direct /
// svo_hip_find_epipolar_match_direct; the Matcher's public scratch members a caller may read afterwards
// (ref_ftr_, search_level_, A_cur_ref_, patch_with_border_, patch_, px_cur_) are filled.  The warp::
// functions, depthFromTriangulation and createPatchFromPatchWithBorder stay in the reference's own
// matcher.cpp (scripts/strip_members.py builds it minus the two members).
//
// The pipeline does not take this path: the drop-in Reprojector and DepthFilter batch all trials of a
// frame into one launch each.  A single trial per launch is latency-bound.
#include <svo/matcher.h>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "marshal.h"

namespace svo {

bool Matcher::findMatchDirect(const Point& pt, const Frame& cur_frame, Vector2d& px_cur) {
  using namespace hip_dropin;
  if (!pt.getCloseViewObs(cur_frame.pos(), ref_ftr_)) return false;  // :137-138, on the host
  svo_hip::Device& dev = ensureDevice(cur_frame);
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(L);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  FrameTable frames(dev, L);
  int32_t *d_cur, *d_ptr; double* d_pos;
  int32_t* cur = a.alloc<int32_t>(1, &d_cur);
  double* pos = a.alloc<double>(3, &d_pos);
  int32_t* ptr = a.alloc<int32_t>(2, &d_ptr);
  FeatureColumns obs;
  obs.alloc(a, 1);
  cur[0] = frames.indexOf(&cur_frame);
  for (int k = 0; k < 3; ++k) pos[k] = pt.pos_[k];
  ptr[0] = 0; ptr[1] = 1;
  obs.set(0, frames.indexOf(ref_ftr_->frame), ref_ftr_);
  svo_hip_frames ft;
  frames.emit(a, &ft);
  a.endInputs();
  double *d_px, *d_A; int32_t *d_ok, *d_ref, *d_lvl; uint8_t* d_patch;
  double* px = a.alloc<double>(2, &d_px);  // in: the estimate, out: the refined pixel
  px[0] = px_cur[0]; px[1] = px_cur[1];
  int32_t* ok = a.alloc<int32_t>(1, &d_ok);
  int32_t* ref = a.alloc<int32_t>(1, &d_ref);
  int32_t* lvl = a.alloc<int32_t>(1, &d_lvl);
  double* A = a.alloc<double>(4, &d_A);
  uint8_t* patch = a.alloc<uint8_t>(100, &d_patch);
  const svo_hip_camera cam = cameraOf(cur_frame.cam_);
  void* ws = dev.workspace(lane, 1);
  a.uploadAll(lane.stream);
  svo_hip::check(svo_hip_find_match_direct(&dev.layout(), dev.store(), &cam, &ft, 1, d_cur, d_pos, d_ptr, &obs.dev, Config::nPyrLevels(),
                                           options_.align_max_iter, d_px, d_ok, d_ref, d_lvl, d_A, d_patch, ws, lane.workspace_bytes,
                                           lane.stream), "svo_hip_find_match_direct");
  a.download(lane.stream);
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  // The reference returns before touching anything when no observation is close enough in view angle or the reference
  // feature sits too close to its image border (:140-145): the kernel then leaves the warp matrix at zero.
  if (*ref < 0 || (A[0] == 0.0 && A[1] == 0.0 && A[2] == 0.0 && A[3] == 0.0)) return false;
  search_level_ = *lvl;
  A_cur_ref_(0, 0) = A[0]; A_cur_ref_(0, 1) = A[1]; A_cur_ref_(1, 0) = A[2]; A_cur_ref_(1, 1) = A[3];
  std::memcpy(patch_with_border_, patch, sizeof(patch_with_border_));
  createPatchFromPatchWithBorder();
  // px_cur = px_scaled*(1<<search_level_) whether or not the alignment converged (:175); px_cur_ is not this function's
  px_cur = Vector2d(px[0], px[1]);
  return *ok != 0;
}

bool Matcher::findEpipolarMatchDirect(const Frame& ref_frame, const Frame& cur_frame, const Feature& ref_ftr, const double d_estimate,
                                      const double d_min, const double d_max, double& depth) {
  using namespace hip_dropin;
  svo_hip::Device& dev = ensureDevice(cur_frame);
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(L);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  FrameTable frames(dev, L);
  int32_t* d_cur; double *d_de, *d_dmin, *d_dmax;
  int32_t* cur = a.alloc<int32_t>(1, &d_cur);
  double* de = a.alloc<double>(1, &d_de);
  double* dmin = a.alloc<double>(1, &d_dmin);
  double* dmax = a.alloc<double>(1, &d_dmax);
  FeatureColumns ftr;
  ftr.alloc(a, 1);
  cur[0] = frames.indexOf(&cur_frame);
  *de = d_estimate; *dmin = d_min; *dmax = d_max;
  ftr.set(0, frames.indexOf(&ref_frame), &ref_ftr);
  svo_hip_frames ft;
  frames.emit(a, &ft);
  a.endInputs();
  int32_t *d_ok, *d_lvl; double *d_depth, *d_px;
  int32_t* ok = a.alloc<int32_t>(1, &d_ok);
  double* z = a.alloc<double>(1, &d_depth);
  double* px = a.alloc<double>(2, &d_px);
  int32_t* lvl = a.alloc<int32_t>(1, &d_lvl);
  svo_hip_depth_filter_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.align_1d = options_.align_1d;
  opt.align_max_iter = options_.align_max_iter;
  opt.max_epi_search_steps = (int32_t)options_.max_epi_search_steps;
  opt.subpix_refinement = options_.subpix_refinement;
  opt.epi_search_edgelet_filtering = options_.epi_search_edgelet_filtering;
  opt.n_pyr_levels = Config::nPyrLevels();
  opt.epi_search_edgelet_max_angle = options_.epi_search_edgelet_max_angle;
  const svo_hip_camera cam = cameraOf(cur_frame.cam_);
  void* ws = dev.workspace(lane, 1);
  a.upload(lane.stream);
  svo_hip::check(svo_hip_find_epipolar_match_direct(&dev.layout(), dev.store(), &cam, &ft, 1, d_cur, &ftr.dev, d_de, d_dmin, d_dmax, &opt,
                                                    d_ok, d_depth, d_px, d_lvl, ws, lane.workspace_bytes, lane.stream),
                 "svo_hip_find_epipolar_match_direct");
  a.download(lane.stream);
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  if (*lvl >= 0) search_level_ = *lvl;  // (-1: returned before matcher.cpp:214, search_level_ keeps its value)
  if (!*ok) return false;
  px_cur_ = Vector2d(px[0], px[1]);
  depth = *z;
  return true;
}

}  // namespace svo
