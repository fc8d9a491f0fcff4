// Drop-in body for svo/src/pose_optimizer.cpp: svo::pose_optimizer::optimizeGaussNewton
// (svo/include/svo/pose_optimizer.h:37-45) on the MI355X through svo_hip_pose_optimize (K4).
// Same signature, same side effects on the frame: T_f_w_, Cov_, and Feature::point reset to
// NULL for observations pruned at reproj_thresh (pose_optimizer.cpp:129-145).
// When Reprojector::reprojectMap's drop-in has already enqueued this very call behind its match kernels
// (svo_hip::Speculation) and the frame is the one it predicted, the result is taken from there: one stream wait.
#include <svo/pose_optimizer.h>

#include <cstring>
#include <functional>

#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "marshal.h"

namespace svo {
namespace pose_optimizer {

namespace {
// T_f_w_, Cov_ and the pruned observations back into the frame (pose_optimizer.cpp:119-145)
void applyResult(FramePtr& frame, const double* T, const double* Cov, const double* stats, const uint8_t* has_out, bool verbose,
                 double& estimated_scale, double& error_init, double& error_final, size_t& num_obs) {
  frame->T_f_w_ = hip_dropin::poseFromRt(T);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) frame->Cov_(r, c) = Cov[r * 6 + c];
  size_t n_deleted_refs = 0;
  size_t i = 0;
  for (Features::iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it, ++i)
    if ((*it)->point != NULL && !has_out[i]) {
      (*it)->point = NULL;  // the point holds no reference to this feature yet (:139-141)
      ++n_deleted_refs;
    }
  estimated_scale = stats[0];
  error_init = stats[1];
  error_final = stats[2];
  num_obs = (size_t)stats[3];
  if (verbose)
    std::cout << "n deleted obs = " << n_deleted_refs << "\t scale = " << estimated_scale << "\t error init = " << error_init
              << "\t error end = " << error_final << std::endl;
}

// Is `frame` what the reprojector predicted -- the same features in the same order, the same starting pose, the same
// parameters?  Then its pose refinement is already running (or done) on the lane's stream.
bool predicted(const svo_hip::Speculation& sp, const FramePtr& frame, double reproj_thresh, size_t n_iter) {
  if (!sp.valid || sp.frame_id != frame->id_ || sp.point.size() != frame->fts_.size() || sp.reproj_thresh != reproj_thresh ||
      sp.n_iter != (int)n_iter)
    return false;
  double T[12];
  hip_dropin::poseToRt(frame->T_f_w_, T);
  if (std::memcmp(T, sp.T_init, sizeof(T)) != 0) return false;
  size_t i = 0;
  for (Features::const_iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it, ++i) {
    const Feature* ftr = *it;
    if (ftr->point != sp.point[i] || ftr->px[0] != sp.px[2 * i] || ftr->px[1] != sp.px[2 * i + 1] || ftr->level != sp.level[i])
      return false;
  }
  return true;
}
}  // namespace

namespace {
void optimizeOnDevice(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtr& frame, double& estimated_scale,
                      double& error_init, double& error_final, size_t& num_obs);
}

void optimizeGaussNewton(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtr& frame,
                         double& estimated_scale, double& error_init, double& error_final, size_t& num_obs) {
  if (frame->fts_.empty()) return;
  // A host that runs the depth filter WITHOUT its thread will hand the frame to updateSeeds a few dozen microseconds of
  // bookkeeping after this call returns (frame_handler_mono.cpp:176-198), with the pose this call fixes.  The filter's
  // drop-in, when it has registered its hook on this thread's mapping lane, enqueues that update from here
  // (dropin/depth_filter.cpp: EarlyUpdate), in two phases: 1 -- its tables are marshalled and uploaded NOW, while the
  // refinement the reprojector predicted is still running on the device and this call would only wait for it; 2 -- the
  // kernels are launched with the final pose (by value: svo_hip_update_seeds_resident_pose).  A frame the reference is
  // about to give up (fewer than 20 observations left, :176) is not worth the launch: what phase 1 holds is then dropped
  // by the next call on the mapping lane.  (The tracking lane's mutex is not held here: the hook takes the mapping
  // lane's and the seed list's.)
  auto early = [&frame](const int phase) {
    if (!svo_hip::Device::earlyMappingEnabled()) return;
    svo_hip::Lane* ml = hip_dropin::ensureDevice(*frame).findLane(svo_hip::Device::LANE_MAPPING);
    if (ml == NULL) return;
    std::function<void(const void*, int)> hook;
    {
      std::lock_guard<std::mutex> g(ml->mut);
      hook = ml->early_hook;
    }
    if (hook) hook(&frame, phase);
  };
  early(1);
  optimizeOnDevice(reproj_thresh, n_iter, verbose, frame, estimated_scale, error_init, error_final, num_obs);
  if (num_obs >= 20) early(2);
}

namespace {
void optimizeOnDevice(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtr& frame, double& estimated_scale,
                      double& error_init, double& error_final, size_t& num_obs) {
  using namespace hip_dropin;
  const size_t n = frame->fts_.size();
  if (n == 0) return;
  svo_hip::Device& dev = ensureDevice(*frame);
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  if (predicted(lane.spec, frame, reproj_thresh, n_iter)) {  // before beginCall(): the result blocks are in the arena
    svo_hip::Speculation& sp = lane.spec;
    bool agree;
    {
      svo_hip::StageTimer wait_timer(dev, lane, svo_hip::Device::STAGE_POSE_OPT);
      wait_timer.device(0);
      sp.valid = false;
      sp.in_flight = false;
      if (sp.stream == lane.stream) dev.finish(lane);
      else svo_hip::check(svo_hip_stream_sync(sp.stream), "svo_hip_stream_sync");
      wait_timer.unmarshal();
      // the device applied the selection rule on its own: it must have picked the trials the host picked; a frame the
      // wave kernel left to the ordered kernel (ran == 2: singular normal equations) takes the ordinary call below
      agree = *sp.n_sel == (int32_t)n && *sp.ran != 2;
      for (size_t k = 0; agree && k < n; ++k) agree = sp.sel[k] == sp.trial[k];
      dev.countSpeculation(agree);
      if (agree) {
        if (*sp.ran) applyResult(frame, sp.T, sp.Cov, sp.stats, sp.has_point, verbose, estimated_scale, error_init, error_final, num_obs);
        return;  // ran == 0: no observation carried a point, the reference returns untouched (:57-58)
      }
    }
  }
  dev.beginCall(L);
  svo_hip::StageTimer stage_timer(dev, lane, svo_hip::Device::STAGE_POSE_OPT);
  svo_hip::Arena& a = lane.arena;
  a.reset();

  // ---- observations, in the order of Frame::fts_ (the order the reference accumulates in) --
  int32_t *d_n, *d_level; double *d_f, *d_pos, *d_Tout; uint8_t* d_has_out;
  int32_t* hn = a.alloc<int32_t>(1, &d_n);
  int32_t* level = a.alloc<int32_t>(n, &d_level);
  double* f = a.alloc<double>(3 * n, &d_f);
  double* pos = a.alloc<double>(3 * n, &d_pos);
  a.endInputs();
  // has_point and the pose are read AND written by the kernel: they live in the output block,
  // are filled here and travel both ways (uploadAll / download), so the call is two copies
  double* Tout = a.alloc<double>(12, &d_Tout);
  uint8_t* has_out = a.alloc<uint8_t>(n, &d_has_out);
  *hn = (int32_t)n;
  size_t i = 0;
  for (Features::const_iterator it = frame->fts_.begin(); it != frame->fts_.end(); ++it, ++i) {
    const Feature* ftr = *it;
    level[i] = ftr->level;
    for (int k = 0; k < 3; ++k) f[3 * i + k] = ftr->f[k];
    has_out[i] = ftr->point != NULL;
    for (int k = 0; k < 3; ++k) pos[3 * i + k] = ftr->point ? ftr->point->pos_[k] : 0.0;
  }
  poseToRt(frame->T_f_w_, Tout);
  double *d_Cov, *d_stats; int32_t* d_ran;
  double* Cov = a.alloc<double>(36, &d_Cov);
  double* stats = a.alloc<double>(4, &d_stats);
  int32_t* ran = a.alloc<int32_t>(1, &d_ran);

  const svo_hip_camera cam = cameraOf(frame->cam_);
  stage_timer.device(a.used());
  a.uploadAll(lane.stream);
  // the wave kernel alone; a frame it hands over (ran == 2: singular normal equations, n_iter == 0) is rare
  // and found out after the sync this call needs anyway, so the single-stream path pays for one launch
  svo_hip::check(svo_hip_pose_optimize_deferred(&cam, 1, d_n, (int)n, d_f, d_level, d_pos, d_has_out, reproj_thresh,
                                                (int)n_iter, d_Tout, d_Cov, d_stats, d_ran, lane.stream),
                 "svo_hip_pose_optimize_deferred");
  a.download(lane.stream);
  dev.finish(lane);
  if (*ran == 2) {  // untouched by the wave kernel: the ordered kernel on the same blocks
    svo_hip::check(svo_hip_pose_optimize_ordered(&cam, 1, d_n, (int)n, d_f, d_level, d_pos, d_has_out, reproj_thresh,
                                                 (int)n_iter, d_Tout, d_Cov, d_stats, d_ran, lane.stream),
                   "svo_hip_pose_optimize_ordered");
    a.download(lane.stream);
    svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  }
  stage_timer.unmarshal();

  if (!*ran) return;  // no observation carried a point: the reference returns untouched (:57-58)
  applyResult(frame, Tout, Cov, stats, has_out, verbose, estimated_scale, error_init, error_final, num_obs);
}
}  // namespace

}  // namespace pose_optimizer
}  // namespace svo
