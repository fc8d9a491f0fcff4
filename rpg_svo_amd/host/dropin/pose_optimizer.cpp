// Drop-in body for svo/src/pose_optimizer.cpp: svo::pose_optimizer::optimizeGaussNewton
// (svo/include/svo/pose_optimizer.h:37-45) on the MI355X through svo_hip_pose_optimize (K4).
// Same signature, same side effects on the frame: T_f_w_, Cov_, and Feature::point reset to
// NULL for observations pruned at reproj_thresh (pose_optimizer.cpp:129-145).
#include <svo/pose_optimizer.h>

#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "marshal.h"

namespace svo {
namespace pose_optimizer {

void optimizeGaussNewton(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtr& frame,
                         double& estimated_scale, double& error_init, double& error_final, size_t& num_obs) {
  using namespace hip_dropin;
  const size_t n = frame->fts_.size();
  if (n == 0) return;
  svo_hip::Device& dev = ensureDevice(*frame);
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
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
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  if (*ran == 2) {  // untouched by the wave kernel: the ordered kernel on the same blocks
    svo_hip::check(svo_hip_pose_optimize_ordered(&cam, 1, d_n, (int)n, d_f, d_level, d_pos, d_has_out, reproj_thresh,
                                                 (int)n_iter, d_Tout, d_Cov, d_stats, d_ran, lane.stream),
                   "svo_hip_pose_optimize_ordered");
    a.download(lane.stream);
    svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  }
  stage_timer.unmarshal();

  if (!*ran) return;  // no observation carried a point: the reference returns untouched (:57-58)
  frame->T_f_w_ = poseFromRt(Tout);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) frame->Cov_(r, c) = Cov[r * 6 + c];
  size_t n_deleted_refs = 0;
  i = 0;
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

}  // namespace pose_optimizer
}  // namespace svo
