// Drop-in body for svo/src/sparse_img_align.cpp: the class svo::SparseImgAlign of
// svo/include/svo/sparse_img_align.h with run() executing on the MI355X through
// svo_hip_sparse_align (K1, include/svo_hip.h).  Replaces the CPU Gauss-Newton of
// svo/src/sparse_img_align.cpp:43-258; callers (frame_handler_mono.cpp:136-138,247-249)
// are untouched.
#include <svo/sparse_img_align.h>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/point.h>

#include "frame_chain.h"
#include "marshal.h"

namespace svo {

SparseImgAlign::SparseImgAlign(int max_level, int min_level, int n_iter, Method method, bool display, bool verbose)
    : display_(display), max_level_(max_level), min_level_(min_level) {
  // same solver settings as the reference constructor (sparse_img_align.cpp:29-41)
  // The kernel runs vk::NLLSSolver's Gauss-Newton loop, the only method the reference pipeline constructs
  // (frame_handler_mono.cpp:136-137, 247-248).  A caller asking for LevenbergMarquardt is told so here instead of
  // silently getting Gauss-Newton (the Python mirror raises the same way, rpg_svo_amd/sparse_img_align.py).
  if (method != GaussNewton) throw svo_hip::Error("SparseImgAlign: only the GaussNewton method runs on the device");
  n_iter_ = n_iter;
  n_iter_init_ = n_iter_;
  method_ = method;
  verbose_ = verbose;
  eps_ = 0.000001;
}

size_t SparseImgAlign::run(FramePtr ref_frame, FramePtr cur_frame) {
  reset();
  if (ref_frame->fts_.empty()) {
    SVO_WARN_STREAM("SparseImgAlign: no features to track!");
    return 0;
  }
  ref_frame_ = ref_frame;
  cur_frame_ = cur_frame;

  using namespace hip_dropin;
  svo_hip::Device& dev = ensureDevice(*ref_frame);
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);
  // The frame's next steps behind this call (frame_chain.h): reprojectMap's drop-in reads the map, which the deferred
  // mapper's pending results feed -- they are written back first, as reprojectMap itself would (before the lane is locked:
  // the join takes the lanes' mutexes).  (chain_hook is written by this thread's own reprojectMap, or by a dying
  // Reprojector under the lane's mutex: re-read under it below.)
  if (lane.chain_hook != NULL && svo_hip::Device::chainEnabled()) svo_hip::Device::joinDeferredAll();
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(L);
  svo_hip::StageTimer stage_timer(dev, lane, svo_hip::Device::STAGE_SPARSE_ALIGN);

  const size_t n = ref_frame->fts_.size();
  if (n > SVO_HIP_MAX_PATCHES) throw svo_hip::Error("SparseImgAlign: more than SVO_HIP_MAX_PATCHES features");
  svo_hip::Arena& a = lane.arena;
  a.reset();
  FrameTable frames(dev, L);
  FrameChain* chain = svo_hip::Device::chainEnabled() ? static_cast<FrameChain*>(lane.chain_hook) : NULL;
  // (no host-visible signals in a mirrored arena; ref == cur: one table entry would have to hold two poses)
  if (chain != NULL && (a.mode() == svo_hip::Arena::MIRRORED || ref_frame.get() == cur_frame.get())) chain = NULL;
  if (chain != NULL) a.reserve(n * 64 + 8192 + chain->outputBytesBound());
  const int i_ref = frames.indexOf(ref_frame.get());
  const int i_cur = frames.indexOf(cur_frame.get());

  // ---- inputs: what precomputeReferencePatches derives per feature (:107-108) ----------
  int32_t *d_slots; double *d_px, *d_xyz, *d_Tin; uint8_t* d_valid;
  int32_t* slots = a.alloc<int32_t>(3, &d_slots);  // ref slot, cur slot, n
  double* px = a.alloc<double>(2 * n, &d_px);
  double* xyz = a.alloc<double>(3 * n, &d_xyz);
  uint8_t* valid = a.alloc<uint8_t>(n, &d_valid);
  double* Tin = a.alloc<double>(12, &d_Tin);
  slots[0] = frames.slot(i_ref); slots[1] = frames.slot(i_cur); slots[2] = (int32_t)n;
  const Vector3d ref_pos = ref_frame->pos();
  size_t i = 0;
  for (Features::const_iterator it = ref_frame->fts_.begin(); it != ref_frame->fts_.end(); ++it, ++i) {
    const Feature* ftr = *it;
    px[2 * i] = ftr->px[0]; px[2 * i + 1] = ftr->px[1];
    valid[i] = ftr->point != NULL;
    if (ftr->point) {
      const double depth = (ftr->point->pos_ - ref_pos).norm();
      for (int k = 0; k < 3; ++k) xyz[3 * i + k] = ftr->f[k] * depth;
    } else {
      xyz[3 * i] = xyz[3 * i + 1] = 0.0; xyz[3 * i + 2] = 1.0;
    }
  }
  const SE3 T_cur_from_ref(cur_frame->T_f_w_ * ref_frame->T_f_w_.inverse());  // prior (:59)
  poseToRt(T_cur_from_ref, Tin);
  a.endInputs();

  // ---- outputs ---------------------------------------------------------------------------
  double *d_Tout, *d_H, *d_chi2; int32_t *d_ntracked, *d_iters, *d_status;
  double* Tout = a.alloc<double>(12, &d_Tout);
  double* H = a.alloc<double>(36, &d_H);
  double* chi2 = a.alloc<double>(1, &d_chi2);
  int32_t* n_tracked = a.alloc<int32_t>(1, &d_ntracked);
  int32_t* iters = a.alloc<int32_t>(SVO_HIP_MAX_LEVELS, &d_iters);
  int32_t* status = a.alloc<int32_t>(1, &d_status);

  const svo_hip_camera cam = cameraOf(ref_frame->cam_);
  svo_hip_sia_params P;
  P.fx = cam.fx; P.fy = cam.fy; P.cx = cam.cx; P.cy = cam.cy;
  P.max_level = max_level_; P.min_level = min_level_; P.n_iter = (int32_t)n_iter_; P.eps = eps_;
  P.cam_model = cam.model;
  for (int k = 0; k < 5; ++k) P.d[k] = cam.d[k];

  stage_timer.device(a.used());
  a.upload(lane.stream);
  svo_hip::check(svo_hip_sparse_align(&dev.layout(), dev.store(), 1, d_slots, d_slots + 1, d_slots + 2, (int)n, d_px, d_xyz,
                                      d_valid, &P, d_Tin, d_Tout, d_H, d_ntracked, d_iters, d_chi2, d_status, lane.stream),
                 "svo_hip_sparse_align");
  // K1 is running: the chain's host work, its inputs (second arena, own copy command) and its launches follow
  bool chained = false;
  if (chain != NULL) {
    struct Abandon {  // an exception between prepare() and enqueue() leaves the chain idle
      FrameChain* c;
      ~Abandon() { if (c) c->abandon(); }
    } abandon = {chain};
    if (chain->prepare(ref_frame, cur_frame, dev, lane)) {
      chain->allocInputs(lane.arena_chain, ref_frame);
      // (behind K1 on the lane's stream: on the second stream, beside K1 and joined by an event, the event's two API calls
      // cost the frame more than the 5 us copy: 0.283 against 0.274 ms, profiles/r06o_*)
      lane.arena_chain.uploadAll(lane.stream);
      chain->allocOutputs(a);
      chain->enqueue(d_Tout);
      chained = true;
    }
    abandon.c = NULL;
    if (!chained) chain->abandon();
  }
  if (chained) {
    // reprojection, matching, selection and the predicted pose refinement follow K1 on the stream; this call returns when
    // K1's results (and the pose the device formed from them) are in host memory
    svo_hip::spinUntil(chain->k1Signal(), 1, lane.stream);
  } else {
    a.download(lane.stream);
    dev.finish(lane);
  }
  stage_timer.unmarshal();

  cur_frame->T_f_w_ = poseFromRt(Tout) * ref_frame->T_f_w_;  // :70
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) H_(r, c) = H[r * 6 + c];
  chi2_ = *chi2;
  stop_ = (*status & SVO_HIP_SIA_STOP) != 0;
  n_meas_ = (size_t)*n_tracked * patch_area_;
  (void)iters;
  return (size_t)*n_tracked;  // n_meas_/patch_area_ (:74)
}

Matrix<double, 6, 6> SparseImgAlign::getFisherInformation() {
  const double sigma_i_sq = 5e-4 * 255 * 255;  // image noise, as the reference assumes (:79)
  Matrix<double, 6, 6> I = H_ / sigma_i_sq;
  return I;
}

// The per-iteration hooks of vk::NLLSSolver are never entered: the whole coarse-to-fine
// optimisation runs inside the kernel.  They exist because the class declares them.
void SparseImgAlign::precomputeReferencePatches() {}
double SparseImgAlign::computeResiduals(const SE3&, bool, bool) { return 0.0; }
int SparseImgAlign::solve() { return 0; }
void SparseImgAlign::update(const ModelType& old_model, ModelType& new_model) { new_model = old_model; }
void SparseImgAlign::startIteration() {}
void SparseImgAlign::finishIteration() {}

}  // namespace svo
