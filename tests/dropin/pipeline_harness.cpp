// tests/dropin/pipeline_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C entry points around the REFERENCE'S OWN svo::FrameHandlerMono (compiled from the
// sources where they lie under /root/reference/svo/src against the dependency shims in
// oracle/shim/), the way svo_ros/src/benchmark_node.cpp:178-256 drives it: first frame
// set at a known pose with FAST features lifted to 3-D through a depth map, every later
// image through addImage().  tests/dropin/Makefile links this file twice:
//
//   _build/libsvo_pipeline_ref.so  all reference translation units          (CPU reference)
//   _build/libsvo_pipeline_hip.so  the reference's control plane (frame handlers, map,
//                                  frame, point, config, matcher) + the drop-in
//                                  bodies of rpg_svo_amd/host/dropin/*.cpp, which replace
//                                  sparse_img_align.cpp, reprojector.cpp, pose_optimizer.cpp,
//                                  depth_filter.cpp and feature_detection.cpp and call
//                                  libsvo_hip.so
//
// so the two trajectories can be compared frame by frame (tests/test_dropin_pipeline_gpu.py,
// bench.py --pipeline dropin).  No arithmetic of the path lives here.
#include <cstdlib>
#include <cstring>
#include <list>
#include <stdexcept>
#include <string>

#include <svo/config.h>
#include <svo/depth_filter.h>
#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/frame_handler_mono.h>
#include <svo/map.h>
#include <svo/point.h>
#include <svo/sparse_img_align.h>
#include <vikit/atan_camera.h>
#include <vikit/pinhole_camera.h>
#include <vikit/vision.h>
#ifdef SVO_PIPELINE_HIP
#include "svo_hip_device.h"
// rpg_svo_amd/host/dropin/reprojector.cpp; absent from the stand-alone-seams flavour (hipm), which links the reference's reprojector
namespace svo { namespace hip_dropin { void mapMirrorStats(uint64_t out[6]) __attribute__((weak)); } }
namespace svo { namespace hip_dropin { void seedStoreStats(uint64_t out[3]) __attribute__((weak)); } }
#endif


using namespace svo;

extern "C" {

typedef struct pipe_config {
  int32_t n_pyr_levels, klt_max_level, klt_min_level, grid_size, max_fts, max_n_kfs;
  int32_t quality_min_fts, quality_max_drop_fts, structureoptim_max_pts, structureoptim_num_iter;
  int32_t poseoptim_num_iter, shuffle_seed;
  int32_t mapper_thread;  // 1: keep DepthFilter's own thread running (asynchronous mapping)
  int32_t pool_slots;     // >0 (hip flavour): size of the device pyramid pool, to exercise LRU eviction
  int32_t defer_mapper;   // 1 (hip flavour, mapper_thread = 0): svo_hip::Device::setDeferredMapping(true) -- updateSeeds
                          // returns with its kernels running; n_seeds of a result is then the count BEFORE that update
  double kfselect_mindist, poseoptim_thresh, triang_min_corner_score;
} pipe_config;

typedef struct pipe_result {
  double T_f_w[12];
  int32_t stage, quality, n_obs, is_keyframe, frame_id, n_kfs, n_candidates, n_seeds;
  // the columns FrameHandlerBase writes to its trace file (frame_handler_base.cpp:46-74)
  double img_align_n_tracked, repr_n_mps, repr_n_new_references, sfba_thresh, sfba_error_init, sfba_error_final,
      sfba_n_edges_final, dropout;
  double t_pyramid_creation, t_sparse_img_align, t_reproject, t_pose_optimizer, t_point_optimizer, t_tot_time;
  // map size seen by Reprojector::reprojectMap of this frame: keyframes with an overlapping field of view
  // (overlap_kfs_, <= max_n_kfs) and how many of their points fell inside the frame (the sum of the pairs' counts)
  int32_t n_overlap_kfs, n_kf_points_in_frame;
} pipe_result;

// FrameHandlerMono::overlap_kfs_ is protected (frame_handler_mono.h:69)
struct ExposedVo : public FrameHandlerMono {
  explicit ExposedVo(vk::AbstractCamera* cam) : FrameHandlerMono(cam) {}
  const std::vector<std::pair<FramePtr, size_t> >& overlap() const { return overlap_kfs_; }
};

struct Pipe {
  vk::AbstractCamera* cam;
  ExposedVo* vo;
};

void pipe_config_default(pipe_config* c) {
  c->n_pyr_levels = 3; c->klt_max_level = 4; c->klt_min_level = 2; c->grid_size = 30; c->max_fts = 120;
  c->max_n_kfs = 10; c->quality_min_fts = 50; c->quality_max_drop_fts = 40; c->structureoptim_max_pts = 20;
  c->structureoptim_num_iter = 5; c->poseoptim_num_iter = 10; c->shuffle_seed = 1; c->mapper_thread = 0; c->pool_slots = 0;
  c->defer_mapper = 0;
  c->kfselect_mindist = 0.12; c->poseoptim_thresh = 2.0; c->triang_min_corner_score = 20.0;
}

// cam_model 0 / 1: vk::PinholeCamera(width, height, p[0..3] = fx fy cx cy, p[4..8] = d0..d4);
// cam_model 2: vk::ATANCamera(width, height, p[0..3] = NORMALISED fx fy cx cy, p[4] = s)
void* pipe_create_cam(int width, int height, int cam_model, const double* p9, const pipe_config* c);

void* pipe_create(int width, int height, double fx, double fy, double cx, double cy, const pipe_config* c) {
  const double p9[9] = {fx, fy, cx, cy, 0, 0, 0, 0, 0};
  return pipe_create_cam(width, height, 0, p9, c);
}

void* pipe_create_cam(int width, int height, int cam_model, const double* p9, const pipe_config* c) {
  Config::nPyrLevels() = c->n_pyr_levels;
  Config::kltMaxLevel() = c->klt_max_level;
  Config::kltMinLevel() = c->klt_min_level;
  Config::gridSize() = c->grid_size;
  Config::maxFts() = c->max_fts;
  Config::maxNKfs() = c->max_n_kfs;
  Config::qualityMinFts() = c->quality_min_fts;
  Config::qualityMaxFtsDrop() = c->quality_max_drop_fts;
  Config::structureOptimMaxPts() = c->structureoptim_max_pts;
  Config::structureOptimNumIter() = c->structureoptim_num_iter;
  Config::poseOptimNumIter() = c->poseoptim_num_iter;
  Config::poseOptimThresh() = c->poseoptim_thresh;
  Config::kfSelectMinDist() = c->kfselect_mindist;
  Config::triangMinCornerScore() = c->triang_min_corner_score;
  Pipe* p = new Pipe;
  if (cam_model == 2)
    p->cam = new vk::ATANCamera(width, height, p9[0], p9[1], p9[2], p9[3], p9[4]);
  else
    p->cam = new vk::PinholeCamera(width, height, p9[0], p9[1], p9[2], p9[3], p9[4], p9[5], p9[6], p9[7], p9[8]);
#ifdef SVO_PIPELINE_HIP
  {
    // the device context of this image geometry, with the pool size this pipeline asks for (contexts outlive the
    // pipelines of a process: an earlier pipeline's small pool must not be inherited)
    const int levels = c->n_pyr_levels > c->klt_max_level + 1 ? c->n_pyr_levels : c->klt_max_level + 1;  // frame.cpp:58
    const int want = c->pool_slots > 0 ? c->pool_slots : 64;
    svo_hip::Device& dev = svo_hip::Device::forGeometry(width, height, levels);
    if (dev.slots() != want) dev.configure(width, height, levels, want);
  }
#endif
#ifdef SVO_PIPELINE_HIP
  svo_hip::Device::setDeferredMapping(c->defer_mapper != 0 && !c->mapper_thread);
#endif
  std::srand((unsigned)c->shuffle_seed);  // Reprojector::initializeGrid's random_shuffle (reprojector.cpp:54)
  p->vo = new ExposedVo(p->cam);
  p->vo->start();
  // run the mapper synchronously inside addFrame()/addKeyframe() (depth_filter.cpp:82-107):
  // deterministic interleaving of tracking and mapping for both libraries
  if (!c->mapper_thread) p->vo->depthFilter()->stopThread();
  return p;
}

void pipe_destroy(void* h) {
  Pipe* p = (Pipe*)h;
#ifdef SVO_PIPELINE_HIP
  svo_hip::Device::joinDeferredAll();  // a deferred update writes into the DepthFilter that is about to go
#endif
  delete p->vo;
#ifdef SVO_TRACE
  // ~FrameHandlerBase deletes the process-global trace monitor (frame_handler_base.cpp:82-84);
  // with two handlers alive the second destructor would delete it again
  g_permon = NULL;
#endif
  delete p->cam;
  delete p;
}

static void fill_result(Pipe* p, pipe_result* r) {
  std::memset(r, 0, sizeof(*r));
  FramePtr f = p->vo->lastFrame();
  if (f) {
    Matrix3d R = f->T_f_w_.rotation_matrix();
    Vector3d t = f->T_f_w_.translation();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r->T_f_w[i * 3 + j] = R(i, j);
    for (int i = 0; i < 3; ++i) r->T_f_w[9 + i] = t[i];
    r->is_keyframe = f->isKeyframe();
    r->frame_id = f->id_;
  }
  r->stage = (int)p->vo->stage();
  r->quality = (int)p->vo->trackingQuality();
  r->n_obs = (int)p->vo->lastNumObservations();
  r->n_kfs = (int)p->vo->map().size();
  r->n_candidates = (int)p->vo->map().point_candidates_.candidates_.size();
  r->n_seeds = (int)p->vo->depthFilter()->getSeeds().size();
  r->n_overlap_kfs = (int)p->vo->overlap().size();
  for (size_t i = 0; i < p->vo->overlap().size(); ++i) r->n_kf_points_in_frame += (int)p->vo->overlap()[i].second;
#ifdef SVO_TRACE
  vk::PerformanceMonitor* m = g_permon;
  r->img_align_n_tracked = m->get("img_align_n_tracked");
  r->repr_n_mps = m->get("repr_n_mps");
  r->repr_n_new_references = m->get("repr_n_new_references");
  r->sfba_thresh = m->get("sfba_thresh");
  r->sfba_error_init = m->get("sfba_error_init");
  r->sfba_error_final = m->get("sfba_error_final");
  r->sfba_n_edges_final = m->get("sfba_n_edges_final");
  r->dropout = m->get("dropout");
  r->t_pyramid_creation = m->get("pyramid_creation");
  r->t_sparse_img_align = m->get("sparse_img_align");
  r->t_reproject = m->get("reproject");
  r->t_pose_optimizer = m->get("pose_optimizer");
  r->t_point_optimizer = m->get("point_optimizer");
  r->t_tot_time = m->get("tot_time");
#endif
}

// benchmark_node.cpp:216-235: reference frame at a known pose, FAST corners of every pyramid
// level lifted through the (level-0) depth map `depth` [height][width] (z-depth along the
// optical axis is NOT what the reference uses: it scales the unit bearing, i.e. range).
int pipe_set_first_frame(void* h, const uint8_t* img, double timestamp, const double T_f_w[12], const float* range_map,
                         pipe_result* out) {
  Pipe* p = (Pipe*)h;
  const int w = p->cam->width(), hh = p->cam->height();
  cv::Mat m(hh, w, CV_8UC1);
  std::memcpy(m.data, img, (size_t)w * hh);
  FramePtr frame_ref(new Frame(p->cam, m, timestamp));
  Matrix3d R;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = T_f_w[i * 3 + j];
  frame_ref->T_f_w_ = SE3(R, Vector3d(T_f_w[9], T_f_w[10], T_f_w[11]));
  feature_detection::FastDetector detector(w, hh, Config::gridSize(), Config::nPyrLevels());
  detector.detect(frame_ref.get(), frame_ref->img_pyr_, Config::triangMinCornerScore(), frame_ref->fts_);
  for (Features::iterator it = frame_ref->fts_.begin(); it != frame_ref->fts_.end(); ++it) {
    Feature* ftr = *it;
    Vector3d pt_pos_cur = ftr->f * (double)range_map[(size_t)((int)ftr->px[1]) * w + (int)ftr->px[0]];
    Vector3d pt_pos_world = frame_ref->T_f_w_.inverse() * pt_pos_cur;
    Point* point = new Point(pt_pos_world, ftr);
    ftr->point = point;
  }
  const int n = (int)frame_ref->nObs();
  p->vo->setFirstFrame(frame_ref);
  if (out) fill_result(p, out);
  return n;
}

static std::string g_last_error;
const char* pipe_last_error(void) { return g_last_error.c_str(); }

// Returns the handler's stage, or -1 when a drop-in body threw (svo_hip::Error: no device, pool too small, a failed
// launch ...): the message is in pipe_last_error() and the pipeline should be closed (the reference's control plane is
// not written to be resumed after an exception).
int pipe_add_image(void* h, const uint8_t* img, double timestamp, pipe_result* out) {
  Pipe* p = (Pipe*)h;
  const int w = p->cam->width(), hh = p->cam->height();
  cv::Mat m(hh, w, CV_8UC1, (void*)img);
  try {
    p->vo->addImage(m, timestamp);  // addImage clones (frame_handler_mono.cpp:69)
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
  fill_result(p, out);
  return (int)p->vo->stage();
}

// pyramid-cache statistics of the device context (hip flavour; zeros otherwise)
void pipe_device_stats(uint64_t out[5]) {
  out[0] = out[1] = out[2] = out[3] = out[4] = 0;
#ifdef SVO_PIPELINE_HIP
  const svo_hip::Device::Stats st = svo_hip::Device::instance().statsSnapshot();
  out[0] = st.uploads; out[1] = st.evictions; out[2] = st.calls;
  out[3] = st.spec_hits; out[4] = st.spec_misses;  // pose refinements taken from / not taken from the reprojector's prediction
#endif
}

// the chain behind the sparse alignment (rpg_svo_amd/host/dropin/frame_chain.h): reprojectMap calls that took their first
// batch from the chain / that found one in flight and could not (hip flavour; zeros otherwise)
void pipe_chain_stats(uint64_t out[8]) {
  for (int i = 0; i < 8; ++i) out[i] = 0;
#ifdef SVO_PIPELINE_HIP
  const svo_hip::Device::Stats st = svo_hip::Device::instance().statsSnapshot();
  out[0] = st.chain_hits; out[1] = st.chain_misses;
  for (int i = 0; i < 6; ++i) out[2 + i] = st.chain_miss_why[i];  // not this frame, pose bits, keyframe ranking, map moved on, capacity
#endif
}

// the depth filter's update enqueued by the pose optimizer's drop-in (dropin/depth_filter.cpp, EarlyUpdate): updates the
// reference's own updateSeeds call then found running and took / early updates that were dropped (hip flavour; zeros otherwise)
void pipe_early_mapper_stats(uint64_t out[3]) {
  out[0] = out[1] = out[2] = 0;
#ifdef SVO_PIPELINE_HIP
  const svo_hip::Device::Stats st = svo_hip::Device::instance().statsSnapshot();
  out[0] = st.early_map_hits; out[1] = st.early_map_misses; out[2] = st.early_map_two_phase;
#endif
}

// the map mirror of the reprojector's drop-in (row N2): calls, rebuilds, fallbacks to the list-walking path, point records
// sent, observation records sent, second batches (hip flavour; zeros otherwise)
void pipe_mirror_stats(uint64_t out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = 0;
#ifdef SVO_PIPELINE_HIP
  if (svo::hip_dropin::mapMirrorStats) svo::hip_dropin::mapMirrorStats(out);
#endif
}

// the resident seed store of the depth filter's drop-in (row N2): calls, seed records sent, rebuilds (hip flavour; zeros otherwise)
void pipe_seed_store_stats(uint64_t out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = 0;
#ifdef SVO_PIPELINE_HIP
  if (svo::hip_dropin::seedStoreStats) svo::hip_dropin::seedStoreStats(out);
#endif
}

// per stage (sparse_align, reproject, pose_opt, depth_filter): calls, marshal us, device round-trip us,
// unmarshal us, payload bytes -- 5 doubles each; out[20] = pyramid upload us (svo_hip::Device::Stats)
void pipe_stage_times(double out[21]) {
  for (int i = 0; i < 21; ++i) out[i] = 0;
#ifdef SVO_PIPELINE_HIP
  const svo_hip::Device::Stats s = svo_hip::Device::instance().statsSnapshot();
  for (int k = 0; k < svo_hip::Device::N_STAGES; ++k) {
    out[5 * k] = (double)s.n[k]; out[5 * k + 1] = s.marshal_us[k]; out[5 * k + 2] = s.device_us[k];
    out[5 * k + 3] = s.unmarshal_us[k]; out[5 * k + 4] = s.payload_bytes[k];
  }
  out[20] = s.pyr_upload_us;
#endif
}

// features of the last frame (px, level, has point), for inspection
int pipe_last_features(void* h, int max_n, double* px, int32_t* level, double* pos) {
  Pipe* p = (Pipe*)h;
  FramePtr f = p->vo->lastFrame();
  int n = 0;
  if (!f) return 0;
  for (Features::iterator it = f->fts_.begin(); it != f->fts_.end() && n < max_n; ++it, ++n) {
    px[2 * n] = (*it)->px[0]; px[2 * n + 1] = (*it)->px[1];
    level[n] = (*it)->level;
    for (int k = 0; k < 3; ++k) pos[3 * n + k] = (*it)->point ? (*it)->point->pos_[k] : 0.0;
  }
  return n;
}

// the depth filter's seed list, for inspection (scripts/dropin_many.py: which seed sits on the convergence threshold where two
// runs part ways): per seed batch_id, the feature's frame id, px, py, a, b, mu, z_range, sigma2 (9 doubles)
int pipe_seeds(void* h, int max_n, double* out) {
  Pipe* p = (Pipe*)h;
  int n = 0;
  std::list<Seed>& seeds = p->vo->depthFilter()->getSeeds();
  for (std::list<Seed>::iterator it = seeds.begin(); it != seeds.end() && n < max_n; ++it, ++n) {
    double* o = out + 9 * n;
    o[0] = it->batch_id; o[1] = it->ftr && it->ftr->frame ? it->ftr->frame->id_ : -1;
    o[2] = it->ftr ? it->ftr->px[0] : 0.0; o[3] = it->ftr ? it->ftr->px[1] : 0.0;
    o[4] = it->a; o[5] = it->b; o[6] = it->mu; o[7] = it->z_range; o[8] = it->sigma2;
  }
  return n;
}

// the last frame's host pyramid: number of levels, and how many of them hold an image (the full drop-in leaves the levels
// above 0 empty: rpg_svo_amd/host/dropin/frame.cpp)
int pipe_last_host_pyramid(void* h, int* n_levels) {
  Pipe* p = (Pipe*)h;
  FramePtr f = p->vo->lastFrame();
  if (!f) return -1;
  *n_levels = (int)f->img_pyr_.size();
  int filled = 0;
  for (size_t i = 0; i < f->img_pyr_.size(); ++i) filled += (f->img_pyr_[i].data != NULL && f->img_pyr_[i].rows > 0) ? 1 : 0;
  return filled;
}

// Does constructing SparseImgAlign with vk::NLLSSolver's LevenbergMarquardt method throw?  The all-CPU reference accepts it
// (0); the drop-in, whose kernel runs the Gauss-Newton loop only, must say so (1) instead of silently running Gauss-Newton.
int pipe_sparse_align_rejects_levenberg_marquardt(void) {
  try {
    SparseImgAlign a(2, 0, 10, SparseImgAlign::LevenbergMarquardt, false, false);
    (void)a;
  } catch (const std::exception&) {
    return 1;
  }
  return 0;
}

}  // extern "C"
