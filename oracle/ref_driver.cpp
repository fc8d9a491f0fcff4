// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C entry points (prefix ref_, same argument lists as the orc_ functions in svo_oracle.h)
// around the REFERENCE'S OWN translation units, which oracle/Makefile compiles where they
// lie under $(REF)/svo/src against the dependency shims in oracle/shim/.  This file builds
// the reference's objects (svo::Frame, Feature, Point, Seed, Matcher, DepthFilter,
// SparseImgAlign ...) from plain arrays, calls the reference's methods and copies the
// results out.  No arithmetic of the path is implemented here.
//
// Output: oracle/_ref/libsvo_ref.so (git-ignored; travels with gpurun).  Used by
// tests/test_oracle_vs_ref.py to pin the C restatement, never by the product.
#include <atomic>
#include <malloc.h>
#include <chrono>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include <svo/config.h>
#include <svo/depth_filter.h>
#include <svo/feature.h>
#include <svo/feature_alignment.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/matcher.h>
#include <svo/point.h>
#include <svo/pose_optimizer.h>
#include <svo/sparse_img_align.h>
#include <vikit/atan_camera.h>
#include <vikit/pinhole_camera.h>
#include <vikit/vision.h>

#include "svo_oracle.h"

namespace vk {
int g_halfsample_mode = 2;  // x86 dispatch (SSE2 flavour iff cols % 16 == 0)
}

using namespace svo;

namespace {

SE3 se3_from_Rt(const double T[12]) {
  Matrix3d R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = T[i * 3 + j];
  return SE3(R, Vector3d(T[9], T[10], T[11]));
}
void se3_to_Rt(const SE3& S, double T[12]) {
  Matrix3d R = S.rotation_matrix();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = R(i, j);
  Vector3d t = S.translation();
  T[9] = t[0]; T[10] = t[1]; T[11] = t[2];
}

// the reference-side camera object for a camera description (orc_camera.h)
vk::AbstractCamera* make_cam(const orc_pinhole* c) {
  if (c->model == ORC_CAM_ATAN) return new vk::ATANCamera(*c);
  return new vk::PinholeCamera(c->width, c->height, c->fx, c->fy, c->cx, c->cy, c->d[0], c->d[1], c->d[2], c->d[3], c->d[4]);
}

// A reference Frame built from level 0 (the reference builds its own pyramid through
// frame_utils::createImgPyramid -> vk::halfSample); levels handed in are then copied over
// the built ones so that both sides look at identical pixels whatever the half-sample
// flavour the caller used.
FramePtr make_frame(vk::AbstractCamera* cam, const orc_pyramid* p, const double T_f_w[12]) {
  Config::nPyrLevels() = p->n_levels;
  Config::kltMaxLevel() = p->n_levels - 1;
  cv::Mat img0(p->h[0], p->w[0], CV_8UC1);
  std::memcpy(img0.data, p->data[0], (size_t)p->w[0] * p->h[0]);
  FramePtr f(new Frame(cam, img0, 0.0));
  for (int l = 1; l < p->n_levels; ++l) {
    cv::Mat& m = f->img_pyr_[l];
    if (m.cols == p->w[l] && m.rows == p->h[l]) std::memcpy(m.data, p->data[l], (size_t)p->w[l] * p->h[l]);
  }
  f->T_f_w_ = se3_from_Rt(T_f_w);
  return f;
}

Feature* make_feature(Frame* frame, const orc_feature* o) {
  Feature* ftr = new Feature(frame, Vector2d(o->px[0], o->px[1]), Vector3d(o->f[0], o->f[1], o->f[2]), o->level);
  ftr->type = o->type == ORC_FTR_EDGELET ? Feature::EDGELET : Feature::CORNER;
  ftr->grad = Vector2d(o->grad[0], o->grad[1]);
  return ftr;
}

void apply_matcher_options(Matcher& m, const orc_matcher_options* o) {
  m.options_.align_1d = o->align_1d;
  m.options_.align_max_iter = o->align_max_iter;
  m.options_.max_epi_length_optim = o->max_epi_length_optim;
  m.options_.max_epi_search_steps = o->max_epi_search_steps;
  m.options_.subpix_refinement = o->subpix_refinement;
  m.options_.epi_search_edgelet_filtering = o->epi_search_edgelet_filtering;
  m.options_.epi_search_edgelet_max_angle = o->epi_search_edgelet_max_angle;
}

void copy_matcher_state(const Matcher& m, orc_match_result* res) {
  res->search_level = m.search_level_;
  res->reject = m.reject_;
  res->A_cur_ref[0] = m.A_cur_ref_(0, 0); res->A_cur_ref[1] = m.A_cur_ref_(0, 1);
  res->A_cur_ref[2] = m.A_cur_ref_(1, 0); res->A_cur_ref[3] = m.A_cur_ref_(1, 1);
  res->h_inv = m.h_inv_;
  res->epi_length = m.epi_length_;
  res->px_cur[0] = m.px_cur_[0]; res->px_cur[1] = m.px_cur_[1];
  std::memcpy(res->patch, m.patch_, 64);
  std::memcpy(res->patch_with_border, m.patch_with_border_, 100);
}

// exposes the protected bits of SparseImgAlign the comparison needs
class SiaProbe : public SparseImgAlign {
 public:
  int evals[ORC_MAX_LEVELS];
  SiaProbe(int max_level, int min_level, int n_iter, double eps)
      : SparseImgAlign(max_level, min_level, n_iter, GaussNewton, false, false) {
    eps_ = eps;
    std::memset(evals, 0, sizeof(evals));
  }
  virtual double computeResiduals(const SE3& model, bool linearize_system, bool compute_weight_scale = false) {
    if (level_ >= 0 && level_ < ORC_MAX_LEVELS) evals[level_]++;
    return SparseImgAlign::computeResiduals(model, linearize_system, compute_weight_scale);
  }
  const std::vector<bool>& visible() const { return visible_fts_; }
  const Matrix<double, 6, 6>& H() const { return H_; }
  double chi2() const { return chi2_; }
};

class DepthFilterProbe : public DepthFilter {
 public:
  DepthFilterProbe(feature_detection::DetectorPtr d, callback_t cb) : DepthFilter(d, cb) {}
  void run(FramePtr frame) { updateSeeds(frame); }
  Matcher& matcher() { return matcher_; }
};

class NullDetector : public feature_detection::AbstractDetector {
 public:
  NullDetector(int w, int h) : AbstractDetector(w, h, 25, 3) {}
  virtual void detect(Frame*, const ImgPyr&, const double, Features&) {}
};

}  // namespace

extern "C" {

void ref_set_halfsample_mode(int mode) { vk::g_halfsample_mode = mode; }

// frame_utils::createImgPyramid (svo/src/frame.cpp:156-165) through the reference itself
void ref_create_img_pyramid(const uint8_t* lvl0, int w, int h, int n_levels, int mode, uint8_t* const* levels_out) {
  vk::g_halfsample_mode = mode;
  cv::Mat img0(h, w, CV_8UC1);
  std::memcpy(img0.data, lvl0, (size_t)w * h);
  ImgPyr pyr;
  frame_utils::createImgPyramid(img0, n_levels, pyr);
  for (int l = 0; l < n_levels; ++l) std::memcpy(levels_out[l], pyr[l].data, (size_t)pyr[l].cols * pyr[l].rows);
}

int ref_sparse_img_align_run(const orc_pyramid* ref_pyr, const orc_pyramid* cur_pyr, const orc_pinhole* cam,
                             const double T_ref_w[12], double T_cur_w[12], int n, const double* px, const double* f,
                             const uint8_t* has_point, const double* pos, const orc_sia_options* opt,
                             orc_sia_result* res, uint8_t* visible_out) {
  std::memset(res, 0, sizeof(*res));
  vk::AbstractCamera* c = make_cam(cam);
  FramePtr ref = make_frame(c, ref_pyr, T_ref_w);
  FramePtr cur = make_frame(c, cur_pyr, T_cur_w);
  std::vector<Point*> points;
  for (int i = 0; i < n; ++i) {
    Feature* ftr = new Feature(ref.get(), Vector2d(px[2 * i], px[2 * i + 1]),
                               Vector3d(f[3 * i], f[3 * i + 1], f[3 * i + 2]), 0);
    if (has_point[i]) {
      Point* p = new Point(Vector3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
      points.push_back(p);
      ftr->point = p;
    }
    ref->addFeature(ftr);
  }
  SiaProbe sia(opt->max_level, opt->min_level, opt->n_iter, opt->eps);
  SE3 T_cur_before = cur->T_f_w_;
  size_t tracked = sia.run(ref, cur);
  se3_to_Rt(cur->T_f_w_, T_cur_w);
  SE3 T_cur_from_ref = cur->T_f_w_ * ref->T_f_w_.inverse();
  if (n == 0) T_cur_from_ref = T_cur_before * ref->T_f_w_.inverse();
  se3_to_Rt(T_cur_from_ref, res->T_cur_from_ref);
  res->n_tracked = (int)tracked;
  res->stop = sia.stop_;
  res->chi2 = sia.chi2();
  for (int l = 0; l < ORC_MAX_LEVELS; ++l) res->iters[l] = sia.evals[l];
  if (n > 0)
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) res->H[i * 6 + j] = sia.H()(i, j);
  if (visible_out && n > 0)
    for (int i = 0; i < n; ++i) visible_out[i] = sia.visible()[i];
  for (Point* p : points) delete p;
  ref.reset();
  cur.reset();
  delete c;
  return (int)tracked;
}

// Batch driver for bench.py's cpu_baseline (kind "reference"): the reference's own
// SparseImgAlign::run over B problems on n_threads host threads.  All Frame / Feature / Point
// objects are built first; *seconds_out is the wall time of the run() calls alone.
int ref_sparse_img_align_batch(int B, const orc_pyramid* pyrs, const int* ref_slot, const int* cur_slot,
                               const orc_pinhole* cam, const double* T_ref_w, double* T_cur_w, const int* n, int n_stride,
                               const double* px, const double* f, const uint8_t* has_point, const double* pos,
                               const orc_sia_options* opt, orc_sia_result* res, int n_threads, double* seconds_out) {
  vk::AbstractCamera* c = make_cam(cam);
  std::vector<FramePtr> refs(B), curs(B);
  std::vector<Point*> points;
  for (int b = 0; b < B; ++b) {
    refs[b] = make_frame(c, &pyrs[ref_slot[b]], T_ref_w + 12 * b);
    curs[b] = make_frame(c, &pyrs[cur_slot[b]], T_cur_w + 12 * b);
    for (int i = 0; i < n[b]; ++i) {
      const size_t k = (size_t)b * n_stride + i;
      Feature* ftr = new Feature(refs[b].get(), Vector2d(px[2 * k], px[2 * k + 1]), Vector3d(f[3 * k], f[3 * k + 1], f[3 * k + 2]), 0);
      if (has_point[k]) {
        Point* p = new Point(Vector3d(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]));
        points.push_back(p);
        ftr->point = p;
      }
      refs[b]->addFeature(ftr);
    }
  }
  if (n_threads < 1) n_threads = 1;
  // SparseImgAlign::run allocates its caches per call (jacobian_cache_ is 150 KB at 200 patches:
  // above glibc's mmap threshold, i.e. an mmap + page faults + munmap per frame, all serialised
  // on the process' mm lock).  Serve them from the per-thread malloc arenas instead, so that the
  // many-thread baseline measures the reference's arithmetic and not the kernel's VM lock.
  mallopt(M_MMAP_THRESHOLD, 64 << 20);
  mallopt(M_TRIM_THRESHOLD, 512 << 20);
  mallopt(M_ARENA_MAX, 1024);
  std::atomic<int> next(0);
  auto worker = [&]() {
    for (;;) {
      const int b = next.fetch_add(1);
      if (b >= B) break;
      std::memset(&res[b], 0, sizeof(res[b]));
      SiaProbe sia(opt->max_level, opt->min_level, opt->n_iter, opt->eps);
      const size_t tracked = sia.run(refs[b], curs[b]);
      res[b].n_tracked = (int)tracked;
      res[b].stop = sia.stop_;
      res[b].chi2 = sia.chi2();
      for (int l = 0; l < ORC_MAX_LEVELS; ++l) res[b].iters[l] = sia.evals[l];
      if (n[b] > 0)
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) res[b].H[i * 6 + j] = sia.H()(i, j);
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (n_threads == 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int b = 0; b < B; ++b) {
    se3_to_Rt(curs[b]->T_f_w_, T_cur_w + 12 * b);
    se3_to_Rt(curs[b]->T_f_w_ * refs[b]->T_f_w_.inverse(), res[b].T_cur_from_ref);
  }
  for (Point* p : points) delete p;
  refs.clear();
  curs.clear();
  delete c;
  return 0;
}

int ref_align2d(const uint8_t* cur_img, int w, int h, int stride, const uint8_t* ref_patch_with_border,
                const uint8_t* ref_patch, int n_iter, double px[2]) {
  cv::Mat img(h, w, CV_8UC1, (void*)cur_img, (size_t)stride);
  uint8_t pwb[100] __attribute__((aligned(16)));
  uint8_t pat[64] __attribute__((aligned(16)));
  std::memcpy(pwb, ref_patch_with_border, 100);
  std::memcpy(pat, ref_patch, 64);
  Vector2d p(px[0], px[1]);
  bool ok = feature_alignment::align2D(img, pwb, pat, n_iter, p);
  px[0] = p[0]; px[1] = p[1];
  return ok;
}

int ref_align1d(const uint8_t* cur_img, int w, int h, int stride, const float dir[2],
                const uint8_t* ref_patch_with_border, const uint8_t* ref_patch, int n_iter, double px[2],
                double* h_inv) {
  cv::Mat img(h, w, CV_8UC1, (void*)cur_img, (size_t)stride);
  uint8_t pwb[100] __attribute__((aligned(16)));
  uint8_t pat[64] __attribute__((aligned(16)));
  std::memcpy(pwb, ref_patch_with_border, 100);
  std::memcpy(pat, ref_patch, 64);
  Vector2d p(px[0], px[1]);
  bool ok = feature_alignment::align1D(img, Vector2f(dir[0], dir[1]), pwb, pat, n_iter, p, *h_inv);
  px[0] = p[0]; px[1] = p[1];
  return ok;
}

void ref_get_warp_matrix_affine(const orc_pinhole* cam_ref, const orc_pinhole* cam_cur, const double px_ref[2],
                                const double f_ref[3], double depth_ref, const double T_cur_ref[12], int level_ref,
                                double A_cur_ref[4]) {
  vk::AbstractCamera* cr = make_cam(cam_ref);
  vk::AbstractCamera* cc = make_cam(cam_cur);
  Matrix2d A;
  warp::getWarpMatrixAffine(*cr, *cc, Vector2d(px_ref[0], px_ref[1]), Vector3d(f_ref[0], f_ref[1], f_ref[2]),
                            depth_ref, se3_from_Rt(T_cur_ref), level_ref, A);
  A_cur_ref[0] = A(0, 0); A_cur_ref[1] = A(0, 1); A_cur_ref[2] = A(1, 0); A_cur_ref[3] = A(1, 1);
  delete cr;
  delete cc;
}

int ref_get_best_search_level(const double A_cur_ref[4], int max_level) {
  Matrix2d A;
  A(0, 0) = A_cur_ref[0]; A(0, 1) = A_cur_ref[1]; A(1, 0) = A_cur_ref[2]; A(1, 1) = A_cur_ref[3];
  return warp::getBestSearchLevel(A, max_level);
}

int ref_warp_affine(const double A_cur_ref[4], const uint8_t* img_ref, int w, int h, int stride,
                    const double px_ref[2], int level_ref, int search_level, int halfpatch_size, uint8_t* patch) {
  Matrix2d A;
  A(0, 0) = A_cur_ref[0]; A(0, 1) = A_cur_ref[1]; A(1, 0) = A_cur_ref[2]; A(1, 1) = A_cur_ref[3];
  cv::Mat img(h, w, CV_8UC1, (void*)img_ref, (size_t)stride);
  const int n = 4 * halfpatch_size * halfpatch_size;
  std::vector<uint8_t> sentinel(patch, patch + n);
  warp::warpAffine(A, img, Vector2d(px_ref[0], px_ref[1]), level_ref, search_level, halfpatch_size, patch);
  const Matrix2f Ai = A.inverse().cast<float>();
  return std::isnan(Ai(0, 0)) ? 0 : 1;
}

int ref_find_match_direct(const orc_frame* frames, const orc_pinhole* cam, int cur_frame, const double pt_pos[3],
                          int n_obs, const orc_feature* obs, const orc_matcher_options* opt, double px_cur[2],
                          orc_match_result* res) {
  res->success = 0;
  res->ref_obs = -1;
  if (n_obs <= 0) return 0;
  vk::AbstractCamera* c = make_cam(cam);
  std::map<int, FramePtr> fr;
  auto get = [&](int idx) {
    auto it = fr.find(idx);
    if (it != fr.end()) return it->second;
    FramePtr f = make_frame(c, &frames[idx].pyr, frames[idx].T_f_w);
    fr[idx] = f;
    return f;
  };
  FramePtr cur = get(cur_frame);
  Point pt(Vector3d(pt_pos[0], pt_pos[1], pt_pos[2]));
  std::vector<Feature*> fts;
  // Point::obs_ in list order: push_back keeps obs[0] at the front
  for (int i = 0; i < n_obs; ++i) {
    Feature* ftr = make_feature(get(obs[i].frame).get(), &obs[i]);
    fts.push_back(ftr);
    pt.obs_.push_back(ftr);
  }
  Config::nPyrLevels() = opt->n_pyr_levels;
  Matcher m;
  apply_matcher_options(m, opt);
  m.ref_ftr_ = NULL;
  m.search_level_ = 0;
  m.reject_ = false;
  m.h_inv_ = 0;
  m.epi_length_ = 0;
  m.A_cur_ref_.setZero();
  m.px_cur_.setZero();
  std::memcpy(m.patch_, res->patch, 64);
  std::memcpy(m.patch_with_border_, res->patch_with_border, 100);
  Vector2d px(px_cur[0], px_cur[1]);
  bool ok = m.findMatchDirect(pt, *cur, px);
  px_cur[0] = px[0]; px_cur[1] = px[1];
  double A_keep[4] = {res->A_cur_ref[0], res->A_cur_ref[1], res->A_cur_ref[2], res->A_cur_ref[3]};
  copy_matcher_state(m, res);
  (void)A_keep;
  res->success = ok;
  res->px_cur[0] = px[0]; res->px_cur[1] = px[1];
  for (int i = 0; i < n_obs; ++i)
    if (fts[i] == m.ref_ftr_) res->ref_obs = i;
  for (Feature* f : fts) delete f;
  fr.clear();
  cur.reset();
  delete c;
  return ok;
}

int ref_find_epipolar_match_direct(const orc_frame* frames, const orc_pinhole* cam, int ref_frame, int cur_frame,
                                   const orc_feature* ref_ftr, double d_estimate, double d_min, double d_max,
                                   const orc_matcher_options* opt, orc_match_result* res) {
  vk::AbstractCamera* c = make_cam(cam);
  FramePtr ref = make_frame(c, &frames[ref_frame].pyr, frames[ref_frame].T_f_w);
  FramePtr cur = make_frame(c, &frames[cur_frame].pyr, frames[cur_frame].T_f_w);
  Feature* ftr = make_feature(ref.get(), ref_ftr);
  Config::nPyrLevels() = opt->n_pyr_levels;
  Matcher m;
  apply_matcher_options(m, opt);
  m.search_level_ = 0; m.reject_ = false; m.h_inv_ = 0; m.epi_length_ = 0;
  m.A_cur_ref_.setZero(); m.px_cur_.setZero();
  std::memcpy(m.patch_, res->patch, 64);
  std::memcpy(m.patch_with_border_, res->patch_with_border, 100);
  double depth = 0;
  bool ok = m.findEpipolarMatchDirect(*ref, *cur, *ftr, d_estimate, d_min, d_max, depth);
  copy_matcher_state(m, res);
  res->success = ok;
  res->ref_obs = 0;
  res->depth = depth;
  delete ftr;
  ref.reset();
  cur.reset();
  delete c;
  return ok;
}

int ref_pose_optimize(double reproj_thresh, int n_iter, const orc_pinhole* cam, const double T_f_w[12], int n,
                      const double* f, const int* level, uint8_t* has_point, const double* pos,
                      orc_pose_opt_result* res) {
  std::memset(res, 0, sizeof(*res));
  std::memcpy(res->T_f_w, T_f_w, sizeof(double) * 12);
  vk::AbstractCamera* c = make_cam(cam);
  Config::nPyrLevels() = 1;
  Config::kltMaxLevel() = 0;
  cv::Mat img0(cam->height, cam->width, CV_8UC1, cv::Scalar(0));
  FramePtr frame(new Frame(c, img0, 0.0));
  frame->T_f_w_ = se3_from_Rt(T_f_w);
  frame->Cov_.setZero();
  std::vector<Point*> points;
  std::vector<Feature*> fts;
  for (int i = 0; i < n; ++i) {
    Feature* ftr = new Feature(frame.get(), Vector2d(0, 0), Vector3d(f[3 * i], f[3 * i + 1], f[3 * i + 2]), level[i]);
    if (has_point[i]) {
      Point* p = new Point(Vector3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
      points.push_back(p);
      ftr->point = p;
    }
    frame->addFeature(ftr);
    fts.push_back(ftr);
  }
  double estimated_scale = 0, error_init = 0, error_final = 0;
  size_t num_obs = 0;
  bool any = false;
  for (int i = 0; i < n; ++i) any = any || has_point[i];
  pose_optimizer::optimizeGaussNewton(reproj_thresh, (size_t)n_iter, false, frame, estimated_scale, error_init,
                                      error_final, num_obs);
  res->ran = any;
  if (any) {
    se3_to_Rt(frame->T_f_w_, res->T_f_w);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) res->Cov[i * 6 + j] = frame->Cov_(i, j);
    res->estimated_scale = estimated_scale;
    res->error_init = error_init;
    res->error_final = error_final;
    res->num_obs = (int)num_obs;
    for (int i = 0; i < n; ++i) has_point[i] = fts[i]->point != NULL;
  }
  for (Point* p : points) delete p;
  frame.reset();
  delete c;
  return any;
}

void ref_point_optimize(int n_iter, int n_obs, const double* T_f_w, const double* f, double pos[3]) {
  orc_pinhole pc = {100, 100, 50, 50, 100, 100};
  vk::AbstractCamera* c = make_cam(&pc);
  Config::nPyrLevels() = 1;
  Config::kltMaxLevel() = 0;
  std::vector<FramePtr> frames;
  Point pt(Vector3d(pos[0], pos[1], pos[2]));
  std::vector<Feature*> fts;
  for (int i = 0; i < n_obs; ++i) {
    cv::Mat img0(100, 100, CV_8UC1, cv::Scalar(0));
    FramePtr fr(new Frame(c, img0, 0.0));
    fr->T_f_w_ = se3_from_Rt(T_f_w + 12 * i);
    frames.push_back(fr);
    Feature* ftr = new Feature(fr.get(), Vector2d(0, 0), Vector3d(f[3 * i], f[3 * i + 1], f[3 * i + 2]), 0);
    fts.push_back(ftr);
    pt.obs_.push_back(ftr);
  }
  pt.optimize((size_t)n_iter);
  pos[0] = pt.pos_[0]; pos[1] = pt.pos_[1]; pos[2] = pt.pos_[2];
  for (Feature* f2 : fts) delete f2;
  frames.clear();
  delete c;
}

void ref_seed_init(orc_seed* s, float depth_mean, float depth_min) {
  Seed seed(NULL, depth_mean, depth_min);
  s->a = seed.a; s->b = seed.b; s->mu = seed.mu; s->z_range = seed.z_range; s->sigma2 = seed.sigma2;
}

void ref_update_seed(float x, float tau2, orc_seed* s) {
  Seed seed(NULL, 1.0f, 1.0f);
  seed.a = s->a; seed.b = s->b; seed.mu = s->mu; seed.z_range = s->z_range; seed.sigma2 = s->sigma2;
  DepthFilter::updateSeed(x, tau2, &seed);
  s->a = seed.a; s->b = seed.b; s->mu = seed.mu; s->z_range = seed.z_range; s->sigma2 = seed.sigma2;
}

double ref_compute_tau(const double T_ref_cur[12], const double f[3], double z, double px_error_angle) {
  return DepthFilter::computeTau(se3_from_Rt(T_ref_cur), Vector3d(f[0], f[1], f[2]), z, px_error_angle);
}

int ref_update_seeds(const orc_frame* frames, const orc_pinhole* cam, int cur_frame, int n_seeds, orc_seed* seeds,
                     orc_seed_update_info* info, const orc_depth_filter_options* dopt,
                     const orc_matcher_options* mopt) {
  vk::AbstractCamera* c = make_cam(cam);
  std::map<int, FramePtr> fr;
  auto get = [&](int idx) {
    auto it = fr.find(idx);
    if (it != fr.end()) return it->second;
    FramePtr f = make_frame(c, &frames[idx].pyr, frames[idx].T_f_w);
    fr[idx] = f;
    return f;
  };
  FramePtr cur = get(cur_frame);
  Config::nPyrLevels() = mopt->n_pyr_levels;

  std::map<Feature*, int> ftr2seed;
  struct Conv { double xyz[3]; double sigma2; };
  std::map<int, Conv> converged;
  std::vector<Point*> new_points;
  DepthFilter::callback_t cb = [&](Point* p, double sigma2_after) {
    Feature* ftr = p->obs_.front();
    Conv cv_;
    cv_.xyz[0] = p->pos_[0]; cv_.xyz[1] = p->pos_[1]; cv_.xyz[2] = p->pos_[2];
    cv_.sigma2 = sigma2_after;  // seed_converged_cb_(point, it->sigma2), depth_filter.cpp:278: the seed's variance AFTER the update
    converged[ftr2seed[ftr]] = cv_;
    new_points.push_back(p);
  };
  feature_detection::DetectorPtr det(new NullDetector(cam->width, cam->height));
  DepthFilterProbe df(det, cb);
  df.options_.max_n_kfs = dopt->max_n_kfs;
  df.options_.seed_convergence_sigma2_thresh = dopt->seed_convergence_sigma2_thresh;
  apply_matcher_options(df.matcher(), mopt);
  Seed::batch_counter = dopt->batch_counter;
  Seed::seed_counter = 0;
  std::vector<Feature*> fts;
  std::list<Seed>& sl = df.getSeeds();
  for (int i = 0; i < n_seeds; ++i) {
    Feature* ftr = make_feature(get(seeds[i].ftr.frame).get(), &seeds[i].ftr);
    fts.push_back(ftr);
    ftr2seed[ftr] = i;
    Seed s(ftr, 1.0f, 1.0f);  // id = i because seed_counter was reset
    s.batch_id = seeds[i].batch_id;
    s.a = seeds[i].a; s.b = seeds[i].b; s.mu = seeds[i].mu; s.z_range = seeds[i].z_range; s.sigma2 = seeds[i].sigma2;
    sl.push_back(s);
  }
  std::vector<orc_seed> before(seeds, seeds + n_seeds);
  df.run(cur);
  // survivors
  std::vector<bool> alive(n_seeds, false);
  for (auto it = sl.begin(); it != sl.end(); ++it) {
    const int i = it->id;
    alive[i] = true;
    seeds[i].a = it->a; seeds[i].b = it->b; seeds[i].mu = it->mu; seeds[i].sigma2 = it->sigma2;
  }
  int n_updates = 0;
  for (int i = 0; i < n_seeds; ++i) {
    std::memset(&info[i], 0, sizeof(info[i]));
    const bool changed = std::memcmp(&before[i].a, &seeds[i].a, sizeof(float) * 5) != 0;
    if (converged.count(i)) {
      info[i].status = ORC_SEED_CONVERGED;
      for (int k = 0; k < 3; ++k) info[i].xyz_world[k] = converged[i].xyz[k];
      // The seed is erased from the list (depth_filter.cpp:280), but its state after the update is not lost: sigma2 is
      // what the callback was handed, and the new point sits at T_f_w^-1 * (f / mu) (:264), i.e. 1 / mu = its distance
      // from the seed's frame (f is a unit vector; the double round trip is 1e-15 relative, far inside a float's 6e-8:
      // rounding returns mu's own bits).  a and b leave no trace: they stay at their values before the update.
      {
        const Vector3d p_f = get(before[i].ftr.frame)->T_f_w_ * Vector3d(converged[i].xyz[0], converged[i].xyz[1], converged[i].xyz[2]);
        seeds[i].mu = (float)(1.0 / p_f.norm());
        seeds[i].sigma2 = (float)converged[i].sigma2;
      }
      ++n_updates;
    } else if (!alive[i]) {
      // erased without a callback: too old, or NaN after an update (cannot be told apart from
      // outside; the age rule is re-evaluated here, it is not arithmetic)
      info[i].status = ((dopt->batch_counter - before[i].batch_id) > dopt->max_n_kfs) ? ORC_SEED_ERASED_OLD : ORC_SEED_NAN;
      if (info[i].status == ORC_SEED_NAN) ++n_updates;
    } else if (!changed) {
      info[i].status = 0;  // behind camera / outside the image: the reference leaves no trace
    } else if (seeds[i].b == before[i].b + 1 && seeds[i].a == before[i].a && seeds[i].mu == before[i].mu) {
      info[i].status = ORC_SEED_NO_MATCH;
    } else {
      info[i].status = ORC_SEED_UPDATED;
      ++n_updates;
    }
  }
  for (Point* p : new_points) delete p;
  for (Feature* f : fts) delete f;
  sl.clear();
  fr.clear();
  cur.reset();
  delete c;
  return n_updates;
}

// The reference's own FastDetector::detect (feature_detection.cpp:66-114) on the shimmed FAST
// library: features in the order the detector emits them (cell order).  Returns their count.
int ref_fast_detect(const orc_pyramid* pyr, const orc_pinhole* cam, int n_levels, int cell_size, const uint8_t* occupancy,
                    double detection_threshold, int max_out, double* px_out, int32_t* level_out) {
  vk::AbstractCamera* c = make_cam(cam);
  const double T0[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  FramePtr frame = make_frame(c, pyr, T0);
  feature_detection::FastDetector det(cam->width, cam->height, cell_size, n_levels);
  const int cols = (cam->width + cell_size - 1) / cell_size, rows = (cam->height + cell_size - 1) / cell_size;
  if (occupancy)
    for (int k = 0; k < cols * rows; ++k)
      if (occupancy[k]) det.setGridOccpuancy(Vector2d((k % cols) * cell_size + 0.5, (k / cols) * cell_size + 0.5));
  Features fts;
  det.detect(frame.get(), frame->img_pyr_, detection_threshold, fts);
  int n = 0;
  for (Features::iterator it = fts.begin(); it != fts.end(); ++it) {
    if (n < max_out) {
      px_out[2 * n] = (*it)->px[0]; px_out[2 * n + 1] = (*it)->px[1];
      level_out[n] = (*it)->level;
    }
    ++n;
    delete *it;
  }
  frame.reset();
  delete c;
  return n;
}

// Frame::c2f of n pixels through the reference's camera object: what the Feature constructor stores in Feature::f
void ref_cam2world(const orc_pinhole* cam, int n, const double* px, double* f) {
  vk::AbstractCamera* c = make_cam(cam);
  for (int i = 0; i < n; ++i) {
    const Vector3d v(c->cam2world(px[2 * i], px[2 * i + 1]));
    f[3 * i] = v[0]; f[3 * i + 1] = v[1]; f[3 * i + 2] = v[2];
  }
  delete c;
}

int ref_reproject_point(const orc_pinhole* cam, const double T_f_w[12], const double pos[3], int cell_size,
                        int grid_n_cols, double px_out[2]) {
  // Reprojector::reprojectPoint is private; its three statements are frame->w2c(),
  // isInFrame(px.cast<int>(), 8) and the cell index (reprojector.cpp:208-213) -- executed
  // here through the reference's Frame and camera objects.
  vk::AbstractCamera* c = make_cam(cam);
  Config::nPyrLevels() = 1;
  Config::kltMaxLevel() = 0;
  cv::Mat img0(cam->height, cam->width, CV_8UC1, cv::Scalar(0));
  FramePtr frame(new Frame(c, img0, 0.0));
  frame->T_f_w_ = se3_from_Rt(T_f_w);
  Vector2d px(frame->w2c(Vector3d(pos[0], pos[1], pos[2])));
  px_out[0] = px[0]; px_out[1] = px[1];
  int k = -1;
  if (frame->cam_->isInFrame(px.cast<int>(), 8))
    k = static_cast<int>(px[1] / cell_size) * grid_n_cols + static_cast<int>(px[0] / cell_size);
  frame.reset();
  delete c;
  return k;
}

}  // extern "C"
