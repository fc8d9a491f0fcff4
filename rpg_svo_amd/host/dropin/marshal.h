// marshal.h -- helpers shared by the drop-in bodies: reference objects (svo::Frame, Feature,
// Point, Sophus::SE3, vk::AbstractCamera) <-> the plain arrays of include/svo_hip.h.
// Compiled against the reference's own headers (svo/include/svo/*.h); only operator[],
// operator()(r,c), rotation_matrix()/translation() and the SE3(R,t) constructor are used,
// so any Eigen/Sophus the reference is built with will do.
#ifndef SVO_HIP_DROPIN_MARSHAL_H_
#define SVO_HIP_DROPIN_MARSHAL_H_

#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/global.h>
#include <svo/point.h>
#include <vikit/abstract_camera.h>

#include "svo_hip_device.h"

namespace svo {
namespace hip_dropin {

inline void poseToRt(const SE3& T, double out[12]) {
  const Matrix3d R = T.rotation_matrix();
  const Vector3d t = T.translation();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[i * 3 + j] = R(i, j);
  out[9] = t[0]; out[10] = t[1]; out[11] = t[2];
}

inline SE3 poseFromRt(const double in[12]) {
  Matrix3d R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = in[i * 3 + j];
  return SE3(R, Vector3d(in[9], in[10], in[11]));
}

// svo_hip_camera of a vk::AbstractCamera, recovered THROUGH THE ABSTRACT INTERFACE ONLY (world2cam on
// probe points), so that it works with whatever accessors the installed rpg_vikit has:
//   1. undistorted pinhole: world2cam(uv) is affine in uv;
//   2. vk::ATANCamera: radially symmetric about the principal point; the focal lengths follow from a
//      probe inside the model's r < 0.001 linear zone, s from one radius by bisection
//      (factor(r) = atan(r * 2 tan(s/2)) / (s r) is monotone in s);
//   3. vk::PinholeCamera with radial-tangential distortion: px - c is LINEAR in
//      (f, f k1, f k2, f k3, f p1, f p2) for known uv -- solved from eight probes per axis.
// Every candidate is verified on independent probes to 1e-9 px; a camera none of them reproduces
// throws.  Recovered parameters agree with the constructor's to ~1e-13 relative; callers that want
// them bit-exact register the block they constructed the camera from (registerCamera).
inline std::map<const vk::AbstractCamera*, svo_hip_camera>& cameraRegistry() {
  static std::map<const vk::AbstractCamera*, svo_hip_camera> r;
  return r;
}
inline void registerCamera(const vk::AbstractCamera* cam, const svo_hip_camera& c) { cameraRegistry()[cam] = c; }

namespace detail {
inline void modelWorld2cam(const svo_hip_camera& c, double x, double y, double px[2]) {
  if (c.model == SVO_HIP_CAM_PINHOLE_RADTAN) {
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + c.d[0] * r2 + c.d[1] * r4 + c.d[4] * r6;
    px[0] = (x * cdist + c.d[2] * a1 + c.d[3] * a2) * c.fx + c.cx;
    px[1] = (y * cdist + c.d[2] * a3 + c.d[3] * a1) * c.fy + c.cy;
  } else if (c.model == SVO_HIP_CAM_ATAN) {
    const double r = std::sqrt(x * x + y * y);
    const double factor = (r < 0.001 || c.d[0] == 0.0) ? 1.0 : (c.d[1] * std::atan(r * c.d[2]) / r);
    px[0] = c.cx + c.fx * (factor * x);
    px[1] = c.cy + c.fy * (factor * y);
  } else {
    px[0] = c.fx * x + c.cx;
    px[1] = c.fy * y + c.cy;
  }
}
inline bool reproduces(const vk::AbstractCamera* cam, const svo_hip_camera& c) {
  static const double probes[6][2] = {{0.31, -0.17}, {-0.45, 0.22}, {0.05, 0.6}, {-0.7, -0.4}, {0.9, 0.1}, {0.0004, -0.0003}};
  for (int i = 0; i < 6; ++i) {
    double px[2];
    modelWorld2cam(c, probes[i][0], probes[i][1], px);
    const Vector2d q = cam->world2cam(Vector2d(probes[i][0], probes[i][1]));
    if (!(std::fabs(px[0] - q[0]) <= 1e-9 && std::fabs(px[1] - q[1]) <= 1e-9)) return false;
  }
  return true;
}
// least squares of an 8 x 6 system by normal equations + Gaussian elimination with pivoting
inline bool solve6(const double A[8][6], const double b[8], double x[6]) {
  double N[6][7];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) {
      N[i][j] = 0;
      for (int k = 0; k < 8; ++k) N[i][j] += A[k][i] * A[k][j];
    }
    N[i][6] = 0;
    for (int k = 0; k < 8; ++k) N[i][6] += A[k][i] * b[k];
  }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (std::fabs(N[r][c]) > std::fabs(N[p][c])) p = r;
    if (std::fabs(N[p][c]) < 1e-300) return false;
    for (int j = 0; j < 7; ++j) std::swap(N[c][j], N[p][j]);
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double k = N[r][c] / N[c][c];
      for (int j = c; j < 7; ++j) N[r][j] -= k * N[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) x[i] = N[i][6] / N[i][i];
  return true;
}
}  // namespace detail

inline svo_hip_camera cameraOf(const vk::AbstractCamera* cam) {
  std::map<const vk::AbstractCamera*, svo_hip_camera>& reg = cameraRegistry();
  std::map<const vk::AbstractCamera*, svo_hip_camera>::const_iterator it = reg.find(cam);
  if (it != reg.end()) return it->second;
  svo_hip_camera c;
  std::memset(&c, 0, sizeof(c));
  c.width = cam->width(); c.height = cam->height();
  const Vector2d o = cam->world2cam(Vector2d(0.0, 0.0));
  c.cx = o[0]; c.cy = o[1];
  // (1) undistorted pinhole
  {
    const Vector2d x = cam->world2cam(Vector2d(1.0, 0.0));
    const Vector2d y = cam->world2cam(Vector2d(0.0, 1.0));
    c.model = SVO_HIP_CAM_PINHOLE;
    c.fx = x[0] - o[0]; c.fy = y[1] - o[1];
    if (detail::reproduces(cam, c)) return reg[cam] = c;
  }
  // (2) ATAN: inside r < 0.001 the model is exactly linear
  {
    const double e = 0.0009765625 * 0.5;  // 2^-11
    c.fx = (cam->world2cam(Vector2d(e, 0.0))[0] - o[0]) / e;
    c.fy = (cam->world2cam(Vector2d(0.0, e))[1] - o[1]) / e;
    const double r = 1.0;  // (r = 0.5 would be useless: atan(2 r tan(s/2)) / (s r) == 1 for every s there)
    const double target = (cam->world2cam(Vector2d(r, 0.0))[0] - o[0]) / (c.fx * r);  // rtrans_factor(r)
    double lo = 1e-6, hi = 3.0;  // s in (0, pi)
    for (int i = 0; i < 200; ++i) {
      const double s = 0.5 * (lo + hi);
      const double fac = std::atan(r * 2.0 * std::tan(s / 2.0)) / (s * r);
      // for r > 0.5 the factor decreases monotonically with s
      if (fac > target) lo = s; else hi = s;
    }
    const double s = 0.5 * (lo + hi);
    svo_hip_camera a = c;
    a.model = SVO_HIP_CAM_ATAN;
    const double tans = 2.0 * std::tan(s / 2.0);
    a.d[0] = s; a.d[1] = 1.0 / s; a.d[2] = tans; a.d[3] = 1.0 / tans; a.d[4] = 0.0;
    if (detail::reproduces(cam, a)) return reg[cam] = a;
  }
  // (3) pinhole + radial-tangential: linear in (f, f k1, f k2, f k3, f p1, f p2) per axis
  {
    static const double P[8][2] = {{0.3, 0.1}, {-0.2, 0.35}, {0.5, -0.3}, {-0.45, -0.25}, {0.15, 0.55}, {0.6, 0.2}, {-0.6, 0.05}, {0.1, -0.5}};
    double Ax[8][6], Ay[8][6], bx[8], by[8];
    for (int k = 0; k < 8; ++k) {
      const double x = P[k][0], y = P[k][1];
      const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
      const Vector2d q = cam->world2cam(Vector2d(x, y));
      // x-axis: f x + (f k1) x r2 + (f k2) x r4 + (f k3) x r6 + (f p1) 2xy + (f p2)(r2 + 2x^2)
      Ax[k][0] = x; Ax[k][1] = x * r2; Ax[k][2] = x * r4; Ax[k][3] = x * r6; Ax[k][4] = 2 * x * y; Ax[k][5] = r2 + 2 * x * x;
      // y-axis: f y + (f k1) y r2 + (f k2) y r4 + (f k3) y r6 + (f p1)(r2 + 2y^2) + (f p2) 2xy
      Ay[k][0] = y; Ay[k][1] = y * r2; Ay[k][2] = y * r4; Ay[k][3] = y * r6; Ay[k][4] = r2 + 2 * y * y; Ay[k][5] = 2 * x * y;
      bx[k] = q[0] - o[0];
      by[k] = q[1] - o[1];
    }
    double sx[6], sy[6];
    if (detail::solve6(Ax, bx, sx) && detail::solve6(Ay, by, sy) && sx[0] != 0.0 && sy[0] != 0.0) {
      svo_hip_camera r = c;
      r.model = SVO_HIP_CAM_PINHOLE_RADTAN;
      r.fx = sx[0]; r.fy = sy[0];
      // the coefficients are shared by both axes: average the two estimates
      r.d[0] = 0.5 * (sx[1] / sx[0] + sy[1] / sy[0]);
      r.d[1] = 0.5 * (sx[2] / sx[0] + sy[2] / sy[0]);
      r.d[4] = 0.5 * (sx[3] / sx[0] + sy[3] / sy[0]);
      r.d[2] = 0.5 * (sx[4] / sx[0] + sy[4] / sy[0]);
      r.d[3] = 0.5 * (sx[5] / sx[0] + sy[5] / sy[0]);
      if (detail::reproduces(cam, r)) return reg[cam] = r;
    }
  }
  throw svo_hip::Error("svo_hip drop-in: the camera is none of the vikit models the device implements "
                       "(pinhole, pinhole + radial-tangential distortion, ATAN)");
}

// Device residency of the frames one call refers to + the frame table the kernels index.
class FrameTable {
 public:
  FrameTable(svo_hip::Device& dev, int lane) : dev_(dev), lane_(lane) {}
  int indexOf(const Frame* f) {
    std::map<const Frame*, int>::iterator it = index_.find(f);
    if (it != index_.end()) return it->second;
    const cv::Mat& img = f->img_pyr_[0];
    const int slot = dev_.slotOf(f->id_, img.data, (int)img.step.p[0], lane_);
    const int idx = (int)frames_.size();
    frames_.push_back(f);
    slots_.push_back(slot);
    index_[f] = idx;
    return idx;
  }
  int slot(int idx) const { return slots_[idx]; }
  int size() const { return (int)frames_.size(); }
  // writes slot[] and T_f_w[] into the arena (inputs) and fills the C struct
  void emit(svo_hip::Arena& a, svo_hip_frames* out) const {
    int32_t* d_slot; double* d_T;
    int32_t* h_slot = a.alloc<int32_t>(frames_.size(), &d_slot);
    double* h_T = a.alloc<double>(frames_.size() * 12, &d_T);
    for (size_t i = 0; i < frames_.size(); ++i) {
      h_slot[i] = slots_[i];
      poseToRt(frames_[i]->T_f_w_, h_T + 12 * i);
    }
    out->n_frames = (int32_t)frames_.size();
    out->reserved = 0;
    out->d_slot = d_slot;
    out->d_T_f_w = d_T;
  }

 private:
  svo_hip::Device& dev_;
  int lane_;
  std::vector<const Frame*> frames_;
  std::vector<int> slots_;
  std::map<const Frame*, int> index_;
};

// SoA writer for svo::Feature records (svo_hip_features)
struct FeatureColumns {
  int32_t *frame, *level; uint8_t* type; double *px, *f, *grad;
  svo_hip_features dev;
  void alloc(svo_hip::Arena& a, size_t n) {
    int32_t *d_frame, *d_level; uint8_t* d_type; double *d_px, *d_f, *d_grad;
    frame = a.alloc<int32_t>(n, &d_frame);
    level = a.alloc<int32_t>(n, &d_level);
    type = a.alloc<uint8_t>(n, &d_type);
    px = a.alloc<double>(2 * n, &d_px);
    f = a.alloc<double>(3 * n, &d_f);
    grad = a.alloc<double>(2 * n, &d_grad);
    dev.d_frame = d_frame; dev.d_level = d_level; dev.d_type = d_type; dev.d_px = d_px; dev.d_f = d_f; dev.d_grad = d_grad;
  }
  void set(size_t i, int frame_idx, const Feature* ftr) {
    frame[i] = frame_idx;
    level[i] = ftr->level;
    type[i] = ftr->type == Feature::EDGELET ? SVO_HIP_FTR_EDGELET : SVO_HIP_FTR_CORNER;
    px[2 * i] = ftr->px[0]; px[2 * i + 1] = ftr->px[1];
    f[3 * i] = ftr->f[0]; f[3 * i + 1] = ftr->f[1]; f[3 * i + 2] = ftr->f[2];
    grad[2 * i] = ftr->grad[0]; grad[2 * i + 1] = ftr->grad[1];
  }
};

// the device context for frames of this geometry (created and sized on first use)
inline svo_hip::Device& ensureDevice(const Frame& f) {
  return svo_hip::Device::forGeometry(f.img_pyr_[0].cols, f.img_pyr_[0].rows, (int)f.img_pyr_.size());
}

}  // namespace hip_dropin
}  // namespace svo
#endif
