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
#include <mutex>
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
//   3. vk::PinholeCamera with radial-tangential distortion: on either axis the odd part of the
//      projection is a cubic in r^2 with coefficients (f, f k1, f k2, f k3), the even part is the
//      tangential term -- four radii per axis.
// Every candidate is verified on independent probes to 1e-8 px; a camera none of them reproduces
// throws.  Recovered parameters agree with the constructor's to ~1e-13 relative; callers that want
// them bit-exact register the block they constructed the camera from (registerCamera).
// The registry is shared by the tracking and the mapping thread (both call cameraOf), so every access
// holds its mutex; an entry is trusted only while it still reproduces the camera behind the pointer (a
// camera freed and another allocated at the same address must not inherit stale intrinsics).
struct CameraRegistry {
  std::mutex mut;
  std::map<const vk::AbstractCamera*, svo_hip_camera> map;
};
inline CameraRegistry& cameraRegistry() {
  static CameraRegistry r;
  return r;
}
inline void registerCamera(const vk::AbstractCamera* cam, const svo_hip_camera& c) {
  CameraRegistry& reg = cameraRegistry();
  std::lock_guard<std::mutex> g(reg.mut);
  reg.map[cam] = c;
}

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
    if (!(std::fabs(px[0] - q[0]) <= 1e-8 && std::fabs(px[1] - q[1]) <= 1e-8)) return false;
  }
  return true;
}
}  // namespace detail

inline svo_hip_camera recoverCamera(const vk::AbstractCamera* cam);

inline svo_hip_camera cameraOf(const vk::AbstractCamera* cam) {
  CameraRegistry& registry = cameraRegistry();
  std::lock_guard<std::mutex> g(registry.mut);
  std::map<const vk::AbstractCamera*, svo_hip_camera>::const_iterator it = registry.map.find(cam);
  if (it != registry.map.end() && it->second.width == cam->width() && it->second.height == cam->height() &&
      detail::reproduces(cam, it->second))
    return it->second;
  return registry.map[cam] = recoverCamera(cam);
}

inline svo_hip_camera recoverCamera(const vk::AbstractCamera* cam) {
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
    if (detail::reproduces(cam, c)) return c;
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
    if (detail::reproduces(cam, a)) return a;
  }
  // (3) pinhole + radial-tangential.  On the x axis (y = 0) the model reads
  //        u(x) - cx = fx (x + k1 x^3 + k2 x^5 + k3 x^7) + fx p2 3 x^2,     v(x) - cy = fy p1 x^2
  //     so the odd part in x isolates (fx, fx k1, fx k2, fx k3) -- a 4 x 4 Vandermonde system in x^2
  //     from four radii -- and the even part p2; the y axis gives fy and p1 the same way.
  {
    static const double X[4] = {0.2, 0.4, 0.6, 0.8};
    long double ax[4], ay[4];
    for (int axis = 0; axis < 2; ++axis) {
      long double M[4][5];
      for (int k = 0; k < 4; ++k) {
        const double x = X[k];
        const Vector2d qp = cam->world2cam(axis == 0 ? Vector2d(x, 0.0) : Vector2d(0.0, x));
        const Vector2d qm = cam->world2cam(axis == 0 ? Vector2d(-x, 0.0) : Vector2d(0.0, -x));
        const long double odd = ((long double)qp[axis] - (long double)qm[axis]) / 2 / x;  // f (1 + k1 t + k2 t^2 + k3 t^3)
        const long double t = (long double)x * x;
        M[k][0] = 1; M[k][1] = t; M[k][2] = t * t; M[k][3] = t * t * t; M[k][4] = odd;
      }
      for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
          if (fabsl(M[r][col]) > fabsl(M[piv][col])) piv = r;
        for (int j = 0; j < 5; ++j) std::swap(M[col][j], M[piv][j]);
        for (int r = 0; r < 4; ++r) {
          if (r == col) continue;
          const long double k = M[r][col] / M[col][col];
          for (int j = col; j < 5; ++j) M[r][j] -= k * M[col][j];
        }
      }
      for (int i = 0; i < 4; ++i) (axis == 0 ? ax : ay)[i] = M[i][4] / M[i][i];
    }
    if (ax[0] != 0 && ay[0] != 0) {
      svo_hip_camera r = c;
      r.model = SVO_HIP_CAM_PINHOLE_RADTAN;
      r.fx = (double)ax[0]; r.fy = (double)ay[0];
      r.d[0] = (double)((ax[1] / ax[0] + ay[1] / ay[0]) / 2);
      r.d[1] = (double)((ax[2] / ax[0] + ay[2] / ay[0]) / 2);
      r.d[4] = (double)((ax[3] / ax[0] + ay[3] / ay[0]) / 2);
      const double x = 0.8;
      const Vector2d qxp = cam->world2cam(Vector2d(x, 0.0)), qxm = cam->world2cam(Vector2d(-x, 0.0));
      const Vector2d qyp = cam->world2cam(Vector2d(0.0, x)), qym = cam->world2cam(Vector2d(0.0, -x));
      // p1 (d2): v on the x axis, and the even part of v on the y axis (fy p1 3 y^2)
      const double p1a = ((qxp[1] + qxm[1]) / 2 - o[1]) / (r.fy * x * x);
      const double p1b = ((qyp[1] + qym[1]) / 2 - o[1]) / (3.0 * r.fy * x * x);
      // p2 (d3): the even part of u on the x axis (fx p2 3 x^2), and u on the y axis
      const double p2a = ((qxp[0] + qxm[0]) / 2 - o[0]) / (3.0 * r.fx * x * x);
      const double p2b = ((qyp[0] + qym[0]) / 2 - o[0]) / (r.fx * x * x);
      r.d[2] = 0.5 * (p1a + p1b);
      r.d[3] = 0.5 * (p2a + p2b);
      if (detail::reproduces(cam, r)) return r;
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
