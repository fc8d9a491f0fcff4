// marshal.h -- helpers shared by the drop-in bodies: reference objects (svo::Frame, Feature,
// Point, Sophus::SE3, vk::AbstractCamera) <-> the plain arrays of include/svo_hip.h.
// Compiled against the reference's own headers (svo/include/svo/*.h); only operator[],
// operator()(r,c), rotation_matrix()/translation() and the SE3(R,t) constructor are used,
// so any Eigen/Sophus the reference is built with will do.
#ifndef SVO_HIP_DROPIN_MARSHAL_H_
#define SVO_HIP_DROPIN_MARSHAL_H_

#include <cmath>
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

// The kernels implement vk::PinholeCamera without distortion.  The intrinsics
// are read through the abstract interface and the model is verified to be that one.
inline svo_hip_camera cameraOf(const vk::AbstractCamera* cam) {
  svo_hip_camera c;
  const Vector2d o = cam->world2cam(Vector2d(0.0, 0.0));
  const Vector2d x = cam->world2cam(Vector2d(1.0, 0.0));
  const Vector2d y = cam->world2cam(Vector2d(0.0, 1.0));
  c.cx = o[0]; c.cy = o[1];
  c.fx = x[0] - o[0]; c.fy = y[1] - o[1];
  c.width = cam->width(); c.height = cam->height();
  const Vector2d probe = cam->world2cam(Vector2d(0.31, -0.17));
  if (std::fabs(probe[0] - (c.fx * 0.31 + c.cx)) > 1e-9 || std::fabs(probe[1] - (c.fy * -0.17 + c.cy)) > 1e-9 ||
      std::fabs(x[1] - o[1]) > 1e-9 || std::fabs(y[0] - o[0]) > 1e-9)
    throw svo_hip::Error("svo_hip drop-in: only the undistorted pinhole camera model is implemented on the device");
  return c;
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
