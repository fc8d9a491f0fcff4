// -*- C++ -*-
// oracle/shim/vikit/atan_camera.h -- TEST INFRASTRUCTURE ONLY.
// vk::ATANCamera restated from rpg_vikit (atan_camera.{h,cpp}), the FOV distortion model of PTAM:
// constructor (width, height, fx, fy, cx, cy, s) with NORMALISED intrinsics (fx_ = width*fx,
// cx_ = cx*width - 0.5, ...), rtrans_factor(r) = atan(r * 2 tan(s/2)) / (s r) for r >= 0.001,
// invrtrans(r) = tan(r s) / (2 tan(s/2)).  The arithmetic lives in orc_camera.h.
#pragma once
#include <vikit/abstract_camera.h>
#include <vikit/math_utils.h>

#include "orc_camera.h"
namespace vk {
class ATANCamera : public AbstractCamera {
  orc_pinhole c_;
 public:
  ATANCamera(double width, double height, double fx, double fy, double dx, double dy, double s)
      : AbstractCamera((int)width, (int)height) {
    orc_cam_init_atan(&c_, (int)width, (int)height, fx, fy, dx, dy, s);
  }
  // shim only: from an already constructed parameter block (the test driver hands the SAME block to
  // every implementation, so that fx_ = width*fx is not re-derived through a division)
  explicit ATANCamera(const orc_pinhole& c) : AbstractCamera(c.width, c.height), c_(c) {}
  virtual Vector3d cam2world(const double& x, const double& y) const {
    double f[3];
    orc_cam_cam2world(&c_, x, y, f);
    return Vector3d(f[0], f[1], f[2]);
  }
  virtual Vector3d cam2world(const Vector2d& px) const { return cam2world(px[0], px[1]); }
  virtual Vector2d world2cam(const Vector3d& xyz_c) const { return world2cam(project2d(xyz_c)); }
  virtual Vector2d world2cam(const Vector2d& uv) const {
    const double in[2] = {uv[0], uv[1]};
    double px[2];
    orc_cam_world2cam_uv(&c_, in, px);
    return Vector2d(px[0], px[1]);
  }
  const Vector2d focal_length() const { return Vector2d(c_.fx, c_.fy); }
  virtual double errorMultiplier2() const { return fabs(c_.fx); }
  virtual double errorMultiplier() const { return fabs(4.0 * c_.fx * c_.fy); }
};
}  // namespace vk
