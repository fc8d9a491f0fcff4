// -*- C++ -*-
// oracle/shim/vikit/pinhole_camera.h -- TEST INFRASTRUCTURE ONLY.
// vk::PinholeCamera restated from rpg_vikit (pinhole_camera.{h,cpp}): constructor
// (width, height, fx, fy, cx, cy, d0..d4), distortion_ = fabs(d0) > 0.0000001; cam2world returns the
// unit-norm bearing (through cv::undistortPoints when distorted); world2cam(xyz) =
// world2cam(project2d(xyz)); errorMultiplier2 = |fx|.  The arithmetic lives in orc_camera.h.
#pragma once
#include <vikit/abstract_camera.h>
#include <vikit/math_utils.h>

#include "orc_camera.h"
namespace vk {
class PinholeCamera : public AbstractCamera {
  orc_pinhole c_;
 public:
  PinholeCamera(double width, double height, double fx, double fy, double cx, double cy, double d0 = 0.0, double d1 = 0.0,
                double d2 = 0.0, double d3 = 0.0, double d4 = 0.0)
      : AbstractCamera((int)width, (int)height) {
    orc_cam_init_pinhole(&c_, (int)width, (int)height, fx, fy, cx, cy, d0, d1, d2, d3, d4);
  }
  virtual Vector3d cam2world(const double& u, const double& v) const {
    double f[3];
    orc_cam_cam2world(&c_, u, v, f);
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
  inline double fx() const { return c_.fx; }
  inline double fy() const { return c_.fy; }
  inline double cx() const { return c_.cx; }
  inline double cy() const { return c_.cy; }
  inline double d0() const { return c_.d[0]; }
  inline double d1() const { return c_.d[1]; }
  inline double d2() const { return c_.d[2]; }
  inline double d3() const { return c_.d[3]; }
};
}  // namespace vk
