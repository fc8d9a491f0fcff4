// -*- C++ -*-
// oracle/shim/vikit/pinhole_camera.h -- TEST INFRASTRUCTURE ONLY.
// vk::PinholeCamera with zero distortion, restated from rpg_vikit (pinhole_camera.cpp):
// cam2world returns the unit-norm bearing of ((x-cx)/fx, (y-cy)/fy, 1); world2cam(xyz) =
// world2cam(project2d(xyz)); world2cam(uv) = (fx*u+cx, fy*v+cy); errorMultiplier2 = |fx|.
#pragma once
#include <vikit/abstract_camera.h>
#include <vikit/math_utils.h>
namespace vk {
class PinholeCamera : public AbstractCamera {
  double fx_, fy_, cx_, cy_;
 public:
  PinholeCamera(double width, double height, double fx, double fy, double cx, double cy)
      : AbstractCamera((int)width, (int)height), fx_(fx), fy_(fy), cx_(cx), cy_(cy) {}
  virtual Vector3d cam2world(const double& u, const double& v) const {
    Vector3d xyz;
    xyz[0] = (u - cx_) / fx_;
    xyz[1] = (v - cy_) / fy_;
    xyz[2] = 1.0;
    return xyz.normalized();
  }
  virtual Vector3d cam2world(const Vector2d& px) const { return cam2world(px[0], px[1]); }
  virtual Vector2d world2cam(const Vector3d& xyz_c) const { return world2cam(project2d(xyz_c)); }
  virtual Vector2d world2cam(const Vector2d& uv) const {
    Vector2d px;
    px[0] = fx_ * uv[0] + cx_;
    px[1] = fy_ * uv[1] + cy_;
    return px;
  }
  virtual double errorMultiplier2() const { return fabs(fx_); }
  virtual double errorMultiplier() const { return fabs(4.0 * fx_ * fy_); }
};
}  // namespace vk
