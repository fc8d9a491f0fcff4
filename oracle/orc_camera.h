/*
 * orc_camera.h -- TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 * The vk::AbstractCamera implementations of rpg_vikit (vikit_common: pinhole_camera.cpp,
 * atan_camera.cpp) restated from the published sources; UNPINNED (no copy of rpg_vikit or of
 * OpenCV exists in this image).  Shared by the C oracle and by the camera shims the reference's
 * own translation units are compiled against (oracle/shim/vikit/{pinhole,atan}_camera.h), so both
 * sides execute the same camera arithmetic.
 *
 *  ORC_CAM_PINHOLE         vk::PinholeCamera, distortion_ == false
 *  ORC_CAM_PINHOLE_RADTAN  vk::PinholeCamera with d0..d4 = k1 k2 p1 p2 k3: world2cam applies the
 *                          radial-tangential model in double; cam2world calls cv::undistortPoints on a
 *                          CV_32FC2 point with FLOAT camera matrix and distortion vector (cvK_, cvD_
 *                          are cv::Mat_<float>): OpenCV 2.4's cvUndistortPoints, five fixed-point
 *                          iterations in double, result stored as float
 *  ORC_CAM_ATAN            vk::ATANCamera (FOV model of PTAM); fx, fy, cx, cy are the constructor's
 *                          fx_ = width*fx, fy_ = height*fy, cx_ = cx*width - 0.5, cy_ = cy*height - 0.5;
 *                          d = {s, 1/s, 2 tan(s/2), 1/(2 tan(s/2))}
 */
#ifndef ORC_CAMERA_H_
#define ORC_CAMERA_H_

#include <math.h>

#define ORC_CAM_PINHOLE 0
#define ORC_CAM_PINHOLE_RADTAN 1
#define ORC_CAM_ATAN 2

typedef struct {
  double fx, fy, cx, cy;
  int width, height;
  int model;
  int pad_;
  double d[5];
} orc_pinhole; /* historical name: any of the three models */

static inline void orc_cam_init_pinhole(orc_pinhole* c, int width, int height, double fx, double fy, double cx, double cy,
                                        double d0, double d1, double d2, double d3, double d4) {
  const int distortion = fabs(d0) > 0.0000001; /* vikit: distortion_(fabs(d0) > 0.0000001) */
  c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy;
  c->width = width; c->height = height;
  c->model = distortion ? ORC_CAM_PINHOLE_RADTAN : ORC_CAM_PINHOLE;
  c->pad_ = 0;
  c->d[0] = distortion ? d0 : 0.0; c->d[1] = distortion ? d1 : 0.0; c->d[2] = distortion ? d2 : 0.0;
  c->d[3] = distortion ? d3 : 0.0; c->d[4] = distortion ? d4 : 0.0;
}

static inline void orc_cam_init_atan(orc_pinhole* c, int width, int height, double fx, double fy, double cx, double cy,
                                     double s) {
  c->fx = (double)width * fx;
  c->fy = (double)height * fy;
  c->cx = cx * (double)width - 0.5;
  c->cy = cy * (double)height - 0.5;
  c->width = width; c->height = height;
  c->model = ORC_CAM_ATAN;
  c->pad_ = 0;
  for (int i = 0; i < 5; ++i) c->d[i] = 0.0;
  if (s != 0.0) {
    const double tans = 2.0 * tan(s / 2.0);
    c->d[0] = s;
    c->d[1] = 1.0 / s;
    c->d[2] = tans;
    c->d[3] = 1.0 / tans;
  }
}

/* round to float and back.  Through a volatile: g++ 11 -O3 drops a plain (double)(float)x pair when it
 * SLP-vectorises the two coordinates of the inlined call (seen in oracle/_ref), which would make the
 * reference-side camera disagree with every other implementation of the same arithmetic. */
static inline double orc_f32(double x) {
  volatile float f = (float)x;
  return (double)f;
}

/* world2cam(const Vector2d& uv) */
static inline void orc_cam_world2cam_uv(const orc_pinhole* c, const double uv[2], double px[2]) {
  if (c->model == ORC_CAM_PINHOLE) {
    px[0] = c->fx * uv[0] + c->cx;
    px[1] = c->fy * uv[1] + c->cy;
  } else if (c->model == ORC_CAM_PINHOLE_RADTAN) {
    double x, y, r2, r4, r6, a1, a2, a3, cdist, xd, yd;
    x = uv[0];
    y = uv[1];
    r2 = x * x + y * y;
    r4 = r2 * r2;
    r6 = r4 * r2;
    a1 = 2 * x * y;
    a2 = r2 + 2 * x * x;
    a3 = r2 + 2 * y * y;
    cdist = 1 + c->d[0] * r2 + c->d[1] * r4 + c->d[4] * r6;
    xd = x * cdist + c->d[2] * a1 + c->d[3] * a2;
    yd = y * cdist + c->d[2] * a3 + c->d[3] * a1;
    px[0] = xd * c->fx + c->cx;
    px[1] = yd * c->fy + c->cy;
  } else {
    const double r = sqrt(uv[0] * uv[0] + uv[1] * uv[1]);
    /* rtrans_factor */
    const double factor = (r < 0.001 || c->d[0] == 0.0) ? 1.0 : (c->d[1] * atan(r * c->d[2]) / r);
    px[0] = c->cx + c->fx * (factor * uv[0]);
    px[1] = c->cy + c->fy * (factor * uv[1]);
  }
}

/* cam2world(u, v): unit bearing */
static inline void orc_cam_cam2world(const orc_pinhole* c, double u, double v, double f[3]) {
  if (c->model == ORC_CAM_PINHOLE) {
    f[0] = (u - c->cx) / c->fx;
    f[1] = (v - c->cy) / c->fy;
    f[2] = 1.0;
  } else if (c->model == ORC_CAM_PINHOLE_RADTAN) {
    const double fx = orc_f32(c->fx), fy = orc_f32(c->fy), cx = orc_f32(c->cx), cy = orc_f32(c->cy);
    const double k0 = orc_f32(c->d[0]), k1 = orc_f32(c->d[1]), k2 = orc_f32(c->d[2]);
    const double k3 = orc_f32(c->d[3]), k4 = orc_f32(c->d[4]);
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = orc_f32(u), y = orc_f32(v), x0, y0;
    x0 = x = (x - cx) * ifx;
    y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = 1. / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
      const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
      const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    f[0] = orc_f32(x);
    f[1] = orc_f32(y);
    f[2] = 1.0;
  } else {
    const double fx_inv = 1.0 / c->fx, fy_inv = 1.0 / c->fy;
    const double dc0 = (u - c->cx) * fx_inv, dc1 = (v - c->cy) * fy_inv;
    const double dist_r = sqrt(dc0 * dc0 + dc1 * dc1);
    const double r = (c->d[0] == 0.0) ? dist_r : tan(dist_r * c->d[0]) * c->d[3]; /* invrtrans */
    const double d_factor = (dist_r > 0.01) ? r / dist_r : 1.0;
    f[0] = d_factor * dc0;
    f[1] = d_factor * dc1;
    f[2] = 1.0;
  }
  {
    const double n = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    f[0] /= n; f[1] /= n; f[2] /= n;
  }
}

#endif /* ORC_CAMERA_H_ */
