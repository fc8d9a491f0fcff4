// track_math.h -- device math for the kernels that follow sparse alignment (matcher,
// feature alignment, pose optimizer, depth filter, point optimizer).
//
// The translation units that include this header are compiled with floating-point
// contraction OFF (#pragma clang fp contract(off) before the includes): the feature
// alignment and the affine warp are float pipelines whose results the reference truncates
// to u8 / compares against thresholds, so every mul/add rounds separately, exactly like the
// x86 build of the reference.  f64 geometry uses IEEE division and sqrt.
//
// Conventions: SE(3) as Sophus stores it (unit quaternion w,x,y,z + translation); poses
// cross the C ABI as 12 doubles [R row-major | t].
#pragma once
#ifndef SVO_HOST_MATH_TEST  // (see device_math.h)
#include <hip/hip_runtime.h>
#endif

#include "device_math.h"
#include "svo_hip.h"

namespace svo_dev {

struct Se3 {
  double q[4];
  double t[3];
};

// vk::AbstractCamera implementations of rpg_vikit, by model tag (include/svo_hip.h):
//   SVO_HIP_CAM_PINHOLE         vk::PinholeCamera, distortion_ == false
//   SVO_HIP_CAM_PINHOLE_RADTAN  vk::PinholeCamera with radial-tangential distortion d[0..4] = k1 k2 p1 p2 k3
//   SVO_HIP_CAM_ATAN            vk::ATANCamera (PTAM's FOV model): d = {s, 1/s, 2 tan(s/2), 1/(2 tan(s/2))}
struct Cam {
  double fx, fy, cx, cy;
  int width, height;
  int model;
  double d[5];
};

__device__ __forceinline__ double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void normalize3(double v[3]) {
  const double n = norm3(v);
  v[0] /= n; v[1] /= n; v[2] /= n;
}
__device__ __forceinline__ double norm2(const double v[2]) { return sqrt(v[0] * v[0] + v[1] * v[1]); }

__device__ __forceinline__ void se3_from_Rt(const double* __restrict__ T, Se3& s) {
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = T[k];
  quat_from_R(R, s.q);
  s.t[0] = T[9]; s.t[1] = T[10]; s.t[2] = T[11];
}
__device__ __forceinline__ void se3_to_Rt(const Se3& s, double* __restrict__ T) {
  double R[9];
  quat_to_R(s.q, R);
#pragma unroll
  for (int k = 0; k < 9; ++k) T[k] = R[k];
  T[9] = s.t[0]; T[10] = s.t[1]; T[11] = s.t[2];
}
// Sophus SE3::operator*: t += so3*other.t ; so3 *= other.so3, renormalised
__device__ __forceinline__ Se3 se3_compose(const Se3& a, const Se3& b) {
  Se3 r;
  double rt[3];
  quat_rot(a.q, b.t, rt);
  r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
  quat_mul(a.q, b.q, r.q);
  const double n = sqrt(r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3]);
  r.q[0] /= n; r.q[1] /= n; r.q[2] /= n; r.q[3] /= n;
  return r;
}
__device__ __forceinline__ Se3 se3_inverse(const Se3& a) {
  Se3 r;
  r.q[0] = a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = -a.q[3];
  const double nt[3] = {a.t[0] * -1., a.t[1] * -1., a.t[2] * -1.};
  quat_rot(r.q, nt, r.t);
  return r;
}
__device__ __forceinline__ void se3_apply(const Se3& a, const double v[3], double o[3]) {
  quat_rot(a.q, v, o);
  o[0] += a.t[0]; o[1] += a.t[1]; o[2] += a.t[2];
}
// Frame::pos() = T_f_w_.inverse().translation()
__device__ __forceinline__ void frame_pos(const Se3& T_f_w, double p[3]) {
  const Se3 inv = se3_inverse(T_f_w);
  p[0] = inv.t[0]; p[1] = inv.t[1]; p[2] = inv.t[2];
}

// ---- vk::PinholeCamera / vk::ATANCamera ---------------------------------------------
// cam2world of the distorted pinhole goes through cv::undistortPoints on CV_32FC2 points with a
// float camera matrix / distortion vector (vikit pinhole_camera.cpp): pixel, intrinsics and
// coefficients are rounded to float first, five fixed-point iterations run in double, the result is
// rounded to float again.
__device__ inline void cam2world(const Cam& c, double u, double v, double f[3]) {
  if (c.model == SVO_HIP_CAM_PINHOLE) {
    f[0] = (u - c.cx) / c.fx;
    f[1] = (v - c.cy) / c.fy;
    f[2] = 1.0;
  } else if (c.model == SVO_HIP_CAM_PINHOLE_RADTAN) {
    const double fx = (double)(float)c.fx, fy = (double)(float)c.fy, cx = (double)(float)c.cx, cy = (double)(float)c.cy;
    const double k0 = (double)(float)c.d[0], k1 = (double)(float)c.d[1], k2 = (double)(float)c.d[2];
    const double k3 = (double)(float)c.d[3], k4 = (double)(float)c.d[4];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = (double)(float)u, y = (double)(float)v;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = 1. / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
      const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
      const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    f[0] = (double)(float)x;
    f[1] = (double)(float)y;
    f[2] = 1.0;
  } else {
    const double fx_inv = 1.0 / c.fx, fy_inv = 1.0 / c.fy;
    const double dc[2] = {(u - c.cx) * fx_inv, (v - c.cy) * fy_inv};
    const double dist_r = sqrt(dc[0] * dc[0] + dc[1] * dc[1]);
    const double r = (c.d[0] == 0.0) ? dist_r : tan(dist_r * c.d[0]) * c.d[3];  // invrtrans
    const double d_factor = (dist_r > 0.01) ? r / dist_r : 1.0;
    f[0] = d_factor * dc[0];
    f[1] = d_factor * dc[1];
    f[2] = 1.0;
  }
  const double n = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  f[0] /= n; f[1] /= n; f[2] /= n;
}
__device__ inline void world2cam_uv(const Cam& c, const double uv[2], double px[2]) {
  if (c.model == SVO_HIP_CAM_PINHOLE) {
    px[0] = c.fx * uv[0] + c.cx;
    px[1] = c.fy * uv[1] + c.cy;
  } else if (c.model == SVO_HIP_CAM_PINHOLE_RADTAN) {
    const double x = uv[0], y = uv[1];
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + c.d[0] * r2 + c.d[1] * r4 + c.d[4] * r6;
    const double xd = x * cdist + c.d[2] * a1 + c.d[3] * a2;
    const double yd = y * cdist + c.d[2] * a3 + c.d[3] * a1;
    px[0] = xd * c.fx + c.cx;
    px[1] = yd * c.fy + c.cy;
  } else {
    const double r = sqrt(uv[0] * uv[0] + uv[1] * uv[1]);
    const double factor = (r < 0.001 || c.d[0] == 0.0) ? 1.0 : (c.d[1] * atan(r * c.d[2]) / r);  // rtrans_factor
    px[0] = c.cx + c.fx * (factor * uv[0]);
    px[1] = c.cy + c.fy * (factor * uv[1]);
  }
}
__device__ __forceinline__ void project2d(const double v[3], double uv[2]) {
  uv[0] = v[0] / v[2];
  uv[1] = v[1] / v[2];
}
__device__ __forceinline__ void world2cam(const Cam& c, const double xyz[3], double px[2]) {
  double uv[2];
  project2d(xyz, uv);
  world2cam_uv(c, uv, px);
}
inline Cam make_cam(const svo_hip_camera* c) {
  Cam k;
  k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy;
  k.width = c->width; k.height = c->height;
  k.model = c->model;
  for (int i = 0; i < 5; ++i) k.d[i] = c->d[i];
  return k;
}
inline bool cam_model_ok(const svo_hip_camera* c) {
  return c->model == SVO_HIP_CAM_PINHOLE || c->model == SVO_HIP_CAM_PINHOLE_RADTAN || c->model == SVO_HIP_CAM_ATAN;
}
__device__ __forceinline__ bool is_in_frame(const Cam& c, int x, int y, int boundary) {
  return x >= boundary && x < c.width - boundary && y >= boundary && y < c.height - boundary;
}
__device__ __forceinline__ bool is_in_frame_level(const Cam& c, int x, int y, int boundary, int level) {
  return x >= boundary && x < c.width / (1 << level) - boundary && y >= boundary &&
         y < c.height / (1 << level) - boundary;
}
// double -> int like a C cast on x86 (cvttsd2si): out-of-range / NaN give INT_MIN
__device__ __forceinline__ int cast_int(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
  return (int)v;
}

// ---- Eigen small inverses (Eigen/src/LU/Inverse.h), row-major ------------------------
template <typename T>
__device__ __forceinline__ T det2(const T m[4]) { return m[0] * m[3] - m[2] * m[1]; }
template <typename T>
__device__ __forceinline__ void inv2(const T m[4], T r[4]) {
  const T invdet = (T)1 / det2(m);
  r[0] = m[3] * invdet;
  r[2] = -m[2] * invdet;
  r[1] = -m[1] * invdet;
  r[3] = m[0] * invdet;
}
#define SVO_COF3(m, i1, i2, j1, j2) ((m)[(i1)*3 + (j1)] * (m)[(i2)*3 + (j2)] - (m)[(i1)*3 + (j2)] * (m)[(i2)*3 + (j1)])
__device__ __forceinline__ void inv3f(const float m[9], float r[9]) {
  // cofactor(i,j): rows (i+1)%3,(i+2)%3 ; cols (j+1)%3,(j+2)%3
  const float c00 = SVO_COF3(m, 1, 2, 1, 2), c10 = SVO_COF3(m, 2, 0, 1, 2), c20 = SVO_COF3(m, 0, 1, 1, 2);
  const float det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
  const float invdet = 1.0f / det;
  r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
  r[3] = SVO_COF3(m, 1, 2, 2, 0) * invdet;
  r[4] = SVO_COF3(m, 2, 0, 2, 0) * invdet;
  r[5] = SVO_COF3(m, 0, 1, 2, 0) * invdet;
  r[6] = SVO_COF3(m, 1, 2, 0, 1) * invdet;
  r[7] = SVO_COF3(m, 2, 0, 0, 1) * invdet;
  r[8] = SVO_COF3(m, 0, 1, 0, 1) * invdet;
}

// ---- Eigen::LDLT<Lower> with diagonal pivoting, register resident -------------------
// Eigen's unblocked LDLT (Eigen/src/Cholesky/LDLT.h) is left-looking: step k picks the largest
// remaining |diagonal|, swaps it to position k symmetrically, then forms column k of L from the
// columns before it; the trailing block is never updated, so it stays symmetric and a full
// row+column swap is the same permutation Eigen applies to the stored lower triangle.
// Everything is unrolled over compile-time indices (pivot positions become predicated
// swaps), so the N x N matrix lives in VGPRs: a dynamically indexed local array would sit in
// scratch memory and turn the few hundred dependent accesses of a solve into ~100 us.
template <int N>
struct LdltReg {
  double m[N * N];  // row-major; lower triangle + diagonal meaningful after factor()
  int tr[N];
  bool all_zero;

  __device__ __forceinline__ static void cswap(bool c, double& a, double& b) {
    const double ta = a, tb = b;
    a = c ? tb : ta;
    b = c ? ta : tb;
  }

  __device__ __forceinline__ void factor(const double* A) {
#pragma unroll
    for (int i = 0; i < N * N; ++i) m[i] = A[i];
    all_zero = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      // pivot: largest |diagonal| in the trailing block, first maximum wins
      int big = k;
      double best = fabs(m[k * N + k]);
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const double v = fabs(m[i * N + i]);
        const bool gt = v > best;
        best = gt ? v : best;
        big = gt ? i : big;
      }
      tr[k] = all_zero ? k : big;
#pragma unroll
      for (int b = k + 1; b < N; ++b) {
        const bool c = (!all_zero) && (big == b);
#pragma unroll
        for (int j = 0; j < N; ++j) cswap(c, m[k * N + j], m[b * N + j]);  // rows
#pragma unroll
        for (int i = 0; i < N; ++i) cswap(c, m[i * N + k], m[i * N + b]);  // columns
      }
      double temp[N];
      double acc = 0;
#pragma unroll
      for (int c = 0; c < k; ++c) {
        temp[c] = m[c * N + c] * m[k * N + c];
        acc += m[k * N + c] * temp[c];
      }
      const double akk_new = (k > 0) ? m[k * N + k] - acc : m[k * N + k];
      if (!all_zero) m[k * N + k] = akk_new;
#pragma unroll
      for (int r = k + 1; r < N; ++r) {
        double a2 = 0;
#pragma unroll
        for (int c = 0; c < k; ++c) a2 += m[r * N + c] * temp[c];
        if (k > 0 && !all_zero) m[r * N + k] -= a2;
      }
      const double akk = m[k * N + k];
      const bool pivot_is_valid = fabs(akk) > 0.0;
      if (k == 0 && !pivot_is_valid) all_zero = true;  // Eigen: identity transpositions, stop
#pragma unroll
      for (int r = k + 1; r < N; ++r)
        if (pivot_is_valid && !all_zero) m[r * N + k] /= akk;
    }
  }

  // LDLT::solve: P, L^-1, D^-1 (tolerance = DBL_MIN, Eigen >= 3.2.2), L^-T, P^T
  __device__ __forceinline__ void solve(const double* b, double* x) const {
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = b[i];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int j = k + 1; j < N; ++j) cswap(tr[k] == j, y[k], y[j]);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int c = 0; c < i; ++c) y[i] -= m[i * N + c] * y[c];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double d = m[i * N + i];
      y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0;
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i)
#pragma unroll
      for (int c = i + 1; c < N; ++c) y[i] -= m[c * N + i] * y[c];
#pragma unroll
    for (int k = N - 1; k >= 0; --k)
#pragma unroll
      for (int j = k + 1; j < N; ++j) cswap(tr[k] == j, y[k], y[j]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = y[i];
  }
};

template <int N>
__device__ __forceinline__ void ldlt_solve_pivoted(const double* A, const double* b, double* x) {
  LdltReg<N> f;
  f.factor(A);
  f.solve(b, x);
}

// inverse of a symmetric matrix through its pivoted LDLT: column j = solve(e_j).  (Eigen
// computes Frame::Cov_ with PartialPivLU, pose_optimizer.cpp:126; the two agree to rounding.)
template <int N>
__device__ __forceinline__ void inv_sym(const double* A, double* out) {
  LdltReg<N> f;
  f.factor(A);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double e[N], x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = (i == j) ? 1.0 : 0.0;
    f.solve(e, x);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i * N + j] = x[i];
  }
}

// Sophus SE3::exp with libm-grade sin/cos (pose optimizer: increments are applied once per
// iteration by one lane, accuracy matters more than speed here)
// theta of an increment (the argument of the trigonometric functions below)
__device__ __forceinline__ double se3_exp_theta(const double xi[6]) {
  return sqrt(xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5]);
}

// The four trigonometric values (sin/cos of theta/2 and of theta) are passed in: the caller
// evaluates them with ONE sincos over two lanes instead of four serial libm calls in one lane.
__device__ inline Se3 se3_exp_full(const double xi[6], double sin_half, double cos_half, double sin_theta,
                                   double cos_theta) {
  Se3 r;
  const double* ups = xi;
  const double* om = xi + 3;
  const double theta = se3_exp_theta(xi);
  double imag_factor;
  const double real_factor = cos_half;
  if (theta < 1e-10) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    imag_factor = sin_half / theta;
  }
  r.q[0] = real_factor;
  r.q[1] = imag_factor * om[0];
  r.q[2] = imag_factor * om[1];
  r.q[3] = imag_factor * om[2];
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double V[9];
  if (theta < 1e-10) {
    quat_to_R(r.q, V);
  } else {
    double O2[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
        O2[i * 3 + j] = s;
      }
    const double theta_sq = theta * theta;
    const double c1 = (1 - cos_theta) / (theta_sq);
    const double c2 = (theta - sin_theta) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
  return r;
}

// Frame::jacobian_xyz2uv (svo/include/svo/frame.h:116-138), 2x6 row-major
__device__ __forceinline__ void frame_jacobian_xyz2uv(const double xyz[3], double J[12]) {
  const double x = xyz[0];
  const double y = xyz[1];
  const double z_inv = 1. / xyz[2];
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv;
  J[1] = 0.0;
  J[2] = x * z_inv_2;
  J[3] = y * J[2];
  J[4] = -(1.0 + x * J[2]);
  J[5] = y * z_inv;
  J[6] = 0.0;
  J[7] = -z_inv;
  J[8] = y * z_inv_2;
  J[9] = 1.0 + y * J[8];
  J[10] = -J[3];
  J[11] = -x * z_inv;
}

}  // namespace svo_dev
