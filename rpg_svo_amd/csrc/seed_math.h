// seed_math.h -- the depth filter's closed-form pieces: depthFromTriangulation (matcher.cpp:109-122), the f32 Bayesian
// update DepthFilter::updateSeed (depth_filter.cpp:309-332) with boost's normal pdf, DepthFilter::computeTau (:334-350).
// Device functions of depth_filter.hip's kernels; also compiled for the CPU by the test suite (SVO_HOST_MATH_TEST, see
// device_math.h) and checked against the oracle there.
#pragma once
#include "track_math.h"

namespace svo_track {
using namespace svo_dev;

constexpr double SVO_PI = 3.14159265;       // svo/include/svo/global.h:78

// depthFromTriangulation, matcher.cpp:109-122
__device__ __forceinline__ bool depth_from_triangulation(const Se3& T_search_ref, const double f_ref[3],
                                                         const double f_cur[3], double* depth) {
  double R[9];
  quat_to_R(T_search_ref.q, R);
  double A[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    A[i][0] = R[i * 3] * f_ref[0] + R[i * 3 + 1] * f_ref[1] + R[i * 3 + 2] * f_ref[2];
    A[i][1] = f_cur[i];
  }
  double AtA[4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) AtA[i * 2 + j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j];
  if (det2<double>(AtA) < 0.000001) return false;
  double inv[4];
  inv2<double>(AtA, inv);
  const double i0 = -inv[0], i1 = -inv[1];
  double m[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) m[k] = i0 * A[k][0] + i1 * A[k][1];
  const double d0 = m[0] * T_search_ref.t[0] + m[1] * T_search_ref.t[1] + m[2] * T_search_ref.t[2];
  *depth = fabs(d0);
  return true;
}

// boost::math::pdf(normal_distribution<float>(mean, sd), x)
__device__ __forceinline__ float normal_pdff(float x, float mean, float sd) {
  float exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sd * sd;
  // host libm's expf is correctly rounded in all but ~0.3 % of its arguments; the f64 exp rounded to
  // float is too (ocml's expf is not): the seed state then matches the reference's bit for bit except
  // for those rare arguments, instead of differing in the last bit every few updates
  float result = (float)exp((double)exponent);
  result /= sd * sqrtf(2 * 3.14159265358979323846f);
  return result;
}

// DepthFilter::updateSeed, depth_filter.cpp:309-332 (all float, mixed with double literals
// exactly as written in the reference)
__device__ __forceinline__ void update_seed(const float x, const float tau2, float& a_, float& b_, float& mu_,
                                            const float z_range, float& sigma2_) {
  const float norm_scale = sqrtf(sigma2_ + tau2);
  if (isnan(norm_scale)) return;
  const float s2 = (float)(1. / (1. / (double)sigma2_ + 1. / (double)tau2));
  const float m = s2 * (mu_ / sigma2_ + x / tau2);
  float C1 = a_ / (a_ + b_) * normal_pdff(x, mu_, norm_scale);
  float C2 = (float)((double)(b_ / (a_ + b_)) * 1. / (double)z_range);
  const float normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  const float f = (float)((double)C1 * ((double)a_ + 1.) / ((double)(a_ + b_) + 1.) +
                          (double)(C2 * a_) / ((double)(a_ + b_) + 1.));
  const float e = (float)((double)C1 * ((double)a_ + 1.) * ((double)a_ + 2.) /
                              (((double)(a_ + b_) + 1.) * ((double)(a_ + b_) + 2.)) +
                          (double)(C2 * a_ * (a_ + 1.0f) / ((a_ + b_ + 1.0f) * (a_ + b_ + 2.0f))));
  const float mu_new = C1 * m + C2 * mu_;
  sigma2_ = C1 * (s2 + m * m) + C2 * (sigma2_ + mu_ * mu_) - mu_new * mu_new;
  mu_ = mu_new;
  a_ = (e - f) / (f - e / f);
  b_ = a_ * (1.0f - f) / f;
}

// DepthFilter::computeTau, depth_filter.cpp:334-350
__device__ __forceinline__ double compute_tau(const Se3& T_ref_cur, const double f[3], const double z,
                                              const double px_error_angle) {
  const double t[3] = {T_ref_cur.t[0], T_ref_cur.t[1], T_ref_cur.t[2]};
  const double av[3] = {f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2]};
  const double t_norm = norm3(t);
  const double a_norm = norm3(av);
  const double alpha = acos(dot3(f, t) / t_norm);
  const double mt[3] = {-t[0], -t[1], -t[2]};
  const double beta = acos(dot3(av, mt) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = SVO_PI - alpha - beta_plus;
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return (z_plus - z);
}

// computeTau without its two acos and two sin -- what seed_finish_kernel runs (233 against 258 us per 3.3 M seeds,
// profiles/r05a_queue_drain.txt; the acos / sin form above stays as the statement of the reference's formula, and
// tests/test_device_math_host.py compares the two).  With c_a = cos(alpha), c_b = cos(beta) -- the two normalised dot
// products the reference hands to acos -- and s = sqrt(1 - c^2) (both angles lie in [0, pi]),
//   sin(beta_plus)  = s_b cos(e) + c_b sin(e)
//   sin(gamma_plus) = sin(SVO_PI - alpha - beta - e) = sin(alpha + beta + e'),  e' = e + (pi - SVO_PI)
//                   = (s_a c_b + c_a s_b) cos(e') + (c_a c_b - s_a s_b) sin(e')
// where e = px_error_angle is the same for every seed of a launch: its sines and cosines come with the arguments.  The
// same function of the same inputs, evaluated differently: tau agrees with the acos / sin form to ~1e-13 relative (the
// subtraction z_plus - z amplifies either form's rounding alike), which moves a seed's f32 state in about one update in a
// million.
struct TauConsts {
  double se, ce, sed, ced;
};
inline TauConsts tau_consts(double px_error_angle) {
  const double delta = 3.14159265358979323846 - SVO_PI;
  return TauConsts{sin(px_error_angle), cos(px_error_angle), sin(px_error_angle + delta), cos(px_error_angle + delta)};
}
__device__ __forceinline__ double compute_tau(const Se3& T_ref_cur, const double f[3], const double z, const TauConsts& k) {
  const double t[3] = {T_ref_cur.t[0], T_ref_cur.t[1], T_ref_cur.t[2]};
  const double av[3] = {f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2]};
  const double t_norm = norm3(t);
  const double a_norm = norm3(av);
  const double ca = dot3(f, t) / t_norm;
  const double mt[3] = {-t[0], -t[1], -t[2]};
  const double cb = dot3(av, mt) / (t_norm * a_norm);
  const double sa = sqrt(1.0 - ca * ca), sb = sqrt(1.0 - cb * cb);  // (NaN where acos would have been)
  const double sin_beta_plus = sb * k.ce + cb * k.se;
  const double sin_gamma_plus = (sa * cb + ca * sb) * k.ced + (ca * cb - sa * sb) * k.sed;
  const double z_plus = t_norm * sin_beta_plus / sin_gamma_plus;
  return (z_plus - z);
}

}  // namespace svo_track
