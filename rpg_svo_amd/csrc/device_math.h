// device_math.h -- small f64 SO(3)/SE(3) + 6x6 LDLT helpers shared by the gfx950
// kernels.  Formulas follow what the reference's dependencies compute (Sophus
// SE3::exp / operator*, Eigen quaternion <-> matrix), written for registers:
// everything is fully unrolled, no dynamically indexed local arrays.
#pragma once
// SVO_HOST_MATH_TEST: the CPU test suite compiles these same functions with g++ (tests/host/device_math_on_host.cpp) and
// checks them against the oracle without a GPU.  Whatever needs the wave (cross-lane moves, v_rsq) is left out there.
#ifdef SVO_HOST_MATH_TEST
#include <cmath>
#include <cstdint>
#include <cstring>
#ifndef __device__  // (tests/host/hip_emu.h has defined them already when whole kernels are compiled for the host)
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__ static const
#endif
using std::fabs;
using std::isnan;
using std::sqrt;
#else
#include <hip/hip_runtime.h>
#endif

// 2^-k (0 <= k < 64) from its exponent bits.  x * pow2_inv(k) == x / 2^k bit for bit (scaling by a power of two is exact
// away from the subnormals), without the division sequence -- 11 f64 instructions -- a run-time k otherwise costs.
__device__ __forceinline__ double pow2_inv_f64(const int k) { return __builtin_bit_cast(double, (unsigned long long)(1023 - k) << 52); }
__device__ __forceinline__ float pow2_inv_f32(const int k) { return __builtin_bit_cast(float, (unsigned)(127 - k) << 23); }

// Hand-over of LDS data between lanes of ONE wave: the DS operations of a wave execute in order, so all it takes is to
// keep the compiler from moving the reads above the writes.  Two names for the same two builtins, by who takes part:
// SVO_WAVE_LDS_HANDOVER: every live lane of the wave reaches this line; SVO_LANES_LDS_HANDOVER: the lanes that took this
// branch (a trial's ten lanes, a seed's eight).  The CPU emulation of the test suite (tests/host/hip_emu.h) gives them
// the barrier each one stands for.
#ifndef SVO_HOST_MATH_TEST
#define SVO_WAVE_LDS_HANDOVER()                             \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
  } while (0)
#define SVO_LANES_LDS_HANDOVER() SVO_WAVE_LDS_HANDOVER()
// the same where the code orders its own instructions and only the compiler has to be held back: the fence alone
#define SVO_WAVE_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define SVO_LANES_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
// a ballot inside a divergent branch: the mask of the lanes that are active there (a caller that only looks at its own
// bit, e.g. a select by lane mask, may stand anywhere); the plain builtin is used where every live lane of the wave votes
#define SVO_BALLOT_ACTIVE(pred) __builtin_amdgcn_ballot_w64(pred)
// the workgroup's dynamically sized LDS (the byte count is the launch's), as an array `name` of `type`
#define SVO_DYNAMIC_LDS(type, name) extern __shared__ type name[]
#endif

namespace svo_dev {

// (int)floorf(x) in one instruction (v_cvt_flr_i32_f32; the compiler emits v_floor_f32 + v_cvt_i32_f32 for the C form)
__device__ __forceinline__ int floor_to_int(float x) {
#ifdef SVO_HOST_MATH_TEST
  return (int)floorf(x);
#else
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#endif
}

// Eigen QuaternionBase::toRotationMatrix; q = (w, x, y, z)
__device__ __forceinline__ void quat_to_R(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Eigen Quaternion(Matrix3).  The four cases of Eigen's algorithm (trace > 0, or the
// largest diagonal element i with j=(i+1)%3, k=(j+1)%3) are evaluated with selects so
// the result stays in registers (an array indexed by i would live in scratch).
__device__ __forceinline__ void quat_from_R(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  // Eigen: i = 0; if (m11 > m00) i = 1; if (m22 > m_ii) i = 2;
  const bool i1 = R[4] > R[0];
  const double mii = i1 ? R[4] : R[0];
  const bool i2 = R[8] > mii;
  const int i = i2 ? 2 : (i1 ? 1 : 0);
  // s = the quantity under the square root in each case
  const double s_tr = tr + 1.0;
  const double s_0 = R[0] - R[4] - R[8] + 1.0;
  const double s_1 = R[4] - R[8] - R[0] + 1.0;
  const double s_2 = R[8] - R[0] - R[4] + 1.0;
  const bool pos = tr > 0.0;
  const double sq = sqrt(pos ? s_tr : (i == 0 ? s_0 : (i == 1 ? s_1 : s_2)));
  const double big = 0.5 * sq;   // the component computed from the square root
  const double f = 0.5 / sq;
  // antisymmetric / symmetric off-diagonal combinations
  const double a21 = R[7] - R[5], a02 = R[2] - R[6], a10 = R[3] - R[1];
  const double s10 = R[3] + R[1], s20 = R[6] + R[2], s21 = R[7] + R[5];
  q[0] = pos ? big : (i == 0 ? a21 * f : (i == 1 ? a02 * f : a10 * f));
  q[1] = pos ? a21 * f : (i == 0 ? big : (i == 1 ? s10 * f : s20 * f));
  q[2] = pos ? a02 * f : (i == 0 ? s10 * f : (i == 1 ? big : s21 * f));
  q[3] = pos ? a10 * f : (i == 0 ? s20 * f : (i == 1 ? s21 * f : big));
}

__device__ __forceinline__ void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

__device__ __forceinline__ void quat_normalize(double q[4]) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double inv = 1.0 / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}

// Eigen _transformVector
__device__ __forceinline__ void quat_rot(const double q[4], const double v[3], double o[3]) {
  double ux = q[2] * v[2] - q[3] * v[1];
  double uy = q[3] * v[0] - q[1] * v[2];
  double uz = q[1] * v[1] - q[2] * v[0];
  ux += ux; uy += uy; uz += uz;
  const double cx = q[2] * uz - q[3] * uy;
  const double cy = q[3] * ux - q[1] * uz;
  const double cz = q[1] * uy - q[2] * ux;
  o[0] = v[0] + q[0] * ux + cx;
  o[1] = v[1] + q[0] * uy + cy;
  o[2] = v[2] + q[0] * uz + cz;
}

// Taylor coefficients of sin (x^3..x^17, sign included, highest first) and cos
// (x^2..x^16).  Kept in constant memory and read through a laundered pointer so the
// compiler loads them into SGPRs at the point of use instead of hoisting sixteen f64
// literals into VGPRs for the lifetime of the calling kernel.  The laundered pointer
// keeps its CONSTANT address space: through a generic pointer the sixteen values came as
// eight flat_load_dwordx4, each waited for on its own inside the Horner chain (round 4:
// found in K4's ISA, 8 vector-memory round trips per SE3::exp); from the constant address
// space they are scalar loads into SGPRs, requested together.
#ifndef SVO_HOST_MATH_TEST
typedef const double __attribute__((address_space(4))) const_as_double;
#endif
__constant__ double kSinCoef[8] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0,
                                   1.0 / 39916800.0,         -1.0 / 362880.0,       1.0 / 5040.0,
                                   -1.0 / 120.0,             1.0 / 6.0};
__constant__ double kCosCoef[8] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0,
                                   -1.0 / 3628800.0,       1.0 / 40320.0,        -1.0 / 720.0,
                                   1.0 / 24.0,             -0.5};

// sin/cos for the angles a Gauss-Newton step produces.  |x| <= 0.5 (always, in
// practice): Taylor/Horner in x^2, truncation < 2e-20 relative.  Larger arguments
// are halved until they fit and rebuilt with the double-angle identities (no
// libm-style argument-reduction tables).
__device__ __forceinline__ void sincos_small(double x, double* s, double* c) {
  int k = 0;
  while (fabs(x) > 0.5 && k < 64) {
    x *= 0.5;
    ++k;
  }
#if defined(SVO_HOST_MATH_TEST)
  const double* cs = kSinCoef;
  const double* cc = kCosCoef;
#else  // (constant address space: scalar loads; a generic pointer measured 3 % slower in K4, profiles/r04v_*)
  const_as_double* cs = (const_as_double*)kSinCoef;
  const_as_double* cc = (const_as_double*)kCosCoef;
  asm volatile("" : "+s"(cs), "+s"(cc));
#endif
  const double z = x * x;
  double ps = cs[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) ps = ps * z + cs[i];
  double sv = x - x * z * ps;
  double pc = cc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) pc = pc * z + cc[i];
  double cv = 1.0 + z * pc;
  for (; k > 0; --k) {
    const double s2 = 2.0 * sv * cv;
    cv = 1.0 - 2.0 * sv * sv;
    sv = s2;
  }
  *s = sv;
  *c = cv;
}

// Sophus SE3::exp (SO3::expAndTheta + V matrix); xi = [upsilon, omega].
__device__ __forceinline__ void se3_exp(const double xi[6], double q[4], double t[3]) {
  const double ox = xi[3], oy = xi[4], oz = xi[5];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  const double theta = sqrt(theta_sq);
  const double half_theta = 0.5 * theta;
  double imag_factor, c1, c2;
  double s_h, c_h;
  sincos_small(half_theta, &s_h, &c_h);
  if (theta < 1e-10) {
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    imag_factor = s_h / theta;
  }
  q[0] = c_h;
  q[1] = imag_factor * ox;
  q[2] = imag_factor * oy;
  q[3] = imag_factor * oz;
  // V = I + c1*Omega + c2*Omega^2 ; V*u = u + c1*(w x u) + c2*(w x (w x u))
  const double ux = xi[0], uy = xi[1], uz = xi[2];
  if (theta < 1e-10) {
    // V = so3.matrix()  (Sophus: "that is an accurate expansion")
    quat_rot(q, xi, t);
  } else {
    // sin(theta) = 2 s c, 1 - cos(theta) = 2 s^2 with s,c of theta/2
    const double s_t = 2.0 * s_h * c_h;
    c1 = (2.0 * s_h * s_h) / theta_sq;
    c2 = (theta - s_t) / (theta_sq * theta);
    const double wx = oy * uz - oz * uy;
    const double wy = oz * ux - ox * uz;
    const double wz = ox * uy - oy * ux;
    const double wwx = oy * wz - oz * wy;
    const double wwy = oz * wx - ox * wz;
    const double wwz = ox * wy - oy * wx;
    t[0] = ux + c1 * wx + c2 * wwx;
    t[1] = uy + c1 * wy + c2 * wwy;
    t[2] = uz + c1 * wz + c2 * wwz;
  }
}

// SE3::exp for a Gauss-Newton increment, in f32 and without sqrt/divisions: the four
// scalar functions of theta that Sophus::SE3::exp evaluates,
//   sin(th/2)/th, cos(th/2), (1-cos th)/th^2, (th-sin th)/th^3,
// are entire in th^2; nine-term Horner forms are exact to f32 rounding for th <= 2 rad
// and stay finite beyond (steps of that size only occur once an alignment has
// diverged).  The increment is tiny against the f64 pose it is composed with, so
// f32 here perturbs an iterate by <= 1e-7 of the step, which Gauss-Newton absorbs.
// SE3_EXP_SHORT (default): a Gauss-Newton increment rotates by a fraction of a degree, and for theta^2 < 0.01 the series
// are exact to f32 rounding after four terms (next terms: y^3/5040 < 4e-12, y^4/40320 < 2e-15, z^4/3628800 < 3e-15,
// z^4/39916800 < 3e-16 of the leading one) -- 11 fused multiply-adds instead of 32 in the solver wave's serial section.
// The argument is wave-uniform in the kernels (the increment comes out of v_readlane), so is the branch.
__device__ __forceinline__ void se3_exp_f32(const float xi[6], float q[4], float t[3]) {
  const float ox = xi[3], oy = xi[4], oz = xi[5];
  const float z = ox * ox + oy * oy + oz * oz;  // theta^2
  const float y = 0.25f * z;                    // (theta/2)^2
  if (z < 0.01f) {
    float a = 1.f / 5040.f;
    a = a * -y + 1.f / 120.f;
    a = a * -y + 1.f / 6.f;
    a = a * -y + 1.f;
    float w = 1.f / 720.f;
    w = w * -y + 1.f / 24.f;
    w = w * -y + 0.5f;
    w = w * -y + 1.f;
    float c1 = 1.f / 40320.f;
    c1 = c1 * -z + 1.f / 720.f;
    c1 = c1 * -z + 1.f / 24.f;
    c1 = c1 * -z + 0.5f;
    float c2 = 1.f / 362880.f;
    c2 = c2 * -z + 1.f / 5040.f;
    c2 = c2 * -z + 1.f / 120.f;
    c2 = c2 * -z + 1.f / 6.f;
    const float imag = 0.5f * a;
    q[0] = w;
    q[1] = imag * ox;
    q[2] = imag * oy;
    q[3] = imag * oz;
    const float ux = xi[0], uy = xi[1], uz = xi[2];
    const float wx = oy * uz - oz * uy, wy = oz * ux - ox * uz, wz = ox * uy - oy * ux;
    const float wwx = oy * wz - oz * wy, wwy = oz * wx - ox * wz, wwz = ox * wy - oy * wx;
    t[0] = ux + c1 * wx + c2 * wwx;
    t[1] = uy + c1 * wy + c2 * wwy;
    t[2] = uz + c1 * wz + c2 * wwz;
    return;
  }
  // sinc(th/2) = sum (-1)^k y^k/(2k+1)!,  cos(th/2) = sum (-1)^k y^k/(2k)!
  float a = 1.f / 355687428096000.f;
  a = a * -y + 1.f / 1307674368000.f;
  a = a * -y + 1.f / 6227020800.f;
  a = a * -y + 1.f / 39916800.f;
  a = a * -y + 1.f / 362880.f;
  a = a * -y + 1.f / 5040.f;
  a = a * -y + 1.f / 120.f;
  a = a * -y + 1.f / 6.f;
  a = a * -y + 1.f;
  float w = 1.f / 20922789888000.f;
  w = w * -y + 1.f / 87178291200.f;
  w = w * -y + 1.f / 479001600.f;
  w = w * -y + 1.f / 3628800.f;
  w = w * -y + 1.f / 40320.f;
  w = w * -y + 1.f / 720.f;
  w = w * -y + 1.f / 24.f;
  w = w * -y + 0.5f;
  w = w * -y + 1.f;
  // (1-cos th)/th^2 = sum (-1)^k z^k/(2k+2)!,  (th-sin th)/th^3 = sum (-1)^k z^k/(2k+3)!
  float c1 = 1.f / 6402373705728000.f;
  c1 = c1 * -z + 1.f / 20922789888000.f;
  c1 = c1 * -z + 1.f / 87178291200.f;
  c1 = c1 * -z + 1.f / 479001600.f;
  c1 = c1 * -z + 1.f / 3628800.f;
  c1 = c1 * -z + 1.f / 40320.f;
  c1 = c1 * -z + 1.f / 720.f;
  c1 = c1 * -z + 1.f / 24.f;
  c1 = c1 * -z + 0.5f;
  float c2 = 1.f / 121645100408832000.f;
  c2 = c2 * -z + 1.f / 355687428096000.f;
  c2 = c2 * -z + 1.f / 1307674368000.f;
  c2 = c2 * -z + 1.f / 6227020800.f;
  c2 = c2 * -z + 1.f / 39916800.f;
  c2 = c2 * -z + 1.f / 362880.f;
  c2 = c2 * -z + 1.f / 5040.f;
  c2 = c2 * -z + 1.f / 120.f;
  c2 = c2 * -z + 1.f / 6.f;
  const float imag = 0.5f * a;
  q[0] = w;
  q[1] = imag * ox;
  q[2] = imag * oy;
  q[3] = imag * oz;
  const float ux = xi[0], uy = xi[1], uz = xi[2];
  const float wx = oy * uz - oz * uy, wy = oz * ux - ox * uz, wz = ox * uy - oy * ux;
  const float wwx = oy * wz - oz * wy, wwy = oz * wx - ox * wz, wwz = ox * wy - oy * wx;
  t[0] = ux + c1 * wx + c2 * wwx;
  t[1] = uy + c1 * wy + c2 * wwy;
  t[2] = uz + c1 * wz + c2 * wwz;
}

// The two halves of se3_exp_f32 on their own, for a solve that is split over two waves (sparse_align.hip): the rotation
// part -- the unit quaternion of exp(omega^) -- and the translation part V(omega) * upsilon.  Same series, same
// coefficients, same order of operations as se3_exp_f32, so each half returns the bits the whole returns.
__device__ __forceinline__ void se3_exp_rot_f32(const float xi[6], float q[4]) {
  const float ox = xi[3], oy = xi[4], oz = xi[5];
  const float z = ox * ox + oy * oy + oz * oz;
  const float y = 0.25f * z;
  float a, w;
  if (z < 0.01f) {
    a = 1.f / 5040.f;
    a = a * -y + 1.f / 120.f;
    a = a * -y + 1.f / 6.f;
    a = a * -y + 1.f;
    w = 1.f / 720.f;
    w = w * -y + 1.f / 24.f;
    w = w * -y + 0.5f;
    w = w * -y + 1.f;
  } else
  {
    a = 1.f / 355687428096000.f;
    a = a * -y + 1.f / 1307674368000.f;
    a = a * -y + 1.f / 6227020800.f;
    a = a * -y + 1.f / 39916800.f;
    a = a * -y + 1.f / 362880.f;
    a = a * -y + 1.f / 5040.f;
    a = a * -y + 1.f / 120.f;
    a = a * -y + 1.f / 6.f;
    a = a * -y + 1.f;
    w = 1.f / 20922789888000.f;
    w = w * -y + 1.f / 87178291200.f;
    w = w * -y + 1.f / 479001600.f;
    w = w * -y + 1.f / 3628800.f;
    w = w * -y + 1.f / 40320.f;
    w = w * -y + 1.f / 720.f;
    w = w * -y + 1.f / 24.f;
    w = w * -y + 0.5f;
    w = w * -y + 1.f;
  }
  const float imag = 0.5f * a;
  q[0] = w;
  q[1] = imag * ox;
  q[2] = imag * oy;
  q[3] = imag * oz;
}
__device__ __forceinline__ void se3_exp_trans_f32(const float xi[6], float t[3]) {
  const float ox = xi[3], oy = xi[4], oz = xi[5];
  const float z = ox * ox + oy * oy + oz * oz;
  float c1, c2;
  if (z < 0.01f) {
    c1 = 1.f / 40320.f;
    c1 = c1 * -z + 1.f / 720.f;
    c1 = c1 * -z + 1.f / 24.f;
    c1 = c1 * -z + 0.5f;
    c2 = 1.f / 362880.f;
    c2 = c2 * -z + 1.f / 5040.f;
    c2 = c2 * -z + 1.f / 120.f;
    c2 = c2 * -z + 1.f / 6.f;
  } else
  {
    c1 = 1.f / 6402373705728000.f;
    c1 = c1 * -z + 1.f / 20922789888000.f;
    c1 = c1 * -z + 1.f / 87178291200.f;
    c1 = c1 * -z + 1.f / 479001600.f;
    c1 = c1 * -z + 1.f / 3628800.f;
    c1 = c1 * -z + 1.f / 40320.f;
    c1 = c1 * -z + 1.f / 720.f;
    c1 = c1 * -z + 1.f / 24.f;
    c1 = c1 * -z + 0.5f;
    c2 = 1.f / 121645100408832000.f;
    c2 = c2 * -z + 1.f / 355687428096000.f;
    c2 = c2 * -z + 1.f / 1307674368000.f;
    c2 = c2 * -z + 1.f / 6227020800.f;
    c2 = c2 * -z + 1.f / 39916800.f;
    c2 = c2 * -z + 1.f / 362880.f;
    c2 = c2 * -z + 1.f / 5040.f;
    c2 = c2 * -z + 1.f / 120.f;
    c2 = c2 * -z + 1.f / 6.f;
  }
  const float ux = xi[0], uy = xi[1], uz = xi[2];
  const float wx = oy * uz - oz * uy, wy = oz * ux - ox * uz, wz = ox * uy - oy * ux;
  const float wwx = oy * wz - oz * wy, wwy = oz * wx - ox * wz, wwz = ox * wy - oy * wx;
  t[0] = ux + c1 * wx + c2 * wwx;
  t[1] = uy + c1 * wy + c2 * wwy;
  t[2] = uz + c1 * wz + c2 * wwz;
}

#if !defined(SVO_HOST_MATH_TEST) || defined(SVO_HIP_EMU)  // (v_rsq_f64, v_readlane: served by tests/host/hip_emu.h only)
// q <- q / |q| for a quaternion that is already close to unit length or not:
// v_rsq_f64 seed + two Newton steps (no f64 sqrt / division sequences).
__device__ __forceinline__ void quat_normalize_fast(double q[4]) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double r = __builtin_amdgcn_rsq(n2);
  r = r * (1.5 - 0.5 * n2 * r * r);
  r = r * (1.5 - 0.5 * n2 * r * r);
  q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}

// broadcast lane `src` (compile-time constant) of a double to the whole wave (SGPRs)
template <int SRC>
__device__ __forceinline__ double readlane_f64(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, SRC);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), SRC);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
#endif

// index into the packed upper triangle of a symmetric 6x6 (row-major, i<=j)
__host__ __device__ constexpr int sym6(int i, int j) {
  return (i <= j) ? (i * 6 - (i * (i - 1)) / 2 + (j - i)) : (j * 6 - (j * (j - 1)) / 2 + (i - j));
}
// index of L(i,j), i>j, in a packed strictly-lower 6x6 (15 entries)
__host__ __device__ constexpr int low6(int i, int j) { return (i * (i - 1)) / 2 + j; }

// LDL^T of a symmetric 6x6 given as packed upper triangle H[21]; no pivoting
// (the systems here are SPD normal equations).  A pivot with |d| <= DBL_MIN is
// treated like Eigen's LDLT::solve treats it: its D^-1 entry becomes 0.
// LD[0..14] = L strictly lower, LD[15..20] = 1/d.
__device__ __forceinline__ void ldlt6_factor(const double H[21], double LD[21]) {
  double d[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dj = H[sym6(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) dj -= LD[low6(j, k)] * LD[low6(j, k)] * d[k];
    d[j] = dj;
    const bool ok = fabs(dj) > 2.2250738585072014e-308;
    const double inv = ok ? 1.0 / dj : 0.0;
    LD[15 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = H[sym6(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= LD[low6(i, k)] * LD[low6(j, k)] * d[k];
      LD[low6(i, j)] = ok ? v * inv : v;
    }
  }
}

__device__ __forceinline__ void ldlt6_solve(const double LD[21], const double b[6], double x[6]) {
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= LD[low6(i, k)] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] *= LD[15 + i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= LD[low6(k, i)] * x[k];
    x[i] = v;
  }
}

#if !defined(SVO_HOST_MATH_TEST) || defined(SVO_HIP_EMU)
// full-wave (64 lanes) sum; every lane receives the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
#endif

}  // namespace svo_dev
