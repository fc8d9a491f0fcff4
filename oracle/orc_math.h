/*
 * orc_math.h -- small fixed-size math used by the oracle (TEST INFRASTRUCTURE,
 * see svo_oracle.h).  Restates, from the published upstream sources, the
 * third-party arithmetic the reference calls but does not vendor:
 *   - Sophus (old, non-templated) SO3/SE3: unit-quaternion storage,
 *     SO3::expAndTheta, SE3::exp, SE3::log, operator*, inverse
 *   - Eigen: Quaternion(Matrix3), toRotationMatrix, _transformVector,
 *     LDLT<Lower> unblocked with diagonal pivoting + solve
 */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <math.h>
#include <string.h>
#include <float.h>

#define ORC_SMALL_EPS 1e-10 /* Sophus SMALL_EPS */

typedef struct {
  double q[4]; /* w, x, y, z  (unit quaternion, as Sophus::SO3 stores it) */
  double t[3];
} orc_se3;

/* Eigen::Quaternion(Matrix3) -- QuaternionBase::operator=(MatrixBase) */
static inline void orc_quat_from_R(const double R[9], double q[4]) {
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3;
    int k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

/* Eigen::QuaternionBase::toRotationMatrix */
static inline void orc_quat_to_R(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* Eigen quaternion product a*b */
static inline void orc_quat_mul(const double a[4], const double b[4], double o[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
  double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

static inline void orc_quat_normalize(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Eigen QuaternionBase::_transformVector: uv = 2 vec x v; v + w uv + vec x uv */
static inline void orc_quat_rot(const double q[4], const double v[3], double o[3]) {
  double ux = q[2] * v[2] - q[3] * v[1];
  double uy = q[3] * v[0] - q[1] * v[2];
  double uz = q[1] * v[1] - q[2] * v[0];
  ux += ux; uy += uy; uz += uz;
  double cx = q[2] * uz - q[3] * uy;
  double cy = q[3] * ux - q[1] * uz;
  double cz = q[1] * uy - q[2] * ux;
  o[0] = v[0] + q[0] * ux + cx;
  o[1] = v[1] + q[0] * uy + cy;
  o[2] = v[2] + q[0] * uz + cz;
}

static inline void orc_se3_from_Rt(const double T[12], orc_se3* s) {
  orc_quat_from_R(T, s->q);
  s->t[0] = T[9]; s->t[1] = T[10]; s->t[2] = T[11];
}
static inline void orc_se3_to_Rt(const orc_se3* s, double T[12]) {
  orc_quat_to_R(s->q, T);
  T[9] = s->t[0]; T[10] = s->t[1]; T[11] = s->t[2];
}

/* Sophus SE3::operator*: t += so3*other.t ; so3 *= other.so3 (normalised) */
static inline orc_se3 orc_se3_compose(const orc_se3* a, const orc_se3* b) {
  orc_se3 r;
  double rt[3];
  orc_quat_rot(a->q, b->t, rt);
  r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
  orc_quat_mul(a->q, b->q, r.q);
  orc_quat_normalize(r.q);
  return r;
}

/* Sophus SE3::inverse: so3^-1 (conjugate), t = so3^-1 * (t * -1) */
static inline orc_se3 orc_se3_inverse(const orc_se3* a) {
  orc_se3 r;
  r.q[0] = a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = -a->q[3];
  double nt[3] = {a->t[0] * -1., a->t[1] * -1., a->t[2] * -1.};
  orc_quat_rot(r.q, nt, r.t);
  return r;
}

/* SE3 * Vector3d */
static inline void orc_se3_apply(const orc_se3* a, const double v[3], double o[3]) {
  orc_quat_rot(a->q, v, o);
  o[0] += a->t[0]; o[1] += a->t[1]; o[2] += a->t[2];
}

/* Sophus SO3::expAndTheta + SE3::exp,  xi = [upsilon(3), omega(3)] */
static inline orc_se3 orc_se3_exp_q(const double xi[6]) {
  orc_se3 r;
  const double* ups = xi;
  const double* om = xi + 3;
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double half_theta = 0.5 * theta;
  double imag_factor;
  const double real_factor = cos(half_theta);
  if (theta < ORC_SMALL_EPS) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    const double sin_half_theta = sin(half_theta);
    imag_factor = sin_half_theta / theta;
  }
  r.q[0] = real_factor;
  r.q[1] = imag_factor * om[0];
  r.q[2] = imag_factor * om[1];
  r.q[3] = imag_factor * om[2];
  /* Omega = hat(omega) */
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
      O2[i * 3 + j] = s;
    }
  double V[9];
  if (theta < ORC_SMALL_EPS) {
    orc_quat_to_R(r.q, V);
  } else {
    const double theta_sq = theta * theta;
    const double c1 = (1 - cos(theta)) / (theta_sq);
    const double c2 = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
  return r;
}

/* Sophus SO3::logAndTheta + SE3::log */
static inline void orc_se3_log_q(const orc_se3* s, double xi[6]) {
  const double n = sqrt(s->q[1] * s->q[1] + s->q[2] * s->q[2] + s->q[3] * s->q[3]);
  const double w = s->q[0];
  const double squared_w = w * w;
  double two_atan_nbyw_by_n;
  if (n < ORC_SMALL_EPS) {
    two_atan_nbyw_by_n = 2. / w - 2. * (n * n) / (w * squared_w);
  } else {
    if (fabs(w) < ORC_SMALL_EPS) {
      if (w > 0) two_atan_nbyw_by_n = M_PI / n;
      else two_atan_nbyw_by_n = -M_PI / n;
    }
    two_atan_nbyw_by_n = 2 * atan(n / w) / n;
  }
  const double theta = two_atan_nbyw_by_n * n;
  double om[3] = {two_atan_nbyw_by_n * s->q[1], two_atan_nbyw_by_n * s->q[2], two_atan_nbyw_by_n * s->q[3]};
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += O[i * 3 + k] * O[k * 3 + j];
      O2[i * 3 + j] = a;
    }
  double Vi[9];
  double c2;
  if (theta < ORC_SMALL_EPS) c2 = 1. / 12.;
  else c2 = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c2 * O2[i];
  for (int i = 0; i < 3; ++i)
    xi[i] = Vi[i * 3] * s->t[0] + Vi[i * 3 + 1] * s->t[1] + Vi[i * 3 + 2] * s->t[2];
  xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

/*
 * Eigen::LDLT<Matrix<double,n,n>, Lower>: unblocked in-place factorisation
 * with largest-diagonal pivoting (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>
 * ::unblocked) followed by LDLT::solve (P, L^-1, D^-1 with the
 * numeric_limits<double>::min() tolerance of Eigen >= 3.2.2, L^-T, P^T).
 * A is n x n row-major (symmetric; only the lower triangle is read).
 * Returns 1; x may contain NaN/Inf exactly like Eigen's result would.
 */
static inline int orc_ldlt_solve(int n, const double* A, const double* b, double* x) {
  double m[36 * 4];
  int tr[12];
  double temp[12];
  if (n > 12) return 0;
  for (int i = 0; i < n * n; ++i) m[i] = A[i];
#define M_(r, c) m[(r) * n + (c)]
  if (n <= 1) {
    tr[0] = 0;
  } else {
    for (int k = 0; k < n; ++k) {
      /* pivot: largest |diagonal| in the trailing block (first max wins) */
      int big = k;
      double best = fabs(M_(k, k));
      for (int i = k + 1; i < n; ++i)
        if (fabs(M_(i, i)) > best) { best = fabs(M_(i, i)); big = i; }
      tr[k] = big;
      if (k != big) {
        int s = n - big - 1;
        for (int c = 0; c < k; ++c) { double t = M_(k, c); M_(k, c) = M_(big, c); M_(big, c) = t; }
        for (int r = 0; r < s; ++r) {
          double t = M_(big + 1 + r, k); M_(big + 1 + r, k) = M_(big + 1 + r, big); M_(big + 1 + r, big) = t;
        }
        { double t = M_(k, k); M_(k, k) = M_(big, big); M_(big, big) = t; }
        for (int i = k + 1; i < big; ++i) { double t = M_(i, k); M_(i, k) = M_(big, i); M_(big, i) = t; }
      }
      int rs = n - k - 1;
      if (k > 0) {
        for (int c = 0; c < k; ++c) temp[c] = M_(c, c) * M_(k, c);
        double acc = 0;
        for (int c = 0; c < k; ++c) acc += M_(k, c) * temp[c];
        M_(k, k) -= acc;
        for (int r = 0; r < rs; ++r) {
          double a2 = 0;
          for (int c = 0; c < k; ++c) a2 += M_(k + 1 + r, c) * temp[c];
          M_(k + 1 + r, k) -= a2;
        }
      }
      double akk = M_(k, k);
      int pivot_is_valid = (fabs(akk) > 0.0);
      if (k == 0 && !pivot_is_valid) {
        /* whole diagonal is zero: Eigen fills the transpositions and stops */
        for (int j = 0; j < n; ++j) tr[j] = j;
        break;
      }
      if (rs > 0 && pivot_is_valid)
        for (int r = 0; r < rs; ++r) M_(k + 1 + r, k) /= akk;
    }
  }
  /* solve: dst = P b */
  for (int i = 0; i < n; ++i) x[i] = b[i];
  for (int k = 0; k < n; ++k)
    if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
  /* L^-1 (unit lower) */
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < i; ++c) x[i] -= M_(i, c) * x[c];
  /* D^-1 (pseudo-inverse with tolerance = DBL_MIN) */
  for (int i = 0; i < n; ++i) {
    if (fabs(M_(i, i)) > DBL_MIN) x[i] /= M_(i, i);
    else x[i] = 0;
  }
  /* L^-T */
  for (int i = n - 1; i >= 0; --i)
    for (int c = i + 1; c < n; ++c) x[i] -= M_(c, i) * x[c];
  /* P^T */
  for (int k = n - 1; k >= 0; --k)
    if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef M_
  return 1;
}

/* ---- Eigen small inverses (Eigen/src/LU/Inverse.h) -------------------------
 * 2x2: invdet = 1/det; adjugate * invdet.
 * 3x3: cofactors of column 0, det = sum(cofactors_col0 .* col(0)), invdet,
 *      result(i,j) = cofactor(j,i) * invdet.
 * Matrices are row-major here: m[r*n+c].                                      */
#define ORC_DEFINE_SMALL_INV(SUF, T)                                              \
  static inline T orc_det2##SUF(const T m[4]) { return m[0] * m[3] - m[2] * m[1]; } \
  static inline void orc_inv2##SUF(const T m[4], T r[4]) {                        \
    const T invdet = (T)1 / orc_det2##SUF(m);                                     \
    r[0] = m[3] * invdet;                                                         \
    r[2] = -m[2] * invdet;                                                        \
    r[1] = -m[1] * invdet;                                                        \
    r[3] = m[0] * invdet;                                                         \
  }                                                                               \
  static inline T orc_cof3##SUF(const T m[9], int i, int j) {                     \
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; \
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];     \
  }                                                                               \
  static inline void orc_inv3##SUF(const T m[9], T r[9]) {                        \
    const T c0 = orc_cof3##SUF(m, 0, 0), c1 = orc_cof3##SUF(m, 1, 0), c2 = orc_cof3##SUF(m, 2, 0); \
    const T det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];                            \
    const T invdet = (T)1 / det;                                                  \
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;                   \
    r[3] = orc_cof3##SUF(m, 0, 1) * invdet;                                       \
    r[4] = orc_cof3##SUF(m, 1, 1) * invdet;                                       \
    r[5] = orc_cof3##SUF(m, 2, 1) * invdet;                                       \
    r[6] = orc_cof3##SUF(m, 0, 2) * invdet;                                       \
    r[7] = orc_cof3##SUF(m, 1, 2) * invdet;                                       \
    r[8] = orc_cof3##SUF(m, 2, 2) * invdet;                                       \
  }
ORC_DEFINE_SMALL_INV(f, float)
ORC_DEFINE_SMALL_INV(d, double)

/* General inverse by partial-pivot LU (Eigen PartialPivLU for n > 4), n <= 12.
 * Only used for Frame::Cov_ (pose_optimizer.cpp:126); compared with a tolerance. */
static inline int orc_inv_lu(int n, const double* A, double* out) {
  double lu[144];
  int perm[12];
  if (n > 12) return 0;
  for (int i = 0; i < n * n; ++i) lu[i] = A[i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(lu[i * n + k]) > best) { best = fabs(lu[i * n + k]); piv = i; }
    if (piv != k) {
      for (int c = 0; c < n; ++c) { double t = lu[k * n + c]; lu[k * n + c] = lu[piv * n + c]; lu[piv * n + c] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < n; ++i) {
      lu[i * n + k] /= lu[k * n + k];
      for (int c = k + 1; c < n; ++c) lu[i * n + c] -= lu[i * n + k] * lu[k * n + c];
    }
  }
  for (int col = 0; col < n; ++col) {
    double y[12];
    for (int i = 0; i < n; ++i) {
      y[i] = (perm[i] == col) ? 1.0 : 0.0;
      for (int c = 0; c < i; ++c) y[i] -= lu[i * n + c] * y[c];
    }
    for (int i = n - 1; i >= 0; --i) {
      for (int c = i + 1; c < n; ++c) y[i] -= lu[i * n + c] * y[c];
      y[i] /= lu[i * n + i];
    }
    for (int i = 0; i < n; ++i) out[i * n + col] = y[i];
  }
  return 1;
}

#endif /* ORC_MATH_H_ */
