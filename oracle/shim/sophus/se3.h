// -*- C++ -*-
// oracle/shim/sophus/se3.h -- TEST INFRASTRUCTURE ONLY.
// Stand-in for the old non-templated Sophus::SE3 (unit-quaternion + translation) used by
// the reference (svo/include/svo/frame.h:20,49).  The arithmetic is ../orc_math.h, i.e.
// the oracle's own restatement of Sophus from the published sources (unpinned).
#ifndef ORC_SHIM_SOPHUS_SE3_H
#define ORC_SHIM_SOPHUS_SE3_H
#include <Eigen/Core>

namespace Sophus {
using namespace Eigen;
using namespace std;
typedef Matrix<double, 6, 1> Vector6d;
typedef Matrix<double, 6, 6> Matrix6d;

class SE3 {
 public:
  orc_se3 s;
  SE3() { s.q[0] = 1; s.q[1] = s.q[2] = s.q[3] = 0; s.t[0] = s.t[1] = s.t[2] = 0; }
  SE3(const Matrix3d& R, const Vector3d& t) {
    double Rr[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rr[i * 3 + j] = R(i, j);
    orc_quat_from_R(Rr, s.q);
    s.t[0] = t[0]; s.t[1] = t[1]; s.t[2] = t[2];
  }
  explicit SE3(const orc_se3& o) : s(o) {}
  SE3 operator*(const SE3& o) const { return SE3(orc_se3_compose(&s, &o.s)); }
  Vector3d operator*(const Vector3d& v) const { Vector3d o; orc_se3_apply(&s, v.data(), o.data()); return o; }
  SE3 inverse() const { return SE3(orc_se3_inverse(&s)); }
  Matrix3d rotation_matrix() const {
    double R[9];
    orc_quat_to_R(s.q, R);
    Matrix3d m;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = R[i * 3 + j];
    return m;
  }
  Vector3d translation() const { return Vector3d(s.t[0], s.t[1], s.t[2]); }
  // so3().unit_quaternion().{w,x,y,z}() as the old Sophus spells them (read-only views)
  struct QuatView {
    const double* q;
    double w() const { return q[0]; }
    double x() const { return q[1]; }
    double y() const { return q[2]; }
    double z() const { return q[3]; }
  };
  struct So3View {
    const double* q;
    QuatView unit_quaternion() const { QuatView v; v.q = q; return v; }
  };
  So3View so3() const { So3View v; v.q = s.q; return v; }
  static SE3 exp(const Vector6d& xi) { return SE3(orc_se3_exp_q(xi.data())); }
  static Vector6d log(const SE3& T) { Vector6d x; orc_se3_log_q(&T.s, x.data()); return x; }
  Vector6d log() const { return log(*this); }
};
}  // namespace Sophus
#endif
