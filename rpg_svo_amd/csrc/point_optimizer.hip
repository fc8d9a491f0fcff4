// point_optimizer.hip -- K6: batched Point::optimize for gfx950.
//
// Replaces svo::Point::optimize (svo/src/point.cpp:119-177; Jacobian point.h:89-103): five
// Gauss-Newton iterations on the 3-D position of a map point over its keyframe observations
// (FrameHandlerBase::optimizeStructure picks <= 20 points per frame,
// frame_handler_base.cpp:178-196).  A point has 2-10 observations and a 3x3 system: one lane
// per point, observations visited in list order, so the sums are the reference's sums.
#pragma clang fp contract(off)
#include "track_kernels.h"
#include "track_math.h"

using namespace svo_capi;
using namespace svo_dev;

namespace {

struct PointArgs {
  int P;
  const double* frame_T;
  const int32_t* obs_ptr;
  const int32_t* obs_frame;
  const double* obs_f;
  int n_iter;
  double* pos;
};

__global__ void __launch_bounds__(64) point_opt_kernel(const PointArgs a) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= a.P) return;
  const int o0 = a.obs_ptr[p], o1 = a.obs_ptr[p + 1];
  double pos[3] = {a.pos[3 * p], a.pos[3 * p + 1], a.pos[3 * p + 2]};
  double old_point[3] = {pos[0], pos[1], pos[2]};
  double chi2 = 0.0;
  for (int it = 0; it < a.n_iter; ++it) {
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    double new_chi2 = 0.0;
    for (int o = o0; o < o1; ++o) {
      Se3 T;
      se3_from_Rt(a.frame_T + 12 * a.obs_frame[o], T);
      double p_in_f[3], R[9];
      se3_apply(T, pos, p_in_f);
      quat_to_R(T.q, R);
      const double z_inv = 1.0 / p_in_f[2];
      const double z_inv_sq = z_inv * z_inv;
      const double pj[6] = {z_inv, 0.0, -p_in_f[0] * z_inv_sq, 0.0, z_inv, -p_in_f[1] * z_inv_sq};
      double J[6];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          J[r * 3 + c] = (-pj[r * 3]) * R[c] + (-pj[r * 3 + 1]) * R[3 + c] + (-pj[r * 3 + 2]) * R[6 + c];
      const double fb[3] = {a.obs_f[3 * o], a.obs_f[3 * o + 1], a.obs_f[3 * o + 2]};
      double u0[2], u1[2];
      project2d(fb, u0);
      project2d(p_in_f, u1);
      const double e[2] = {u0[0] - u1[0], u0[1] - u1[1]};
      new_chi2 += e[0] * e[0] + e[1] * e[1];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[r * 3 + c] += J[r] * J[c] + J[3 + r] * J[3 + c];
#pragma unroll
      for (int r = 0; r < 3; ++r) b[r] -= J[r] * e[0] + J[3 + r] * e[1];
    }
    double dp[3];
    ldlt_solve_pivoted<3>(A, b, dp);
    if ((it > 0 && new_chi2 > chi2) || isnan(dp[0])) {
      pos[0] = old_point[0]; pos[1] = old_point[1]; pos[2] = old_point[2];
      break;
    }
    const double np[3] = {pos[0] + dp[0], pos[1] + dp[1], pos[2] + dp[2]};
    old_point[0] = pos[0]; old_point[1] = pos[1]; old_point[2] = pos[2];
    pos[0] = np[0]; pos[1] = np[1]; pos[2] = np[2];
    chi2 = new_chi2;
    double nm = -1;
    for (int k = 0; k < 3; ++k)
      if (fabs(dp[k]) > nm) nm = fabs(dp[k]);
    if (nm <= 0.0000000001) break;
  }
  a.pos[3 * p] = pos[0];
  a.pos[3 * p + 1] = pos[1];
  a.pos[3 * p + 2] = pos[2];
}

}  // namespace

extern "C" int svo_hip_point_optimize(const svo_hip_frames* frames, int P, const int32_t* d_obs_ptr,
                                      const int32_t* d_obs_frame, const double* d_obs_f, int n_iter, double* d_pos,
                                      void* stream) {
  if (!frames || P < 0 || n_iter < 0) return SVO_HIP_EINVAL;
  if (P == 0) return SVO_HIP_OK;
  if (!frames->d_T_f_w || !d_obs_ptr || !d_obs_frame || !d_obs_f || !d_pos) return SVO_HIP_EINVAL;
  PointArgs a;
  a.P = P;
  a.frame_T = frames->d_T_f_w;
  a.obs_ptr = d_obs_ptr;
  a.obs_frame = d_obs_frame;
  a.obs_f = d_obs_f;
  a.n_iter = n_iter;
  a.pos = d_pos;
  hipLaunchKernelGGL(point_opt_kernel, dim3((P + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}
