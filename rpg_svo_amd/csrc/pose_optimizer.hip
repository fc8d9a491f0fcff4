// pose_optimizer.hip -- K4: batched pose_optimizer::optimizeGaussNewton for gfx950.
//
// Replaces svo::pose_optimizer::optimizeGaussNewton (svo/src/pose_optimizer.cpp:28-161):
// robust (Tukey / MAD) Gauss-Newton refinement of Frame::T_f_w_ over the reprojection
// errors of the frame's matched features, the covariance (:126), the outlier pruning
// (:129-145) and the reported medians (:147-152).
//
// Mapping: one workgroup per frame, one lane per observation.  An iteration needs 28 f64
// sums (21 unique entries of A, 6 of b, chi2).  Lanes drop their contributions into an LDS
// tile [28][BLOCK]; 28 lanes then add their row in observation order, chunk after chunk --
// the same order the reference's `for(fts_)` loop adds them in, so the normal equations are
// reproduced to the bit wherever the per-observation arithmetic is (no reassociation).
// Lane 0 runs the 6x6 pivoted LDLT (Eigen's algorithm), the rollback / convergence rules
// and SE3::exp(dT)*T; medians are found by exact rank counting in LDS (nth_element's value).
#pragma clang fp contract(off)
#include "track_kernels.h"
#include "track_math.h"

using namespace svo_capi;
using namespace svo_dev;

namespace {

constexpr int PO_MINW = 4;
constexpr int PO_HALF = 32;
constexpr int PO_BLOCK = 64;  // one wave per frame: the serial parts dominate, occupancy comes from many small workgroups
constexpr int PO_MAXN = 1024;
constexpr double SVO_EPS = 0.0000000001;  // svo/include/svo/global.h:77

struct PoseArgs {
  Cam cam;
  const int32_t* n;
  int n_stride;
  const double* f;
  const int32_t* level;
  const double* pos;
  uint8_t* has_point;
  double reproj_thresh;
  int n_iter;
  double* T;
  double* Cov;
  double* stats;
  int32_t* ran;
  int only_flagged;  // 1: work only on frames the wave kernel handed over (ran[b] == 2)
};

struct PoseLds {
  double tile[28][PO_HALF];  // half a chunk at a time: 7 KB instead of 14 KB per frame doubles the frames per CU
  unsigned long long live[PO_BLOCK / 64];  // which lanes of the current chunk contributed
  double acc[28];
  double dT[6];
  double new_chi2;
  Se3 T, T_old;
  double scale;
  double median_d;
  double chi2;
  float median_f;
  int n_err;
  int flag;  // 0 continue, 1 stop
};

// tukey: vk::robust_cost::TukeyWeightFunction::value
__device__ __forceinline__ float tukey_weight(float x) {
  const float b_square = 4.6851f * 4.6851f;
  const float x_square = x * x;
  if (x_square <= b_square) {
    const float tmp = 1.0f - x_square / b_square;
    return tmp * tmp;
  }
  return 0.0f;
}

// value with rank k among the entries of v[0..n) flagged in `hp` (ties broken by index):
// what nth_element(begin, begin+k, end) leaves at position k.
template <typename T>
__device__ __forceinline__ void rank_select(const T* v, int n, int k, T* out) {
  // entries that do not take part hold +inf: they are never "smaller" and never selected
  for (int i = threadIdx.x; i < n; i += PO_BLOCK) {
    const T mine = v[i];
    if (!(mine < (T)INFINITY)) continue;
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) {
      const T o = v[j];
      rank += (o < mine || (o == mine && j < i)) ? 1 : 0;
    }
    if (rank == k) *out = mine;
  }
}

// per-observation arrays, sized by n_stride at launch (dynamic LDS):
//   vals f64 [ns]: chi2_vec_init, later chi2_vec_final (+inf where there is no point)
//   err  f32 [ns]: |e| for the MAD scale, later the pruning marks
//   hp   u8  [ns]: has_point of this frame (global loads inside the serial sums would dominate)
struct PoseDyn {
  double* vals;
  float* err;
  uint8_t* hp;
};

__global__ void __launch_bounds__(PO_BLOCK, PO_MINW) pose_opt_kernel(const PoseArgs a) {
  __shared__ PoseLds s;
  SVO_DYNAMIC_LDS(double, po_dyn);
  const int ns8 = (a.n_stride + 7) & ~7;
  PoseDyn d;
  d.vals = po_dyn;
  d.err = reinterpret_cast<float*>(po_dyn + ns8);
  d.hp = reinterpret_cast<uint8_t*>(d.err + ns8);
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (a.only_flagged && a.ran[b] != 2) return;
  int n = a.n[b];
  n = n < 0 ? 0 : (n > a.n_stride ? a.n_stride : n);  // contract: n <= n_stride; never index past the row
  const size_t base = (size_t)b * a.n_stride;
  const double focal = fabs(a.cam.fx);

  if (tid == 0) {
    se3_from_Rt(a.T + 12 * b, s.T);
    s.T_old = s.T;
    s.n_err = 0;
    s.flag = 0;
    s.chi2 = 0.0;
  }
  for (int i = tid; i < n; i += PO_BLOCK) d.hp[i] = a.has_point[base + i];
  __syncthreads();
  // ---- error scale (:45-60) ---------------------------------------------------------
  {
    const Se3 T = s.T;
    int cnt = 0;
    for (int i = tid; i < n; i += PO_BLOCK) {
      d.err[i] = INFINITY;
      if (!d.hp[i]) continue;
      const double p[3] = {a.pos[3 * (base + i)], a.pos[3 * (base + i) + 1], a.pos[3 * (base + i) + 2]};
      const double fb[3] = {a.f[3 * (base + i)], a.f[3 * (base + i) + 1], a.f[3 * (base + i) + 2]};
      double pf[3], u0[2], u1[2];
      se3_apply(T, p, pf);
      project2d(fb, u0);
      project2d(pf, u1);
      double e[2] = {u0[0] - u1[0], u0[1] - u1[1]};
      const double k = 1.0 / (double)(1 << a.level[base + i]);
      e[0] *= k;
      e[1] *= k;
      d.err[i] = (float)norm2(e);
      ++cnt;
    }
    if (cnt) atomicAdd(&s.n_err, cnt);
  }
  __syncthreads();
  const int n_err = s.n_err;
  if (n_err == 0) {  // errors.empty(): return before touching anything (:57-58)
    if (tid == 0) a.ran[b] = 0;
    return;
  }
  rank_select<float>(d.err, n, n_err / 2, &s.median_f);
  __syncthreads();
  const double estimated_scale = (double)(1.48f * s.median_f);  // MADScaleEstimator::compute
  if (tid == 0) s.scale = estimated_scale;

  if (tid < 28) s.acc[tid] = 0.0;
  // ---- Gauss-Newton (:66-121) -------------------------------------------------------
  for (int iter = 0; iter < a.n_iter; ++iter) {
    if (tid == 0 && iter == 5) s.scale = 0.85 / focal;
    if (tid < 28) s.acc[tid] = 0.0;
    __syncthreads();
    const Se3 T = s.T;
    const double scale = s.scale;
    for (int c0 = 0; c0 < n; c0 += PO_BLOCK) {
      const int i = c0 + tid;
      double A21[21], bb[6], c2 = 0.0;
#pragma unroll
      for (int k = 0; k < 21; ++k) A21[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) bb[k] = 0.0;
      bool live = false;
      if (i < n && d.hp[i]) {
        live = true;
        const double p[3] = {a.pos[3 * (base + i)], a.pos[3 * (base + i) + 1], a.pos[3 * (base + i) + 2]};
        const double fb[3] = {a.f[3 * (base + i)], a.f[3 * (base + i) + 1], a.f[3 * (base + i) + 2]};
        double J[12], xyz_f[3], u0[2], u1[2];
        se3_apply(T, p, xyz_f);
        frame_jacobian_xyz2uv(xyz_f, J);
        project2d(fb, u0);
        project2d(xyz_f, u1);
        double e[2] = {u0[0] - u1[0], u0[1] - u1[1]};
        const double sqrt_inv_cov = 1.0 / (double)(1 << a.level[base + i]);
        e[0] *= sqrt_inv_cov;
        e[1] *= sqrt_inv_cov;
        const double e2 = e[0] * e[0] + e[1] * e[1];
        if (iter == 0) d.vals[i] = e2;  // chi2_vec_init
#pragma unroll
        for (int k = 0; k < 12; ++k) J[k] *= sqrt_inv_cov;
        const double weight = (double)tukey_weight((float)(norm2(e) / scale));
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) A21[q++] = (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
#pragma unroll
        for (int r = 0; r < 6; ++r) bb[r] = (J[r] * e[0] + J[6 + r] * e[1]) * weight;
        c2 = e2 * weight;
      } else if (iter == 0 && i < n) {
        d.vals[i] = INFINITY;
      }
      {
        const unsigned long long mk = __ballot(live);
        if ((tid & 63) == 0) s.live[tid >> 6] = mk;
      }
      __syncthreads();  // s.live visible
      for (int half = 0; half < 2; ++half) {
        if ((tid >> 5) == half) {
          const int t = tid & 31;
#pragma unroll
          for (int k = 0; k < 21; ++k) s.tile[k][t] = A21[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) s.tile[21 + k][t] = bb[k];
          s.tile[27][t] = c2;
        }
        __syncthreads();
        if (tid < 28) {
          // A += ..., b -= ..., new_chi2 += ...   in observation order.  Observations without a
          // point are skipped (`continue` in the reference), here by a select so that the LDS
          // reads of eight elements are in flight together instead of one per branch.
          int m = n - c0 - PO_HALF * half;
          m = m < 0 ? 0 : (m > PO_HALF ? PO_HALF : m);
          double acc = s.acc[tid];
          const bool neg = (tid >= 21 && tid < 27);
          const double* row = s.tile[tid];
          const unsigned long long lv = s.live[0] >> (PO_HALF * half);
          for (int j0 = 0; j0 < m; j0 += 8) {
            const unsigned long long mk = lv >> j0;
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[(j0 + u) < PO_HALF ? (j0 + u) : 0];
            // only the additions form the dependent chain: sign and skip are folded into the
            // addend (a - b == a + (-b); adding +0.0 to a sum that is never -0.0 changes nothing)
            double addend[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const bool on = ((mk >> u) & 1ull) != 0 && (j0 + u) < m;
              addend[u] = on ? (neg ? -v[u] : v[u]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += addend[u];
          }
          s.acc[tid] = acc;
        }
        __syncthreads();
      }
    }
    if (tid == 0) {
      double A[36], bv[6], dT[6];
      int q = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
          A[r * 6 + c] = s.acc[q];
          A[c * 6 + r] = s.acc[q];
          ++q;
        }
      for (int r = 0; r < 6; ++r) bv[r] = s.acc[21 + r];
      const double new_chi2 = s.acc[27];
      ldlt_solve_pivoted<6>(A, bv, dT);
#pragma unroll
      for (int k = 0; k < 6; ++k) s.dT[k] = dT[k];
      s.new_chi2 = new_chi2;
    }
    __syncthreads();
    // sin/cos of theta/2 (lane 0) and theta (lane 1) in one evaluation
    double sn, cs;
    {
      const double dTl[6] = {s.dT[0], s.dT[1], s.dT[2], s.dT[3], s.dT[4], s.dT[5]};
      const double theta = se3_exp_theta(dTl);
      sincos((tid & 1) ? theta : 0.5 * theta, &sn, &cs);
    }
    const double sin_half = readlane_f64<0>(sn), cos_half = readlane_f64<0>(cs);
    const double sin_theta = readlane_f64<1>(sn), cos_theta = readlane_f64<1>(cs);
    if (tid == 0) {
      const double dT[6] = {s.dT[0], s.dT[1], s.dT[2], s.dT[3], s.dT[4], s.dT[5]};
      const double new_chi2 = s.new_chi2;
      if ((iter > 0 && new_chi2 > s.chi2) || isnan(dT[0])) {
        s.T = s.T_old;  // roll-back
        s.flag = 1;
      } else {
        const Se3 ex = se3_exp_full(dT, sin_half, cos_half, sin_theta, cos_theta);
        const Se3 T_new = se3_compose(ex, s.T);
        s.T_old = s.T;
        s.T = T_new;
        s.chi2 = new_chi2;
        double nm = -1;
        for (int k = 0; k < 6; ++k)
          if (fabs(dT[k]) > nm) nm = fabs(dT[k]);
        if (nm <= SVO_EPS) s.flag = 1;
      }
    }
    __syncthreads();
    if (s.flag) break;
  }
  __syncthreads();
  // ---- covariance (:124-126) --------------------------------------------------------
  // (A f^2)^-1 column by column through the pivoted LDLT: lane j solves for e_j, so the six
  // solves run side by side (every lane factors the same matrix: same instructions, same bits)
  if (a.Cov) {
    double Af[36];
    {
      const double f2 = focal * focal;  // std::pow(f, 2)
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) {
          const double v = s.acc[q] * f2;
          Af[r * 6 + c] = v;
          Af[c * 6 + r] = v;
          ++q;
        }
    }
    LdltReg<6> fac;
    fac.factor(Af);
    double e[6], x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = (i == tid) ? 1.0 : 0.0;
    fac.solve(e, x);
    if (tid < 6) {
#pragma unroll
      for (int i = 0; i < 6; ++i) a.Cov[36 * b + i * 6 + tid] = 1.0 * x[i];
    }
  }
  if (tid == 0) se3_to_Rt(s.T, a.T + 12 * b);
  // chi2_vec_init median before vals is overwritten
  rank_select<double>(d.vals, n, n_err / 2, &s.median_d);
  __syncthreads();
  const double med_init = s.median_d;
  __syncthreads();
  // ---- prune outliers (:128-145) ----------------------------------------------------
  const double reproj_thresh_scaled = a.reproj_thresh / focal;
  {
    const Se3 T = s.T;
    if (tid == 0) s.n_err = 0;
    __syncthreads();
    int deleted = 0;
    for (int i = tid; i < n; i += PO_BLOCK) {
      d.err[i] = 0.f;
      d.vals[i] = INFINITY;
      if (!d.hp[i]) continue;
      const double p[3] = {a.pos[3 * (base + i)], a.pos[3 * (base + i) + 1], a.pos[3 * (base + i) + 2]};
      const double fb[3] = {a.f[3 * (base + i)], a.f[3 * (base + i) + 1], a.f[3 * (base + i) + 2]};
      double pf[3], u0[2], u1[2];
      se3_apply(T, p, pf);
      project2d(fb, u0);
      project2d(pf, u1);
      double e[2] = {u0[0] - u1[0], u0[1] - u1[1]};
      const double k = 1.0 / (double)(1 << a.level[base + i]);
      e[0] *= k;
      e[1] *= k;
      d.vals[i] = e[0] * e[0] + e[1] * e[1];  // chi2_vec_final
      d.err[i] = 1.f;                         // member of chi2_vec_final
      if (norm2(e) > reproj_thresh_scaled) {
        d.err[i] = 2.f;  // pruned after the median is taken
        ++deleted;
      }
    }
    if (deleted) atomicAdd(&s.n_err, deleted);
  }
  __syncthreads();
  rank_select<double>(d.vals, n, n_err / 2, &s.median_d);
  __syncthreads();
  const int n_deleted = s.n_err;
  for (int i = tid; i < n; i += PO_BLOCK)
    if (d.err[i] == 2.f) a.has_point[base + i] = 0;  // (*it)->point = NULL
  if (tid == 0) {
    a.stats[4 * b + 0] = estimated_scale * focal;
    a.stats[4 * b + 1] = a.n_iter > 0 ? sqrt(med_init) * focal : 0.0;  // chi2_vec_init empty without an iteration
    a.stats[4 * b + 2] = sqrt(s.median_d) * focal;
    a.stats[4 * b + 3] = (double)(n_err - n_deleted);
    a.ran[b] = 1;
  }
}

}  // namespace

static int pose_args_check(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f,
                           const int32_t* d_level, const double* d_pos, uint8_t* d_has_point, int n_iter,
                           double* d_T_f_w, double* d_stats, int32_t* d_ran) {
  if (!cam || !cam_model_ok(cam) || B < 0 || n_stride < 1 || n_iter < 0) return SVO_HIP_EINVAL;
  if (n_stride > PO_MAXN) return SVO_HIP_ERANGE;
  if (B == 0) return 1;
  if (!d_n || !d_f || !d_level || !d_pos || !d_has_point || !d_T_f_w || !d_stats || !d_ran) return SVO_HIP_EINVAL;
  return SVO_HIP_OK;
}

static int launch_ordered(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f,
                          const int32_t* d_level, const double* d_pos, uint8_t* d_has_point, double reproj_thresh,
                          int n_iter, double* d_T_f_w, double* d_Cov, double* d_stats, int32_t* d_ran, int only_flagged,
                          void* stream) {
  PoseArgs a;
  a.cam = make_cam(cam);
  a.n = d_n;
  a.n_stride = n_stride;
  a.f = d_f;
  a.level = d_level;
  a.pos = d_pos;
  a.has_point = d_has_point;
  a.reproj_thresh = reproj_thresh;
  a.n_iter = n_iter;
  a.T = d_T_f_w;
  a.Cov = d_Cov;
  a.stats = d_stats;
  a.ran = d_ran;
  a.only_flagged = only_flagged;
  const int ns8 = (n_stride + 7) & ~7;
  const size_t dyn = (size_t)ns8 * (sizeof(double) + sizeof(float) + 1);
  hipLaunchKernelGGL(pose_opt_kernel, dim3(B), dim3(PO_BLOCK), dyn, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

// finish: also run the ordered kernel on the frames the wave kernel flags (ran = 2)
static int pose_optimize_impl(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f,
                              const int32_t* d_level, const double* d_pos, uint8_t* d_has_point, double reproj_thresh,
                              int n_iter, double* d_T_f_w, double* d_Cov, double* d_stats, int32_t* d_ran, bool finish,
                              void* stream) {
  const int rc = pose_args_check(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, n_iter, d_T_f_w, d_stats, d_ran);
  if (rc != SVO_HIP_OK) return rc > 0 ? SVO_HIP_OK : rc;
  if (n_stride > svo_track::POSE_WAVE_MAX_STRIDE)  // more than 4 observations per lane: ordered kernel
    return launch_ordered(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T_f_w, d_Cov,
                          d_stats, d_ran, 0, stream);
  svo_track::PoseWaveArgs w;
  w.cam = *cam;
  w.B = B;
  w.n = d_n;
  w.n_stride = n_stride;
  w.f = d_f;
  w.level = d_level;
  w.pos = d_pos;
  w.has_point = d_has_point;
  w.reproj_thresh = reproj_thresh;
  w.n_iter = n_iter;
  w.T = d_T_f_w;
  w.Cov = d_Cov;
  w.stats = d_stats;
  w.ran = d_ran;
  const int r2 = svo_track::launch_pose_wave(w, static_cast<hipStream_t>(stream));
  if (r2 != SVO_HIP_OK || !finish) return r2;
  // frames whose normal equations are (nearly) singular were left untouched and flagged ran = 2:
  // the ordered kernel takes exactly those (its other workgroups exit at once)
  return launch_ordered(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T_f_w, d_Cov,
                        d_stats, d_ran, 1, stream);
}

extern "C" int svo_hip_pose_optimize(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride,
                                     const double* d_f, const int32_t* d_level, const double* d_pos,
                                     uint8_t* d_has_point, double reproj_thresh, int n_iter, double* d_T_f_w,
                                     double* d_Cov, double* d_stats, int32_t* d_ran, void* stream) {
  return pose_optimize_impl(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T_f_w, d_Cov,
                            d_stats, d_ran, true, stream);
}

extern "C" int svo_hip_pose_optimize_deferred(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride,
                                              const double* d_f, const int32_t* d_level, const double* d_pos,
                                              uint8_t* d_has_point, double reproj_thresh, int n_iter, double* d_T_f_w,
                                              double* d_Cov, double* d_stats, int32_t* d_ran, void* stream) {
  return pose_optimize_impl(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T_f_w, d_Cov,
                            d_stats, d_ran, false, stream);
}

extern "C" int svo_hip_pose_optimize_ordered(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride,
                                             const double* d_f, const int32_t* d_level, const double* d_pos,
                                             uint8_t* d_has_point, double reproj_thresh, int n_iter, double* d_T_f_w,
                                             double* d_Cov, double* d_stats, int32_t* d_ran, void* stream) {
  const int rc = pose_args_check(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, n_iter, d_T_f_w, d_stats, d_ran);
  if (rc != SVO_HIP_OK) return rc > 0 ? SVO_HIP_OK : rc;
  return launch_ordered(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T_f_w, d_Cov,
                        d_stats, d_ran, 0, stream);
}
