// pose_optimizer_wave.hip -- K4 (default form): batched pose_optimizer::optimizeGaussNewton, one
// WAVE per frame, several frames per workgroup, no workgroup barrier and no LDS.
//
// Replaces svo::pose_optimizer::optimizeGaussNewton (svo/src/pose_optimizer.cpp:28-161) to a
// stated tolerance (pose within 1e-9 of the reference, SE(3) log-map norm; pruning decisions on
// the residuals of that pose: e2 > thresh^2).  The bit-ordered variant that reproduces the
// reference's summation order (pose_optimizer.hip, svo_hip_pose_optimize_ordered) stays as the
// checker; this kernel is what the pipeline runs:
//
//   * a lane keeps up to 4 observations (point, normalised measurement, level scale) in
//     registers for the whole call: nothing is re-read from HBM between Gauss-Newton iterations;
//   * the 27 non-zero f64 sums of an iteration (20 entries of A -- A(0,1) is identically zero,
//     b[6], chi2) are accumulated per lane and reduced over the wave with a TRANSPOSING
//     butterfly (v_permlane32_swap / v_permlane16_swap / DPP): 32 f64 additions instead of
//     27 x 6, no LDS round trip, no ordered chain;
//   * the 6x6 solve, SE3::exp(dT)*T and the stop / rollback rules run redundantly in every lane
//     on wave-uniform values (readlane broadcasts): no serial lane, no barrier.  The solve is an
//     unpivoted LDL^T; only when a pivot collapses (rank-deficient systems of a handful of
//     observations) the wave falls back to Eigen's pivoted algorithm, as the ordered kernel uses;
//   * MAD scale and the two reported medians are exact order statistics found by a bitwise
//     radix select over the wave (ballot + popcount per bit), not by O(n^2) rank counting.
#include "track_kernels.h"
#include "track_math.h"
#include "wave_reduce.h"

using namespace svo_capi;
using namespace svo_dev;

namespace {

using svo_track::PoseWaveArgs;

constexpr int PW_WAVES = 4;  // frames per workgroup
constexpr int PW_MINW = 3;  // waves per SIMD asked of the register allocator (<= 168 VGPRs)
constexpr double SVO_EPS = 0.0000000001;  // svo/include/svo/global.h:77


// A wave-uniform value computed by the VALU lives in a VGPR pair; reading it back through
// v_readfirstlane moves it to SGPRs.  The pose (R, t, quaternion, rollback copy) is 38 doubles:
// kept in VGPRs it would cost 76 registers per lane next to the resident observations.
__device__ __forceinline__ double uni(double v) {
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo32(v));
  const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi32(v));
  return mk_f64(l, h);
}
__device__ __forceinline__ void uni_se3(Se3& T) {
#pragma unroll
  for (int k = 0; k < 4; ++k) T.q[k] = uni(T.q[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) T.t[k] = uni(T.t[k]);
}


// Sums v[0..31] over the 64 lanes.  On return lanes 2j and 2j+1 hold the wave total of v[j].
__device__ __forceinline__ double wave_reduce32_f64(const double v[32], int lane) {
  double r[16], q[8], t[4], u[2];
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = swap32_add(v[k], v[k + 16]);
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = swap16_add(r[k], r[k + 8]);
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double mine = hi ? q[k + 4] : q[k], send = hi ? q[k] : q[k + 4];
      t[k] = mine + dpp_f64<DPP_ROW_ROR8>(send);
    }
  }
  {
    const bool hi = (lane & 4) != 0;  // partner 7-i inside each 8 lanes has the other bit 2
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double mine = hi ? t[k + 2] : t[k], send = hi ? t[k] : t[k + 2];
      u[k] = mine + dpp_f64<DPP_ROW_HALF_MIRROR>(send);
    }
  }
  const bool hi = (lane & 2) != 0;
  const double mine = hi ? u[1] : u[0], send = hi ? u[0] : u[1];
  double w = mine + dpp_f64<DPP_QUAD_XOR2>(send);
  w += dpp_f64<DPP_QUAD_XOR1>(w);
  return w;
}

__device__ __forceinline__ double fast_rcp(double z) {
  double r = __builtin_amdgcn_rcp(z);
  r = fma(fma(-z, r, 1.0), r, r);
  r = fma(fma(-z, r, 1.0), r, r);
  return r;
}
// sqrt for x >= 0 (v_rsq_f64 seed + two coupled Newton steps), 0 for x == 0
__device__ __forceinline__ double fast_sqrt(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  double g = x * r;
  const double h = 0.5 * r;
  g = fma(fma(-g, g, x), h, g);
  g = fma(fma(-g, g, x), h, g);
  return fmax(g, 0.0);  // x == 0: rsq is +inf and g a NaN, which the maximum drops (one v_max_f64 for a compare and two selects)
}

// ldlt6_factor (device_math.h) with the six pivot reciprocals by v_rcp_f64 + Newton.  A vanished pivot leaves LD[15 + j] = 0
// and a column that means nothing (v * 0: ldlt6_factor keeps v): the only caller hands such a frame to the ordered kernel
// (pivots_ok) before anything reads the factor -- fifteen selects per Gauss-Newton iteration less.
// (Skipping the fourth observation slot of a frame of <= 192 observations behind a scalar branch: five more 8-byte spills,
// 0.232 -> 0.235 ms, profiles/r06ag_*.)
__device__ __forceinline__ void ldlt6_factor_fast(const double H[21], double LD[21]) {
  double d[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dj = H[sym6(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) dj -= LD[low6(j, k)] * LD[low6(j, k)] * d[k];
    d[j] = dj;
    const bool ok = fabs(dj) > 2.2250738585072014e-308;
    const double inv = ok ? fast_rcp(dj) : 0.0;
    LD[15 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = H[sym6(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= LD[low6(i, k)] * LD[low6(j, k)] * d[k];
      LD[low6(i, j)] = v * inv;
    }
  }
}

// Sophus SE3::exp (device_math.h se3_exp) with one reciprocal instead of three divisions
__device__ __forceinline__ void se3_exp_fast(const double xi[6], double q[4], double t[3]) {
  const double ox = xi[3], oy = xi[4], oz = xi[5];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  const double theta = fast_sqrt(theta_sq);
  double s_h, c_h;
  sincos_small(0.5 * theta, &s_h, &c_h);
  const double ux = xi[0], uy = xi[1], uz = xi[2];
  if (theta < 1e-10) {
    const double theta_po4 = theta_sq * theta_sq;
    const double imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    q[0] = c_h; q[1] = imag_factor * ox; q[2] = imag_factor * oy; q[3] = imag_factor * oz;
    quat_rot(q, xi, t);
  } else {
    const double it = fast_rcp(theta);
    const double imag_factor = s_h * it;
    q[0] = c_h; q[1] = imag_factor * ox; q[2] = imag_factor * oy; q[3] = imag_factor * oz;
    const double it2 = it * it;
    const double s_t = 2.0 * s_h * c_h;
    const double c1 = (2.0 * s_h * s_h) * it2;
    const double c2 = (theta - s_t) * (it2 * it);
    const double wx = oy * uz - oz * uy, wy = oz * ux - ox * uz, wz = ox * uy - oy * ux;
    const double wwx = oy * wz - oz * wy, wwy = oz * wx - ox * wz, wwz = ox * wy - oy * wx;
    t[0] = ux + c1 * wx + c2 * wwx;
    t[1] = uy + c1 * wy + c2 * wwy;
    t[2] = uz + c1 * wz + c2 * wwz;
  }
}

// Sophus SE3::operator* with the renormalisation by v_rsq_f64 + Newton
__device__ __forceinline__ Se3 se3_compose_fast(const Se3& a, const Se3& b) {
  Se3 r;
  double rt[3];
  quat_rot(a.q, b.t, rt);
  r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
  quat_mul(a.q, b.q, r.q);
  quat_normalize_fast(r.q);
  return r;
}

// vk::robust_cost::TukeyWeightFunction::value
__device__ __forceinline__ float tukey_w(float x) {
  const float b_square = 4.6851f * 4.6851f;
  const float x_square = x * x;
  const float tmp = 1.0f - x_square / b_square;
  return (x_square <= b_square) ? tmp * tmp : 0.0f;
}

// Exact order statistic: the value of rank k (0-based) among the wave's NPL*64 keys.  Keys are
// bit patterns of non-negative floats/doubles (order-isomorphic to unsigned integers);
// non-participants hold +inf.  One ballot + popcount per key and bit; the walk down the bits ends as
// soon as the interval [prefix, prefix + 2^(bit+1)) holds a single key, which is then fetched from
// its owner lane (distinct values: ~20 of the 31 / 63 steps).
template <int NPL, typename K>
__device__ __forceinline__ K radix_select(const K key[NPL], int k, int top_bit) {
  K prefix = 0;
  int cnt_lo = 0, cnt_hi = NPL * 64;  // #keys < prefix, #keys < prefix + 2^(bit+1)
  for (int bit = top_bit; bit >= 0; --bit) {
    const K cand = prefix | ((K)1 << bit);
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) cnt += __popcll(__ballot(key[j] < cand));  // (64-bit keys compared as doubles instead: no difference, profiles/r06al_*)
    if (cnt <= k) {
      prefix = cand;
      cnt_lo = cnt;
    } else {
      cnt_hi = cnt;
    }
    if (cnt_hi - cnt_lo == 1 && bit > 0) {
      const K upper = prefix + ((K)1 << bit);  // exclusive; bit is the width still undecided
      K found = prefix;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const unsigned long long m = __ballot(key[j] >= prefix && key[j] < upper);
        if (m) {
          const int src = __ffsll((long long)m) - 1;
          if (sizeof(K) == 8) {
            const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)key[j], src);
            const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)key[j] >> 32), src);
            found = (K)(((unsigned long long)hi << 32) | lo);
          } else {
            found = (K)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)key[j], src);
          }
        }
      }
      return found;
    }
  }
  return prefix;
}

// every pivot of the unpivoted LDL^T (LD[15+k] = 1/d_k, 0 where d_k vanished) above 1e-9 of the
// largest diagonal entry
__device__ __forceinline__ bool pivots_ok(const double A[21], const double LD[21]) {
  double dmax = 0.0, imax = 0.0;
  bool zero = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    dmax = fmax(dmax, fabs(A[sym6(k, k)]));
    const double id = fabs(LD[15 + k]);
    zero = zero || !(id > 0.0);
    imax = fmax(imax, id);
  }
  // d_min = 1/imax > 1e-9 * dmax
  return !zero && (imax * dmax * 1e-9 < 1.0);
}

// broadcast of a runtime-indexed pair of lanes is not needed: indices are compile-time
template <int J>
__device__ __forceinline__ double bcast(double v) { return readlane_f64<2 * J>(v); }

// The 27 sums in reduction order:
//   0..4   A(0,0) A(0,2) A(0,3) A(0,4) A(0,5)
//   5..9   A(1,1) A(1,2) A(1,3) A(1,4) A(1,5)
//   10..19 A(2,2) A(2,3) A(2,4) A(2,5) A(3,3) A(3,4) A(3,5) A(4,4) A(4,5) A(5,5)
//   20..25 b[0..5]   26 chi2
template <int NPL>
__global__ void __launch_bounds__(64 * PW_WAVES, PW_MINW) pose_opt_wave_kernel(const PoseWaveArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * PW_WAVES + (threadIdx.x >> 6);
  if (b >= a.B) return;
  int n = a.n[b];
  n = n < 0 ? 0 : (n > a.n_stride ? a.n_stride : n);  // contract: n <= n_stride (svo_hip.h); never read past the row
  const size_t base = (size_t)b * a.n_stride;
  const double focal = fabs(a.cam.fx);

  // ---- this lane's observations, resident for the whole call ---------------------------
  double px[NPL], py[NPL], pz[NPL], ux[NPL], uy[NPL];
  float kk[NPL];
  bool live[NPL];
  int n_err = 0;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int i = j * 64 + lane;
    live[j] = (i < n) && a.has_point[base + i] != 0;
    px[j] = py[j] = 0.0;
    pz[j] = 1.0;
    ux[j] = uy[j] = 0.0;
    kk[j] = 1.f;
    if (live[j]) {
      const double* p = a.pos + 3 * (base + i);
      const double* f = a.f + 3 * (base + i);
      px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
      ux[j] = f[0] / f[2];  // vk::project2d(f), once
      uy[j] = f[1] / f[2];
      kk[j] = pow2_inv_f32(a.level[base + i]);  // == 1.0f / (float)(1 << level), from exponent bits
    }
    n_err += __popcll(__ballot(live[j]));
  }
  if (n_err == 0) {  // errors.empty(): return before touching anything (:57-58)
    if (lane == 0) a.ran[b] = 0;
    return;
  }

  // model, wave-uniform (every lane loads the same 12 doubles)
  Se3 T, T_old;
  se3_from_Rt(a.T + 12 * b, T);
  uni_se3(T);
  T_old = T;
  double R[9];
  quat_to_R(T.q, R);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = uni(R[k]);

  // ---- error scale (:45-60) and the median of chi2_vec_init (same residuals as iteration 0) ----
  double estimated_scale, med_init;
  {
    unsigned long long kd[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const double x = R[0] * px[j] + R[1] * py[j] + R[2] * pz[j] + T.t[0];
      const double y = R[3] * px[j] + R[4] * py[j] + R[5] * pz[j] + T.t[1];
      const double z = R[6] * px[j] + R[7] * py[j] + R[8] * pz[j] + T.t[2];
      const double k = (double)kk[j];
      const double zi = fast_rcp(z);  // (as in the iterations below: the same residuals as iteration 0)
      const double e0 = (ux[j] - x * zi) * k, e1 = (uy[j] - y * zi) * k;
      const double e2 = e0 * e0 + e1 * e1;
      kd[j] = live[j] ? (unsigned long long)__double_as_longlong(e2) : 0x7ff0000000000000ull;
    }
    med_init = __longlong_as_double((long long)radix_select<NPL, unsigned long long>(kd, n_err / 2, 62));
    // The scale estimator's median is over errors = (float)sqrt(e2) of the same observations (:52-56), rank n / 2 as well:
    // sqrt and the conversion are monotone, so the rank-k error IS the image of the rank-k e2 -- one selection, not two
    // (equal images of different e2 are the same value).
    const float median_f = (float)sqrt(med_init);
    estimated_scale = (double)(1.48f * median_f);  // MADScaleEstimator::compute
  }

  // ---- Gauss-Newton (:66-121) ----------------------------------------------------------
  double scale = estimated_scale;
  double chi2 = 0.0;
  if (a.n_iter == 0) {  // Cov_ of a never-formed A: the ordered kernel reproduces what the reference leaves behind
    if (lane == 0) a.ran[b] = 2;
    return;
  }
  for (int iter = 0; iter < a.n_iter; ++iter) {
    if (iter == 5) scale = 0.85 / focal;
    const double inv_scale = 1.0 / scale;
    double A[21];  // packed upper triangle of this iteration's normal matrix (sym6 order)
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const double x = R[0] * px[j] + R[1] * py[j] + R[2] * pz[j] + T.t[0];
      const double y = R[3] * px[j] + R[4] * py[j] + R[5] * pz[j] + T.t[1];
      const double z = R[6] * px[j] + R[7] * py[j] + R[8] * pz[j] + T.t[2];
      const double zi = fast_rcp(z);
      const double xn = x * zi, yn = y * zi;
      const double k = (double)kk[j];
      const double e0 = (ux[j] - xn) * k, e1 = (uy[j] - yn) * k;  // e *= sqrt_inv_cov
      const double e2 = e0 * e0 + e1 * e1;
      const float wf = tukey_w((float)(fast_sqrt(e2) * inv_scale));
      const double w = live[j] ? (double)wf : 0.0;
      // Frame::jacobian_xyz2uv (frame.h:116-138), rows J0 / J1 scaled by sqrt_inv_cov
      const double j00 = -zi * k, j02 = xn * zi * k, j03 = xn * yn * k, j04 = -(1.0 + xn * xn) * k, j05 = yn * k;
      const double j11 = -zi * k, j12 = yn * zi * k, j13 = (1.0 + yn * yn) * k, j14 = -j03, j15 = -xn * k;
      const double w00 = w * j00, w02 = w * j02, w03 = w * j03, w04 = w * j04, w05 = w * j05;
      const double w11 = w * j11, w12 = w * j12, w13 = w * j13, w14 = w * j14, w15 = w * j15;
      acc[0] = fma(w00, j00, acc[0]);
      acc[1] = fma(w00, j02, acc[1]);
      acc[2] = fma(w00, j03, acc[2]);
      acc[3] = fma(w00, j04, acc[3]);
      acc[4] = fma(w00, j05, acc[4]);
      acc[5] = fma(w11, j11, acc[5]);
      acc[6] = fma(w11, j12, acc[6]);
      acc[7] = fma(w11, j13, acc[7]);
      acc[8] = fma(w11, j14, acc[8]);
      acc[9] = fma(w11, j15, acc[9]);
      acc[10] = fma(w02, j02, fma(w12, j12, acc[10]));
      acc[11] = fma(w02, j03, fma(w12, j13, acc[11]));
      acc[12] = fma(w02, j04, fma(w12, j14, acc[12]));
      acc[13] = fma(w02, j05, fma(w12, j15, acc[13]));
      acc[14] = fma(w03, j03, fma(w13, j13, acc[14]));
      acc[15] = fma(w03, j04, fma(w13, j14, acc[15]));
      acc[16] = fma(w03, j05, fma(w13, j15, acc[16]));
      acc[17] = fma(w04, j04, fma(w14, j14, acc[17]));
      acc[18] = fma(w04, j05, fma(w14, j15, acc[18]));
      acc[19] = fma(w05, j05, fma(w15, j15, acc[19]));
      // b -= J' e w
      acc[20] = fma(-w00, e0, acc[20]);
      acc[21] = fma(-w11, e1, acc[21]);
      acc[22] = fma(-w02, e0, fma(-w12, e1, acc[22]));
      acc[23] = fma(-w03, e0, fma(-w13, e1, acc[23]));
      acc[24] = fma(-w04, e0, fma(-w14, e1, acc[24]));
      acc[25] = fma(-w05, e0, fma(-w15, e1, acc[25]));
      acc[26] = fma(e2, w, acc[26]);
    }
    const double tot = wave_reduce32_f64(acc, lane);
    const double s0 = bcast<0>(tot), s1 = bcast<1>(tot), s2 = bcast<2>(tot), s3 = bcast<3>(tot), s4 = bcast<4>(tot);
    const double s5 = bcast<5>(tot), s6 = bcast<6>(tot), s7 = bcast<7>(tot), s8 = bcast<8>(tot), s9 = bcast<9>(tot);
    const double s10 = bcast<10>(tot), s11 = bcast<11>(tot), s12 = bcast<12>(tot), s13 = bcast<13>(tot);
    const double s14 = bcast<14>(tot), s15 = bcast<15>(tot), s16 = bcast<16>(tot), s17 = bcast<17>(tot);
    const double s18 = bcast<18>(tot), s19 = bcast<19>(tot);
    const double bv[6] = {bcast<20>(tot), bcast<21>(tot), bcast<22>(tot), bcast<23>(tot), bcast<24>(tot), bcast<25>(tot)};
    const double new_chi2 = bcast<26>(tot);
    A[sym6(0, 0)] = s0; A[sym6(0, 1)] = 0.0; A[sym6(0, 2)] = s1; A[sym6(0, 3)] = s2; A[sym6(0, 4)] = s3; A[sym6(0, 5)] = s4;
    A[sym6(1, 1)] = s5; A[sym6(1, 2)] = s6; A[sym6(1, 3)] = s7; A[sym6(1, 4)] = s8; A[sym6(1, 5)] = s9;
    A[sym6(2, 2)] = s10; A[sym6(2, 3)] = s11; A[sym6(2, 4)] = s12; A[sym6(2, 5)] = s13;
    A[sym6(3, 3)] = s14; A[sym6(3, 4)] = s15; A[sym6(3, 5)] = s16;
    A[sym6(4, 4)] = s17; A[sym6(4, 5)] = s18; A[sym6(5, 5)] = s19;

    // dT = A.ldlt().solve(b) (:97).  Unpivoted LDL^T (stable for the SPD normal equations).  A
    // collapsed pivot means a (nearly) rank-deficient system -- fewer observations than degrees
    // of freedom -- where the reference's answer is decided by the rounding of its own ordered
    // sums inside Eigen's pivoted algorithm: such a frame is handed to the ordered kernel
    // untouched (ran = 2, nothing has been written yet).
    double dT[6];
    double LD[21];
    ldlt6_factor_fast(A, LD);
    if (!pivots_ok(A, LD)) {
      if (lane == 0) a.ran[b] = 2;
      return;
    }
    ldlt6_solve(LD, bv, dT);

    bool last = false;
    if ((iter > 0 && new_chi2 > chi2) || isnan(dT[0])) {  // check if error increased (:100-107)
      T = T_old;  // roll-back
      last = true;
    } else {
      // update the model: T_new = SE3::exp(dT) * T  (:110)
      Se3 ex;
      se3_exp_fast(dT, ex.q, ex.t);
      T_old = T;
      T = se3_compose_fast(ex, T);
      uni_se3(T);
      chi2 = new_chi2;
      double nm = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) nm = fmax(nm, fabs(dT[k]));
      last = nm <= SVO_EPS;  // stop when converged (:120)
    }
    quat_to_R(T.q, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = uni(R[k]);
    last = last || iter + 1 == a.n_iter;
    if (last) {
      // covariance (:124-126): (A f^2)^-1 = A^-1 / f^2 with the A of the last evaluated iteration,
      // from the factor that is still in registers; lane j (< 6) solves for column j
      if (a.Cov) {
        double e[6], x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) e[i] = (i == lane) ? 1.0 : 0.0;
        ldlt6_solve(LD, e, x);
        const double if2 = 1.0 / (focal * focal);
        if (lane < 6) {
#pragma unroll
          for (int i = 0; i < 6; ++i) a.Cov[36 * b + i * 6 + lane] = x[i] * if2;
        }
      }
      break;
    }
  }
  if (lane == 0) se3_to_Rt(T, a.T + 12 * b);

  // ---- prune outliers (:128-145) and the median of chi2_vec_final ---------------------------
  const double reproj_thresh_scaled = a.reproj_thresh / focal;
  const double reproj_thresh2 = reproj_thresh_scaled * reproj_thresh_scaled;
  int n_deleted = 0;
  double med_final;
  {
    unsigned long long kd[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const double x = R[0] * px[j] + R[1] * py[j] + R[2] * pz[j] + T.t[0];
      const double y = R[3] * px[j] + R[4] * py[j] + R[5] * pz[j] + T.t[1];
      const double z = R[6] * px[j] + R[7] * py[j] + R[8] * pz[j] + T.t[2];
      const double k = (double)kk[j];
      const double zi = fast_rcp(z);
      const double e0 = (ux[j] - x * zi) * k, e1 = (uy[j] - y * zi) * k;
      const double e2 = e0 * e0 + e1 * e1;
      kd[j] = live[j] ? (unsigned long long)__double_as_longlong(e2) : 0x7ff0000000000000ull;
      // e.norm() > reproj_thresh_scaled (:136) as e2 > thresh^2: the pose this kernel arrives at differs from the reference's
      // by up to 1e-9, which moves e2 by far more than the rounding of a division or a square root -- no decision hangs on them
      const bool prune = live[j] && e2 > reproj_thresh2;
      if (prune) a.has_point[base + j * 64 + lane] = 0;  // (*it)->point = NULL
      n_deleted += __popcll(__ballot(prune));
    }
    med_final = __longlong_as_double((long long)radix_select<NPL, unsigned long long>(kd, n_err / 2, 62));
  }
  if (lane == 0) {
    a.stats[4 * b + 0] = estimated_scale * focal;
    a.stats[4 * b + 1] = a.n_iter > 0 ? sqrt(med_init) * focal : 0.0;  // chi2_vec_init is empty without an iteration
    a.stats[4 * b + 2] = sqrt(med_final) * focal;
    a.stats[4 * b + 3] = (double)(n_err - n_deleted);
    a.ran[b] = 1;
  }
}

}  // namespace

namespace svo_track {

// n_stride <= 256 only (4 observations per lane); the caller falls back to the ordered kernel beyond
int launch_pose_wave(const PoseWaveArgs& a, hipStream_t s) {
  const int grid = (a.B + PW_WAVES - 1) / PW_WAVES;
  const dim3 g(grid), blk(64 * PW_WAVES);
  if (a.n_stride <= 64) hipLaunchKernelGGL(pose_opt_wave_kernel<1>, g, blk, 0, s, a);
  else if (a.n_stride <= 128) hipLaunchKernelGGL(pose_opt_wave_kernel<2>, g, blk, 0, s, a);
  else hipLaunchKernelGGL(pose_opt_wave_kernel<4>, g, blk, 0, s, a);
  return check_launch();
}

}  // namespace svo_track
