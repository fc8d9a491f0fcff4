// device_math_on_host.cpp -- TEST INFRASTRUCTURE.  The device math of the kernels (rpg_svo_amd/csrc/device_math.h,
// track_math.h, matcher_device.h) compiled by g++ for the CPU (SVO_HOST_MATH_TEST: the headers drop the HIP include and
// whatever needs the wave), behind a C interface tests/test_device_math_host.py loads with ctypes and checks against the
// oracle.  What this shows is that the FORMULAS the kernels run are the reference's; the bits the GPU produces (fused
// multiply-adds where a translation unit allows contraction) are the GPU parity tests' business.
#define SVO_HOST_MATH_TEST
#include "matcher_device.h"
#include "seed_math.h"
#include "align_lanes.h"
#include "warp_sample.h"

using namespace svo_dev;

namespace {
Cam make_cam(const double* k /* fx fy cx cy */, int width, int height, int model, const double* d) {
  Cam c;
  c.fx = k[0]; c.fy = k[1]; c.cx = k[2]; c.cy = k[3];
  c.width = width; c.height = height; c.model = model;
  for (int i = 0; i < 5; ++i) c.d[i] = d[i];
  return c;
}
}  // namespace

extern "C" {

// Sophus SE3::exp as the kernels run it (quaternion + translation), returned as [R row-major | t]
void hm_se3_exp(const double xi[6], double T_out[12]) {
  Se3 s;
  se3_exp(xi, s.q, s.t);
  se3_to_Rt(s, T_out);
}
// the f32 variant of K1's solver (short series below |omega| < 0.01 unless long_only)
void hm_se3_exp_f32(const float xi[6], float q[4], float t[3]) { se3_exp_f32(xi, q, t); }

void hm_se3_mul(const double A[12], const double B[12], double out[12]) {
  Se3 a, b;
  se3_from_Rt(A, a);
  se3_from_Rt(B, b);
  se3_to_Rt(se3_compose(a, b), out);
}
void hm_se3_inv(const double A[12], double out[12]) {
  Se3 a;
  se3_from_Rt(A, a);
  se3_to_Rt(se3_inverse(a), out);
}
void hm_frame_pos(const double T_f_w[12], double p[3]) {
  Se3 a;
  se3_from_Rt(T_f_w, a);
  frame_pos(a, p);
}
void hm_quat_round_trip(const double R[9], double R_out[9]) {
  double q[4];
  quat_from_R(R, q);
  quat_to_R(q, R_out);
}

void hm_world2cam(const double k[4], int width, int height, int model, const double d[5], const double xyz[3], double px[2]) {
  const Cam c = make_cam(k, width, height, model, d);
  world2cam(c, xyz, px);
}
void hm_cam2world(const double k[4], int width, int height, int model, const double d[5], const double px[2], double f[3]) {
  const Cam c = make_cam(k, width, height, model, d);
  cam2world(c, px[0], px[1], f);
}

void hm_warp_matrix_affine(const double k[4], int width, int height, int model, const double d[5], const double px_ref[2],
                           const double f_ref[3], double depth_ref, const double T_cur_ref[12], int level_ref, double A[4]) {
  const Cam c = make_cam(k, width, height, model, d);
  Se3 T;
  se3_from_Rt(T_cur_ref, T);
  svo_track::warp_matrix_affine(c, px_ref, f_ref, depth_ref, T, level_ref, A);
}
int hm_best_search_level(const double A[4], int max_level) { return svo_track::best_search_level(A, max_level); }

// Eigen's LDLT with pivoting as the pose optimizer's kernels run it (6 x 6)
void hm_ldlt6_solve_pivoted(const double A[36], const double b[6], double x[6]) { ldlt_solve_pivoted<6>(A, b, x); }
// the unpivoted packed LDLT of K1's solver: H as the packed upper triangle (21 values, row-major, i <= j)
void hm_ldlt6_solve_packed(const double H21[21], const double b[6], double x[6]) {
  double LD[21];
  ldlt6_factor(H21, LD);
  ldlt6_solve(LD, b, x);
}
void hm_inv3f(const float m[9], float r[9]) { inv3f(m, r); }
int hm_floor_to_int(float x) { return floor_to_int(x); }
void hm_sincos_small(double x, double* s, double* c) { sincos_small(x, s, c); }

// ---- the depth filter's closed-form pieces (seed_math.h) ------------------------------------------------------------
int hm_depth_from_triangulation(const double T_search_ref[12], const double f_ref[3], const double f_cur[3], double* depth) {
  Se3 T;
  se3_from_Rt(T_search_ref, T);
  return svo_track::depth_from_triangulation(T, f_ref, f_cur, depth) ? 1 : 0;
}
// state = {a, b, mu, sigma2} in and out
void hm_update_seed(float x, float tau2, float z_range, float state[4]) {
  svo_track::update_seed(x, tau2, state[0], state[1], state[2], z_range, state[3]);
}
double hm_compute_tau(const double T_ref_cur[12], const double f[3], double z, double px_error_angle) {
  Se3 T;
  se3_from_Rt(T_ref_cur, T);
  return svo_track::compute_tau(T, f, z, px_error_angle);
}
double hm_compute_tau_algebraic(const double T_ref_cur[12], const double f[3], double z, double px_error_angle) {
  Se3 T;
  se3_from_Rt(T_ref_cur, T);
  return svo_track::compute_tau(T, f, z, svo_track::tau_consts(px_error_angle));
}
float hm_normal_pdf(float x, float mean, float sd) { return svo_track::normal_pdff(x, mean, sd); }

// ---- K3's lane bodies (align_lanes.h) on a level of the TILED store (pyr_addr.h) ---------------------------------------
// pwb: the 10 x 10 template with border; px: in / out (level coordinates); returns the verdict of the reference's function
int hm_align2d(const uint8_t* level, int cols, int rows, int pitch, const uint8_t pwb[100], int n_iter, int phase, double px[2]) {
  uint32_t g[25];
  std::memcpy(g, pwb, 100);
  svo_track::AlignState st;
  st.u = (float)px[0]; st.v = (float)px[1]; st.mean_diff = 0.f; st.chi2 = 0.f; st.up0 = st.up1 = 0.f;
  bool converged = false, wrote = true;
  int n_eval = 0;
  // `phase` > 0: the iterations in runs of `phase` with the state parked in between, as the phased launches do
  const int step = phase > 0 ? phase : n_iter;
  for (int it0 = 0; it0 < n_iter; it0 += step) {
    const bool go_on = svo_track::align2d_lane(level, cols, rows, pitch, g, n_iter, it0, it0 + step, st, converged, wrote, n_eval);
    if (!go_on) break;
  }
  px[0] = (double)st.u;
  px[1] = (double)st.v;
  return converged ? 1 : 0;
}
int hm_align1d(const uint8_t* level, int cols, int rows, int pitch, const uint8_t pwb[100], const float dir[2], int n_iter, int phase,
               double px[2], double* h_inv) {
  uint32_t g[25];
  std::memcpy(g, pwb, 100);
  svo_track::AlignState st;
  st.u = (float)px[0]; st.v = (float)px[1]; st.mean_diff = 0.f; st.chi2 = 0.f; st.up0 = st.up1 = 0.f;
  bool converged = false, wrote = true;
  int n_eval = 0;
  *h_inv = 0.0;
  const int step = phase > 0 ? phase : n_iter;
  for (int it0 = 0; it0 < n_iter; it0 += step) {
    const bool go_on = svo_track::align1d_lane(level, cols, rows, pitch, g, dir[0], dir[1], n_iter, it0, it0 + step, st, *h_inv, converged,
                                               wrote, n_eval);
    if (!go_on) break;
  }
  px[0] = (double)st.u;
  px[1] = (double)st.v;
  return converged ? 1 : 0;
}
// byte offset of pixel (x, y) in a level of the store (for the test to lay an image out)
unsigned hm_px_off(int x, int y, int pitch) { return svo_pyr::px_off(x, y, pitch); }
long long hm_level_bytes(int pitch, int h) { return (long long)svo_pyr::level_bytes(pitch, h); }

// ---- warp_kernel's sample arithmetic (warp_sample.h) -----------------------------------------------------------------
// The level is a 48-byte-wide image, i.e. it IS a region in the kernel's layout (reg_o = level, xlo = ylo = 0).
// mode 0: warp_column<true> (per-sample bounds test), 1: warp_column<false>; mode 1 needs the
// box of the four corner samples inside the image, as in the kernel: -1 when it is not.  Returns 0 when A^-1 is NaN.
int hm_warp_patch(const uint8_t* level48, int rows, const double A_cur_ref[4], const double px_ref[2], int level_ref, int search_level,
                  int mode, uint8_t out[100]) {
  const int cols = 48;
  double Ainv[4];
  inv2<double>(A_cur_ref, Ainv);
  const float Ax = (float)Ainv[0], Ay = (float)Ainv[1], Az = (float)Ainv[2], Aw = (float)Ainv[3];
  if (Ax != Ax) return 0;
  const float pyrx = (float)px_ref[0] / (float)(1 << level_ref), pyry = (float)px_ref[1] / (float)(1 << level_ref);
  const float sc = (float)(1 << search_level);
  float bx0 = 3.0e38f, bx1 = -3.0e38f, by0 = 3.0e38f, by1 = -3.0e38f;
  for (int k = 0; k < 4; ++k) {
    float pp0 = (float)((k & 1) ? 4 : -5), pp1 = (float)((k & 2) ? 4 : -5);
    pp0 *= sc;
    pp1 *= sc;
    const float q0 = (Ax * pp0 + Ay * pp1) + pyrx;
    const float q1 = (Az * pp0 + Aw * pp1) + pyry;
    bx0 = fminf(bx0, q0); bx1 = fmaxf(bx1, q0);
    by0 = fminf(by0, q1); by1 = fmaxf(by1, q1);
  }
  const bool all_in = bx0 >= 0.f && by0 >= 0.f && bx1 < (float)(cols - 1) && by1 < (float)(rows - 1);
  if (mode != 0 && !all_in) return -1;
  for (int x = 0; x < 10; ++x) {
    uint8_t col[10];
    if (mode == 0) svo_track::warp_column<true>(Ax, Ay, Az, Aw, pyrx, pyry, sc, x, cols, rows, 0, 0, level48, col);
    else svo_track::warp_column<false>(Ax, Ay, Az, Aw, pyrx, pyry, sc, x, cols, rows, 0, 0, level48, col);
    for (int y = 0; y < 10; ++y) out[y * 10 + x] = col[y];
  }
  return 1;
}

}  // extern "C"
