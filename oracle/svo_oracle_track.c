/*
 * svo_oracle_track.c -- CPU restatement of the tracking steps that follow sparse image
 * alignment: warp + feature alignment + matcher, pose
 * optimizer, point optimizer, depth filter.  TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 * Plain C99; build with -ffp-contract=off so float/double expressions round exactly like
 * the reference's written C++ (mixed float/double promotion rules are the same in C).
 *
 * Third-party pieces (Eigen small inverses / LDLT, Sophus SE3, vikit camera / ZMSSD /
 * interpolation / Tukey / MAD, boost normal pdf) come from orc_math.h and orc_vikit.h and
 * are evaluated in the order Eigen's expression templates evaluate each coefficient.
 */
#include "svo_oracle.h"
#include "orc_math.h"
#include "orc_vikit.h"

#include <stdlib.h>
#include <stdio.h>

#define SVO_EPS 0.0000000001 /* svo/include/svo/global.h:77 */
#define SVO_PI 3.14159265    /* svo/include/svo/global.h:78 */

/* ---- small vector helpers (left-fold reductions like Eigen's unrolled redux) ---- */
static inline double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void normalize3(double v[3]) { double n = norm3(v); v[0] /= n; v[1] /= n; v[2] /= n; }
static inline double norm2(const double v[2]) { return sqrt(v[0] * v[0] + v[1] * v[1]); }

/* ---- vk::AbstractCamera (orc_camera.h: pinhole, pinhole + radial-tangential, ATAN) ---- */
static inline void cam2world(const orc_pinhole* c, double u, double v, double f[3]) { orc_cam_cam2world(c, u, v, f); }
static inline void world2cam_uv(const orc_pinhole* c, const double uv[2], double px[2]) { orc_cam_world2cam_uv(c, uv, px); }
static inline void project2d(const double v[3], double uv[2]) { uv[0] = v[0] / v[2]; uv[1] = v[1] / v[2]; }
static inline void world2cam(const orc_pinhole* c, const double xyz[3], double px[2]) {
  double uv[2];
  project2d(xyz, uv);
  world2cam_uv(c, uv, px);
}
static inline int is_in_frame(const orc_pinhole* c, int x, int y, int boundary) {
  return x >= boundary && x < c->width - boundary && y >= boundary && y < c->height - boundary;
}
static inline int is_in_frame_level(const orc_pinhole* c, int x, int y, int boundary, int level) {
  return x >= boundary && x < c->width / (1 << level) - boundary && y >= boundary &&
         y < c->height / (1 << level) - boundary;
}
/* Frame::pos() = T_f_w_.inverse().translation()  (svo/include/svo/frame.h:110) */
static inline void frame_pos(const orc_se3* T_f_w, double p[3]) {
  orc_se3 inv = orc_se3_inverse(T_f_w);
  p[0] = inv.t[0]; p[1] = inv.t[1]; p[2] = inv.t[2];
}

void orc_matcher_options_default(orc_matcher_options* o) { /* matcher.h:84-92, config.cpp:60 */
  o->align_1d = 0;
  o->align_max_iter = 10;
  o->max_epi_length_optim = 2.0;
  o->max_epi_search_steps = 1000;
  o->subpix_refinement = 1;
  o->epi_search_edgelet_filtering = 1;
  o->epi_search_edgelet_max_angle = 0.7;
  o->n_pyr_levels = 3;
  o->pad_ = 0;
}

/* ======================================================================== */
/* feature_alignment::align1D, svo/src/feature_alignment.cpp:30-147          */
/* ======================================================================== */
int orc_align1d(const uint8_t* cur_img, int w, int h, int stride, const float dir[2],
                const uint8_t* ref_patch_with_border, const uint8_t* ref_patch, int n_iter,
                double px[2], double* h_inv) {
  const int halfpatch_size_ = 4;
  const int patch_size = 8;
  int converged = 0;
  float ref_patch_dv[64];
  float H[4] = {0, 0, 0, 0}; /* row-major 2x2 */
  const int ref_step = patch_size + 2;
  float* it_dv = ref_patch_dv;
  for (int y = 0; y < patch_size; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size; ++x, ++it, ++it_dv) {
      float J[2];
      J[0] = 0.5 * (dir[0] * (it[1] - it[-1]) + dir[1] * (it[ref_step] - it[-ref_step]));
      J[1] = 1;
      *it_dv = J[0];
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) H[r * 2 + c] += J[r] * J[c];
    }
  }
  *h_inv = 1.0 / H[0] * patch_size * patch_size;
  float Hinv[4];
  orc_inv2f(H, Hinv);
  float mean_diff = 0;
  float u = px[0];
  float v = px[1];
  const float min_update_squared = 0.03 * 0.03;
  const int cur_step = stride;
  float chi2 = 0;
  float update[2] = {0, 0};
  for (int iter = 0; iter < n_iter; ++iter) {
    int u_r = floorf(u);
    int v_r = floorf(v);
    if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= w - halfpatch_size_ || v_r >= h - halfpatch_size_)
      break;
    if (isnan(u) || isnan(v)) return 0;
    float subpix_x = u - u_r;
    float subpix_y = v - v_r;
    float wTL = (1.0 - subpix_x) * (1.0 - subpix_y);
    float wTR = subpix_x * (1.0 - subpix_y);
    float wBL = (1.0 - subpix_x) * subpix_y;
    float wBR = subpix_x * subpix_y;
    const uint8_t* it_ref = ref_patch;
    const float* it_ref_dv = ref_patch_dv;
    float new_chi2 = 0.0;
    float Jres[2] = {0, 0};
    for (int y = 0; y < patch_size; ++y) {
      const uint8_t* it = cur_img + (v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
      for (int x = 0; x < patch_size; ++x, ++it, ++it_ref, ++it_ref_dv) {
        float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        float res = search_pixel - *it_ref + mean_diff;
        Jres[0] -= res * (*it_ref_dv);
        Jres[1] -= res;
        new_chi2 += res * res;
      }
    }
    if (iter > 0 && new_chi2 > chi2) {
      u -= update[0]; /* sic: the reference undoes with the raw update, :116-117 */
      v -= update[1];
      break;
    }
    chi2 = new_chi2;
    update[0] = Hinv[0] * Jres[0] + Hinv[1] * Jres[1];
    update[1] = Hinv[2] * Jres[0] + Hinv[3] * Jres[1];
    u += update[0] * dir[0];
    v += update[0] * dir[1];
    mean_diff += update[1];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
      converged = 1;
      break;
    }
  }
  px[0] = u;
  px[1] = v;
  return converged;
}

/* ======================================================================== */
/* feature_alignment::align2D (x86: the plain float path), :149-277           */
/* ======================================================================== */
int orc_align2d(const uint8_t* cur_img, int w, int h, int stride, const uint8_t* ref_patch_with_border,
                const uint8_t* ref_patch, int n_iter, double px[2]) {
  const int halfpatch_size_ = 4;
  const int patch_size_ = 8;
  int converged = 0;
  float ref_patch_dx[64];
  float ref_patch_dy[64];
  float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int ref_step = patch_size_ + 2;
  float* it_dx = ref_patch_dx;
  float* it_dy = ref_patch_dy;
  for (int y = 0; y < patch_size_; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size_; ++x, ++it, ++it_dx, ++it_dy) {
      float J[3];
      J[0] = 0.5 * (it[1] - it[-1]);
      J[1] = 0.5 * (it[ref_step] - it[-ref_step]);
      J[2] = 1;
      *it_dx = J[0];
      *it_dy = J[1];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r * 3 + c] += J[r] * J[c];
    }
  }
  float Hinv[9];
  orc_inv3f(H, Hinv);
  float mean_diff = 0;
  float u = px[0];
  float v = px[1];
  const float min_update_squared = 0.03 * 0.03;
  const int cur_step = stride;
  float update[3] = {0, 0, 0};
  for (int iter = 0; iter < n_iter; ++iter) {
    int u_r = floorf(u);
    int v_r = floorf(v);
    if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= w - halfpatch_size_ || v_r >= h - halfpatch_size_)
      break;
    if (isnan(u) || isnan(v)) return 0; /* px is NOT written back on this path, :209 */
    float subpix_x = u - u_r;
    float subpix_y = v - v_r;
    float wTL = (1.0 - subpix_x) * (1.0 - subpix_y);
    float wTR = subpix_x * (1.0 - subpix_y);
    float wBL = (1.0 - subpix_x) * subpix_y;
    float wBR = subpix_x * subpix_y;
    const uint8_t* it_ref = ref_patch;
    const float* it_ref_dx = ref_patch_dx;
    const float* it_ref_dy = ref_patch_dy;
    float Jres[3] = {0, 0, 0};
    for (int y = 0; y < patch_size_; ++y) {
      const uint8_t* it = cur_img + (v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
      for (int x = 0; x < patch_size_; ++x, ++it, ++it_ref, ++it_ref_dx, ++it_ref_dy) {
        float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        float res = search_pixel - *it_ref + mean_diff;
        Jres[0] -= res * (*it_ref_dx);
        Jres[1] -= res * (*it_ref_dy);
        Jres[2] -= res;
      }
    }
    for (int r = 0; r < 3; ++r)
      update[r] = Hinv[r * 3] * Jres[0] + Hinv[r * 3 + 1] * Jres[1] + Hinv[r * 3 + 2] * Jres[2];
    u += update[0];
    v += update[1];
    mean_diff += update[2];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
      converged = 1;
      break;
    }
  }
  px[0] = u;
  px[1] = v;
  return converged;
}

/* ======================================================================== */
/* warp::, svo/src/matcher.cpp:33-105                                         */
/* ======================================================================== */
static void warp_matrix_affine_q(const orc_pinhole* cam_ref, const orc_pinhole* cam_cur, const double px_ref[2],
                                 const double f_ref[3], double depth_ref, const orc_se3* T_cur_ref,
                                 int level_ref, double A[4]) {
  const int halfpatch_size = 5;
  const double xyz_ref[3] = {f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref};
  double xyz_du_ref[3], xyz_dv_ref[3];
  /* px_ref + Vector2d(halfpatch_size,0)*(1<<level_ref) */
  const double s = (double)(1 << level_ref);
  cam2world(cam_ref, px_ref[0] + (double)halfpatch_size * s, px_ref[1] + 0.0 * s, xyz_du_ref);
  cam2world(cam_ref, px_ref[0] + 0.0 * s, px_ref[1] + (double)halfpatch_size * s, xyz_dv_ref);
  const double ku = xyz_ref[2] / xyz_du_ref[2];
  xyz_du_ref[0] *= ku; xyz_du_ref[1] *= ku; xyz_du_ref[2] *= ku;
  const double kv = xyz_ref[2] / xyz_dv_ref[2];
  xyz_dv_ref[0] *= kv; xyz_dv_ref[1] *= kv; xyz_dv_ref[2] *= kv;
  double p[3], px_cur[2], px_du[2], px_dv[2];
  orc_se3_apply(T_cur_ref, xyz_ref, p);
  world2cam(cam_cur, p, px_cur);
  orc_se3_apply(T_cur_ref, xyz_du_ref, p);
  world2cam(cam_cur, p, px_du);
  orc_se3_apply(T_cur_ref, xyz_dv_ref, p);
  world2cam(cam_cur, p, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size; /* col(0) */
  A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size; /* col(1) */
  A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;
}

void orc_get_warp_matrix_affine(const orc_pinhole* cam_ref, const orc_pinhole* cam_cur, const double px_ref[2],
                                const double f_ref[3], double depth_ref, const double T_cur_ref[12],
                                int level_ref, double A_cur_ref[4]) {
  orc_se3 T;
  orc_se3_from_Rt(T_cur_ref, &T);
  warp_matrix_affine_q(cam_ref, cam_cur, px_ref, f_ref, depth_ref, &T, level_ref, A_cur_ref);
}

int orc_get_best_search_level(const double A[4], int max_level) {
  int search_level = 0;
  double D = orc_det2d(A);
  while (D > 3.0 && search_level < max_level) {
    search_level += 1;
    D *= 0.25;
  }
  return search_level;
}

int orc_warp_affine(const double A_cur_ref[4], const uint8_t* img_ref, int w, int h, int stride,
                    const double px_ref[2], int level_ref, int search_level, int halfpatch_size,
                    uint8_t* patch) {
  const int patch_size = halfpatch_size * 2;
  double Ainv[4];
  orc_inv2d(A_cur_ref, Ainv);
  const float A_ref_cur[4] = {(float)Ainv[0], (float)Ainv[1], (float)Ainv[2], (float)Ainv[3]};
  if (isnan(A_ref_cur[0])) return 0; /* "Affine warp is NaN", patch untouched */
  uint8_t* patch_ptr = patch;
  const float px_ref_pyr[2] = {(float)px_ref[0] / (1 << level_ref), (float)px_ref[1] / (1 << level_ref)};
  for (int y = 0; y < patch_size; ++y) {
    for (int x = 0; x < patch_size; ++x, ++patch_ptr) {
      float px_patch[2] = {(float)(x - halfpatch_size), (float)(y - halfpatch_size)};
      px_patch[0] *= (1 << search_level);
      px_patch[1] *= (1 << search_level);
      const float px0 = (A_ref_cur[0] * px_patch[0] + A_ref_cur[1] * px_patch[1]) + px_ref_pyr[0];
      const float px1 = (A_ref_cur[2] * px_patch[0] + A_ref_cur[3] * px_patch[1]) + px_ref_pyr[1];
      if (px0 < 0 || px1 < 0 || px0 >= w - 1 || px1 >= h - 1)
        *patch_ptr = 0;
      else
        *patch_ptr = (uint8_t)orc_interpolate_mat_8u(img_ref, stride, px0, px1);
    }
  }
  return 1;
}

/* Matcher::createPatchFromPatchWithBorder, matcher.cpp:124-133 */
static void create_patch_from_patch_with_border(const uint8_t pwb[100], uint8_t patch[64]) {
  for (int y = 1; y < 9; ++y)
    for (int x = 0; x < 8; ++x) patch[(y - 1) * 8 + x] = pwb[y * 10 + 1 + x];
}

/* depthFromTriangulation, matcher.cpp:109-122 */
static int depth_from_triangulation(const orc_se3* T_search_ref, const double f_ref[3], const double f_cur[3],
                                    double* depth) {
  double R[9];
  orc_quat_to_R(T_search_ref->q, R);
  double A[3][2];
  for (int i = 0; i < 3; ++i) {
    A[i][0] = R[i * 3] * f_ref[0] + R[i * 3 + 1] * f_ref[1] + R[i * 3 + 2] * f_ref[2];
    A[i][1] = f_cur[i];
  }
  double AtA[4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) AtA[i * 2 + j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j];
  if (orc_det2d(AtA) < 0.000001) return 0;
  double inv[4];
  orc_inv2d(AtA, inv);
  for (int i = 0; i < 4; ++i) inv[i] = -inv[i];
  /* (-AtA^-1 * A^T) * t  -- row 0 only */
  double m[3];
  for (int k = 0; k < 3; ++k) m[k] = inv[0] * A[k][0] + inv[1] * A[k][1];
  const double d0 = m[0] * T_search_ref->t[0] + m[1] * T_search_ref->t[1] + m[2] * T_search_ref->t[2];
  *depth = fabs(d0);
  return 1;
}

/* Point::getCloseViewObs, svo/src/point.cpp:97-117 */
static int get_close_view_obs(const orc_se3* frames_q, const double framepos[3], const double pt_pos[3],
                              int n_obs, const orc_feature* obs, int* best) {
  double obs_dir[3] = {framepos[0] - pt_pos[0], framepos[1] - pt_pos[1], framepos[2] - pt_pos[2]};
  normalize3(obs_dir);
  int min_it = 0;
  double min_cos_angle = 0;
  for (int i = 0; i < n_obs; ++i) {
    double fp[3];
    frame_pos(&frames_q[obs[i].frame], fp);
    double dir[3] = {fp[0] - pt_pos[0], fp[1] - pt_pos[1], fp[2] - pt_pos[2]};
    normalize3(dir);
    double cos_angle = dot3(obs_dir, dir);
    if (cos_angle > min_cos_angle) {
      min_cos_angle = cos_angle;
      min_it = i;
    }
  }
  *best = min_it;
  if (min_cos_angle < 0.5) return 0;
  return 1;
}

static int max_frame_index(int cur, int n_obs, const orc_feature* obs) {
  int m = cur;
  for (int i = 0; i < n_obs; ++i)
    if (obs[i].frame > m) m = obs[i].frame;
  return m;
}

/* ======================================================================== */
/* Matcher::findMatchDirect, svo/src/matcher.cpp:135-177                      */
/* ======================================================================== */
int orc_find_match_direct(const orc_frame* frames, const orc_pinhole* cam, int cur_frame,
                          const double pt_pos[3], int n_obs, const orc_feature* obs,
                          const orc_matcher_options* opt, double px_cur[2], orc_match_result* res) {
  res->success = 0;
  res->ref_obs = -1;
  if (n_obs <= 0) return 0;
  const int nf = max_frame_index(cur_frame, n_obs, obs) + 1;
  orc_se3* fq = (orc_se3*)malloc(sizeof(orc_se3) * (size_t)nf);
  for (int i = 0; i < nf; ++i) orc_se3_from_Rt(frames[i].T_f_w, &fq[i]);
  int ok = 0;
  double cur_pos[3];
  frame_pos(&fq[cur_frame], cur_pos);
  int best = 0;
  const int close = get_close_view_obs(fq, cur_pos, pt_pos, n_obs, obs, &best);
  res->ref_obs = best;
  if (!close) goto done;
  {
    const orc_feature* ref = &obs[best];
    const orc_frame* rf = &frames[ref->frame];
    /* isInFrame(px.cast<int>()/(1<<level), halfpatch_size_+2, level) */
    const int pxi = (int)ref->px[0] / (1 << ref->level);
    const int pyi = (int)ref->px[1] / (1 << ref->level);
    if (!is_in_frame_level(cam, pxi, pyi, 4 + 2, ref->level)) goto done;
    double ref_pos[3];
    frame_pos(&fq[ref->frame], ref_pos);
    const double d[3] = {ref_pos[0] - pt_pos[0], ref_pos[1] - pt_pos[1], ref_pos[2] - pt_pos[2]};
    orc_se3 Tri = orc_se3_inverse(&fq[ref->frame]);
    orc_se3 T_cur_ref = orc_se3_compose(&fq[cur_frame], &Tri);
    warp_matrix_affine_q(cam, cam, ref->px, ref->f, norm3(d), &T_cur_ref, ref->level, res->A_cur_ref);
    res->search_level = orc_get_best_search_level(res->A_cur_ref, opt->n_pyr_levels - 1);
    orc_warp_affine(res->A_cur_ref, rf->pyr.data[ref->level], rf->pyr.w[ref->level], rf->pyr.h[ref->level],
                    rf->pyr.w[ref->level], ref->px, ref->level, res->search_level, 4 + 1,
                    res->patch_with_border);
    create_patch_from_patch_with_border(res->patch_with_border, res->patch);
    double px_scaled[2] = {px_cur[0] / (1 << res->search_level), px_cur[1] / (1 << res->search_level)};
    const orc_pyramid* cp = &frames[cur_frame].pyr;
    const int sl = res->search_level;
    if (ref->type == ORC_FTR_EDGELET) {
      double dir_cur[2] = {res->A_cur_ref[0] * ref->grad[0] + res->A_cur_ref[1] * ref->grad[1],
                           res->A_cur_ref[2] * ref->grad[0] + res->A_cur_ref[3] * ref->grad[1]};
      const double n = norm2(dir_cur);
      dir_cur[0] /= n; dir_cur[1] /= n;
      const float dirf[2] = {(float)dir_cur[0], (float)dir_cur[1]};
      ok = orc_align1d(cp->data[sl], cp->w[sl], cp->h[sl], cp->w[sl], dirf, res->patch_with_border, res->patch,
                       opt->align_max_iter, px_scaled, &res->h_inv);
    } else {
      ok = orc_align2d(cp->data[sl], cp->w[sl], cp->h[sl], cp->w[sl], res->patch_with_border, res->patch,
                       opt->align_max_iter, px_scaled);
    }
    px_cur[0] = px_scaled[0] * (1 << sl);
    px_cur[1] = px_scaled[1] * (1 << sl);
  }
done:
  free(fq);
  res->success = ok;
  res->px_cur[0] = px_cur[0];
  res->px_cur[1] = px_cur[1];
  return ok;
}

/* ======================================================================== */
/* Matcher::findEpipolarMatchDirect, svo/src/matcher.cpp:179-321              */
/* ======================================================================== */
static int align_in_level(const orc_pyramid* cp, int sl, const orc_matcher_options* opt, const double px_A[2],
                          const double px_B[2], orc_match_result* res, double px_scaled[2]) {
  if (opt->align_1d) {
    float d[2] = {(float)(px_A[0] - px_B[0]), (float)(px_A[1] - px_B[1])};
    const float n = sqrtf(d[0] * d[0] + d[1] * d[1]);
    d[0] /= n; d[1] /= n;
    return orc_align1d(cp->data[sl], cp->w[sl], cp->h[sl], cp->w[sl], d, res->patch_with_border, res->patch,
                       opt->align_max_iter, px_scaled, &res->h_inv);
  }
  return orc_align2d(cp->data[sl], cp->w[sl], cp->h[sl], cp->w[sl], res->patch_with_border, res->patch,
                     opt->align_max_iter, px_scaled);
}

static int find_epipolar_q(const orc_frame* frames, const orc_pinhole* cam, const orc_se3* T_ref_w,
                           const orc_se3* T_cur_w, int ref_frame, int cur_frame, const orc_feature* ref_ftr,
                           double d_estimate, double d_min, double d_max, const orc_matcher_options* opt,
                           orc_match_result* res) {
  orc_se3 Tri = orc_se3_inverse(T_ref_w);
  orc_se3 T_cur_ref = orc_se3_compose(T_cur_w, &Tri);
  int zmssd_best = ORC_ZMSSD_THRESHOLD;
  double uv_best[2] = {0, 0};
  res->success = 0;
  res->ref_obs = 0;

  double p[3], q[3], A[2], B[2];
  p[0] = ref_ftr->f[0] * d_min; p[1] = ref_ftr->f[1] * d_min; p[2] = ref_ftr->f[2] * d_min;
  orc_se3_apply(&T_cur_ref, p, q);
  project2d(q, A);
  p[0] = ref_ftr->f[0] * d_max; p[1] = ref_ftr->f[1] * d_max; p[2] = ref_ftr->f[2] * d_max;
  orc_se3_apply(&T_cur_ref, p, q);
  project2d(q, B);
  const double epi_dir[2] = {A[0] - B[0], A[1] - B[1]};

  warp_matrix_affine_q(cam, cam, ref_ftr->px, ref_ftr->f, d_estimate, &T_cur_ref, ref_ftr->level, res->A_cur_ref);

  res->reject = 0;
  if (ref_ftr->type == ORC_FTR_EDGELET && opt->epi_search_edgelet_filtering) {
    double g[2] = {res->A_cur_ref[0] * ref_ftr->grad[0] + res->A_cur_ref[1] * ref_ftr->grad[1],
                   res->A_cur_ref[2] * ref_ftr->grad[0] + res->A_cur_ref[3] * ref_ftr->grad[1]};
    const double gn = norm2(g);
    g[0] /= gn; g[1] /= gn;
    double e[2] = {epi_dir[0], epi_dir[1]};
    const double en = norm2(e);
    e[0] /= en; e[1] /= en;
    const double cosangle = fabs(g[0] * e[0] + g[1] * e[1]);
    if (cosangle < opt->epi_search_edgelet_max_angle) {
      res->reject = 1;
      return 0;
    }
  }

  res->search_level = orc_get_best_search_level(res->A_cur_ref, opt->n_pyr_levels - 1);
  const int sl = res->search_level;

  double px_A[2], px_B[2];
  world2cam_uv(cam, A, px_A);
  world2cam_uv(cam, B, px_B);
  const double dAB[2] = {px_A[0] - px_B[0], px_A[1] - px_B[1]};
  res->epi_length = norm2(dAB) / (1 << sl);

  const orc_frame* rf = &frames[ref_frame];
  orc_warp_affine(res->A_cur_ref, rf->pyr.data[ref_ftr->level], rf->pyr.w[ref_ftr->level],
                  rf->pyr.h[ref_ftr->level], rf->pyr.w[ref_ftr->level], ref_ftr->px, ref_ftr->level, sl, 4 + 1,
                  res->patch_with_border);
  create_patch_from_patch_with_border(res->patch_with_border, res->patch);
  const orc_pyramid* cp = &frames[cur_frame].pyr;

  if (res->epi_length < 2.0) {
    res->px_cur[0] = (px_A[0] + px_B[0]) / 2.0;
    res->px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
    double px_scaled[2] = {res->px_cur[0] / (1 << sl), res->px_cur[1] / (1 << sl)};
    const int ok = align_in_level(cp, sl, opt, px_A, px_B, res, px_scaled);
    if (ok) {
      res->px_cur[0] = px_scaled[0] * (1 << sl);
      res->px_cur[1] = px_scaled[1] * (1 << sl);
      double fc[3];
      cam2world(cam, res->px_cur[0], res->px_cur[1], fc);
      if (depth_from_triangulation(&T_cur_ref, ref_ftr->f, fc, &res->depth)) return 1;
    }
    return 0;
  }

  size_t n_steps = res->epi_length / 0.7;
  const double step[2] = {epi_dir[0] / n_steps, epi_dir[1] / n_steps};
  if (n_steps > (size_t)opt->max_epi_search_steps) return 0; /* "WARNING: skip epipolar search" */

  int sumA, sumAA;
  orc_zmssd_init(res->patch, &sumA, &sumAA);
  double uv[2] = {B[0] - step[0], B[1] - step[1]};
  int last_x = 0, last_y = 0;
  ++n_steps;
  const int cols = cp->w[sl];
  for (size_t i = 0; i < n_steps; ++i, uv[0] += step[0], uv[1] += step[1]) {
    double px[2];
    world2cam_uv(cam, uv, px);
    const int pxi0 = (int)(px[0] / (1 << sl) + 0.5);
    const int pxi1 = (int)(px[1] / (1 << sl) + 0.5);
    if (pxi0 == last_x && pxi1 == last_y) continue;
    last_x = pxi0;
    last_y = pxi1;
    if (!is_in_frame_level(cam, pxi0, pxi1, 8, sl)) continue;
    const uint8_t* cur_patch_ptr = cp->data[sl] + (pxi1 - 4) * cols + (pxi0 - 4);
    const int zmssd = orc_zmssd_score(res->patch, sumA, sumAA, cur_patch_ptr, cols);
    if (zmssd < zmssd_best) {
      zmssd_best = zmssd;
      uv_best[0] = uv[0];
      uv_best[1] = uv[1];
    }
  }

  if (zmssd_best < ORC_ZMSSD_THRESHOLD) {
    if (opt->subpix_refinement) {
      world2cam_uv(cam, uv_best, res->px_cur);
      double px_scaled[2] = {res->px_cur[0] / (1 << sl), res->px_cur[1] / (1 << sl)};
      const int ok = align_in_level(cp, sl, opt, px_A, px_B, res, px_scaled);
      if (ok) {
        res->px_cur[0] = px_scaled[0] * (1 << sl);
        res->px_cur[1] = px_scaled[1] * (1 << sl);
        double fc[3];
        cam2world(cam, res->px_cur[0], res->px_cur[1], fc);
        if (depth_from_triangulation(&T_cur_ref, ref_ftr->f, fc, &res->depth)) return 1;
      }
      return 0;
    }
    world2cam_uv(cam, uv_best, res->px_cur);
    double fc[3] = {uv_best[0], uv_best[1], 1.0};
    normalize3(fc);
    if (depth_from_triangulation(&T_cur_ref, ref_ftr->f, fc, &res->depth)) return 1;
  }
  return 0;
}

int orc_find_epipolar_match_direct(const orc_frame* frames, const orc_pinhole* cam, int ref_frame, int cur_frame,
                                   const orc_feature* ref_ftr, double d_estimate, double d_min, double d_max,
                                   const orc_matcher_options* opt, orc_match_result* res) {
  orc_se3 Tr, Tc;
  orc_se3_from_Rt(frames[ref_frame].T_f_w, &Tr);
  orc_se3_from_Rt(frames[cur_frame].T_f_w, &Tc);
  res->success = find_epipolar_q(frames, cam, &Tr, &Tc, ref_frame, cur_frame, ref_ftr, d_estimate, d_min, d_max,
                                 opt, res);
  return res->success;
}

/* ======================================================================== */
/* Frame::jacobian_xyz2uv, svo/include/svo/frame.h:116-138                    */
/* ======================================================================== */
static void frame_jacobian_xyz2uv(const double xyz[3], double J[12] /* 2x6 row-major */) {
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

/* ======================================================================== */
/* pose_optimizer::optimizeGaussNewton, svo/src/pose_optimizer.cpp:28-161     */
/* ======================================================================== */
int orc_pose_optimize(double reproj_thresh, int n_iter, const orc_pinhole* cam, const double T_f_w_in[12], int n,
                      const double* f, const int* level, uint8_t* has_point, const double* pos,
                      orc_pose_opt_result* res) {
  memset(res, 0, sizeof(*res));
  memcpy(res->T_f_w, T_f_w_in, sizeof(double) * 12);
  double chi2 = 0.0;
  orc_se3 T_f_w;
  orc_se3_from_Rt(T_f_w_in, &T_f_w);
  orc_se3 T_old = T_f_w;
  double A[36];
  double b[6];
  const double focal = fabs(cam->fx); /* errorMultiplier2 */

  float* errors = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  double* chi2_vec_init = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double* chi2_vec_final = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  int n_err = 0, n_init = 0, n_final = 0;
  for (int i = 0; i < n; ++i) {
    if (!has_point[i]) continue;
    double pf[3], a[2], c[2];
    orc_se3_apply(&T_f_w, pos + 3 * i, pf);
    project2d(f + 3 * i, a);
    project2d(pf, c);
    double e[2] = {a[0] - c[0], a[1] - c[1]};
    const double k = 1.0 / (1 << level[i]);
    e[0] *= k; e[1] *= k;
    errors[n_err++] = (float)norm2(e);
  }
  if (n_err == 0) {
    free(errors); free(chi2_vec_init); free(chi2_vec_final);
    return 0;
  }
  res->ran = 1;
  double estimated_scale = ORC_MAD_NORMALIZER * orc_median_float(errors, n_err);
  int num_obs = n_err;
  double scale = estimated_scale;
  for (int iter = 0; iter < n_iter; iter++) {
    if (iter == 5) scale = 0.85 / focal;
    memset(b, 0, sizeof(b));
    memset(A, 0, sizeof(A));
    double new_chi2 = 0.0;
    for (int i = 0; i < n; ++i) {
      if (!has_point[i]) continue;
      double J[12];
      double xyz_f[3], a[2], c[2];
      orc_se3_apply(&T_f_w, pos + 3 * i, xyz_f);
      frame_jacobian_xyz2uv(xyz_f, J);
      project2d(f + 3 * i, a);
      project2d(xyz_f, c);
      double e[2] = {a[0] - c[0], a[1] - c[1]};
      double sqrt_inv_cov = 1.0 / (1 << level[i]);
      e[0] *= sqrt_inv_cov; e[1] *= sqrt_inv_cov;
      if (iter == 0) chi2_vec_init[n_init++] = e[0] * e[0] + e[1] * e[1];
      for (int k = 0; k < 12; ++k) J[k] *= sqrt_inv_cov;
      double weight = orc_tukey_weight((float)(norm2(e) / scale));
      for (int r = 0; r < 6; ++r)
        for (int c2 = 0; c2 < 6; ++c2) A[r * 6 + c2] += (J[r] * J[c2] + J[6 + r] * J[6 + c2]) * weight;
      for (int r = 0; r < 6; ++r) b[r] -= (J[r] * e[0] + J[6 + r] * e[1]) * weight;
      new_chi2 += (e[0] * e[0] + e[1] * e[1]) * weight;
    }
    double dT[6];
    orc_ldlt_solve(6, A, b, dT);
    if ((iter > 0 && new_chi2 > chi2) || isnan(dT[0])) {
      T_f_w = T_old; /* roll-back */
      break;
    }
    orc_se3 ex = orc_se3_exp_q(dT);
    orc_se3 T_new = orc_se3_compose(&ex, &T_f_w);
    T_old = T_f_w;
    T_f_w = T_new;
    chi2 = new_chi2;
    res->n_iter_done++;
    double nm = -1;
    for (int k = 0; k < 6; ++k)
      if (fabs(dT[k]) > nm) nm = fabs(dT[k]);
    if (nm <= SVO_EPS) break;
  }
  /* frame->Cov_ = pixel_variance*(A*std::pow(f,2)).inverse(); */
  {
    double Af[36];
    const double f2 = pow(focal, 2);
    for (int k = 0; k < 36; ++k) Af[k] = A[k] * f2;
    orc_inv_lu(6, Af, res->Cov);
    for (int k = 0; k < 36; ++k) res->Cov[k] = 1.0 * res->Cov[k];
  }
  double reproj_thresh_scaled = reproj_thresh / focal;
  int n_deleted_refs = 0;
  for (int i = 0; i < n; ++i) {
    if (!has_point[i]) continue;
    double pf[3], a[2], c[2];
    orc_se3_apply(&T_f_w, pos + 3 * i, pf);
    project2d(f + 3 * i, a);
    project2d(pf, c);
    double e[2] = {a[0] - c[0], a[1] - c[1]};
    double sqrt_inv_cov = 1.0 / (1 << level[i]);
    e[0] *= sqrt_inv_cov; e[1] *= sqrt_inv_cov;
    chi2_vec_final[n_final++] = e[0] * e[0] + e[1] * e[1];
    if (norm2(e) > reproj_thresh_scaled) {
      has_point[i] = 0;
      ++n_deleted_refs;
    }
  }
  res->error_init = 0.0;
  res->error_final = 0.0;
  if (n_init > 0) res->error_init = sqrt(orc_median_double(chi2_vec_init, n_init)) * focal;
  if (n_final > 0) res->error_final = sqrt(orc_median_double(chi2_vec_final, n_final)) * focal;
  estimated_scale *= focal;
  res->estimated_scale = estimated_scale;
  num_obs -= n_deleted_refs;
  res->num_obs = num_obs;
  orc_se3_to_Rt(&T_f_w, res->T_f_w);
  free(errors); free(chi2_vec_init); free(chi2_vec_final);
  return 1;
}

/* ======================================================================== */
/* Point::optimize, svo/src/point.cpp:119-177; Jacobian point.h:89-103         */
/* ======================================================================== */
void orc_point_optimize(int n_iter, int n_obs, const double* T_f_w, const double* f, double pos[3]) {
  double old_point[3] = {pos[0], pos[1], pos[2]};
  double chi2 = 0.0;
  double A[9], b[3];
  orc_se3* Tq = (orc_se3*)malloc(sizeof(orc_se3) * (size_t)(n_obs > 0 ? n_obs : 1));
  for (int i = 0; i < n_obs; ++i) orc_se3_from_Rt(T_f_w + 12 * i, &Tq[i]);
  for (int it = 0; it < n_iter; it++) {
    memset(A, 0, sizeof(A));
    memset(b, 0, sizeof(b));
    double new_chi2 = 0.0;
    for (int o = 0; o < n_obs; ++o) {
      double p_in_f[3], R[9];
      orc_se3_apply(&Tq[o], pos, p_in_f);
      orc_quat_to_R(Tq[o].q, R);
      const double z_inv = 1.0 / p_in_f[2];
      const double z_inv_sq = z_inv * z_inv;
      double pj[6] = {z_inv, 0.0, -p_in_f[0] * z_inv_sq, 0.0, z_inv, -p_in_f[1] * z_inv_sq};
      double J[6]; /* 2x3 = (-pj) * R */
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
          J[r * 3 + c] = (-pj[r * 3]) * R[c] + (-pj[r * 3 + 1]) * R[3 + c] + (-pj[r * 3 + 2]) * R[6 + c];
      double a[2], c2[2];
      project2d(f + 3 * o, a);
      project2d(p_in_f, c2);
      const double e[2] = {a[0] - c2[0], a[1] - c2[1]};
      new_chi2 += e[0] * e[0] + e[1] * e[1];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r * 3 + c] += J[r] * J[c] + J[3 + r] * J[3 + c];
      for (int r = 0; r < 3; ++r) b[r] -= J[r] * e[0] + J[3 + r] * e[1];
    }
    double dp[3];
    orc_ldlt_solve(3, A, b, dp);
    if ((it > 0 && new_chi2 > chi2) || isnan(dp[0])) {
      pos[0] = old_point[0]; pos[1] = old_point[1]; pos[2] = old_point[2];
      break;
    }
    double new_point[3] = {pos[0] + dp[0], pos[1] + dp[1], pos[2] + dp[2]};
    old_point[0] = pos[0]; old_point[1] = pos[1]; old_point[2] = pos[2];
    pos[0] = new_point[0]; pos[1] = new_point[1]; pos[2] = new_point[2];
    chi2 = new_chi2;
    double nm = -1;
    for (int k = 0; k < 3; ++k)
      if (fabs(dp[k]) > nm) nm = fabs(dp[k]);
    if (nm <= SVO_EPS) break;
  }
  free(Tq);
}

/* ======================================================================== */
/* DepthFilter, svo/src/depth_filter.cpp                                      */
/* ======================================================================== */
void orc_seed_init(orc_seed* s, float depth_mean, float depth_min) { /* :37-46 */
  s->a = 10;
  s->b = 10;
  s->mu = 1.0 / depth_mean;
  s->z_range = 1.0 / depth_min;
  s->sigma2 = s->z_range * s->z_range / 36;
}

void orc_update_seed(const float x, const float tau2, orc_seed* seed) { /* :309-332 */
  float norm_scale = sqrtf(seed->sigma2 + tau2);
  if (isnan(norm_scale)) return;
  float s2 = 1. / (1. / seed->sigma2 + 1. / tau2);
  float m = s2 * (seed->mu / seed->sigma2 + x / tau2);
  float C1 = seed->a / (seed->a + seed->b) * orc_normal_pdff(x, seed->mu, norm_scale);
  float C2 = seed->b / (seed->a + seed->b) * 1. / seed->z_range;
  float normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  float f = C1 * (seed->a + 1.) / (seed->a + seed->b + 1.) + C2 * seed->a / (seed->a + seed->b + 1.);
  float e = C1 * (seed->a + 1.) * (seed->a + 2.) / ((seed->a + seed->b + 1.) * (seed->a + seed->b + 2.)) +
            C2 * seed->a * (seed->a + 1.0f) / ((seed->a + seed->b + 1.0f) * (seed->a + seed->b + 2.0f));
  float mu_new = C1 * m + C2 * seed->mu;
  seed->sigma2 = C1 * (s2 + m * m) + C2 * (seed->sigma2 + seed->mu * seed->mu) - mu_new * mu_new;
  seed->mu = mu_new;
  seed->a = (e - f) / (f - e / f);
  seed->b = seed->a * (1.0f - f) / f;
}

static double compute_tau_q(const orc_se3* T_ref_cur, const double f[3], const double z, const double px_error_angle) {
  const double t[3] = {T_ref_cur->t[0], T_ref_cur->t[1], T_ref_cur->t[2]};
  const double a[3] = {f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2]};
  double t_norm = norm3(t);
  double a_norm = norm3(a);
  double alpha = acos(dot3(f, t) / t_norm);
  const double mt[3] = {-t[0], -t[1], -t[2]};
  double beta = acos(dot3(a, mt) / (t_norm * a_norm));
  double beta_plus = beta + px_error_angle;
  double gamma_plus = SVO_PI - alpha - beta_plus;
  double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return (z_plus - z);
}
double orc_compute_tau(const double T_ref_cur[12], const double f[3], double z, double px_error_angle) { /* :334-350 */
  orc_se3 T;
  orc_se3_from_Rt(T_ref_cur, &T);
  return compute_tau_q(&T, f, z, px_error_angle);
}

int orc_update_seeds(const orc_frame* frames, const orc_pinhole* cam, int cur_frame, int n_seeds, orc_seed* seeds,
                     orc_seed_update_info* info, const orc_depth_filter_options* dopt,
                     const orc_matcher_options* mopt) { /* :197-291 */
  int n_updates = 0;
  const double focal_length = fabs(cam->fx);
  double px_noise = 1.0;
  double px_error_angle = atan(px_noise / (2.0 * focal_length)) * 2.0;
  orc_se3 Tc;
  orc_se3_from_Rt(frames[cur_frame].T_f_w, &Tc);
  orc_se3 Tc_inv = orc_se3_inverse(&Tc);
  orc_match_result mr;
  memset(&mr, 0, sizeof(mr));
  for (int i = 0; i < n_seeds; ++i) {
    orc_seed* it = &seeds[i];
    memset(&info[i], 0, sizeof(info[i]));
    if ((dopt->batch_counter - it->batch_id) > dopt->max_n_kfs) {
      info[i].status = ORC_SEED_ERASED_OLD;
      continue;
    }
    orc_se3 Tr;
    orc_se3_from_Rt(frames[it->ftr.frame].T_f_w, &Tr);
    orc_se3 T_ref_cur = orc_se3_compose(&Tr, &Tc_inv);
    orc_se3 T_cur_ref = orc_se3_inverse(&T_ref_cur);
    const double k = 1.0 / it->mu;
    const double pr[3] = {k * it->ftr.f[0], k * it->ftr.f[1], k * it->ftr.f[2]};
    double xyz_f[3];
    orc_se3_apply(&T_cur_ref, pr, xyz_f);
    if (xyz_f[2] < 0.0) {
      info[i].status = ORC_SEED_BEHIND;
      continue;
    }
    double pxp[2];
    world2cam(cam, xyz_f, pxp);
    if (!is_in_frame(cam, (int)pxp[0], (int)pxp[1], 0)) {
      info[i].status = ORC_SEED_NOT_IN_FRAME;
      continue;
    }
    float z_inv_min = it->mu + sqrtf(it->sigma2);
    const float zlo = it->mu - sqrtf(it->sigma2);
    float z_inv_max = (zlo < 0.00000001f) ? 0.00000001f : zlo; /* std::max(a,b) = a<b ? b : a */
    if (!find_epipolar_q(frames, cam, &Tr, &Tc, it->ftr.frame, cur_frame, &it->ftr, 1.0 / it->mu, 1.0 / z_inv_min,
                         1.0 / z_inv_max, mopt, &mr)) {
      it->b++;
      info[i].status = ORC_SEED_NO_MATCH;
      info[i].search_level = mr.search_level;
      continue;
    }
    const double z = mr.depth;
    double tau = compute_tau_q(&T_ref_cur, it->ftr.f, z, px_error_angle);
    const double zmt = (0.0000001 < z - tau) ? z - tau : 0.0000001; /* std::max */
    double tau_inverse = 0.5 * (1.0 / zmt - 1.0 / (z + tau));
    orc_update_seed(1. / z, tau_inverse * tau_inverse, it);
    ++n_updates;
    info[i].z = z;
    info[i].tau = tau;
    info[i].px_cur[0] = mr.px_cur[0];
    info[i].px_cur[1] = mr.px_cur[1];
    info[i].search_level = mr.search_level;
    if (sqrtf(it->sigma2) < it->z_range / dopt->seed_convergence_sigma2_thresh) {
      orc_se3 Tr_inv = orc_se3_inverse(&Tr);
      const double kk = 1.0 / it->mu;
      const double pw[3] = {it->ftr.f[0] * kk, it->ftr.f[1] * kk, it->ftr.f[2] * kk};
      orc_se3_apply(&Tr_inv, pw, info[i].xyz_world);
      info[i].status = ORC_SEED_CONVERGED;
    } else if (isnan(z_inv_min)) {
      info[i].status = ORC_SEED_NAN;
    } else {
      info[i].status = ORC_SEED_UPDATED;
    }
  }
  return n_updates;
}

/* ======================================================================== */
/* Reprojector::reprojectPoint, svo/src/reprojector.cpp:206-217               */
/* ======================================================================== */
int orc_reproject_point(const orc_pinhole* cam, const double T_f_w[12], const double pos[3], int cell_size,
                        int grid_n_cols, double px_out[2]) {
  orc_se3 T;
  orc_se3_from_Rt(T_f_w, &T);
  double p[3];
  orc_se3_apply(&T, pos, p);
  world2cam(cam, p, px_out);
  if (is_in_frame(cam, (int)px_out[0], (int)px_out[1], 8))
    return (int)(px_out[1] / cell_size) * grid_n_cols + (int)(px_out[0] / cell_size);
  return -1;
}

/* ======================================================================== */
/* The cell loop of Reprojector::reprojectMap (svo/src/reprojector.cpp:131-139) with                 */
/* reprojectCell (:150-200), given the outcome of every findMatchDirect trial.  Trials in visiting   */
/* order, those of one cell adjacent.  Emits what the new Feature holds for the pose optimizer:      */
/* f = cam2world(px) (feature.h:44-52), level, point position.  Returns the number of new features.  */
/* ======================================================================== */
int orc_select_matches(const orc_pinhole* cam, int M, const int32_t* cell, const int32_t* ok, const double* px,
                       const int32_t* level, const double* pos, int max_fts, int32_t* sel, double* f, int32_t* level_out,
                       double* pos_out) {
  int n_matches = 0, n = 0, m = 0;
  while (m < M) {
    const int c = cell[m];
    int matched = 0;
    for (; m < M && cell[m] == c; ++m) {  /* reprojectCell: walks the list until the first success (:153-199) */
      if (matched || !ok[m]) continue;
      sel[n] = m;
      cam2world(cam, px[2 * m], px[2 * m + 1], f + 3 * n);
      level_out[n] = level[m];
      for (int k = 0; k < 3; ++k) pos_out[3 * n + k] = pos[3 * m + k];
      ++n;
      matched = 1;
    }
    if (matched) ++n_matches;          /* :135-136 */
    if (n_matches > max_fts) break;    /* :137-138 */
  }
  return n;
}

/* ======================================================================== */
/* Reprojector::reprojectMap up to the first findMatchDirect (svo/src/reprojector.cpp:64-142), on the  */
/* plain-array form of the map (the device mirror's records, include/svo_hip.h svo_hip_map): the        */
/* keyframe loop with its "project a point only once" flag (:85-102), the candidate loop (:108-123),    */
/* reprojectPoint into grid cells (:206-217), the per-cell stable sort by point quality (:151-153), and  */
/* -- what the batched drop-in adds -- Point::getCloseViewObs (point.cpp:97-117) for every candidate    */
/* of the cells the visiting loop can reach: cells in visiting order from first_cell until              */
/* max_cells_with_trials of them hold a candidate with a close view.                                     */
/* Written the way the reference does it (lists filled in walking order, a stable sort per cell), NOT   */
/* the way the kernel does it (keys, counting, ranking).                                                 */
/* ======================================================================== */
typedef struct rm_item { int ord, p; } rm_item;
static int rm_item_cmp(const void* a, const void* b) { return ((const rm_item*)a)->ord - ((const rm_item*)b)->ord; }

int orc_reproject_map(const orc_pinhole* cam, int n_frames, const double* frame_T, int cur_frame, const int32_t* kf_rank,
                      int P, const double* pos, const int32_t* type, const int32_t* order, const int32_t* obs_begin,
                      const int32_t* obs_count, const int32_t* obs_frame, const int32_t* obs_order, int cell_size,
                      int n_cols, int n_cells, const int32_t* cell_rank, int first_cell, int max_cells_with_trials,
                      int32_t* header, int32_t* point_cell, double* point_px, int32_t* kf_count, int32_t* visit_point,
                      int32_t* visit_cell, int32_t* visit_trial, int32_t* trial_obs, int32_t* trial_cell, double* trial_px,
                      double* trial_pos) {
  /* cells: Reprojector::Grid::cells, each a list of Candidate(pt, px) in push_back order */
  int* cell_n = (int*)calloc((size_t)n_cells, sizeof(int));
  int** cell_items = (int**)calloc((size_t)n_cells, sizeof(int*));
  for (int k = 0; k < n_cells; ++k) cell_items[k] = (int*)malloc(sizeof(int) * (size_t)(P > 0 ? P : 1));
  char* last_projected = (char*)calloc((size_t)(P > 0 ? P : 1), 1);  /* Point::last_projected_kf_id_ == frame->id_ */
  rm_item* fts = (rm_item*)malloc(sizeof(rm_item) * (size_t)(P > 0 ? P : 1));
  orc_se3* fq = (orc_se3*)malloc(sizeof(orc_se3) * (size_t)n_frames);
  for (int i = 0; i < n_frames; ++i) orc_se3_from_Rt(frame_T + 12 * i, &fq[i]);
  for (int p = 0; p < P; ++p) point_cell[p] = -2;
  for (int i = 0; i < n_frames; ++i) kf_count[i] = 0;
  int n_in_frame = 0;
  /* :82-102 the closest keyframes first; every feature of the keyframe that has a map point */
  for (int rank = 0; rank < n_frames; ++rank) {
    int f = -1;
    for (int i = 0; i < n_frames; ++i)
      if (kf_rank[i] == rank) f = i;
    if (f < 0) continue;
    int n = 0;  /* ref_frame->fts_ in list order */
    for (int p = 0; p < P; ++p) {
      if (type[p] < 2) continue;  /* a Feature::point of a keyframe is a live UNKNOWN / GOOD point */
      for (int o = obs_begin[p]; o < obs_begin[p] + obs_count[p]; ++o)
        if (obs_frame[o] == f && obs_order[o] >= 0) { fts[n].ord = obs_order[o]; fts[n].p = p; ++n; }
    }
    qsort(fts, (size_t)n, sizeof(rm_item), rm_item_cmp);
    for (int i = 0; i < n; ++i) {
      const int p = fts[i].p;
      if (last_projected[p]) continue;      /* :97-99 */
      last_projected[p] = 1;
      const int k = orc_reproject_point(cam, frame_T + 12 * cur_frame, pos + 3 * p, cell_size, n_cols, point_px + 2 * p);
      point_cell[p] = k;
      if (k >= 0) {
        cell_items[k][cell_n[k]++] = p;
        ++kf_count[f];                      /* overlap_kfs.back().second++ */
        ++n_in_frame;
      }
    }
  }
  /* :108-123 all point candidates, in list order */
  {
    int n = 0;
    for (int p = 0; p < P; ++p)
      if (type[p] == 1) { fts[n].ord = order[p]; fts[n].p = p; ++n; }
    qsort(fts, (size_t)n, sizeof(rm_item), rm_item_cmp);
    for (int i = 0; i < n; ++i) {
      const int p = fts[i].p;
      const int k = orc_reproject_point(cam, frame_T + 12 * cur_frame, pos + 3 * p, cell_size, n_cols, point_px + 2 * p);
      point_cell[p] = k;
      if (k >= 0) { cell_items[k][cell_n[k]++] = p; ++n_in_frame; }
    }
  }
  /* the visiting order: grid_.cell_order[i] = the cell whose rank is i */
  int* cell_of_rank = (int*)malloc(sizeof(int) * (size_t)n_cells);
  for (int k = 0; k < n_cells; ++k) cell_of_rank[cell_rank[k]] = k;
  double cur_pos[3];
  frame_pos(&fq[cur_frame], cur_pos);
  int V = 0, M = 0, cells_with_trials = 0, i = first_cell;
  for (; i < n_cells && cells_with_trials < max_cells_with_trials; ++i) {
    const int k = cell_of_rank[i];
    int* it = cell_items[k];
    const int n = cell_n[k];
    /* :153 cell.sort(pointQualityComparator): stable, "lhs.type_ > rhs.type_" first */
    for (int a = 1; a < n; ++a) {
      const int v = it[a];
      int b = a - 1;
      while (b >= 0 && type[v] > type[it[b]]) { it[b + 1] = it[b]; --b; }
      it[b + 1] = v;
    }
    const int before = M;
    for (int a = 0; a < n; ++a) {
      const int p = it[a];
      visit_point[V] = p;
      visit_cell[V] = i;
      /* Point::getCloseViewObs (point.cpp:97-117) */
      int best = -1, close = 0;
      if (obs_count[p] > 0) {
        double obs_dir[3] = {cur_pos[0] - pos[3 * p], cur_pos[1] - pos[3 * p + 1], cur_pos[2] - pos[3 * p + 2]};
        normalize3(obs_dir);
        double min_cos_angle = 0;
        best = obs_begin[p];
        for (int o = obs_begin[p]; o < obs_begin[p] + obs_count[p]; ++o) {
          double fp[3];
          frame_pos(&fq[obs_frame[o]], fp);
          double dir[3] = {fp[0] - pos[3 * p], fp[1] - pos[3 * p + 1], fp[2] - pos[3 * p + 2]};
          normalize3(dir);
          const double cos_angle = dot3(obs_dir, dir);
          if (cos_angle > min_cos_angle) { min_cos_angle = cos_angle; best = o; }
        }
        close = !(min_cos_angle < 0.5);
      }
      if (close) {
        visit_trial[V] = M;
        trial_obs[M] = best;
        trial_cell[M] = i;
        trial_px[2 * M] = point_px[2 * p]; trial_px[2 * M + 1] = point_px[2 * p + 1];
        for (int c = 0; c < 3; ++c) trial_pos[3 * M + c] = pos[3 * p + c];
        ++M;
      } else {
        visit_trial[V] = -1;
      }
      ++V;
    }
    if (M > before) ++cells_with_trials;
  }
  header[0] = 0; header[1] = n_in_frame; header[2] = V; header[3] = M; header[4] = i;
  for (int k = 0; k < n_cells; ++k) free(cell_items[k]);
  free(cell_items); free(cell_n); free(last_projected); free(fts); free(fq); free(cell_of_rank);
  return 0;
}

/* vk::AbstractCamera::cam2world(px): the unit bearing (vikit pinhole_camera.cpp / atan_camera.cpp) */
void orc_cam2world(const orc_pinhole* cam, int n, const double* px, double* f) {
  for (int i = 0; i < n; ++i) cam2world(cam, px[2 * i], px[2 * i + 1], f + 3 * i);
}

/* ---- FastDetector::detect (svo/src/feature_detection.cpp:66-114) ----------------------- */
#include "orc_fast.h"
int orc_fast_detect_grid(const orc_pyramid* pyr, int n_levels, int fast_threshold, int cell_size, int grid_n_cols,
                         int grid_n_rows, const uint8_t* occupancy, double detection_threshold, int32_t* corner_xy,
                         int32_t* corner_level, float* corner_score) {
  const int n_cells = grid_n_cols * grid_n_rows;
  for (int k = 0; k < n_cells; ++k) {  /* Corner(0,0,detection_threshold,0,0.0f), :72 */
    corner_xy[2 * k] = corner_xy[2 * k + 1] = -1;
    corner_level[k] = -1;
    corner_score[k] = (float)detection_threshold;
  }
  for (int L = 0; L < n_levels; ++L) {
    const int scale = 1 << L;
    const int w = pyr->w[L], h = pyr->h[L];
    const uint8_t* img = pyr->data[L];
    int cap = w * h / 4 + 16;
    short* xy = (short*)malloc(sizeof(short) * 2 * (size_t)cap);
    int n = orc_fast10_detect(img, w, h, w, fast_threshold, xy, cap);
    if (n > cap) {
      free(xy);
      cap = n;
      xy = (short*)malloc(sizeof(short) * 2 * (size_t)cap);
      n = orc_fast10_detect(img, w, h, w, fast_threshold, xy, cap);
    }
    int* scores = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int* keep = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) scores[i] = orc_fast10_score(img + xy[2 * i + 1] * w + xy[2 * i], w, fast_threshold);
    const int m = n ? orc_fast_nonmax_3x3(xy, scores, n, w, h, keep) : 0;
    for (int j = 0; j < m; ++j) {
      const int x = xy[2 * keep[j]], y = xy[2 * keep[j] + 1];
      const int k = ((y * scale) / cell_size) * grid_n_cols + (x * scale) / cell_size;
      if (occupancy && occupancy[k]) continue;
      const float score = orc_shi_tomasi_score(img, w, h, w, x, y);
      if (score > corner_score[k]) {
        corner_xy[2 * k] = x * scale;
        corner_xy[2 * k + 1] = y * scale;
        corner_level[k] = L;
        corner_score[k] = score;
      }
    }
    free(xy); free(scores); free(keep);
  }
  int n_fts = 0;
  for (int k = 0; k < n_cells; ++k) n_fts += corner_score[k] > detection_threshold;
  return n_fts;
}
