/*
 * svo_oracle.c -- CPU restatement of SparseImgAlign + image pyramid.
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).  Plain C99, no dependencies.
 * Build with -ffp-contract=off so float expressions round like the written
 * C++ of the reference.
 */
#include "svo_oracle.h"
#include "orc_math.h"
#include "orc_vikit.h"

#include <stdlib.h>
#include <stdio.h>
#include <pthread.h>

/* ------------------------------------------------------------------------ */
/* exported SE(3) helpers                                                    */
/* ------------------------------------------------------------------------ */
void orc_se3_exp(const double xi[6], double T_out[12]) {
  orc_se3 s = orc_se3_exp_q(xi);
  orc_se3_to_Rt(&s, T_out);
}
void orc_se3_log(const double T[12], double xi_out[6]) {
  orc_se3 s;
  orc_se3_from_Rt(T, &s);
  orc_se3_log_q(&s, xi_out);
}
void orc_se3_mul(const double A[12], const double B[12], double out[12]) {
  orc_se3 a, b;
  orc_se3_from_Rt(A, &a);
  orc_se3_from_Rt(B, &b);
  orc_se3 r = orc_se3_compose(&a, &b);
  orc_se3_to_Rt(&r, out);
}
void orc_se3_inv(const double A[12], double out[12]) {
  orc_se3 a;
  orc_se3_from_Rt(A, &a);
  orc_se3 r = orc_se3_inverse(&a);
  orc_se3_to_Rt(&r, out);
}
int orc_ldlt6_solve(const double H[36], const double b[6], double x[6]) {
  return orc_ldlt_solve(6, H, b, x);
}
int orc_ldlt_solve_n(int n, const double* H, const double* b, double* x) {
  return orc_ldlt_solve(n, H, b, x);
}

/* ------------------------------------------------------------------------ */
/* vk::halfSample (rpg_vikit vision.cpp), called from                        */
/* frame_utils::createImgPyramid, svo/src/frame.cpp:156-165                  */
/* ------------------------------------------------------------------------ */
void orc_half_sample(const uint8_t* in, int in_w, int in_h, int in_stride,
                     uint8_t* out, int out_stride, int mode) {
  orc_half_sample_impl(in, in_w, in_h, in_stride, out, out_stride, mode); /* orc_vikit.h */
}

void orc_create_img_pyramid(const uint8_t* lvl0, int w, int h, int n_levels, int mode,
                            uint8_t* const* levels_out) {
  memcpy(levels_out[0], lvl0, (size_t)w * h);
  int cw = w, ch = h;
  for (int i = 1; i < n_levels; ++i) {
    /* pyr[i] = cv::Mat(pyr[i-1].rows/2, pyr[i-1].cols/2, CV_8U) */
    orc_half_sample(levels_out[i - 1], cw, ch, cw, levels_out[i], cw / 2, mode);
    cw /= 2;
    ch /= 2;
  }
}

/* ------------------------------------------------------------------------ */
/* SparseImgAlign, svo/src/sparse_img_align.cpp                              */
/* ------------------------------------------------------------------------ */
#define PATCH_HALFSIZE 2                 /* sparse_img_align.h:35 */
#define PATCH_SIZE (2 * PATCH_HALFSIZE)  /* :36 */
#define PATCH_AREA (PATCH_SIZE * PATCH_SIZE)

typedef struct {
  /* inputs */
  const orc_pyramid* ref_pyr;
  const orc_pyramid* cur_pyr;
  const orc_pinhole* cam;
  orc_se3 T_ref_w;
  double ref_pos[3];
  int n;
  const double* px;
  const double* f;
  const uint8_t* has_point;
  const double* pos;
  /* NLLSSolver state (rpg_vikit nlls_solver.h) */
  double H[36];
  double Jres[6];
  double x[6];
  double chi2;
  size_t n_meas;
  int n_iter;
  int iter;
  int stop;
  double eps;
  /* SparseImgAlign state */
  int level;
  int have_ref_patch_cache;
  float* ref_patch_cache; /* n x 16 */
  double* jacobian_cache; /* 6 x (16 n), column-major */
  uint8_t* visible_fts;
} sia_t;

/* Frame::jacobian_xyz2uv, svo/include/svo/frame.h:116-138 */
static void jacobian_xyz2uv(const double xyz[3], double J[12] /* 2x6 row-major */) {
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

/* sparse_img_align.cpp:84-145 */
static void precompute_reference_patches(sia_t* s) {
  const int border = PATCH_HALFSIZE + 1;
  const uint8_t* ref_img = s->ref_pyr->data[s->level];
  const int cols = s->ref_pyr->w[s->level];
  const int rows = s->ref_pyr->h[s->level];
  const int stride = cols;
  const float scale = 1.0f / (1 << s->level);
  const double focal_length = fabs(s->cam->fx); /* vk::PinholeCamera::errorMultiplier2 */
  for (int i = 0; i < s->n; ++i) {
    const float u_ref = s->px[2 * i] * scale;
    const float v_ref = s->px[2 * i + 1] * scale;
    const int u_ref_i = floorf(u_ref);
    const int v_ref_i = floorf(v_ref);
    if (!s->has_point[i] || u_ref_i - border < 0 || v_ref_i - border < 0 ||
        u_ref_i + border >= cols || v_ref_i + border >= rows)
      continue;
    s->visible_fts[i] = 1;

    const double dpx = s->pos[3 * i] - s->ref_pos[0];
    const double dpy = s->pos[3 * i + 1] - s->ref_pos[1];
    const double dpz = s->pos[3 * i + 2] - s->ref_pos[2];
    const double depth = sqrt(dpx * dpx + dpy * dpy + dpz * dpz);
    const double xyz_ref[3] = {s->f[3 * i] * depth, s->f[3 * i + 1] * depth, s->f[3 * i + 2] * depth};

    double frame_jac[12];
    jacobian_xyz2uv(xyz_ref, frame_jac);

    const float subpix_u_ref = u_ref - u_ref_i;
    const float subpix_v_ref = v_ref - v_ref_i;
    const float w_ref_tl = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
    const float w_ref_tr = subpix_u_ref * (1.0 - subpix_v_ref);
    const float w_ref_bl = (1.0 - subpix_u_ref) * subpix_v_ref;
    const float w_ref_br = subpix_u_ref * subpix_v_ref;
    size_t pixel_counter = 0;
    float* cache_ptr = s->ref_patch_cache + PATCH_AREA * i;
    for (int y = 0; y < PATCH_SIZE; ++y) {
      const uint8_t* p = ref_img + (v_ref_i + y - PATCH_HALFSIZE) * stride + (u_ref_i - PATCH_HALFSIZE);
      for (int x = 0; x < PATCH_SIZE; ++x, ++p, ++cache_ptr, ++pixel_counter) {
        *cache_ptr = w_ref_tl * p[0] + w_ref_tr * p[1] + w_ref_bl * p[stride] + w_ref_br * p[stride + 1];
        float dx = 0.5f * ((w_ref_tl * p[1] + w_ref_tr * p[2] + w_ref_bl * p[stride + 1] + w_ref_br * p[stride + 2]) -
                           (w_ref_tl * p[-1] + w_ref_tr * p[0] + w_ref_bl * p[stride - 1] + w_ref_br * p[stride]));
        float dy = 0.5f * ((w_ref_tl * p[stride] + w_ref_tr * p[1 + stride] + w_ref_bl * p[stride * 2] + w_ref_br * p[stride * 2 + 1]) -
                           (w_ref_tl * p[-stride] + w_ref_tr * p[1 - stride] + w_ref_bl * p[0] + w_ref_br * p[1]));
        double* col = s->jacobian_cache + 6 * ((size_t)i * PATCH_AREA + pixel_counter);
        const double fl = focal_length / (1 << s->level);
        for (int k = 0; k < 6; ++k) col[k] = (dx * frame_jac[k] + dy * frame_jac[6 + k]) * fl;
      }
    }
  }
  s->have_ref_patch_cache = 1;
}

/* sparse_img_align.cpp:147-243 (use_weights_ == false, display_ == false) */
static double compute_residuals(sia_t* s, const orc_se3* T_cur_from_ref, int linearize_system) {
  const uint8_t* cur_img = s->cur_pyr->data[s->level];
  const int cols = s->cur_pyr->w[s->level];
  const int rows = s->cur_pyr->h[s->level];

  if (!s->have_ref_patch_cache) precompute_reference_patches(s);

  const int stride = cols;
  const int border = PATCH_HALFSIZE + 1;
  const float scale = 1.0f / (1 << s->level);
  float chi2 = 0.0;
  for (int i = 0; i < s->n; ++i) {
    if (!s->visible_fts[i]) continue;

    const double dpx = s->pos[3 * i] - s->ref_pos[0];
    const double dpy = s->pos[3 * i + 1] - s->ref_pos[1];
    const double dpz = s->pos[3 * i + 2] - s->ref_pos[2];
    const double depth = sqrt(dpx * dpx + dpy * dpy + dpz * dpz);
    const double xyz_ref[3] = {s->f[3 * i] * depth, s->f[3 * i + 1] * depth, s->f[3 * i + 2] * depth};
    double xyz_cur[3];
    orc_se3_apply(T_cur_from_ref, xyz_ref, xyz_cur);
    /* cur_frame_->cam_->world2cam(xyz_cur) = world2cam(project2d(xyz_cur)) */
    const double uvn[2] = {xyz_cur[0] / xyz_cur[2], xyz_cur[1] / xyz_cur[2]};
    double pxy[2];
    orc_cam_world2cam_uv(s->cam, uvn, pxy);
    const double pxd = pxy[0];
    const double pyd = pxy[1];
    const float u_cur = (float)pxd * scale;
    const float v_cur = (float)pyd * scale;
    const int u_cur_i = floorf(u_cur);
    const int v_cur_i = floorf(v_cur);

    if (u_cur_i < 0 || v_cur_i < 0 || u_cur_i - border < 0 || v_cur_i - border < 0 ||
        u_cur_i + border >= cols || v_cur_i + border >= rows)
      continue;

    const float subpix_u_cur = u_cur - u_cur_i;
    const float subpix_v_cur = v_cur - v_cur_i;
    const float w_cur_tl = (1.0 - subpix_u_cur) * (1.0 - subpix_v_cur);
    const float w_cur_tr = subpix_u_cur * (1.0 - subpix_v_cur);
    const float w_cur_bl = (1.0 - subpix_u_cur) * subpix_v_cur;
    const float w_cur_br = subpix_u_cur * subpix_v_cur;
    const float* ref_patch_cache_ptr = s->ref_patch_cache + PATCH_AREA * i;
    size_t pixel_counter = 0;
    for (int y = 0; y < PATCH_SIZE; ++y) {
      const uint8_t* p = cur_img + (v_cur_i + y - PATCH_HALFSIZE) * stride + (u_cur_i - PATCH_HALFSIZE);
      for (int x = 0; x < PATCH_SIZE; ++x, ++pixel_counter, ++p, ++ref_patch_cache_ptr) {
        const float intensity_cur = w_cur_tl * p[0] + w_cur_tr * p[1] + w_cur_bl * p[stride] + w_cur_br * p[stride + 1];
        const float res = intensity_cur - (*ref_patch_cache_ptr);
        float weight = 1.0;
        chi2 += res * res * weight;
        s->n_meas++;
        if (linearize_system) {
          const double* J = s->jacobian_cache + 6 * ((size_t)i * PATCH_AREA + pixel_counter);
          /* H_.noalias() += J*J.transpose()*weight;  Jres_.noalias() -= J*res*weight; */
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) s->H[r * 6 + c] += J[r] * J[c] * weight;
          for (int r = 0; r < 6; ++r) s->Jres[r] -= J[r] * res * weight;
        }
      }
    }
  }
  return chi2 / s->n_meas;
}

/* sparse_img_align.cpp:245-251 */
static int sia_solve(sia_t* s) {
  orc_ldlt_solve(6, s->H, s->Jres, s->x);
  if (isnan(s->x[0])) return 0;
  return 1;
}

/* vk::NLLSSolver<6,SE3>::optimizeGaussNewton (rpg_vikit nlls_solver_impl.hpp),
 * with use_weights_=false, have_prior_=false. */
static int optimize_gauss_newton(sia_t* s, orc_se3* model) {
  orc_se3 old_model = *model;
  int evals = 0;
  for (s->iter = 0; s->iter < s->n_iter; ++s->iter) {
    memset(s->H, 0, sizeof(s->H));
    memset(s->Jres, 0, sizeof(s->Jres));
    s->n_meas = 0;
    double new_chi2 = compute_residuals(s, model, 1);
    ++evals;
    if (!sia_solve(s)) s->stop = 1;
    if ((s->iter > 0 && new_chi2 > s->chi2) || s->stop) {
      *model = old_model; /* rollback */
      break;
    }
    /* update(): T_new = T_old * SE3::exp(-x_), sparse_img_align.cpp:253-258 */
    double mx[6];
    for (int k = 0; k < 6; ++k) mx[k] = -s->x[k];
    orc_se3 e = orc_se3_exp_q(mx);
    orc_se3 new_model = orc_se3_compose(model, &e);
    old_model = *model;
    *model = new_model;
    s->chi2 = new_chi2;
    /* vk::norm_max(x_) <= eps_ */
    double nm = 0;
    for (int k = 0; k < 6; ++k)
      if (fabs(s->x[k]) > nm) nm = fabs(s->x[k]);
    if (nm <= s->eps) break;
  }
  return evals;
}

int orc_sparse_img_align_run(const orc_pyramid* ref_pyr, const orc_pyramid* cur_pyr,
                             const orc_pinhole* cam, const double T_ref_w[12], double T_cur_w[12],
                             int n, const double* px, const double* f, const uint8_t* has_point,
                             const double* pos, const orc_sia_options* opt, orc_sia_result* res,
                             uint8_t* visible_out) {
  sia_t s;
  memset(&s, 0, sizeof(s));
  memset(res, 0, sizeof(*res));
  /* reset(): chi2_=1e10, n_meas_=0, n_iter_=n_iter_init_, iter_=0, stop_=false */
  s.chi2 = 1e10;
  s.n_meas = 0;
  s.n_iter = opt->n_iter;
  s.iter = 0;
  s.stop = 0;
  s.eps = opt->eps;
  res->chi2 = s.chi2;

  orc_se3 T_cur;
  orc_se3_from_Rt(T_cur_w, &T_cur);
  orc_se3_from_Rt(T_ref_w, &s.T_ref_w);

  if (n == 0) { /* sparse_img_align.cpp:47-51: warn, return 0, pose untouched */
    orc_se3 Tri = orc_se3_inverse(&s.T_ref_w);
    orc_se3 Tcr = orc_se3_compose(&T_cur, &Tri);
    orc_se3_to_Rt(&Tcr, res->T_cur_from_ref);
    return 0;
  }

  s.ref_pyr = ref_pyr;
  s.cur_pyr = cur_pyr;
  s.cam = cam;
  s.n = n;
  s.px = px;
  s.f = f;
  s.has_point = has_point;
  s.pos = pos;
  s.ref_patch_cache = (float*)calloc((size_t)n * PATCH_AREA, sizeof(float));
  s.jacobian_cache = (double*)calloc((size_t)n * PATCH_AREA * 6, sizeof(double));
  s.visible_fts = (uint8_t*)calloc((size_t)n, 1);

  /* Frame::pos(): T_f_w_.inverse().translation() */
  orc_se3 T_ref_inv = orc_se3_inverse(&s.T_ref_w);
  s.ref_pos[0] = T_ref_inv.t[0];
  s.ref_pos[1] = T_ref_inv.t[1];
  s.ref_pos[2] = T_ref_inv.t[2];

  /* SE3 T_cur_from_ref(cur_frame_->T_f_w_ * ref_frame_->T_f_w_.inverse()); :59 */
  orc_se3 T_cur_from_ref = orc_se3_compose(&T_cur, &T_ref_inv);

  for (s.level = opt->max_level; s.level >= opt->min_level; --s.level) {
    memset(s.jacobian_cache, 0, (size_t)n * PATCH_AREA * 6 * sizeof(double)); /* :64 */
    s.have_ref_patch_cache = 0;
    int ev = optimize_gauss_newton(&s, &T_cur_from_ref);
    if (s.level < ORC_MAX_LEVELS) res->iters[s.level] = ev;
  }
  /* cur_frame_->T_f_w_ = T_cur_from_ref * ref_frame_->T_f_w_; :70 */
  orc_se3 T_new = orc_se3_compose(&T_cur_from_ref, &s.T_ref_w);
  orc_se3_to_Rt(&T_new, T_cur_w);
  orc_se3_to_Rt(&T_cur_from_ref, res->T_cur_from_ref);

  res->n_tracked = (int)(s.n_meas / PATCH_AREA);
  res->stop = s.stop;
  res->chi2 = s.chi2;
  memcpy(res->H, s.H, sizeof(s.H));
  if (visible_out) memcpy(visible_out, s.visible_fts, (size_t)n);
  free(s.ref_patch_cache);
  free(s.jacobian_cache);
  free(s.visible_fts);
  return res->n_tracked;
}

/* ---- batch driver with optional std threads (cpu_baseline "all cores") -- */
typedef struct {
  int b0, b1;
  const orc_pyramid* pyrs;
  const int *ref_slot, *cur_slot;
  const orc_pinhole* cam;
  const double* T_ref_w;
  double* T_cur_w;
  const int* n;
  int n_stride;
  const double *px, *f, *pos;
  const uint8_t* has_point;
  const orc_sia_options* opt;
  orc_sia_result* res;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*)arg;
  for (int b = j->b0; b < j->b1; ++b) {
    size_t o = (size_t)b * j->n_stride;
    orc_sparse_img_align_run(&j->pyrs[j->ref_slot[b]], &j->pyrs[j->cur_slot[b]], j->cam,
                             j->T_ref_w + 12 * b, j->T_cur_w + 12 * b, j->n[b], j->px + 2 * o,
                             j->f + 3 * o, j->has_point + o, j->pos + 3 * o, j->opt, &j->res[b], NULL);
  }
  return NULL;
}

int orc_sparse_img_align_batch(int B, const orc_pyramid* pyrs, const int* ref_slot,
                               const int* cur_slot, const orc_pinhole* cam, const double* T_ref_w,
                               double* T_cur_w, const int* n, int n_stride, const double* px,
                               const double* f, const uint8_t* has_point, const double* pos,
                               const orc_sia_options* opt, orc_sia_result* res, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > B) n_threads = B > 0 ? B : 1;
  batch_job* jobs = (batch_job*)calloc((size_t)n_threads, sizeof(batch_job));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  for (int t = 0; t < n_threads; ++t) {
    batch_job* j = &jobs[t];
    j->b0 = (int)((long long)B * t / n_threads);
    j->b1 = (int)((long long)B * (t + 1) / n_threads);
    j->pyrs = pyrs; j->ref_slot = ref_slot; j->cur_slot = cur_slot; j->cam = cam;
    j->T_ref_w = T_ref_w; j->T_cur_w = T_cur_w; j->n = n; j->n_stride = n_stride;
    j->px = px; j->f = f; j->pos = pos; j->has_point = has_point; j->opt = opt; j->res = res;
    if (n_threads == 1) batch_worker(j);
    else pthread_create(&th[t], NULL, batch_worker, j);
  }
  if (n_threads > 1)
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
  return 0;
}
