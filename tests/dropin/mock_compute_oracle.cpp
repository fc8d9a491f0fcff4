// tests/dropin/mock_compute_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The COMPUTE entry points of the C ABI (include/svo_hip.h) that the drop-in bodies under rpg_svo_amd/host/dropin/ call,
// served by the CPU oracle (oracle/svo_oracle.h) -- together with tests/host/mock_svo_hip.cpp (memory, streams, a
// row-major pyramid store) this gives tests/dropin/_build/libsvo_pipeline_hipmock.so: the reference's control plane +
// the product's drop-in HOST code + a mock device.  It exists so that the host logic of the drop-ins -- marshalling,
// trial ordering, the predicted pose refinement, the deferred mapper's replay, slot bookkeeping -- runs end to end in
// the CPU test suite, where there is no GPU (tests/test_dropin_pipeline.py::test_dropin_host_logic_on_the_mock_device).
// It is NOT a fallback: nothing of the product links it, libsvo_hip.so has no such path, and the GPU tests run the same
// host code against the real kernels.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <svo_hip.h>

extern "C" {
#include "svo_oracle.h"
#include "orc_math.h"
}

namespace {

orc_pinhole camOf(const svo_hip_camera* c) {
  orc_pinhole o;
  o.fx = c->fx; o.fy = c->fy; o.cx = c->cx; o.cy = c->cy;
  o.width = c->width; o.height = c->height;
  o.model = c->model;  // SVO_HIP_CAM_* and ORC_CAM_* share their values
  o.pad_ = 0;
  for (int i = 0; i < 5; ++i) o.d[i] = c->d[i];
  return o;
}

// the mock store (tests/host/mock_svo_hip.cpp) is row-major with pitch == width: a level is a continuous image
orc_pyramid pyrOf(const svo_hip_pyr_layout* L, const uint8_t* store, int slot) {
  orc_pyramid p;
  std::memset(&p, 0, sizeof(p));
  p.n_levels = L->n_levels;
  for (int l = 0; l < L->n_levels; ++l) {
    p.w[l] = L->w[l];
    p.h[l] = L->h[l];
    p.data[l] = store + (int64_t)slot * L->slot_bytes + L->offset[l];
  }
  return p;
}

std::vector<orc_frame> framesOf(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_frames* ft) {
  std::vector<orc_frame> fr((size_t)ft->n_frames);
  for (int i = 0; i < ft->n_frames; ++i) {
    fr[i].pyr = pyrOf(L, store, ft->d_slot[i]);
    std::memcpy(fr[i].T_f_w, ft->d_T_f_w + 12 * i, 12 * sizeof(double));
  }
  return fr;
}

orc_feature featureOf(const svo_hip_features* f, int i) {
  orc_feature o;
  std::memset(&o, 0, sizeof(o));
  o.frame = f->d_frame[i];
  o.level = f->d_level[i];
  o.type = f->d_type ? f->d_type[i] : ORC_FTR_CORNER;
  o.px[0] = f->d_px[2 * i]; o.px[1] = f->d_px[2 * i + 1];
  for (int k = 0; k < 3; ++k) o.f[k] = f->d_f[3 * i + k];
  o.grad[0] = f->d_grad ? f->d_grad[2 * i] : 1.0;
  o.grad[1] = f->d_grad ? f->d_grad[2 * i + 1] : 0.0;
  return o;
}

int poseOptimize(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f, const int32_t* d_level,
                 const double* d_pos, uint8_t* d_has_point, double reproj_thresh, int n_iter, double* d_T, double* d_Cov,
                 double* d_stats, int32_t* d_ran) {
  if (!cam || B < 0 || n_stride < 1) return SVO_HIP_EINVAL;
  const orc_pinhole c = camOf(cam);
  for (int b = 0; b < B; ++b) {
    const int n = d_n[b];
    std::vector<int> level(d_level + (size_t)b * n_stride, d_level + (size_t)b * n_stride + n);
    orc_pose_opt_result r;
    std::memset(&r, 0, sizeof(r));
    orc_pose_optimize(reproj_thresh, n_iter, &c, d_T + 12 * b, n, d_f + (size_t)3 * b * n_stride, level.data(),
                      d_has_point + (size_t)b * n_stride, d_pos + (size_t)3 * b * n_stride, &r);
    d_ran[b] = r.ran;
    if (!r.ran) continue;
    std::memcpy(d_T + 12 * b, r.T_f_w, 12 * sizeof(double));
    if (d_Cov) std::memcpy(d_Cov + 36 * b, r.Cov, 36 * sizeof(double));
    d_stats[4 * b] = r.estimated_scale; d_stats[4 * b + 1] = r.error_init; d_stats[4 * b + 2] = r.error_final;
    d_stats[4 * b + 3] = (double)r.num_obs;
  }
  return SVO_HIP_OK;
}

}  // namespace

extern "C" {

int svo_hip_sparse_align(const svo_hip_pyr_layout* L, const uint8_t* store, int B, const int32_t* d_ref_slot,
                         const int32_t* d_cur_slot, const int32_t* d_n, int n_stride, const double* d_px, const double* d_xyz_ref,
                         const uint8_t* d_valid, const svo_hip_sia_params* P, const double* d_T_in, double* d_T_out,
                         double* d_H_out, int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2, int32_t* d_status, void*) {
  orc_pinhole cam;
  std::memset(&cam, 0, sizeof(cam));
  cam.fx = P->fx; cam.fy = P->fy; cam.cx = P->cx; cam.cy = P->cy;
  cam.width = L->w[0]; cam.height = L->h[0];
  cam.model = P->cam_model;
  for (int i = 0; i < 5; ++i) cam.d[i] = P->d[i];
  orc_sia_options opt;
  opt.max_level = P->max_level; opt.min_level = P->min_level; opt.n_iter = P->n_iter; opt.eps = P->eps;
  const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    const int n = d_n[b];
    const orc_pyramid rp = pyrOf(L, store, d_ref_slot[b]), cp = pyrOf(L, store, d_cur_slot[b]);
    // the reference frame is the world: a point's position is f * depth (sparse_img_align.cpp:107-108), the prior on
    // the current frame's pose is T_cur_from_ref
    std::vector<double> f((size_t)3 * n), pos((size_t)3 * n);
    std::vector<uint8_t> has((size_t)n);
    for (int i = 0; i < n; ++i) {
      const double* x = d_xyz_ref + ((size_t)b * n_stride + i) * 3;
      const double nn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      has[i] = (d_valid ? d_valid[(size_t)b * n_stride + i] : 1) && nn > 0;
      for (int k = 0; k < 3; ++k) { pos[3 * i + k] = x[k]; f[3 * i + k] = nn > 0 ? x[k] / nn : 0.0; }
    }
    double T[12];
    std::memcpy(T, d_T_in + 12 * b, sizeof(T));
    orc_sia_result r;
    std::memset(&r, 0, sizeof(r));
    orc_sparse_img_align_run(&rp, &cp, &cam, I12, T, n, d_px + (size_t)2 * b * n_stride, f.data(), has.data(), pos.data(), &opt,
                             &r, NULL);
    std::memcpy(d_T_out + 12 * b, r.T_cur_from_ref, 12 * sizeof(double));
    if (d_H_out) std::memcpy(d_H_out + 36 * b, r.H, 36 * sizeof(double));
    d_n_tracked[b] = r.n_tracked;
    if (d_iters)
      for (int l = 0; l < SVO_HIP_MAX_LEVELS; ++l) d_iters[(size_t)b * SVO_HIP_MAX_LEVELS + l] = l < ORC_MAX_LEVELS ? r.iters[l] : 0;
    if (d_chi2) d_chi2[b] = r.chi2;
    if (d_status) d_status[b] = r.stop ? SVO_HIP_SIA_STOP : 0;
  }
  return SVO_HIP_OK;
}

int svo_hip_frame_pose_compose(const double* d_T_cur_ref, const double* d_q_ref, const double* d_t_ref, double* d_frame_T, int cur_frame,
                               double* d_T_copy, double* d_T_out, const svo_hip_camera* cam, int n_frames, int n_kf,
                               const double* d_key_pos, const uint8_t* d_key_valid, int max_n_kfs, int32_t* d_rank, int32_t* d_rank_out,
                               int32_t* d_signal, int32_t signal_value, void*) {
  if (!d_T_cur_ref || !d_q_ref || !d_t_ref || !d_frame_T || cur_frame < 0) return SVO_HIP_EINVAL;
  orc_se3 x, y;
  orc_se3_from_Rt(d_T_cur_ref, &x);
  for (int k = 0; k < 4; ++k) y.q[k] = d_q_ref[k];
  for (int k = 0; k < 3; ++k) y.t[k] = d_t_ref[k];
  const orc_se3 r = orc_se3_compose(&x, &y);
  double T[12];
  orc_se3_to_Rt(&r, T);
  std::memcpy(d_frame_T + 12 * cur_frame, T, sizeof(T));
  if (d_T_copy) std::memcpy(d_T_copy, T, sizeof(T));
  if (d_T_out) std::memcpy(d_T_out, T, sizeof(T));
  if (d_rank) {  // Map::getCloseKeyframes + the reprojector's closest-first sort, cut at max_n_kfs
    if (!cam || n_frames < 1 || n_frames > 64 || n_kf < 0 || n_kf > n_frames) return SVO_HIP_EINVAL;
    const orc_pinhole c = camOf(cam);
    std::vector<double> dist((size_t)n_frames, -1.0);
    for (int i = 0; i < n_kf; ++i)
      for (int k = 0; k < 5; ++k) {
        if (!d_key_valid[5 * i + k]) continue;
        double xyz_f[3], uv[2], px[2];
        orc_se3_apply(&r, d_key_pos + 3 * (5 * i + k), xyz_f);
        if (xyz_f[2] < 0.0) continue;
        uv[0] = xyz_f[0] / xyz_f[2]; uv[1] = xyz_f[1] / xyz_f[2];
        orc_cam_world2cam_uv(&c, uv, px);
        if (px[0] >= 0.0 && px[1] >= 0.0 && px[0] < (double)c.width && px[1] < (double)c.height) {
          const double* tk = d_frame_T + 12 * i + 9;
          const double d[3] = {r.t[0] - tk[0], r.t[1] - tk[1], r.t[2] - tk[2]};
          dist[(size_t)i] = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
          break;
        }
      }
    for (int i = 0; i < n_frames; ++i) {
      int rk = -1;
      if (dist[(size_t)i] >= 0.0) {
        rk = 0;
        for (int j = 0; j < n_kf; ++j)
          if (dist[(size_t)j] >= 0.0 && (dist[(size_t)j] < dist[(size_t)i] || (dist[(size_t)j] == dist[(size_t)i] && j < i))) ++rk;
        if (rk >= max_n_kfs) rk = -1;
      }
      d_rank[i] = rk;
      if (d_rank_out) d_rank_out[i] = rk;
    }
  }
  if (d_signal) *d_signal = signal_value;
  return SVO_HIP_OK;
}

int svo_hip_find_match_direct(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_camera* cam,
                              const svo_hip_frames* frames, int M, const int32_t* d_cur_frame, const double* d_pt_pos,
                              const int32_t* d_obs_ptr, const svo_hip_features* obs, int n_pyr_levels, int align_max_iter,
                              double* d_px_cur, int32_t* d_ok, int32_t* d_ref_obs, int32_t* d_search_level, double* d_A_cur_ref,
                              uint8_t* d_patch_out, void*, size_t, void*) {
  const orc_pinhole c = camOf(cam);
  const std::vector<orc_frame> fr = framesOf(L, store, frames);
  orc_matcher_options opt;
  orc_matcher_options_default(&opt);
  opt.n_pyr_levels = n_pyr_levels;
  opt.align_max_iter = align_max_iter;
  for (int m = 0; m < M; ++m) {
    const int o0 = d_obs_ptr[m], n_obs = d_obs_ptr[m + 1] - o0;
    std::vector<orc_feature> ob((size_t)n_obs);
    for (int k = 0; k < n_obs; ++k) ob[k] = featureOf(obs, o0 + k);
    orc_match_result r;
    std::memset(&r, 0, sizeof(r));
    double px[2] = {d_px_cur[2 * m], d_px_cur[2 * m + 1]};
    const int ok = orc_find_match_direct(fr.data(), &c, d_cur_frame[m], d_pt_pos + 3 * m, n_obs, ob.data(), &opt, px, &r);
    d_px_cur[2 * m] = px[0]; d_px_cur[2 * m + 1] = px[1];
    d_ok[m] = ok;
    d_ref_obs[m] = r.ref_obs >= 0 ? o0 + r.ref_obs : -1;
    d_search_level[m] = r.search_level;
    if (d_A_cur_ref) std::memcpy(d_A_cur_ref + 4 * m, r.A_cur_ref, 4 * sizeof(double));
    if (d_patch_out) std::memcpy(d_patch_out + 100 * m, r.patch_with_border, 100);
  }
  return SVO_HIP_OK;
}

// the device-resident map mirror (row N2): the mock's "device memory" is host memory, so the patch is a plain copy and
// the walk is the oracle's restatement of Reprojector::reprojectMap
int svo_hip_reproject_map(const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame, const int32_t* d_kf_rank,
                          const svo_hip_map* map, const svo_hip_map_patch* patch, const svo_hip_grid* grid, int first_cell,
                          int max_cells_with_trials, int max_visits, int max_trials, const svo_hip_reprojection* out, void*) {
  if (patch) {
    for (int i = 0; i < patch->n_obs; ++i) {
      const int o = patch->d_obs_index[i];
      map->d_obs_frame[o] = patch->obs.d_frame[i];
      map->d_obs_order[o] = patch->d_obs_order[i];
      map->d_obs_level[o] = patch->obs.d_level[i];
      map->d_obs_type[o] = patch->obs.d_type[i];
      for (int k = 0; k < 2; ++k) map->d_obs_px[2 * o + k] = patch->obs.d_px[2 * i + k];
      for (int k = 0; k < 3; ++k) map->d_obs_f[3 * o + k] = patch->obs.d_f[3 * i + k];
      for (int k = 0; k < 2; ++k) map->d_obs_grad[2 * o + k] = patch->obs.d_grad[2 * i + k];
    }
    for (int i = 0; i < patch->n_points; ++i) {
      const int p = patch->d_index[i];
      for (int k = 0; k < 3; ++k) map->d_pos[3 * p + k] = patch->d_pos[3 * i + k];
      map->d_type[p] = patch->d_type[i];
      map->d_order[p] = patch->d_order[i];
      map->d_obs_begin[p] = patch->d_obs_begin[i];
      map->d_obs_count[p] = patch->d_obs_count[i];
    }
  }
  const orc_pinhole c = camOf(cam);
  const int P = map->n_points;
  const size_t cap = (size_t)(P > 0 ? P : 1);
  std::vector<int32_t> vp(cap), vc(cap), vt(cap), tobs(cap), tcell(cap);
  std::vector<double> tpx(2 * cap), tpos(3 * cap);
  int32_t header[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  orc_reproject_map(&c, frames->n_frames, frames->d_T_f_w, cur_frame, d_kf_rank, P, map->d_pos, map->d_type, map->d_order,
                    map->d_obs_begin, map->d_obs_count, map->d_obs_frame, map->d_obs_order, grid->cell_size, grid->n_cols,
                    grid->n_cells, grid->d_cell_rank, first_cell, max_cells_with_trials, header, out->d_point_cell, out->d_point_px,
                    out->d_kf_count, vp.data(), vc.data(), vt.data(), tobs.data(), tcell.data(), tpx.data(), tpos.data());
  const int V = header[2], M = header[3];
  if (header[1] > SVO_HIP_REPROJ_MAX_IN_FRAME || V > max_visits || M > max_trials) {  // the kernel's capacity rule
    header[0] = 1; header[2] = header[3] = 0; header[4] = first_cell;
  } else {
    for (int v = 0; v < V; ++v) { out->d_visit_point[v] = vp[v]; out->d_visit_cell[v] = vc[v]; out->d_visit_trial[v] = vt[v]; }
    for (int m = 0; m < M; ++m) {
      out->d_trial_cur[m] = cur_frame;
      for (int k = 0; k < 3; ++k) out->d_trial_pos[3 * m + k] = tpos[3 * m + k];
      out->d_trial_obs_begin[m] = tobs[m];
      out->d_trial_obs_end[m] = tobs[m] + 1;
      out->d_trial_cell[m] = tcell[m];
      out->d_trial_px[2 * m] = tpx[2 * m]; out->d_trial_px[2 * m + 1] = tpx[2 * m + 1];
    }
  }
  for (int k = 0; k < SVO_HIP_REPROJ_HEADER; ++k) out->d_header[k] = header[k];
  return SVO_HIP_OK;
}

int svo_hip_find_match_direct_indirect(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_camera* cam,
                                       const svo_hip_frames* frames, int M_cap, const int32_t* d_M, const int32_t* d_cur_frame,
                                       const double* d_pt_pos, const int32_t* d_obs_begin, const int32_t* d_obs_end,
                                       const svo_hip_features* obs, int n_pyr_levels, int align_max_iter, double* d_px_cur,
                                       int32_t* d_ok, int32_t* d_ref_obs, int32_t* d_search_level, double* d_A_cur_ref,
                                       uint8_t* d_patch_out, void*, size_t, void*) {
  const int M = d_M[0] < M_cap ? d_M[0] : M_cap;
  const orc_pinhole c = camOf(cam);
  const std::vector<orc_frame> fr = framesOf(L, store, frames);
  orc_matcher_options opt;
  orc_matcher_options_default(&opt);
  opt.n_pyr_levels = n_pyr_levels;
  opt.align_max_iter = align_max_iter;
  for (int m = 0; m < M; ++m) {
    const int o0 = d_obs_begin[m], n_obs = d_obs_end[m] - o0;
    std::vector<orc_feature> ob((size_t)(n_obs > 0 ? n_obs : 0));
    for (int k = 0; k < n_obs; ++k) ob[k] = featureOf(obs, o0 + k);
    orc_match_result r;
    std::memset(&r, 0, sizeof(r));
    double px[2] = {d_px_cur[2 * m], d_px_cur[2 * m + 1]};
    const int ok = orc_find_match_direct(fr.data(), &c, d_cur_frame[m], d_pt_pos + 3 * m, n_obs, ob.data(), &opt, px, &r);
    d_px_cur[2 * m] = px[0]; d_px_cur[2 * m + 1] = px[1];
    d_ok[m] = ok;
    d_ref_obs[m] = r.ref_obs >= 0 ? o0 + r.ref_obs : -1;
    d_search_level[m] = r.search_level;
    if (d_A_cur_ref) std::memcpy(d_A_cur_ref + 4 * m, r.A_cur_ref, 4 * sizeof(double));
    if (d_patch_out) std::memcpy(d_patch_out + 100 * m, r.patch_with_border, 100);
  }
  return SVO_HIP_OK;
}

int svo_hip_select_matches_indirect(const svo_hip_camera* cam, int M_cap, const int32_t* d_M, const int32_t* d_cell,
                                    const int32_t* d_ok, const double* d_px, const int32_t* d_level, const double* d_pos, int max_fts,
                                    int32_t* d_n, int32_t* d_sel, double* d_f, int32_t* d_level_out, double* d_pos_out,
                                    uint8_t* d_has_point, int32_t* d_signal, int32_t signal_value, void*) {
  if (d_signal) *d_signal = signal_value;
  const int M = d_M[0] < M_cap ? d_M[0] : M_cap;
  const orc_pinhole c = camOf(cam);
  const int n = M > 0 ? orc_select_matches(&c, M, d_cell, d_ok, d_px, d_level, d_pos, max_fts, d_sel, d_f, d_level_out, d_pos_out) : 0;
  for (int i = 0; i < n; ++i) d_has_point[i] = 1;
  d_n[0] = n;
  return SVO_HIP_OK;
}

int svo_hip_select_matches(const svo_hip_camera* cam, int M, const int32_t* d_cell, const int32_t* d_ok, const double* d_px,
                           const int32_t* d_level, const double* d_pos, int max_fts, int32_t* d_n, int32_t* d_sel, double* d_f,
                           int32_t* d_level_out, double* d_pos_out, uint8_t* d_has_point, int32_t* d_signal, int32_t signal_value,
                           void*) {
  if (d_signal) *d_signal = signal_value;  // the mock's streams are synchronous: everything before is complete
  const orc_pinhole c = camOf(cam);
  const int n = M > 0 ? orc_select_matches(&c, M, d_cell, d_ok, d_px, d_level, d_pos, max_fts, d_sel, d_f, d_level_out, d_pos_out) : 0;
  for (int i = 0; i < n; ++i) d_has_point[i] = 1;
  d_n[0] = n;
  return SVO_HIP_OK;
}

int svo_hip_pose_optimize(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f,
                          const int32_t* d_level, const double* d_pos, uint8_t* d_has_point, double reproj_thresh, int n_iter,
                          double* d_T, double* d_Cov, double* d_stats, int32_t* d_ran, void*) {
  return poseOptimize(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T, d_Cov, d_stats, d_ran);
}
int svo_hip_pose_optimize_deferred(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f,
                                   const int32_t* d_level, const double* d_pos, uint8_t* d_has_point, double reproj_thresh,
                                   int n_iter, double* d_T, double* d_Cov, double* d_stats, int32_t* d_ran, void*) {
  return poseOptimize(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T, d_Cov, d_stats, d_ran);
}
int svo_hip_pose_optimize_ordered(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride, const double* d_f,
                                  const int32_t* d_level, const double* d_pos, uint8_t* d_has_point, double reproj_thresh,
                                  int n_iter, double* d_T, double* d_Cov, double* d_stats, int32_t* d_ran, void*) {
  return poseOptimize(cam, B, d_n, n_stride, d_f, d_level, d_pos, d_has_point, reproj_thresh, n_iter, d_T, d_Cov, d_stats, d_ran);
}

int svo_hip_update_seeds(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_camera* cam, const svo_hip_frames* frames,
                         int S, const int32_t* d_cur_frame, const svo_hip_features* ftr, const svo_hip_seeds* seeds,
                         const svo_hip_depth_filter_options* opt, int32_t* d_status, double* d_xyz_world, double* d_px_cur, void*,
                         size_t, void*) {
  {  // fault injection (tests/test_dropin_pipeline.py: what the seed store does when an update fails): the k-th call fails
    static const int fail_at = [] { const char* v = std::getenv("SVO_MOCK_FAIL_UPDATE_SEEDS_AT"); return v ? std::atoi(v) : -1; }();
    static int n_calls = 0;
    if (++n_calls == fail_at) return SVO_HIP_EHIP;
  }
  const orc_pinhole c = camOf(cam);
  const std::vector<orc_frame> fr = framesOf(L, store, frames);
  orc_depth_filter_options dopt;
  dopt.max_n_kfs = opt->max_n_kfs;
  dopt.batch_counter = opt->batch_counter;
  dopt.seed_convergence_sigma2_thresh = opt->seed_convergence_sigma2_thresh;
  orc_matcher_options mopt;
  orc_matcher_options_default(&mopt);
  mopt.align_1d = opt->align_1d;
  mopt.align_max_iter = opt->align_max_iter;
  mopt.max_epi_search_steps = opt->max_epi_search_steps;
  mopt.subpix_refinement = opt->subpix_refinement;
  mopt.epi_search_edgelet_filtering = opt->epi_search_edgelet_filtering;
  mopt.epi_search_edgelet_max_angle = opt->epi_search_edgelet_max_angle;
  mopt.n_pyr_levels = opt->n_pyr_levels;
  // orc_update_seeds takes one measurement frame for the whole list, as DepthFilter::updateSeeds does: runs of equal
  // d_cur_frame are handed over one by one (the drop-in passes a single frame)
  int s0 = 0;
  while (s0 < S) {
    int s1 = s0;
    while (s1 < S && d_cur_frame[s1] == d_cur_frame[s0]) ++s1;
    const int n = s1 - s0;
    std::vector<orc_seed> sd((size_t)n);
    std::vector<orc_seed_update_info> info((size_t)n);
    for (int i = 0; i < n; ++i) {
      const int s = s0 + i;
      std::memset(&sd[i], 0, sizeof(orc_seed));
      sd[i].ftr = featureOf(ftr, s);
      sd[i].batch_id = seeds->d_batch_id ? seeds->d_batch_id[s] : 0;
      sd[i].a = seeds->d_a[s]; sd[i].b = seeds->d_b[s]; sd[i].mu = seeds->d_mu[s];
      sd[i].z_range = seeds->d_z_range[s]; sd[i].sigma2 = seeds->d_sigma2[s];
    }
    std::memset(info.data(), 0, info.size() * sizeof(orc_seed_update_info));
    orc_update_seeds(fr.data(), &c, d_cur_frame[s0], n, sd.data(), info.data(), &dopt, &mopt);
    for (int i = 0; i < n; ++i) {
      const int s = s0 + i;
      seeds->d_a[s] = sd[i].a; seeds->d_b[s] = sd[i].b; seeds->d_mu[s] = sd[i].mu; seeds->d_sigma2[s] = sd[i].sigma2;
      d_status[s] = info[i].status;  // SVO_HIP_SEED_* and ORC_SEED_* share their values
      for (int k = 0; k < 3; ++k) d_xyz_world[3 * s + k] = info[i].xyz_world[k];
      if (d_px_cur) { d_px_cur[2 * s] = info[i].px_cur[0]; d_px_cur[2 * s + 1] = info[i].px_cur[1]; }
    }
    s0 = s1;
  }
  return SVO_HIP_OK;
}

// row N2, seeds: the resident store on host memory (the mock's "device" memory is the host's)
int svo_hip_seed_store_patch(const svo_hip_seed_patch* p, const svo_hip_features* sf, const svo_hip_seeds* ss, void*) {
  if (!p || !sf || !ss || p->n < 0) return SVO_HIP_EINVAL;
  {  // fault injection: the k-th patch fails (its records never reach the store)
    static const int fail_at = [] { const char* v = std::getenv("SVO_MOCK_FAIL_SEED_PATCH_AT"); return v ? std::atoi(v) : -1; }();
    static int n_calls = 0;
    if (++n_calls == fail_at) return SVO_HIP_EHIP;
  }
  for (int i = 0; i < p->n; ++i) {
    const int q = p->d_slot[i];
    const_cast<int32_t*>(sf->d_frame)[q] = p->src_ftr.d_frame[i];
    const_cast<int32_t*>(sf->d_level)[q] = p->src_ftr.d_level[i];
    const_cast<uint8_t*>(sf->d_type)[q] = p->src_ftr.d_type ? p->src_ftr.d_type[i] : (uint8_t)SVO_HIP_FTR_CORNER;
    for (int k = 0; k < 2; ++k) const_cast<double*>(sf->d_px)[2 * q + k] = p->src_ftr.d_px[2 * i + k];
    for (int k = 0; k < 3; ++k) const_cast<double*>(sf->d_f)[3 * q + k] = p->src_ftr.d_f[3 * i + k];
    for (int k = 0; k < 2; ++k) const_cast<double*>(sf->d_grad)[2 * q + k] = p->src_ftr.d_grad ? p->src_ftr.d_grad[2 * i + k] : (k == 0 ? 1.0 : 0.0);
    ss->d_a[q] = p->src_seeds.d_a[i]; ss->d_b[q] = p->src_seeds.d_b[i]; ss->d_mu[q] = p->src_seeds.d_mu[i];
    ss->d_z_range[q] = p->src_seeds.d_z_range[i]; ss->d_sigma2[q] = p->src_seeds.d_sigma2[i];
    const_cast<int32_t*>(ss->d_batch_id)[q] = p->src_seeds.d_batch_id[i];
  }
  return SVO_HIP_OK;
}

int svo_hip_update_seeds_resident(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_camera* cam, const svo_hip_frames* frames,
                                  int cur_frame, int S, const int32_t* d_slot_of, const svo_hip_features* ftr, const svo_hip_seeds* seeds,
                                  const svo_hip_depth_filter_options* opt, int32_t* d_status, double* d_xyz_world, double* d_px_cur,
                                  float* d_state_out, void* ws, size_t ws_bytes, void* stream) {
  // gather the list's records, run the flattened call, scatter the state back
  std::vector<int32_t> frame((size_t)S), level((size_t)S), batch((size_t)S), cur((size_t)S, cur_frame);
  std::vector<uint8_t> type((size_t)S);
  std::vector<double> px(2 * (size_t)S), f(3 * (size_t)S), grad(2 * (size_t)S);
  std::vector<float> a((size_t)S), b((size_t)S), mu((size_t)S), zr((size_t)S), s2((size_t)S);
  for (int s = 0; s < S; ++s) {
    const int q = d_slot_of[s];
    frame[s] = ftr->d_frame[q]; level[s] = ftr->d_level[q]; type[s] = ftr->d_type[q]; batch[s] = seeds->d_batch_id[q];
    for (int k = 0; k < 2; ++k) { px[2 * s + k] = ftr->d_px[2 * q + k]; grad[2 * s + k] = ftr->d_grad[2 * q + k]; }
    for (int k = 0; k < 3; ++k) f[3 * s + k] = ftr->d_f[3 * q + k];
    a[s] = seeds->d_a[q]; b[s] = seeds->d_b[q]; mu[s] = seeds->d_mu[q]; zr[s] = seeds->d_z_range[q]; s2[s] = seeds->d_sigma2[q];
  }
  svo_hip_features ff;
  ff.d_frame = frame.data(); ff.d_level = level.data(); ff.d_type = type.data(); ff.d_px = px.data(); ff.d_f = f.data(); ff.d_grad = grad.data();
  svo_hip_seeds sd;
  sd.d_a = a.data(); sd.d_b = b.data(); sd.d_mu = mu.data(); sd.d_z_range = zr.data(); sd.d_sigma2 = s2.data(); sd.d_batch_id = batch.data();
  const int rc = svo_hip_update_seeds(L, store, cam, frames, S, cur.data(), &ff, &sd, opt, d_status, d_xyz_world, d_px_cur, ws, ws_bytes, stream);
  if (rc) return rc;
  for (int s = 0; s < S; ++s) {
    const int q = d_slot_of[s];
    seeds->d_a[q] = a[s]; seeds->d_b[q] = b[s]; seeds->d_mu[q] = mu[s]; seeds->d_sigma2[q] = s2[s];
    if (d_state_out) { d_state_out[s] = a[s]; d_state_out[S + s] = b[s]; d_state_out[2 * S + s] = mu[s]; d_state_out[3 * S + s] = s2[s]; }
  }
  return SVO_HIP_OK;
}

int svo_hip_update_seeds_resident_pose(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_camera* cam, const svo_hip_frames* frames,
                                       int cur_frame, const double* T_cur_f_w, int S, const int32_t* d_slot_of, const svo_hip_features* ftr,
                                       const svo_hip_seeds* seeds, const svo_hip_depth_filter_options* opt, int32_t* d_status,
                                       double* d_xyz_world, double* d_px_cur, float* d_state_out, void* ws, size_t ws_bytes, void* stream) {
  // the frame table with row cur_frame replaced by the pose handed over by value
  if (!T_cur_f_w || !frames || cur_frame < 0 || cur_frame >= frames->n_frames) return SVO_HIP_EINVAL;
  std::vector<double> T(frames->d_T_f_w, frames->d_T_f_w + 12 * (size_t)frames->n_frames);
  for (int k = 0; k < 12; ++k) T[12 * (size_t)cur_frame + k] = T_cur_f_w[k];
  svo_hip_frames fr = *frames;
  fr.d_T_f_w = T.data();
  return svo_hip_update_seeds_resident(L, store, cam, &fr, cur_frame, S, d_slot_of, ftr, seeds, opt, d_status, d_xyz_world, d_px_cur,
                                       d_state_out, ws, ws_bytes, stream);
}

int svo_hip_update_seed_batch(int S, const float* d_x, const float* d_tau2, const svo_hip_seeds* seeds, void*) {
  for (int s = 0; s < S; ++s) {
    orc_seed sd;
    std::memset(&sd, 0, sizeof(sd));
    sd.a = seeds->d_a[s]; sd.b = seeds->d_b[s]; sd.mu = seeds->d_mu[s]; sd.z_range = seeds->d_z_range[s]; sd.sigma2 = seeds->d_sigma2[s];
    orc_update_seed(d_x[s], d_tau2[s], &sd);
    seeds->d_a[s] = sd.a; seeds->d_b[s] = sd.b; seeds->d_mu[s] = sd.mu; seeds->d_sigma2[s] = sd.sigma2;
  }
  return SVO_HIP_OK;
}

int svo_hip_compute_tau_batch(int S, const double* d_t, const double* d_f, const double* d_z, double px_error_angle, double* d_tau,
                              void*) {
  for (int s = 0; s < S; ++s) {
    double T[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, d_t[3 * s], d_t[3 * s + 1], d_t[3 * s + 2]};
    d_tau[s] = orc_compute_tau(T, d_f + 3 * s, d_z[s], px_error_angle);
  }
  return SVO_HIP_OK;
}

// the stand-alone seams (hipm flavour: Matcher::findEpipolarMatchDirect, feature_alignment::align1D / align2D)
int svo_hip_find_epipolar_match_direct(const svo_hip_pyr_layout* L, const uint8_t* store, const svo_hip_camera* cam,
                                       const svo_hip_frames* frames, int S, const int32_t* d_cur_frame, const svo_hip_features* ftr,
                                       const double* d_d_estimate, const double* d_d_min, const double* d_d_max,
                                       const svo_hip_depth_filter_options* opt, int32_t* d_ok, double* d_depth, double* d_px_cur,
                                       int32_t* d_search_level, void*, size_t, void*) {
  const orc_pinhole c = camOf(cam);
  const std::vector<orc_frame> fr = framesOf(L, store, frames);
  orc_matcher_options mopt;
  orc_matcher_options_default(&mopt);
  mopt.align_1d = opt->align_1d;
  mopt.align_max_iter = opt->align_max_iter;
  mopt.max_epi_search_steps = opt->max_epi_search_steps;
  mopt.subpix_refinement = opt->subpix_refinement;
  mopt.epi_search_edgelet_filtering = opt->epi_search_edgelet_filtering;
  mopt.epi_search_edgelet_max_angle = opt->epi_search_edgelet_max_angle;
  mopt.n_pyr_levels = opt->n_pyr_levels;
  for (int s = 0; s < S; ++s) {
    const orc_feature f = featureOf(ftr, s);
    orc_match_result r;
    std::memset(&r, 0, sizeof(r));
    const int ok = orc_find_epipolar_match_direct(fr.data(), &c, f.frame, d_cur_frame[s], &f, d_d_estimate[s], d_d_min[s], d_d_max[s],
                                                  &mopt, &r);
    d_ok[s] = ok;
    d_depth[s] = ok ? r.depth : 0.0;
    if (d_px_cur) { d_px_cur[2 * s] = r.px_cur[0]; d_px_cur[2 * s + 1] = r.px_cur[1]; }
    if (d_search_level) d_search_level[s] = r.reject ? -1 : r.search_level;  // rejected before matcher.cpp:214
  }
  return SVO_HIP_OK;
}

int svo_hip_align_batch(const svo_hip_pyr_layout* L, const uint8_t* store, int M, const int32_t* d_slot, const int32_t* d_level,
                        const uint8_t* d_pwb, const float* d_dir, const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok,
                        double* d_h_inv, void*) {
  for (int t = 0; t < M; ++t) {
    const int l = d_level[t];
    const uint8_t* img = store + (int64_t)d_slot[t] * L->slot_bytes + L->offset[l];
    const uint8_t* pwb = d_pwb + 100 * t;
    uint8_t patch[64];  // Matcher::createPatchFromPatchWithBorder: the interior of the 10x10
    for (int y = 0; y < 8; ++y) std::memcpy(patch + 8 * y, pwb + 10 * (y + 1) + 1, 8);
    double px[2] = {d_px[2 * t], d_px[2 * t + 1]};
    double hinv = 0.0;
    if (d_use_1d && d_use_1d[t]) d_ok[t] = orc_align1d(img, L->w[l], L->h[l], L->pitch[l], d_dir + 2 * t, pwb, patch, n_iter, px, &hinv);
    else d_ok[t] = orc_align2d(img, L->w[l], L->h[l], L->pitch[l], pwb, patch, n_iter, px);
    d_px[2 * t] = px[0]; d_px[2 * t + 1] = px[1];
    if (d_h_inv) d_h_inv[t] = hinv;
  }
  return SVO_HIP_OK;
}

size_t svo_hip_fast_workspace_bytes(const svo_hip_pyr_layout*, int, int) { return 256; }

int svo_hip_fast_detect(const svo_hip_pyr_layout* L, const uint8_t* store, int n_frames, const int32_t* d_slot, int n_levels,
                        int fast_threshold, int cell_size, int grid_n_cols, int grid_n_rows, const uint8_t* d_occupancy,
                        double detection_threshold, int32_t* d_corner_xy, int32_t* d_corner_level, float* d_corner_score, void*,
                        size_t, void*) {
  const size_t cells = (size_t)grid_n_cols * grid_n_rows;
  for (int i = 0; i < n_frames; ++i) {
    const orc_pyramid p = pyrOf(L, store, d_slot[i]);
    orc_fast_detect_grid(&p, n_levels, fast_threshold, cell_size, grid_n_cols, grid_n_rows, d_occupancy ? d_occupancy + i * cells : NULL,
                         detection_threshold, d_corner_xy + 2 * i * cells, d_corner_level + i * cells, d_corner_score + i * cells);
  }
  return SVO_HIP_OK;
}

}  // extern "C"
