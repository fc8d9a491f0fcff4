// seed_finish.h -- the depth filter's workspace / argument blocks (shared by depth_filter.hip's kernels, the scan of
// epi_scan.h and the alignment's epilogue in feature_align.hip) and the LAST step of DepthFilter::updateSeeds for one seed:
// depthFromTriangulation (matcher.cpp:109-122), computeTau (depth_filter.cpp:334-350), updateSeed (:309-332), the
// convergence test (:261-262, :283-287).  seed_finish_kernel runs it over a batch of seeds; for large batches it is the
// epilogue of the lane-per-trial alignment kernel instead (a seed's last alignment launch hands over the refined pixel and
// its verdict in registers: no align_ok / px_cur round trip through HBM, one launch less).
#pragma once
#include "track_math.h"
#include "matcher_device.h"
#include "seed_math.h"

namespace svo_track {
using namespace svo_dev;

constexpr int ZMSSD_THRESHOLD = 2000 * 64;  // vk::patch_score::ZMSSD<4>::threshold()

enum : int { MODE_NONE = 0, MODE_SHORT = 1, MODE_SCAN = 2 };

struct SeedWs {
  uint8_t* align_active;  // [S]
  uint8_t* use_1d;        // [S]
  int32_t* status;        // [S] preliminary status (0 = still running)
  int32_t* mode;          // [S]
  int32_t* ref_slot;
  int32_t* ref_level;
  int32_t* cur_slot;
  int32_t* search_level;
  int32_t* n_steps;
  float* A_ref_cur;   // [S][4]
  float* px_ref_pyr;  // [S][2]
  float* dir;         // [S][2]
  float* z_inv_min;   // [S]
  double* B;          // [S][2] epipolar start (unit plane)
  double* step;       // [S][2]
  double* px_scaled;  // [S][2] align start, level coordinates
  double* px_cur;     // [S][2] Matcher::px_cur_
  double* uv_best;    // [S][2]
  uint8_t* pwb;       // [S][100]
  int32_t* align_ok;  // [S] written by the alignment kernel only
  uint8_t* accepted_raw;  // [S] 1: scan match accepted without sub-pixel refinement (subpix_refinement == false)
  // T_cur_ref and T_ref_cur (q0..3, t0..2 each) of the seed's (reference keyframe, current frame) pair live at
  // pair_T[14 * pair_index[s]]: formed once per run of seeds by seed_prepare_kernel.
  int32_t* pair_index;    // [S]
  double* pair_T;         // [S][14], written at the first seed of a run only
};

struct SeedArgs {
  svo_hip_pyr_layout L;
  const uint8_t* store;
  Cam cam;
  int S;
  const int32_t* frame_slot;
  const double* frame_T;
  const int32_t* cur_frame;  // [S], or NULL: every seed is updated with frame `cur_index` of the table
  int cur_index;
  // the pose of frame `cur_index` handed over BY VALUE (svo_hip_update_seeds_resident_pose: a host that uploaded the call's
  // tables before the frame's pose was known); 0: row cur_index of frame_T
  int cur_T_set;
  double cur_T[12];
  const int32_t* slot_of;    // NULL: seed s is record s of ftr / seeds; else the resident store's slot of seed s (row N2)
  float* state_out;          // [4][S] a, b, mu, sigma2 after the update, dense (resident store only; may be NULL)
  svo_hip_features ftr;
  svo_hip_seeds seeds;
  svo_hip_depth_filter_options opt;
  int32_t* status_out;
  double* xyz_world;
  double* px_cur_out;
  // Matcher::findEpipolarMatchDirect on its own (svo_hip_find_epipolar_match_direct): the depth interval
  // is given, nothing of DepthFilter::updateSeeds runs around it
  int match_only;
  const double* d_est;
  const double* d_min;
  const double* d_max;
  double* depth_out;
  int32_t* ok_out;
  int32_t* search_level_out;
  double px_error_angle;  // atan(1 / (2 |fx|)) * 2
  TauConsts tau_k;         // its sines and cosines (seed_math.h)
  SeedWs ws;
  // seeds per workgroup of the scan kernel (epi_scan.h: SCAN_CHUNK for batches, SCAN_CHUNK_SMALL for a camera frame's few
  // hundred seeds, which one workgroup of 32 groups would otherwise work through round by round)
  int scan_chunk;
};


// FROM_ALIGN = false: the verdict and the pixel of the sub-pixel alignment are read from the workspace (align_ok, px_cur);
// true: they arrive in registers -- `aligned_here` says whether THIS seed went through the alignment (then ok_here /
// px0_here, px1_here are its outputs), a seed that did not takes the workspace values as before.
template <bool FROM_ALIGN>
__device__ __forceinline__ void seed_finish_seed(const SeedArgs& a, const int s, const bool aligned_here, const int ok_here,
                                                 const double px0_here, const double px1_here) {
  const SeedWs& w = a.ws;
  // every record the seed may need is requested before the first early exit, as in seed_prepare_kernel (259 -> 243 us).
  // Workspace words of a seed that did not get that far hold whatever they held: read, not used.
  const int rec = a.slot_of ? a.slot_of[s] : s;  // the seed's record (resident store: its slot)
  const int rfi = a.ftr.d_frame[rec];
  // (a seed that went through the alignment in this very launch: its results are in registers)
  const int aok_early = FROM_ALIGN && aligned_here ? ok_here : w.align_ok[s];
  const double pxc0 = FROM_ALIGN && aligned_here ? px0_here : w.px_cur[2 * s], pxc1 = FROM_ALIGN && aligned_here ? px1_here : w.px_cur[2 * s + 1];
  const double f[3] = {a.ftr.d_f[3 * rec], a.ftr.d_f[3 * rec + 1], a.ftr.d_f[3 * rec + 2]};
  float sa = 0.f, sb = 0.f, smu = 0.f, ssig = 0.f, zr_early = 0.f;
  if (!a.match_only) {
    sa = a.seeds.d_a[rec]; sb = a.seeds.d_b[rec]; smu = a.seeds.d_mu[rec]; ssig = a.seeds.d_sigma2[rec];
    zr_early = a.seeds.d_z_range[rec];
  }
  // T_cur_ref and T_ref_cur of the seed's (reference, current) pair: formed once per run of seeds by seed_prepare_kernel
  // (rounds 1-5 rebuilt them per seed from the frame table: two quaternions from matrices, two products, three inverses)
  double PT[14];
  {
    const double* const pt = w.pair_T + 14 * (size_t)w.pair_index[s];
#pragma unroll
    for (int k = 0; k < 14; ++k) PT[k] = pt[k];
  }
  int status = w.status[s];
  const bool aligned = w.align_active[s] != 0;  // a short segment, or a scan match handed to the sub-pixel alignment
  if (a.px_cur_out) {
    // Matcher::px_cur_ exists once the seed reached the alignment (set by seed_prepare for a short segment, by the scan
    // for a match, refined by the alignment) or was accepted straight from the scan; 0 otherwise
    const bool has_px = aligned || w.accepted_raw[s] != 0;
    a.px_cur_out[2 * s] = has_px ? pxc0 : 0.0;
    a.px_cur_out[2 * s + 1] = has_px ? pxc1 : 0.0;
  }
  if (status == SVO_HIP_SEED_ERASED_OLD || status == SVO_HIP_SEED_BEHIND || status == SVO_HIP_SEED_NOT_IN_FRAME) {
    a.status_out[s] = status;
    return;
  }
  Se3 T_cur_ref, T_ref_cur;
#pragma unroll
  for (int k = 0; k < 4; ++k) { T_cur_ref.q[k] = PT[k]; T_ref_cur.q[k] = PT[7 + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) { T_cur_ref.t[k] = PT[4 + k]; T_ref_cur.t[k] = PT[11 + k]; }
  bool matched = false;
  double z = 0;
  if (status == 0) {
    const int aok = aligned ? aok_early : 0;
    if (aligned && aok == 1) {
      // px_cur_ = px_scaled*(1<<search_level_) was written by the alignment kernel
      double fc[3];
      cam2world(a.cam, pxc0, pxc1, fc);
      matched = depth_from_triangulation(T_cur_ref, f, fc, &z);
    } else if (w.accepted_raw[s]) {
      // subpix_refinement == false: vk::unproject2d(uv_best).normalized()
      double fc[3] = {w.uv_best[2 * s], w.uv_best[2 * s + 1], 1.0};
      normalize3(fc);
      matched = depth_from_triangulation(T_cur_ref, f, fc, &z);
    }
  }
  if (a.match_only) {  // findEpipolarMatchDirect's own outputs: the verdict, depth, px_cur_, search_level_
    a.ok_out[s] = matched ? 1 : 0;
    a.depth_out[s] = matched ? z : 0.0;
    if (a.search_level_out) a.search_level_out[s] = w.search_level[s];
    return;
  }
  const float zr = zr_early;
  if (!matched) {
    a.seeds.d_b[rec] = sb + 1.0f;  // it->b++ (:240)
    if (a.state_out) {
      a.state_out[s] = sa; a.state_out[a.S + s] = sb + 1.0f; a.state_out[2 * a.S + s] = smu; a.state_out[3 * a.S + s] = ssig;
    }
    a.status_out[s] = SVO_HIP_SEED_NO_MATCH;
    return;
  }
  // law of chord (depth_filter.cpp:252-255): atan(px_noise / (2 * focal_length)) * 2 depends on the camera alone -- computed
  // once on the host (run_seed_chain), by the libm the reference itself runs on
  const double tau = compute_tau(T_ref_cur, f, z, a.tau_k);
  const double zmt = (0.0000001 < z - tau) ? z - tau : 0.0000001;
  const double tau_inverse = 0.5 * (1.0 / zmt - 1.0 / (z + tau));
  update_seed((float)(1. / z), (float)(tau_inverse * tau_inverse), sa, sb, smu, zr, ssig);
  a.seeds.d_a[rec] = sa;
  a.seeds.d_b[rec] = sb;
  a.seeds.d_mu[rec] = smu;
  a.seeds.d_sigma2[rec] = ssig;
  if (a.state_out) {
    a.state_out[s] = sa; a.state_out[a.S + s] = sb; a.state_out[2 * a.S + s] = smu; a.state_out[3 * a.S + s] = ssig;
  }
  if ((double)sqrtf(ssig) < (double)zr / a.opt.seed_convergence_sigma2_thresh) {
    Se3 Tr;  // (rare: a seed converges once in its life)
    {
      double Rt[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) Rt[k] = a.frame_T[12 * rfi + k];
      se3_from_Rt(Rt, Tr);
    }
    const Se3 Tr_inv = se3_inverse(Tr);
    const double kk = 1.0 / (double)smu;
    const double pw[3] = {f[0] * kk, f[1] * kk, f[2] * kk};
    double xw[3];
    se3_apply(Tr_inv, pw, xw);
    if (a.xyz_world) {
      a.xyz_world[3 * s] = xw[0];
      a.xyz_world[3 * s + 1] = xw[1];
      a.xyz_world[3 * s + 2] = xw[2];
    }
    status = SVO_HIP_SEED_CONVERGED;
  } else if (isnan(w.z_inv_min[s])) {
    status = SVO_HIP_SEED_NAN;
  } else {
    status = SVO_HIP_SEED_UPDATED;
  }
  a.status_out[s] = status;
}

}  // namespace svo_track
