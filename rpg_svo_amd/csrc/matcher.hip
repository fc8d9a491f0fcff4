// matcher.hip -- K2: batched Matcher::findMatchDirect front end for gfx950, plus
// Reprojector::reprojectPoint.
//
// Replaces, per candidate map point (svo/src/matcher.cpp:135-177):
//   Point::getCloseViewObs            (svo/src/point.cpp:97-117)     \  match_prepare_kernel
//   isInFrame test on the ref feature (matcher.cpp:143-145)          |  one lane per candidate,
//   warp::getWarpMatrixAffine         (matcher.cpp:33-55)            |  f64 geometry
//   warp::getBestSearchLevel          (matcher.cpp:57-70)            /
//   warp::warpAffine 10x10            (matcher.cpp:72-105)              warp_kernel, 8 lanes per
//                                                                       candidate (warp_group.h)
//   feature_alignment::align2D/1D     (feature_alignment.cpp)           K3 (feature_align.hip)
//
// The map (Point -> list of observing Features -> Frames) reaches the device as a CSR
// structure over SoA feature records and a frame table; no pointer graph on the GPU.
// Everything a later stage needs travels through a caller-provided workspace, so the whole
// chain is three launches on one stream with no host synchronisation.
#pragma clang fp contract(off)
#include <type_traits>

#include "track_kernels.h"
#include "track_math.h"
#include "matcher_device.h"
#include "warp_sample.h"
#include "warp_group.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_track;


namespace {

struct PrepArgs {
  Cam cam;
  int M;
  int n_pyr_levels;
  const int32_t* frame_slot;
  const double* frame_T;
  const int32_t* cur_frame;
  const double* pt_pos;
  const int32_t* obs_ptr;    // [M+1] CSR, or NULL and then:
  const int32_t* obs_begin;  // [M]
  const int32_t* obs_end;    // [M]
  const int32_t* M_dev;      // NULL, or the batch size lives on the device (min(*M_dev, M))
  svo_hip_features obs;
  const double* px_cur;  // [M][2] level-0 estimate
  // outputs for the caller
  int32_t* ref_obs;
  int32_t* search_level;
  double* A_cur_ref;  // may be NULL
  // workspace
  uint8_t* active;
  int32_t* ref_slot;
  int32_t* ref_level;
  int32_t* cur_slot;
  float* A_ref_cur;
  float* px_ref_pyr;
  float* dir;
  uint8_t* use_1d;
  double* px_scaled;
};

__global__ void __launch_bounds__(64) match_prepare_kernel(const PrepArgs a) {
  const int m = blockIdx.x * 64 + threadIdx.x;
  if (m >= (a.M_dev ? min(*a.M_dev, a.M) : a.M)) return;
  // The trial's own records -- frame, projection, observation range, position -- are requested before anything is stored
  // or looked at, the frame's pose and slot in a second round; in the closest-view loop the frame index of observation
  // o + 1 is requested while observation o is worked on, so that an observation costs one dependent memory round trip, not
  // two (reading each value where it is first needed made ~8 round trips before the loop and two per observation: 262
  // against 223 us per 16 384 frames, profiles/r05a_queue_drain.txt).
  const int cf = a.cur_frame[m];
  const double pxc0 = a.px_cur[2 * m], pxc1 = a.px_cur[2 * m + 1];
  // (either CSR offsets or begin / end arrays: read through one pointer each, a load under a condition is waited for in its branch)
  const int o0 = (a.obs_ptr ? a.obs_ptr : a.obs_begin)[m], o1 = (a.obs_ptr ? a.obs_ptr + 1 : a.obs_end)[m];
  const double pt[3] = {a.pt_pos[3 * m], a.pt_pos[3 * m + 1], a.pt_pos[3 * m + 2]};
  double RtC[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) RtC[k] = a.frame_T[12 * cf + k];
  const int cur_slot_v = a.frame_slot[cf];
  int fi_next = o1 > o0 ? a.obs.d_frame[o0] : 0;
  a.active[m] = 0;
  a.ref_obs[m] = -1;
  a.search_level[m] = 0;
  a.ref_slot[m] = 0;
  a.ref_level[m] = 0;
  a.use_1d[m] = 0;
  a.dir[2 * m] = 1.f;
  a.dir[2 * m + 1] = 0.f;
  a.cur_slot[m] = cur_slot_v;
  a.px_scaled[2 * m] = pxc0;
  a.px_scaled[2 * m + 1] = pxc1;
  if (a.A_cur_ref) {
    a.A_cur_ref[4 * m] = a.A_cur_ref[4 * m + 1] = a.A_cur_ref[4 * m + 2] = a.A_cur_ref[4 * m + 3] = 0.0;
  }
  if (o1 <= o0) return;
  Se3 Tc;
  se3_from_Rt(RtC, Tc);
  double cur_pos[3];
  frame_pos(Tc, cur_pos);
  // Point::getCloseViewObs (point.cpp:97-117)
  double obs_dir[3] = {cur_pos[0] - pt[0], cur_pos[1] - pt[1], cur_pos[2] - pt[2]};
  normalize3(obs_dir);
  int best = o0;
  double min_cos_angle = 0;
  for (int o = o0; o < o1; ++o) {
    const int fi = fi_next;
    double RtF[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) RtF[k] = a.frame_T[12 * fi + k];
    fi_next = a.obs.d_frame[o + 1 < o1 ? o + 1 : o];  // (the next observation's frame, on its way while this one is worked on)
    Se3 Tf;
    se3_from_Rt(RtF, Tf);
    double fp[3];
    frame_pos(Tf, fp);
    double dir[3] = {fp[0] - pt[0], fp[1] - pt[1], fp[2] - pt[2]};
    normalize3(dir);
    const double cos_angle = dot3(obs_dir, dir);
    if (cos_angle > min_cos_angle) {
      min_cos_angle = cos_angle;
      best = o;
    }
  }
  a.ref_obs[m] = best;
  if (min_cos_angle < 0.5) return;
  const int rfi = a.obs.d_frame[best];
  const int rlevel = a.obs.d_level[best];
  const double rpx[2] = {a.obs.d_px[2 * best], a.obs.d_px[2 * best + 1]};
  // isInFrame(px.cast<int>()/(1<<level), halfpatch_size_+2, level)  (matcher.cpp:143-145)
  if (!is_in_frame_level(a.cam, cast_int(rpx[0]) / (1 << rlevel), cast_int(rpx[1]) / (1 << rlevel), 4 + 2, rlevel))
    return;
  Se3 Tr;
  se3_from_Rt(a.frame_T + 12 * rfi, Tr);
  double ref_pos[3];
  frame_pos(Tr, ref_pos);
  const double d[3] = {ref_pos[0] - pt[0], ref_pos[1] - pt[1], ref_pos[2] - pt[2]};
  const Se3 T_cur_ref = se3_compose(Tc, se3_inverse(Tr));
  const double rf[3] = {a.obs.d_f[3 * best], a.obs.d_f[3 * best + 1], a.obs.d_f[3 * best + 2]};
  double A[4];
  warp_matrix_affine(a.cam, rpx, rf, norm3(d), T_cur_ref, rlevel, A);
  const int sl = best_search_level(A, a.n_pyr_levels - 1);
  a.search_level[m] = sl;
  if (a.A_cur_ref) {
    a.A_cur_ref[4 * m] = A[0]; a.A_cur_ref[4 * m + 1] = A[1]; a.A_cur_ref[4 * m + 2] = A[2]; a.A_cur_ref[4 * m + 3] = A[3];
  }
  double Ainv[4];
  inv2<double>(A, Ainv);
  a.A_ref_cur[4 * m] = (float)Ainv[0];
  a.A_ref_cur[4 * m + 1] = (float)Ainv[1];
  a.A_ref_cur[4 * m + 2] = (float)Ainv[2];
  a.A_ref_cur[4 * m + 3] = (float)Ainv[3];
  a.px_ref_pyr[2 * m] = (float)rpx[0] * pow2_inv_f32(rlevel);  // (/ 2^level, same bits)
  a.px_ref_pyr[2 * m + 1] = (float)rpx[1] * pow2_inv_f32(rlevel);
  a.ref_slot[m] = a.frame_slot[rfi];
  a.ref_level[m] = rlevel;
  a.px_scaled[2 * m] = a.px_cur[2 * m] * pow2_inv_f64(sl);
  a.px_scaled[2 * m + 1] = a.px_cur[2 * m + 1] * pow2_inv_f64(sl);
  const bool edgelet = a.obs.d_type && a.obs.d_type[best] == SVO_HIP_FTR_EDGELET;
  if (edgelet) {
    const double gx = a.obs.d_grad[2 * best], gy = a.obs.d_grad[2 * best + 1];
    double dir_cur[2] = {A[0] * gx + A[1] * gy, A[2] * gx + A[3] * gy};
    const double n = norm2(dir_cur);
    dir_cur[0] /= n;
    dir_cur[1] /= n;
    a.dir[2 * m] = (float)dir_cur[0];
    a.dir[2 * m + 1] = (float)dir_cur[1];
    a.use_1d[m] = 1;
  }
  a.active[m] = 1;
}

// warp::warpAffine (matcher.cpp:72-105), halfpatch_size = 5: a wave owns EIGHT trials, 8 lanes each (warp_group.h: the
// source region through LDS, 13 sample slots per lane, the patch assembled in LDS); a group stores its 100 bytes as 25 dwords.
// Arithmetic per sample unchanged (bit-identical patches).  The depth filter no longer comes here: its scan kernel warps
// its own seeds with the same function (epi_scan.h).
constexpr int WARP_TPW = 8;   // trials per wave
constexpr int WARP_MINW = 4;  // waves per SIMD the register budget is held to
constexpr int WARP_WGS_PER_CU = 16;
__global__ void __launch_bounds__(256, WARP_MINW) warp_kernel(const WarpArgs a) {
  __shared__ long long s_off[SVO_HIP_MAX_LEVELS];
  __shared__ int s_w[SVO_HIP_MAX_LEVELS], s_h[SVO_HIP_MAX_LEVELS], s_p[SVO_HIP_MAX_LEVELS];
  __shared__ __attribute__((aligned(16))) uint32_t s_lds[4 * WARP_TPW][WG_BOX_DWORDS + WG_PATCH_DWORDS];
  if (threadIdx.x < SVO_HIP_MAX_LEVELS) {
    s_off[threadIdx.x] = a.L.offset[threadIdx.x];
    s_w[threadIdx.x] = a.L.w[threadIdx.x];
    s_h[threadIdx.x] = a.L.h[threadIdx.x];
    s_p[threadIdx.x] = a.L.pitch[threadIdx.x];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = lane >> 3, gl = lane & 7;
  uint32_t* const region = s_lds[threadIdx.x >> 3];
  uint32_t* const patch = region + WG_BOX_DWORDS;
  const int M = a.M_dev ? min(*a.M_dev, a.M) : a.M;
  const long long n_wave_groups = ((long long)M + WARP_TPW - 1) / WARP_TPW;
  // An iteration of a wave is a chain of two memory round trips -- the parameters of its trials, then their source
  // regions -- and the arithmetic: the parameters of the NEXT iteration are requested before this iteration's work starts.
  struct TrialParams {
    float4 A;
    float2 pyr;
    int act, level, slot, slev;
  };
  auto load_params = [&](long long gw) {  // (the 8 lanes of a trial read the same words)
    TrialParams p;
    p.A = make_float4(0.f, 0.f, 0.f, 0.f);
    p.pyr = make_float2(0.f, 0.f);
    p.act = p.level = p.slot = p.slev = 0;
    const long long m = gw * WARP_TPW + t;
    if (m < M) {
      p.A = *reinterpret_cast<const float4*>(a.A_ref_cur + 4 * (size_t)m);
      p.pyr = *reinterpret_cast<const float2*>(a.px_ref_pyr + 2 * (size_t)m);
      p.act = a.active[m];
      p.level = a.ref_level[m];  // (masked where it is used: nothing here may wait for a load)
      p.slot = a.ref_slot[m];
      p.slev = a.search_level[m];
    }
    return p;
  };
  const long long gw_step = (long long)gridDim.x * 4;
  long long gw = (long long)blockIdx.x * 4 + wave;
  TrialParams nxt = load_params(gw < n_wave_groups ? gw : 0);
  for (; gw < n_wave_groups; gw += gw_step) {
    const long long m = gw * WARP_TPW + t;
    const TrialParams cur = nxt;
    if (gw + gw_step < n_wave_groups) nxt = load_params(gw + gw_step);
    if (m < M) {
      const int level = cur.level & (SVO_HIP_MAX_LEVELS - 1);
      const uint8_t* img = a.store + (int64_t)cur.slot * a.L.slot_bytes + s_off[level];
      // an inactive trial's patch is zeros (a NaN warp likewise: warp_group.h), as in rounds 1-5
      const float Ax = cur.act ? cur.A.x : __builtin_nanf("");
      warp_patch_group8(img, s_w[level], s_h[level], s_p[level], Ax, cur.A.y, cur.A.z, cur.A.w, cur.pyr.x, cur.pyr.y,
                        cur.slev & 31, gl, region, patch);
      warp_patch_store_group8(patch, gl, a.pwb + (size_t)m * 100);
    }
  }
}

struct ReprojArgs {
  Cam cam;
  int M;
  const double* frame_T;
  const int32_t* cur_frame;
  const double* pt_pos;
  int cell_size, grid_n_cols;
  int32_t* cell;
  double* px;
};
__global__ void __launch_bounds__(256) reproject_kernel(const ReprojArgs a) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  Se3 T;
  se3_from_Rt(a.frame_T + 12 * a.cur_frame[m], T);
  const double pos[3] = {a.pt_pos[3 * m], a.pt_pos[3 * m + 1], a.pt_pos[3 * m + 2]};
  double p[3], px[2];
  se3_apply(T, pos, p);
  world2cam(a.cam, p, px);
  int k = -1;
  if (is_in_frame(a.cam, cast_int(px[0]), cast_int(px[1]), 8))
    k = cast_int(px[1] / a.cell_size) * a.grid_n_cols + cast_int(px[0] / a.cell_size);
  a.cell[m] = k;
  if (a.px) {
    a.px[2 * m] = px[0];
    a.px[2 * m + 1] = px[1];
  }
}

struct GlueArgs {
  Cam cam;
  int n;
  const double* A;
  const double* B;
  double* out;
  const int32_t* out_index;
  const double* px;
  double* f;
};
__global__ void __launch_bounds__(256) compose_kernel(const GlueArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  Se3 x, y;
  se3_from_Rt(a.A + 12 * i, x);
  se3_from_Rt(a.B + 12 * i, y);
  const Se3 r = se3_compose(x, y);
  const int o = a.out_index ? a.out_index[i] : i;
  se3_to_Rt(r, a.out + 12 * o);
}
// svo_hip_frame_pose_compose: lane 0 forms the pose (a dozen dependent f64 operations; what matters is that it sits on the
// stream), then lane i ranks keyframe i among the keyframes whose field of view overlaps the frame's: Map::getCloseKeyframes
// (svo/src/map.cpp:106-131: the first of a keyframe's key points that Frame::isVisible, frame.cpp:115-123, puts it on the
// list with the distance of the two T_f_w translations) followed by the reprojector's closest-first sort and its cut at
// max_n_kfs (reprojector.cpp:78-84; std::list::sort is stable: equal distances keep the map's order).
struct FramePoseArgs {
  const double *T_cur_ref, *q_ref, *t_ref;
  double *frame_T, *T_copy, *T_out;
  int cur_frame;
  int32_t* signal;
  int32_t signal_value;
  Cam cam;
  int n_frames, n_kf, max_n_kfs;
  const double* key_pos;     // [n_kf][5][3]
  const uint8_t* key_valid;  // [n_kf][5]
  int32_t *rank, *rank_out;  // [n_frames]
};
__global__ void __launch_bounds__(64) frame_pose_compose_kernel(const FramePoseArgs a) {
  __shared__ Se3 s_T;
  __shared__ double s_dist[64];
  const int i = (int)threadIdx.x;
  // what the ranking reads does not depend on the pose: the five key points and the translation of keyframe i are requested
  // before lane 0 starts on the product (one memory round trip, hidden behind its chain of divisions and square roots)
  const bool ranks = a.rank != nullptr, mine = ranks && i < a.n_kf;  // (keyframes come first in the table: i != cur_frame)
  uint8_t valid[5] = {0, 0, 0, 0, 0};
  double kp[5][3], tk[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    valid[k] = mine ? a.key_valid[5 * i + k] : (uint8_t)0;
#pragma unroll
    for (int c = 0; c < 3; ++c) kp[k][c] = mine ? a.key_pos[3 * (5 * i + k) + c] : 0.0;
  }
  if (mine) {
#pragma unroll
    for (int c = 0; c < 3; ++c) tk[c] = a.frame_T[12 * i + 9 + c];
  }
  if (i == 0) {
    Se3 x, y;
    se3_from_Rt(a.T_cur_ref, x);  // SE3(R, t): the quaternion of the rotation matrix (Eigen's Quaternion(Matrix3))
#pragma unroll
    for (int k = 0; k < 4; ++k) y.q[k] = a.q_ref[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) y.t[k] = a.t_ref[k];
    const Se3 r = se3_compose(x, y);
    s_T = r;
    double T[12];
    se3_to_Rt(r, T);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      a.frame_T[12 * a.cur_frame + k] = T[k];
      if (a.T_copy) a.T_copy[k] = T[k];
      if (a.T_out) a.T_out[k] = T[k];
    }
  }
  if (ranks) {  // (uniform)
    __syncthreads();
    const Se3 T = s_T;  // the pose as the host's object holds it: quaternion and translation, not re-derived from R
    double dist = -1.0;   // < 0: no key point of the keyframe is visible
    bool found = false;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (found || !valid[k]) continue;
      double xyz_f[3], px[2];
      se3_apply(T, kp[k], xyz_f);
      if (xyz_f[2] < 0.0) continue;  // behind the camera
      world2cam(a.cam, xyz_f, px);
      if (px[0] >= 0.0 && px[1] >= 0.0 && px[0] < (double)a.cam.width && px[1] < (double)a.cam.height) {
        const double d[3] = {T.t[0] - tk[0], T.t[1] - tk[1], T.t[2] - tk[2]};
        dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        found = true;
      }
    }
    s_dist[i] = dist;
    __syncthreads();
    if (i < a.n_frames) {
      int r = -1;
      if (dist >= 0.0) {
        r = 0;
        for (int j = 0; j < a.n_kf; ++j) {
          const double dj = s_dist[j];
          if (dj >= 0.0 && (dj < dist || (dj == dist && j < i))) ++r;
        }
        if (r >= a.max_n_kfs) r = -1;
      }
      a.rank[i] = r;
      if (a.rank_out) a.rank_out[i] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (system scope) every lane's rank is out before lane 0 signals
    __syncthreads();
  }
  if (i == 0 && a.signal) __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) cam2world_kernel(const GlueArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  double f[3];
  cam2world(a.cam, a.px[2 * i], a.px[2 * i + 1], f);
  a.f[3 * i] = f[0];
  a.f[3 * i + 1] = f[1];
  a.f[3 * i + 2] = f[2];
}

// ---- the selection rule of Reprojector::reprojectMap's cell loop over a batch of trials ----------------------------
// reprojector.cpp:131-139 visits the cells in grid_.cell_order and takes, per cell, the first candidate of the sorted
// list whose findMatchDirect succeeds (reprojectCell, :150-200); it stops once more than maxFts cells have matched.  The
// trials arrive in that visiting order (those of one cell next to each other), so "first success of its cell" is a look
// back over the cell's run and the position in Frame::fts_ is the number of selected trials before it.  A selected
// trial becomes the observation the pose optimizer reads: Feature(frame, px, level) with f = cam2world(px)
// (feature.h:44-52), point->pos_.
struct SelectArgs {
  Cam cam;
  int M;
  const int32_t* cell;
  const int32_t* ok;
  const double* px;
  const int32_t* level;
  const double* pos;
  int max_selected;
  int32_t* n;
  int32_t* sel;
  double* f;
  int32_t* level_out;
  double* pos_out;
  uint8_t* has_point;
  int32_t* signal;  // or NULL
  int32_t signal_value;
  const int32_t* M_dev;  // NULL, or the batch size lives on the device (min(*M_dev, M))
};
// NT threads take NT trials per pass.  The match results may live in host-mapped memory (the single-stream drop-in), where
// every pass costs a round trip over the link: batches of more than 256 trials run with 1024 threads (one pass for the
// ~700 trials of a frame on a full map instead of three).
template <int NT>
__global__ void __launch_bounds__(NT) match_select_kernel(const SelectArgs a) {
  __shared__ int s_wave[NT / 64];
  __shared__ int s_ok[NT], s_cell[NT];
  __shared__ int s_carry[2];  // cell of the last trial of the pass before, and whether that cell has matched already
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // everything enqueued before this kernel has completed (stream order): tell a host that polls mapped memory
  if (a.signal && tid == 0) __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (tid == 0) { s_carry[0] = -1; s_carry[1] = 0; }
  const int M = a.M_dev ? min(*a.M_dev, a.M) : a.M;
  int base = 0;  // selected trials before this pass (the same in every thread)
  for (int m0 = 0; m0 < M && base < a.max_selected; m0 += NT) {
    const int m = m0 + tid;
    const bool valid = m < M;
    // every global read of the pass is issued here, before anything depends on one: the match results may live in
    // host-mapped memory, where a dependent chain of loads costs a link round trip per link
    const int ok = valid ? (a.ok[m] != 0) : 0;
    const int cell = valid ? a.cell[m] : -2;
    const double u = valid ? a.px[2 * m] : 0.0, v = valid ? a.px[2 * m + 1] : 0.0;
    const int lvl = valid ? a.level[m] : 0;
    double pos[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[k] = valid ? a.pos[3 * m + k] : 0.0;
    s_ok[tid] = ok;
    s_cell[tid] = cell;
    __syncthreads();
    // first success of its cell: nothing matched among the cell's earlier trials (this pass: LDS; earlier passes: carry)
    bool first = ok != 0;
    if (first) {
      int j = tid - 1;
      for (; j >= 0 && s_cell[j] == cell; --j)
        if (s_ok[j]) { first = false; break; }
      if (first && j < 0 && cell == s_carry[0] && s_carry[1]) first = false;
    }
    int carry_cell = 0, carry_matched = 0;
    if (tid == NT - 1 && m0 + NT < M) {  // only when another pass follows (a partial pass ends in idle lanes that all share
                                         // the cell id -2: one lane walking back over them cost 80 us at NT = 1024)
      carry_cell = cell;
      carry_matched = ok;
      int j = NT - 2;
      for (; !carry_matched && j >= 0 && s_cell[j] == cell; --j) carry_matched = s_ok[j];
      if (!carry_matched && j < 0 && cell == s_carry[0]) carry_matched = s_carry[1];
    }
    const unsigned long long b = __ballot(first);
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int rank = base + __popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += s_wave[w];
    for (int w = 0; w < NT / 64; ++w) base += s_wave[w];
    if (tid == NT - 1) { s_carry[0] = carry_cell; s_carry[1] = carry_matched; }
    __syncthreads();
    if (first && rank < a.max_selected) {
      double f[3];
      cam2world(a.cam, u, v, f);
      a.sel[rank] = m;
      a.level_out[rank] = lvl;
      a.has_point[rank] = 1;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a.f[3 * rank + k] = f[k];
        a.pos_out[3 * rank + k] = pos[k];
      }
    }
  }
  if (tid == 0) a.n[0] = base < a.max_selected ? base : a.max_selected;
}

}  // namespace

namespace svo_track {
int launch_warp(const WarpArgs& a, hipStream_t s) {
  if (a.M <= 0) return SVO_HIP_OK;
  // 4 waves x WARP_TPW trials per workgroup pass; at most WARP_WGS_PER_CU workgroups per CU walk over them
  const long long n_groups = ((long long)a.M + 4 * WARP_TPW - 1) / (4 * WARP_TPW);
  const long long cap = 256ll * WARP_WGS_PER_CU;
  hipLaunchKernelGGL(warp_kernel, dim3((unsigned)(n_groups < cap ? n_groups : cap)), dim3(256), 0, s, a);
  return check_launch();
}
}  // namespace svo_track

extern "C" size_t svo_hip_match_workspace_bytes(int M) {
  if (M < 0) return 0;
  const size_t m = (size_t)M;
  // generous: every per-trial scratch array of the matcher and of the depth filter
  size_t b = 0;
  b += Carver::round(m * 100);                 // patches
  b += 9 * Carver::round(m * sizeof(int32_t)); // slots, levels, flags, the alignment's evaluation counts
  b += 4 * Carver::round(m);                   // u8 flags
  b += Carver::round(m * 4 * sizeof(float)) + 2 * Carver::round(m * 2 * sizeof(float));
  b += 8 * Carver::round(m * 2 * sizeof(double));  // px arrays, epipolar geometry
  b += 4 * Carver::round(m * sizeof(double));
  b += Carver::round(m * 12 * sizeof(double));     // T_cur_ref (quaternion + t padded)
  b += align_phase_workspace_bytes(M) + 256;       // queues + parked loop state of the phased alignment
  b += Carver::round(m * sizeof(uint16_t));        // (spare)
  b += Carver::round(m * sizeof(int32_t)) + Carver::round(m * 14 * sizeof(double));  // the depth filter's pair poses (round 6)
  return b + 4096;
}

static int find_match_direct_impl(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                  const svo_hip_camera* cam, const svo_hip_frames* frames, int M, const int32_t* d_M,
                                  const int32_t* d_cur_frame, const double* d_pt_pos,
                                  const int32_t* d_obs_ptr, const int32_t* d_obs_begin, const int32_t* d_obs_end,
                                  const svo_hip_features* obs, int n_pyr_levels,
                                  int align_max_iter, double* d_px_cur, int32_t* d_ok, int32_t* d_ref_obs,
                                  int32_t* d_search_level, double* d_A_cur_ref, uint8_t* d_patch_out,
                                  void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!layout_ok(layout) || !d_store || !cam || !cam_model_ok(cam) || !frames || !obs || M < 0) return SVO_HIP_EINVAL;
  if (M == 0) return SVO_HIP_OK;
  if (!d_cur_frame || !d_pt_pos || !(d_obs_ptr || (d_obs_begin && d_obs_end)) || !d_px_cur || !d_ok || !d_ref_obs || !d_search_level ||
      !frames->d_slot || !frames->d_T_f_w || !obs->d_frame || !obs->d_level || !obs->d_px || !obs->d_f)
    return SVO_HIP_EINVAL;
  if (obs->d_type && !obs->d_grad) return SVO_HIP_EINVAL;
  if (n_pyr_levels < 1 || n_pyr_levels > layout->n_levels || align_max_iter < 0) return SVO_HIP_EINVAL;
  if (!d_workspace || workspace_bytes < svo_hip_match_workspace_bytes(M)) return SVO_HIP_ERANGE;
  if (d_patch_out && (reinterpret_cast<uintptr_t>(d_patch_out) & 3)) return SVO_HIP_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver ws(d_workspace, workspace_bytes);
  const size_t m = (size_t)M;
  PrepArgs p;
  p.cam = make_cam(cam);
  p.M = M;
  p.M_dev = d_M;
  p.n_pyr_levels = n_pyr_levels;
  p.frame_slot = frames->d_slot;
  p.frame_T = frames->d_T_f_w;
  p.cur_frame = d_cur_frame;
  p.pt_pos = d_pt_pos;
  p.obs_ptr = d_obs_ptr;
  p.obs_begin = d_obs_begin;
  p.obs_end = d_obs_end;
  p.obs = *obs;
  p.px_cur = d_px_cur;
  p.ref_obs = d_ref_obs;
  p.search_level = d_search_level;
  p.A_cur_ref = d_A_cur_ref;
  uint8_t* pwb = d_patch_out ? d_patch_out : ws.take<uint8_t>(m * 100);
  p.active = ws.take<uint8_t>(m);
  p.ref_slot = ws.take<int32_t>(m);
  p.ref_level = ws.take<int32_t>(m);
  p.cur_slot = ws.take<int32_t>(m);
  p.A_ref_cur = ws.take<float>(4 * m);
  p.px_ref_pyr = ws.take<float>(2 * m);
  p.dir = ws.take<float>(2 * m);
  p.use_1d = ws.take<uint8_t>(m);
  p.px_scaled = ws.take<double>(2 * m);
  if (!ws.ok || !pwb) return SVO_HIP_ERANGE;
  hipLaunchKernelGGL(match_prepare_kernel, dim3((M + 63) / 64), dim3(64), 0, s, p);
  int rc = check_launch();
  if (rc) return rc;
  WarpArgs w;
  w.L = *layout;
  w.store = d_store;
  w.M = M;
  w.active = p.active;
  w.ref_slot = p.ref_slot;
  w.ref_level = p.ref_level;
  w.search_level = d_search_level;
  w.A_ref_cur = p.A_ref_cur;
  w.px_ref_pyr = p.px_ref_pyr;
  w.pwb = pwb;
  w.M_dev = d_M;
  rc = launch_warp(w, s);
  if (rc) return rc;
  AlignArgs al;
  al.L = *layout;
  al.store = d_store;
  al.M = M;
  al.slot = p.cur_slot;
  al.level = d_search_level;
  al.pwb = pwb;
  al.dir = p.dir;
  al.use_1d = p.use_1d;
  al.active = p.active;
  al.n_iter = align_max_iter;
  al.px_in = p.px_scaled;
  al.px_out = d_px_cur;
  al.scale_out = 1;
  al.ok = d_ok;
  al.h_inv = nullptr;
  al.M_dev = d_M;
  const size_t phase_bytes = d_M ? 0 : align_phase_workspace_bytes(M);  // (phases compact by queues sized from M)
  void* phase_ws = phase_bytes ? ws.take<uint8_t>(phase_bytes) : nullptr;
  return launch_align(al, s, ws.ok ? phase_ws : nullptr, phase_bytes);
}

extern "C" int svo_hip_find_match_direct(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                         const svo_hip_camera* cam, const svo_hip_frames* frames, int M,
                                         const int32_t* d_cur_frame, const double* d_pt_pos,
                                         const int32_t* d_obs_ptr, const svo_hip_features* obs, int n_pyr_levels,
                                         int align_max_iter, double* d_px_cur, int32_t* d_ok, int32_t* d_ref_obs,
                                         int32_t* d_search_level, double* d_A_cur_ref, uint8_t* d_patch_out,
                                         void* d_workspace, size_t workspace_bytes, void* stream) {
  if (M > 0 && !d_obs_ptr) return SVO_HIP_EINVAL;
  return find_match_direct_impl(layout, d_store, cam, frames, M, nullptr, d_cur_frame, d_pt_pos, d_obs_ptr, nullptr, nullptr, obs,
                                n_pyr_levels, align_max_iter, d_px_cur, d_ok, d_ref_obs, d_search_level, d_A_cur_ref, d_patch_out,
                                d_workspace, workspace_bytes, stream);
}

extern "C" int svo_hip_find_match_direct_indirect(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                                  const svo_hip_camera* cam, const svo_hip_frames* frames, int M_cap,
                                                  const int32_t* d_M, const int32_t* d_cur_frame, const double* d_pt_pos,
                                                  const int32_t* d_obs_begin, const int32_t* d_obs_end,
                                                  const svo_hip_features* obs, int n_pyr_levels, int align_max_iter,
                                                  double* d_px_cur, int32_t* d_ok, int32_t* d_ref_obs, int32_t* d_search_level,
                                                  double* d_A_cur_ref, uint8_t* d_patch_out, void* d_workspace,
                                                  size_t workspace_bytes, void* stream) {
  if (M_cap < 0 || M_cap > 65536 || !d_M || (M_cap > 0 && (!d_obs_begin || !d_obs_end))) return SVO_HIP_EINVAL;
  return find_match_direct_impl(layout, d_store, cam, frames, M_cap, d_M, d_cur_frame, d_pt_pos, nullptr, d_obs_begin, d_obs_end, obs,
                                n_pyr_levels, align_max_iter, d_px_cur, d_ok, d_ref_obs, d_search_level, d_A_cur_ref, d_patch_out,
                                d_workspace, workspace_bytes, stream);
}

extern "C" int svo_hip_reproject_points(const svo_hip_camera* cam, const svo_hip_frames* frames, int M,
                                        const int32_t* d_cur_frame, const double* d_pt_pos, int cell_size,
                                        int grid_n_cols, int32_t* d_cell, double* d_px, void* stream) {
  if (!cam || !cam_model_ok(cam) || !frames || M < 0 || cell_size < 1 || grid_n_cols < 1) return SVO_HIP_EINVAL;
  if (M == 0) return SVO_HIP_OK;
  if (!d_cur_frame || !d_pt_pos || !d_cell || !frames->d_T_f_w) return SVO_HIP_EINVAL;
  ReprojArgs a;
  a.cam = make_cam(cam);
  a.M = M;
  a.frame_T = frames->d_T_f_w;
  a.cur_frame = d_cur_frame;
  a.pt_pos = d_pt_pos;
  a.cell_size = cell_size;
  a.grid_n_cols = grid_n_cols;
  a.cell = d_cell;
  a.px = d_px;
  hipLaunchKernelGGL(reproject_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

extern "C" int svo_hip_compose_poses(int n, const double* d_A, const double* d_B, double* d_out,
                                     const int32_t* d_out_index, void* stream) {
  if (n < 0) return SVO_HIP_EINVAL;
  if (n == 0) return SVO_HIP_OK;
  if (!d_A || !d_B || !d_out) return SVO_HIP_EINVAL;
  GlueArgs a{};
  a.n = n;
  a.A = d_A;
  a.B = d_B;
  a.out = d_out;
  a.out_index = d_out_index;
  hipLaunchKernelGGL(compose_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

extern "C" int svo_hip_frame_pose_compose(const double* d_T_cur_ref, const double* d_q_ref, const double* d_t_ref, double* d_frame_T,
                                          int cur_frame, double* d_T_copy, double* d_T_out, const svo_hip_camera* cam, int n_frames,
                                          int n_kf, const double* d_key_pos, const uint8_t* d_key_valid, int max_n_kfs,
                                          int32_t* d_rank, int32_t* d_rank_out, int32_t* d_signal, int32_t signal_value,
                                          void* stream) {
  if (!d_T_cur_ref || !d_q_ref || !d_t_ref || !d_frame_T || cur_frame < 0) return SVO_HIP_EINVAL;
  FramePoseArgs a{};
  a.T_cur_ref = d_T_cur_ref; a.q_ref = d_q_ref; a.t_ref = d_t_ref;
  a.frame_T = d_frame_T; a.T_copy = d_T_copy; a.T_out = d_T_out;
  a.cur_frame = cur_frame;
  a.signal = d_signal;
  a.signal_value = signal_value;
  if (d_rank != nullptr) {
    if (!cam || !cam_model_ok(cam) || n_frames < 1 || n_frames > 64 || n_kf < 0 || n_kf > n_frames || cur_frame >= n_frames ||
        (n_kf > 0 && (!d_key_pos || !d_key_valid)) || max_n_kfs < 0)
      return SVO_HIP_EINVAL;
    a.cam = make_cam(cam);
    a.n_frames = n_frames; a.n_kf = n_kf; a.max_n_kfs = max_n_kfs;
    a.key_pos = d_key_pos; a.key_valid = d_key_valid;
    a.rank = d_rank; a.rank_out = d_rank_out;
  }
  hipLaunchKernelGGL(frame_pose_compose_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

extern "C" int svo_hip_cam2world(const svo_hip_camera* cam, int n, const double* d_px, double* d_f, void* stream) {
  if (!cam || !cam_model_ok(cam) || n < 0) return SVO_HIP_EINVAL;
  if (n == 0) return SVO_HIP_OK;
  if (!d_px || !d_f) return SVO_HIP_EINVAL;
  GlueArgs a{};
  a.cam = make_cam(cam);
  a.n = n;
  a.px = d_px;
  a.f = d_f;
  hipLaunchKernelGGL(cam2world_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

static int select_matches_impl(const svo_hip_camera* cam, int M, const int32_t* d_M, const int32_t* d_cell, const int32_t* d_ok,
                               const double* d_px, const int32_t* d_level, const double* d_pos, int max_fts,
                               int32_t* d_n, int32_t* d_sel, double* d_f, int32_t* d_level_out, double* d_pos_out,
                               uint8_t* d_has_point, int32_t* d_signal, int32_t signal_value, void* stream) {
  if (!cam || !cam_model_ok(cam) || M < 0 || max_fts < 0 || !d_n) return SVO_HIP_EINVAL;
  if (M > 0 && (!d_cell || !d_ok || !d_px || !d_level || !d_pos || !d_sel || !d_f || !d_level_out || !d_pos_out || !d_has_point))
    return SVO_HIP_EINVAL;
  SelectArgs a{};
  a.cam = make_cam(cam);
  a.M = M;
  a.M_dev = d_M;
  a.cell = d_cell;
  a.ok = d_ok;
  a.px = d_px;
  a.level = d_level;
  a.pos = d_pos;
  a.max_selected = max_fts + 1;  // "if (n_matches_ > maxFts) break" lets the (maxFts+1)-th match in (:137-138)
  a.n = d_n;
  a.sel = d_sel;
  a.f = d_f;
  a.level_out = d_level_out;
  a.pos_out = d_pos_out;
  a.has_point = d_has_point;
  a.signal = d_signal;
  a.signal_value = signal_value;
  // (M == 0: n = 0)
  if (M <= 256) hipLaunchKernelGGL(match_select_kernel<256>, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(match_select_kernel<1024>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

extern "C" int svo_hip_select_matches(const svo_hip_camera* cam, int M, const int32_t* d_cell, const int32_t* d_ok,
                                      const double* d_px, const int32_t* d_level, const double* d_pos, int max_fts,
                                      int32_t* d_n, int32_t* d_sel, double* d_f, int32_t* d_level_out, double* d_pos_out,
                                      uint8_t* d_has_point, int32_t* d_signal, int32_t signal_value, void* stream) {
  return select_matches_impl(cam, M, nullptr, d_cell, d_ok, d_px, d_level, d_pos, max_fts, d_n, d_sel, d_f, d_level_out, d_pos_out,
                             d_has_point, d_signal, signal_value, stream);
}

extern "C" int svo_hip_select_matches_indirect(const svo_hip_camera* cam, int M_cap, const int32_t* d_M, const int32_t* d_cell,
                                               const int32_t* d_ok, const double* d_px, const int32_t* d_level,
                                               const double* d_pos, int max_fts, int32_t* d_n, int32_t* d_sel, double* d_f,
                                               int32_t* d_level_out, double* d_pos_out, uint8_t* d_has_point, int32_t* d_signal,
                                               int32_t signal_value, void* stream) {
  if (!d_M) return SVO_HIP_EINVAL;
  return select_matches_impl(cam, M_cap, d_M, d_cell, d_ok, d_px, d_level, d_pos, max_fts, d_n, d_sel, d_f, d_level_out, d_pos_out,
                             d_has_point, d_signal, signal_value, stream);
}
