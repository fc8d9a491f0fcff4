// map_mirror.hip -- row N2: Reprojector::reprojectMap up to the first findMatchDirect, on a device-resident mirror
// of the map (svo/src/reprojector.cpp:64-142, 151-153, 206-217; Point::getCloseViewObs, svo/src/point.cpp:97-117).
//
// The reference walks Map -> keyframes -> Frame::fts_ -> Point (and MapPointCandidates::candidates_) on the host for
// every frame: a reprojectPoint per point (w2c, isInFrame, a std::list node pushed into the point's grid cell), a
// stable list sort per cell, getCloseViewObs per candidate it gets to try.  Here the records that walk reads live in
// HBM (svo_hip_map, patched incrementally by the host) and ONE workgroup does the walk for a frame:
//
//   patch -> project every live point once (at the place the reference's keyframe loop meets it first) -> count per
//   cell (LDS atomics) -> scan over the cells in VISITING order -> scatter into cell buckets -> rank inside the bucket
//   by (type descending, binning order) -> close-view test per candidate -> cells until `max_cells_with_trials` hold a
//   trial -> visit list + trial list in the order reprojectCell would walk them.
//
// A single frame is ~2000 points: the kernel is latency-bound (a dozen barriers), not throughput-bound; what it buys
// is that the match kernels follow on the stream with no host pass in between ("Row N2").
// The projection and the close-view arithmetic are the ones of reproject_kernel / match_prepare_kernel (matcher.hip).
#pragma clang fp contract(off)
#include "track_kernels.h"
#include "track_math.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_track;

namespace {

constexpr int RM_BLOCK = 1024;
constexpr int RM_MAX_E = SVO_HIP_REPROJ_MAX_IN_FRAME;  // points inside the frame
constexpr int RM_MAX_CELLS = SVO_HIP_REPROJ_MAX_CELLS;
constexpr int RM_MAX_FRAMES = 64;
constexpr int RM_EPT = RM_MAX_E / RM_BLOCK;  // sorted elements per thread in the ordered phases

struct ReprojMapArgs {
  Cam cam;
  int n_frames, cur_frame;
  const double* frame_T;
  const int32_t* kf_rank;
  svo_hip_map map;
  svo_hip_map_patch patch;
  svo_hip_grid grid;
  int first_cell, max_cells_with_trials, max_visits, max_trials;
  svo_hip_reprojection out;
};

// exclusive scan of one int per thread over the workgroup; *total = sum.  Two barriers.
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave /*[RM_BLOCK/64 + 1]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < RM_BLOCK / 64; ++w) {
      const int t = s_wave[w];
      s_wave[w] = run;
      run += t;
    }
    s_wave[RM_BLOCK / 64] = run;
  }
  __syncthreads();
  *total = s_wave[RM_BLOCK / 64];
  const int r = s_wave[wave] + inc - v;
  __syncthreads();  // s_wave is reused by the next scan
  return r;
}

// Where the keyframe loop meets a map point first: the smallest (keyframe rank, position in that keyframe's fts_) over
// the point's observations in overlapping keyframes.  An observation record's d_obs_order word is Feature::frame << 16
// | position (0xffff: in no keyframe's list), written by the patch step.  The records of a point are contiguous: up to
// eight are fetched by independent loads before any is looked at (one memory round trip, not one per record).
__device__ __forceinline__ uint32_t first_meeting(const int32_t* __restrict__ obs_word, int o0, int n, const int* s_kfrank,
                                                  int* first_frame) {
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = k < n ? (uint32_t)obs_word[o0 + k] : 0xffffffffu;
  uint32_t best = 0xffffffffu;
  int ff = -1;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t fr = w[k] >> 16, ord = w[k] & 0xffffu;
    if (k < n && ord != 0xffffu) {
      // (a stale record whose frame field lies beyond the table must not alias a live slot: it never wins)
      const int r = fr < (uint32_t)RM_MAX_FRAMES ? s_kfrank[fr] : -1;
      const uint32_t key = (uint32_t)r << 12 | (ord & 0xfffu);
      if (r >= 0 && key < best) { best = key; ff = (int)fr; }
    }
  }
  for (int k = 8; k < n; ++k) {  // (a point seen from more than eight keyframes)
    const uint32_t x = (uint32_t)obs_word[o0 + k];
    const uint32_t fr = x >> 16, ord = x & 0xffffu;
    if (ord != 0xffffu) {
      const int r = fr < (uint32_t)RM_MAX_FRAMES ? s_kfrank[fr] : -1;
      const uint32_t key = (uint32_t)r << 12 | (ord & 0xfffu);
      if (r >= 0 && key < best) { best = key; ff = (int)fr; }
    }
  }
  *first_frame = ff;  // < RM_MAX_FRAMES when >= 0: it indexes s_kfcount
  return best;
}

// PPT: map entries per thread (n_points <= PPT * RM_BLOCK): what a thread learns about its entries in the projection
// phase -- cell, sort key -- stays in registers for the scatter phase.
template <int PPT>
__global__ void __launch_bounds__(RM_BLOCK) reproject_map_kernel(const ReprojMapArgs a) {
  __shared__ int s_cnt[RM_MAX_CELLS];    // per visiting rank: points binned; later: 1 where the cell holds a trial
  __shared__ int s_base[RM_MAX_CELLS + 1];  // exclusive scan of s_cnt
  __shared__ uint32_t s_bkey[RM_MAX_E];  // bucketed: (3 - type) << 28 | binning order
  __shared__ uint16_t s_bidx[RM_MAX_E];  //           map entry
  __shared__ uint16_t s_sidx[RM_MAX_E];  // sorted:   map entry
  __shared__ uint16_t s_srank[RM_MAX_E]; //           visiting rank of its cell
  __shared__ double s_fpos[RM_MAX_FRAMES][3];  // Frame::pos() of the frame table
  __shared__ int s_kfcount[RM_MAX_FRAMES], s_kfrank[RM_MAX_FRAMES];
  __shared__ int s_wave[RM_BLOCK / 64 + 1];
  __shared__ int s_E, s_end_cell;
  const int tid = threadIdx.x;
  const svo_hip_map& mp = a.map;
  const int P = mp.n_points, n_cells = a.grid.n_cells;

  // ---- 0. the host's changes since the last call -------------------------------------------------------------------
  // Every field of a record is read before the first store: the loops interleave loads and stores to arrays the compiler
  // cannot prove distinct, so a load placed after a store is waited for on its own -- one memory round trip each, on the
  // critical path of a single-stream frame.
  for (int i = tid; i < a.patch.n_obs; i += RM_BLOCK) {
    const int o = a.patch.d_obs_index[i];
    const int fr = a.patch.obs.d_frame[i], ord = a.patch.d_obs_order[i];
    const int lv = a.patch.obs.d_level[i];
    const uint8_t ty = a.patch.obs.d_type[i];
    const double px0 = a.patch.obs.d_px[2 * i], px1 = a.patch.obs.d_px[2 * i + 1];
    const double f0 = a.patch.obs.d_f[3 * i], f1 = a.patch.obs.d_f[3 * i + 1], f2 = a.patch.obs.d_f[3 * i + 2];
    const double g0 = a.patch.obs.d_grad[2 * i], g1 = a.patch.obs.d_grad[2 * i + 1];
    mp.d_obs_frame[o] = fr;
    mp.d_obs_order[o] = (int32_t)((uint32_t)fr << 16 | (ord < 0 ? 0xffffu : (uint32_t)ord & 0xffffu));
    mp.d_obs_level[o] = lv;
    mp.d_obs_type[o] = ty;
    mp.d_obs_px[2 * o] = px0;
    mp.d_obs_px[2 * o + 1] = px1;
    mp.d_obs_f[3 * o] = f0; mp.d_obs_f[3 * o + 1] = f1; mp.d_obs_f[3 * o + 2] = f2;
    mp.d_obs_grad[2 * o] = g0;
    mp.d_obs_grad[2 * o + 1] = g1;
  }
  for (int i = tid; i < a.patch.n_points; i += RM_BLOCK) {
    const int p = a.patch.d_index[i];
    const double x0 = a.patch.d_pos[3 * i], x1 = a.patch.d_pos[3 * i + 1], x2 = a.patch.d_pos[3 * i + 2];
    const int ty = a.patch.d_type[i], od = a.patch.d_order[i], ob = a.patch.d_obs_begin[i], oc = a.patch.d_obs_count[i];
    mp.d_pos[3 * p] = x0; mp.d_pos[3 * p + 1] = x1; mp.d_pos[3 * p + 2] = x2;
    mp.d_type[p] = ty;
    mp.d_order[p] = od;
    mp.d_obs_begin[p] = ob;
    mp.d_obs_count[p] = oc;
  }
  for (int k = tid; k < n_cells; k += RM_BLOCK) s_cnt[k] = 0;
  if (tid < RM_MAX_FRAMES) {
    // (ranks are the positions of the overlapping keyframes, 0..15: four bits of the sort key.  A larger value in the
    // device-resident array -- the C entry point cannot see it -- would spill into the key's type bits: such a frame is skipped)
    const int r = tid < a.n_frames ? a.kf_rank[tid] : -1;
    s_kfrank[tid] = r < 16 ? r : -1;
    s_kfcount[tid] = 0;
  }
  if (tid < a.n_frames) {
    Se3 T;
    se3_from_Rt(a.frame_T + 12 * tid, T);
    double fp[3];
    frame_pos(T, fp);
    s_fpos[tid][0] = fp[0]; s_fpos[tid][1] = fp[1]; s_fpos[tid][2] = fp[2];
  }
  if (tid == 0) { s_E = 0; s_end_cell = n_cells; }
  __syncthreads();  // (also orders the patch stores before the reads below: one workgroup, workgroup-scope fence)

  // ---- 1. reprojectPoint for every live point, once (:85-123, :206-217) -------------------------------------------
  Se3 Tc;
  se3_from_Rt(a.frame_T + 12 * a.cur_frame, Tc);
  int my_rank[PPT];        // visiting rank of the point's cell, -1: not binned
  uint32_t my_key[PPT];
  {
    // every record of the thread's entries first (independent loads), then the dependent round: the observation words
    int type[PPT], o0[PPT], on[PPT];
    double pos[PPT][3];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int p = tid + RM_BLOCK * j;
      const bool in = p < P;
      type[j] = in ? mp.d_type[p] : 0;
      o0[j] = in ? mp.d_obs_begin[p] : 0;
      on[j] = in ? mp.d_obs_count[p] : 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) pos[j][k] = in ? mp.d_pos[3 * p + k] : 0.0;
      my_key[j] = in && type[j] == 1 ? (uint32_t)(mp.d_order[p] & 0xffff) : 0u;
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int p = tid + RM_BLOCK * j;
      my_rank[j] = -1;
      if (p < P) {
        bool projected = false;
        int first_frame = -1;
        uint32_t key = 0;
        if (type[j] == 1) {  // a candidate: after every keyframe point, in list order
          projected = true;
          key = (3u - 1u) << 28 | my_key[j];
        } else if (type[j] >= 2) {
          const uint32_t best = first_meeting(mp.d_obs_order, o0[j], on[j], s_kfrank, &first_frame);
          projected = best != 0xffffffffu;
          key = (uint32_t)(3 - type[j]) << 28 | best;
        }
        my_key[j] = key;
        int cell = -2;
        if (projected) {
          double q[3], px[2];
          se3_apply(Tc, pos[j], q);
          world2cam(a.cam, q, px);
          a.out.d_point_px[2 * p] = px[0];
          a.out.d_point_px[2 * p + 1] = px[1];
          cell = -1;
          if (is_in_frame(a.cam, cast_int(px[0]), cast_int(px[1]), 8)) {
            const int k = cast_int(px[1] / a.grid.cell_size) * a.grid.n_cols + cast_int(px[0] / a.grid.cell_size);
            if (k >= 0 && k < n_cells) {  // (always, for a grid that covers the image)
              cell = k;
              my_rank[j] = a.grid.d_cell_rank[k];
              atomicAdd(&s_cnt[my_rank[j]], 1);
              atomicAdd(&s_E, 1);
              if (first_frame >= 0) atomicAdd(&s_kfcount[first_frame], 1);
            }
          }
        }
        a.out.d_point_cell[p] = cell;
      }
    }
  }
  __syncthreads();
  const int E = s_E;
  if (E > RM_MAX_E) {  // more points inside the frame than one call orders: the host takes its list-walking path
    if (tid == 0) {
      a.out.d_header[0] = 1;
      a.out.d_header[1] = E;
      a.out.d_header[2] = a.out.d_header[3] = a.out.d_header[5] = 0;
      a.out.d_header[4] = a.first_cell;
    }
    return;
  }

  // ---- 2. exclusive scan of the per-cell counts over the VISITING order --------------------------------------------
  {
    constexpr int CPT = RM_MAX_CELLS / RM_BLOCK;
    int c[CPT], sum = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int r = tid * CPT + k;
      c[k] = r < n_cells ? s_cnt[r] : 0;
      sum += c[k];
    }
    int total;
    int run = block_exclusive_scan(sum, s_wave, &total);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int r = tid * CPT + k;
      if (r < n_cells) { s_base[r] = run; s_cnt[r] = 0; }
      run += c[k];
    }
    if (tid == 0) s_base[n_cells] = total;
  }
  __syncthreads();

  // ---- 3. scatter into the cell buckets (any order inside a bucket) -------------------------------------------------
#pragma unroll
  for (int j = 0; j < PPT; ++j)
    if (my_rank[j] >= 0) {
      const int at = s_base[my_rank[j]] + atomicAdd(&s_cnt[my_rank[j]], 1);
      s_bkey[at] = my_key[j];
      s_bidx[at] = (uint16_t)(tid + RM_BLOCK * j);
    }
  __syncthreads();

  // ---- 4. order inside a bucket: type descending, then binning order (the stable sort of :153) ----------------------
  // keys of one cell are distinct, so the position is the number of smaller keys
  {
    constexpr int CPT = RM_MAX_CELLS / RM_BLOCK;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int r = tid + RM_BLOCK * k;  // (neighbouring threads take neighbouring cells)
      if (r < n_cells) {
        const int b0 = s_base[r], b1 = s_base[r + 1];
        for (int i = b0; i < b1; ++i) {
          const uint32_t key = s_bkey[i];
          int smaller = 0;
          for (int j = b0; j < b1; ++j) smaller += s_bkey[j] < key ? 1 : 0;
          s_sidx[b0 + smaller] = s_bidx[i];
          s_srank[b0 + smaller] = (uint16_t)r;
        }
      }
    }
  }
  for (int k = tid; k < n_cells; k += RM_BLOCK) s_cnt[k] = 0;  // from here: 1 where the cell holds a trial
  __syncthreads();

  // ---- 5. Point::getCloseViewObs per binned point (point.cpp:97-117; matcher.cpp:137-138) ---------------------------
  int best_obs[RM_EPT];
  bool has_view[RM_EPT];
  {
    int p_[RM_EPT], o0[RM_EPT], on[RM_EPT];
    double pt[RM_EPT][3];
#pragma unroll
    for (int e = 0; e < RM_EPT; ++e) {
      const int i = tid + RM_BLOCK * e;
      const bool in = i < E;
      p_[e] = in ? (int)s_sidx[i] : 0;
      o0[e] = in ? mp.d_obs_begin[p_[e]] : 0;
      on[e] = in ? mp.d_obs_count[p_[e]] : 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) pt[e][k] = in ? mp.d_pos[3 * p_[e] + k] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < RM_EPT; ++e) {
      const int i = tid + RM_BLOCK * e;
      best_obs[e] = -1;
      has_view[e] = false;
      if (i < E && on[e] > 0) {
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = k < on[e] ? (uint32_t)mp.d_obs_order[o0[e] + k] : 0u;
        double obs_dir[3] = {s_fpos[a.cur_frame][0] - pt[e][0], s_fpos[a.cur_frame][1] - pt[e][1], s_fpos[a.cur_frame][2] - pt[e][2]};
        normalize3(obs_dir);
        int best = o0[e];
        double min_cos_angle = 0;
        auto look = [&](const uint32_t word, const int k) {
          const int fr = (int)(word >> 16) & (RM_MAX_FRAMES - 1);
          double dir[3] = {s_fpos[fr][0] - pt[e][0], s_fpos[fr][1] - pt[e][1], s_fpos[fr][2] - pt[e][2]};
          normalize3(dir);
          const double cos_angle = dot3(obs_dir, dir);
          if (cos_angle > min_cos_angle) {
            min_cos_angle = cos_angle;
            best = o0[e] + k;
          }
        };
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < on[e]) look(w[k], k);
        for (int k = 8; k < on[e]; ++k) look((uint32_t)mp.d_obs_order[o0[e] + k], k);
        best_obs[e] = best;
        has_view[e] = !(min_cos_angle < 0.5);
        if (has_view[e]) s_cnt[s_srank[i]] = 1;  // (benign race: every writer stores 1)
      }
    }
  }
  __syncthreads();

  // ---- 6. cells from first_cell until max_cells_with_trials of them hold a trial -----------------------------------
  {
    constexpr int CPT = RM_MAX_CELLS / RM_BLOCK;
    int c[CPT], sum = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int r = tid * CPT + k;
      c[k] = (r < n_cells && r >= a.first_cell) ? s_cnt[r] : 0;
      sum += c[k];
    }
    int total;
    int run = block_exclusive_scan(sum, s_wave, &total);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int r = tid * CPT + k;
      run += c[k];
      // the loop of the host stops right after the cell that brought the count to the maximum
      if (c[k] && run == a.max_cells_with_trials) atomicMin(&s_end_cell, r + 1);
    }
    if (tid == 0 && a.max_cells_with_trials <= 0) s_end_cell = a.first_cell;
  }
  __syncthreads();
  const int end_cell = s_end_cell < a.first_cell ? a.first_cell : s_end_cell;
  const int v0 = a.first_cell < n_cells ? s_base[a.first_cell] : E;  // sorted positions [v0, v1) are the visits
  const int v1 = s_base[end_cell];
  const int V = v1 - v0;

  // ---- 7. trial numbers: exclusive count of the visits with a close view -------------------------------------------
  // thread t owns the sorted positions t, t + RM_BLOCK, ...: scan pass by pass, carrying the running total
  int trial[RM_EPT];
  int M = 0;
#pragma unroll
  for (int e = 0; e < RM_EPT; ++e) {
    const int i = tid + RM_BLOCK * e;
    trial[e] = -1;
    if (RM_BLOCK * e < v1 && RM_BLOCK * (e + 1) > v0) {  // (uniform: a pass without visits costs no barrier)
      const int flag = (i >= v0 && i < v1 && has_view[e]) ? 1 : 0;
      int total;
      const int ex = block_exclusive_scan(flag, s_wave, &total);
      trial[e] = flag ? M + ex : -1;
      M += total;
    }
  }
  const bool overflow = V > a.max_visits || M > a.max_trials;
  if (tid == 0) {
    a.out.d_header[0] = overflow ? 1 : 0;
    a.out.d_header[1] = E;
    a.out.d_header[2] = overflow ? 0 : V;
    a.out.d_header[3] = overflow ? 0 : M;
    a.out.d_header[4] = overflow ? a.first_cell : end_cell;
    a.out.d_header[5] = a.out.d_header[6] = a.out.d_header[7] = 0;
  }
  if (tid < a.n_frames) a.out.d_kf_count[tid] = s_kfcount[tid];
  if (overflow) return;
#pragma unroll
  for (int e = 0; e < RM_EPT; ++e) {
    const int i = tid + RM_BLOCK * e;
    if (i >= v0 && i < v1) {
      const int v = i - v0, p = s_sidx[i], r = s_srank[i], m = trial[e];
      a.out.d_visit_point[v] = p;
      a.out.d_visit_cell[v] = r;
      a.out.d_visit_trial[v] = m;
      if (m >= 0) {
        a.out.d_trial_cur[m] = a.cur_frame;
#pragma unroll
        for (int k = 0; k < 3; ++k) a.out.d_trial_pos[3 * m + k] = mp.d_pos[3 * p + k];
        a.out.d_trial_obs_begin[m] = best_obs[e];
        a.out.d_trial_obs_end[m] = best_obs[e] + 1;
        a.out.d_trial_cell[m] = r;
        a.out.d_trial_px[2 * m] = a.out.d_point_px[2 * p];
        a.out.d_trial_px[2 * m + 1] = a.out.d_point_px[2 * p + 1];
      }
    }
  }
}

}  // namespace

extern "C" int svo_hip_reproject_map(const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame,
                                     const int32_t* d_kf_rank, const svo_hip_map* map, const svo_hip_map_patch* patch,
                                     const svo_hip_grid* grid, int first_cell, int max_cells_with_trials, int max_visits,
                                     int max_trials, const svo_hip_reprojection* out, void* stream) {
  if (!cam || !cam_model_ok(cam) || !frames || !map || !grid || !out || !d_kf_rank) return SVO_HIP_EINVAL;
  if (frames->n_frames < 1 || frames->n_frames > RM_MAX_FRAMES || cur_frame < 0 || cur_frame >= frames->n_frames || !frames->d_T_f_w)
    return SVO_HIP_EINVAL;
  // a thread carries <= 8 entries in registers (a 1024-thread workgroup has 128 VGPRs per thread: the 16-entry form spilled
  // 188 dwords).  8192 entries are three times what the reference's map holds (10 keyframes + candidates: ~2600)
  if (map->n_points < 0 || map->n_points > 8 * RM_BLOCK || map->n_obs < 0) return SVO_HIP_ERANGE;
  if (grid->cell_size < 1 || grid->n_cols < 1 || grid->n_cells < 1 || grid->n_cells > RM_MAX_CELLS || !grid->d_cell_rank)
    return grid->n_cells > RM_MAX_CELLS ? SVO_HIP_ERANGE : SVO_HIP_EINVAL;
  if (first_cell < 0 || first_cell > grid->n_cells || max_visits < 0 || max_trials < 0) return SVO_HIP_EINVAL;
  if (map->n_points > 0 && (!map->d_pos || !map->d_type || !map->d_order || !map->d_obs_begin || !map->d_obs_count)) return SVO_HIP_EINVAL;
  if (map->n_obs > 0 && (!map->d_obs_frame || !map->d_obs_order || !map->d_obs_level || !map->d_obs_type || !map->d_obs_px ||
                         !map->d_obs_f || !map->d_obs_grad))
    return SVO_HIP_EINVAL;
  if (!out->d_header || !out->d_kf_count || (map->n_points > 0 && (!out->d_point_cell || !out->d_point_px))) return SVO_HIP_EINVAL;
  if (max_visits > 0 && (!out->d_visit_point || !out->d_visit_cell || !out->d_visit_trial)) return SVO_HIP_EINVAL;
  if (max_trials > 0 && (!out->d_trial_cur || !out->d_trial_pos || !out->d_trial_obs_begin || !out->d_trial_obs_end ||
                         !out->d_trial_cell || !out->d_trial_px))
    return SVO_HIP_EINVAL;
  ReprojMapArgs a;
  a.cam = make_cam(cam);
  a.n_frames = frames->n_frames;
  a.cur_frame = cur_frame;
  a.frame_T = frames->d_T_f_w;
  a.kf_rank = d_kf_rank;
  a.map = *map;
  if (patch) {
    a.patch = *patch;
    if (patch->n_points < 0 || patch->n_obs < 0) return SVO_HIP_EINVAL;
    if (patch->n_points > 0 && (!patch->d_index || !patch->d_pos || !patch->d_type || !patch->d_order || !patch->d_obs_begin ||
                                !patch->d_obs_count))
      return SVO_HIP_EINVAL;
    if (patch->n_obs > 0 && (!patch->d_obs_index || !patch->d_obs_order || !patch->obs.d_frame || !patch->obs.d_level ||
                             !patch->obs.d_type || !patch->obs.d_px || !patch->obs.d_f || !patch->obs.d_grad))
      return SVO_HIP_EINVAL;
  } else {
    a.patch = svo_hip_map_patch{};
  }
  a.grid = *grid;
  a.first_cell = first_cell;
  a.max_cells_with_trials = max_cells_with_trials;
  a.max_visits = max_visits;
  a.max_trials = max_trials;
  a.out = *out;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (map->n_points <= 4 * RM_BLOCK) hipLaunchKernelGGL(reproject_map_kernel<4>, dim3(1), dim3(RM_BLOCK), 0, st, a);
  else hipLaunchKernelGGL(reproject_map_kernel<8>, dim3(1), dim3(RM_BLOCK), 0, st, a);
  return check_launch();
}
