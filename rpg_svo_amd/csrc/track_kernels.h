// track_kernels.h -- internal launchers shared by the matcher / depth-filter entry points.
#pragma once
#include "capi_common.h"

namespace svo_track {

// K3 (feature_align.hip): one lane per trial
struct AlignArgs {
  svo_hip_pyr_layout L;
  const uint8_t* store;
  int M;
  const int32_t* slot;     // [M] pyramid slot of the image searched
  const int32_t* level;    // [M] its level
  const uint8_t* pwb;      // [M][100]
  const float* dir;        // [M][2] or NULL
  const uint8_t* use_1d;   // [M] or NULL
  const uint8_t* active;   // [M] or NULL: trials with 0 are skipped (ok = 0, px untouched)
  int n_iter;
  const double* px_in;     // [M][2] level coordinates
  double* px_out;          // [M][2]
  int scale_out;           // 1: px_out = px * (1 << level)  (Matcher: px_cur = px_scaled*(1<<search_level_))
  int32_t* ok;             // [M]
  double* h_inv;           // [M] or NULL
  int32_t* iters = nullptr;  // [M] or NULL: residual evaluations per trial (instrumented kernel variant)
  const int32_t* M_dev = nullptr;  // or the batch size lives on the device: min(*M_dev, M) trials (launches sized for M)
  // Phased run (launch_align fills these; see feature_align.hip): iterations [it0, it1) of the trials listed in
  // queue_in (NULL: all M trials); a trial that has neither converged nor failed by it1 < n_iter parks its loop
  // state in `state` and its index in queue_out for the next launch.
  int it0 = 0, it1 = 1 << 30;
  const int32_t* queue_in = nullptr;   // [ALIGN_NQ][queue_cap]
  const int32_t* n_in = nullptr;       // [ALIGN_NQ]
  int32_t* queue_out = nullptr;
  int32_t* n_out = nullptr;
  int queue_cap = 0;
  float* state = nullptr;              // [M][6]: u, v, mean_diff, chi2, up0, up1
};
// Scratch of a phased alignment (compaction of the trials still iterating between launches): with it (and M large enough
// for the extra launches to pay) launch_align runs the iterations in three launches, 0-2 / 3-5 / 6..., each over the
// trials still alive, so that a wave no longer runs as long as its slowest trial.  Results do not depend on it.
constexpr int ALIGN_NQ = 64;           // queues (one atomic counter each: no single hot address)
size_t align_phase_workspace_bytes(int M);
struct SeedArgs;  // seed_finish.h
// finish: the depth filter's last step (seed_finish.h) as the epilogue of every trial's LAST alignment launch -- trial t is
// seed t.  Only the lane-per-trial kernel takes it: align_takes_finish(M) says whether launch_align will (the caller runs
// seed_finish_kernel itself otherwise).
bool align_takes_finish(int M);
int launch_align(const AlignArgs& a, hipStream_t s, void* d_phase_ws = nullptr, size_t phase_ws_bytes = 0,
                 const SeedArgs* finish = nullptr);

// K2b (matcher.hip): warp::warpAffine for M trials, 10x10 output, 32 lanes per trial
struct WarpArgs {
  svo_hip_pyr_layout L;
  const uint8_t* store;
  int M;
  const uint8_t* active;      // [M]: 0 -> patch zeroed
  const int32_t* ref_slot;    // [M]
  const int32_t* ref_level;   // [M]
  const int32_t* search_level;  // [M]
  const float* A_ref_cur;     // [M][4] row-major (A_cur_ref^-1 cast to float)
  const float* px_ref_pyr;    // [M][2] px_ref.cast<float>() / (1 << level_ref)
  uint8_t* pwb;               // [M][100]
  const int32_t* M_dev = nullptr;  // see AlignArgs::M_dev
};
int launch_warp(const WarpArgs& a, hipStream_t s);

// K4 (pose_optimizer_wave.hip): one wave per frame, n_stride <= 256
struct PoseWaveArgs {
  svo_hip_camera cam;
  int B;
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
};
constexpr int POSE_WAVE_MAX_STRIDE = 256;
int launch_pose_wave(const PoseWaveArgs& a, hipStream_t s);

// carve 256-byte aligned arrays out of a caller-provided workspace
struct Carver {
  uint8_t* p;
  size_t left;
  bool ok = true;
  Carver(void* base, size_t bytes) : p(static_cast<uint8_t*>(base)), left(bytes) {}
  template <typename T>
  T* take(size_t n) {
    const size_t need = (n * sizeof(T) + 255) & ~size_t(255);
    if (need > left) { ok = false; return nullptr; }
    T* r = reinterpret_cast<T*>(p);
    p += need;
    left -= need;
    return r;
  }
  static size_t round(size_t n_bytes) { return (n_bytes + 255) & ~size_t(255); }
};

}  // namespace svo_track
