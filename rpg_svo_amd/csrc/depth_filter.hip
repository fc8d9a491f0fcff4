// depth_filter.hip -- K5: batched DepthFilter::updateSeeds for gfx950.
//
// Replaces, per seed (svo/src/depth_filter.cpp:197-291):
//   age / visibility tests, inverse-depth interval (:216-235)            seed_prepare_kernel
//   Matcher::findEpipolarMatchDirect (svo/src/matcher.cpp:179-321):        (lane per seed, f64)
//     epipolar segment, affine warp matrix, edgelet filter, search level
//     warp::warpAffine 10x10                                            \ epi_scan_kernel (8 lanes per seed; the patch
//     ZMSSD scan along the epipolar line (:248-291)                     /  stays in LDS between the two: warp_group.h)
//     sub-pixel refinement align2D / align1D (:295-315)                   K3 (feature_align.hip)
//     depthFromTriangulation (:109-122)                                  \ seed_finish_kernel
//   DepthFilter::computeTau (:334-350), updateSeed (:309-332),           | (lane per seed)
//   convergence test (:261-262, :283-287)                                /
//
// The epipolar scan is the only part with real per-seed parallelism (up to 1000 candidate
// positions x 64 pixels of integer ZMSSD): 8 lanes share a seed and take its steps
// round-robin, score them with v_dot4_u32_u8, and a lexicographic (score, step) minimum over the
// group reproduces the reference's "first strictly smaller score wins".  A lane replays only the
// chain of f64 additions (uv += step) up to its own step and keeps the position of the step before
// (the last_checked_pxi rule compares with step i-1), so the positions visited are the
// reference's, rounding included; see epi_scan_kernel.
//
// List surgery (erasing seeds, creating svo::Point objects, the converged callback) stays on
// the host: the kernel reports a status per seed and the new point's position.
#pragma clang fp contract(off)
#include <atomic>

#include "track_kernels.h"
#include "track_math.h"
#include "matcher_device.h"
#include "seed_math.h"
#include "epi_scan.h"
#include "wave_reduce.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_track;

namespace {

// The relative poses of a seed -- T_ref_cur, T_cur_ref -- depend on its (reference keyframe, current frame) PAIR alone, and
// a seed list holds its seeds keyframe by keyframe.  Rounds 1-5 rebuilt the two quaternions from the frame table's matrices,
// inverted and composed them per seed here and AGAIN in seed_finish (a third of its instructions: two sqrt and sixteen f64
// divisions among them).  Round 6: this kernel forms them once per seed, before its early exits, the first seed of every
// RUN of equal pairs inside a wave files them in `pair_T`, every seed notes where (`pair_index`), and seed_finish reads
// 14 doubles.  (Forming them once per run HERE as well -- workgroups of 256, one lane per run, the poses handed out
// through LDS -- measured slower for this kernel: the one wave that forms a workgroup's poses is a bubble the other
// three wait behind, 0.75 -> 0.86 ms per 13 M seeds, profiles/r06u_*, r06z_*.)
constexpr int PREP_BLOCK = 64;
__device__ __forceinline__ void pair_store(double* __restrict__ dst, const Se3& T_cur_ref, const Se3& T_ref_cur) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { dst[k] = T_cur_ref.q[k]; dst[7 + k] = T_ref_cur.q[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) { dst[4 + k] = T_cur_ref.t[k]; dst[11 + k] = T_ref_cur.t[k]; }
}

__global__ void __launch_bounds__(PREP_BLOCK) seed_prepare_kernel(const SeedArgs a) {
  const int s = blockIdx.x * PREP_BLOCK + threadIdx.x;
  if (s >= a.S) return;
  const SeedWs& w = a.ws;
  // What every seed needs whatever happens to it.  Arrays only the warp / scan / alignment of an ACTIVE seed read
  // (ref_slot, ref_level, dir, px_scaled, px_cur) are written where the seed becomes active, and seed_finish reads
  // px_cur / align_ok only for seeds that got that far: an early exit costs 19 bytes of workspace, not 71 (round 3:
  // 230 B per seed through these arrays against 36 B of seed state).
  w.align_active[s] = 0;
  w.use_1d[s] = a.opt.align_1d ? 1 : 0;
  w.mode[s] = MODE_NONE;
  w.status[s] = 0;
  w.search_level[s] = -1;  // not reached yet (an edgelet rejected by the angle filter returns before matcher.cpp:214)
  w.n_steps[s] = 0;
  w.accepted_raw[s] = 0;
  const int cf = a.cur_frame ? a.cur_frame[s] : a.cur_index;
  const int rec = a.slot_of ? a.slot_of[s] : s;  // the seed's record (resident store: its slot)
  // Every record of the seed is requested before the first early exit -- two memory round trips, (cf, rfi, batch id, mu,
  // sigma2, f) then (the two poses, the slot), instead of three or four: a load below an early exit cannot be issued above
  // it by the compiler (360 -> 315 us per 3.3 M seeds, profiles/r05a_queue_drain.txt).
  const int rfi = a.ftr.d_frame[rec];
  // (the seed state does not exist in match-only calls: read through stand-in pointers, see below)
  const int batch_raw = (a.match_only ? a.ftr.d_level : a.seeds.d_batch_id)[rec];
  const float mu_raw = (a.match_only ? reinterpret_cast<const float*>(a.ftr.d_px) : a.seeds.d_mu)[rec];
  const float sigma2_raw = (a.match_only ? reinterpret_cast<const float*>(a.ftr.d_px) : a.seeds.d_sigma2)[rec];
  const int batch_id = a.match_only ? 0 : batch_raw;
  const float mu = a.match_only ? 1.f : mu_raw, sigma2 = a.match_only ? 0.f : sigma2_raw;
  const double f[3] = {a.ftr.d_f[3 * rec], a.ftr.d_f[3 * rec + 1], a.ftr.d_f[3 * rec + 2]};
  // (what the epipolar set-up reads further down, behind two more early exits: the feature's level, pixel, type and
  // gradient -- the optional type array through a stand-in pointer, a load under a condition is waited for in its branch)
  const int rlevel_early = a.ftr.d_level[rec];
  const double rpx0_early = a.ftr.d_px[2 * rec], rpx1_early = a.ftr.d_px[2 * rec + 1];
  const uint8_t type_early = (a.ftr.d_type ? a.ftr.d_type : reinterpret_cast<const uint8_t*>(a.ftr.d_px))[rec];
  const double* const gradp = (a.ftr.d_type && a.ftr.d_grad) ? a.ftr.d_grad : a.ftr.d_px;
  const double gx_early = gradp[2 * rec], gy_early = gradp[2 * rec + 1];
  double RtR[12], RtC[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    RtR[k] = a.frame_T[12 * rfi + k];
    RtC[k] = a.frame_T[12 * cf + k];
  }
  if (a.cur_T_set && cf == a.cur_index) {  // (the current frame's pose arrived with the launch: uniform in practice)
#pragma unroll
    for (int k = 0; k < 12; ++k) RtC[k] = a.cur_T[k];
  }
  const int ref_slot_early = a.frame_slot[rfi];
  w.cur_slot[s] = a.frame_slot[cf];
  // the pair's poses, for this kernel and (through pair_T) for seed_finish: before the early exits, so that the first seed
  // of a run files them whatever happens to it
  Se3 Tr, Tc;
  se3_from_Rt(RtR, Tr);
  se3_from_Rt(RtC, Tc);
  const Se3 T_ref_cur = se3_compose(Tr, se3_inverse(Tc));
  const Se3 T_cur_ref = se3_compose(Tc, se3_inverse(Tr));
  {
    // a run starts at lane 0 of the wave and wherever the (reference, current) pair differs from the lane before
    const int lane = threadIdx.x & 63;
    const int rfi_left = __shfl_up(rfi, 1, 64), cf_left = __shfl_up(cf, 1, 64);
    const bool starts = lane == 0 || rfi_left != rfi || cf_left != cf;
    const uint64_t flags = __builtin_amdgcn_ballot_w64(starts);
    const int start_lane = 63 - __builtin_clzll(flags & (~0ull >> (63 - lane)));  // the last start at or before this lane
    w.pair_index[s] = s - lane + start_lane;
    if (starts) pair_store(w.pair_T + 14 * (size_t)s, T_cur_ref, T_ref_cur);
  }
  // check if seed is not already too old (:216-219)
  if (!a.match_only && (a.opt.batch_counter - batch_id) > a.opt.max_n_kfs) {
    w.status[s] = SVO_HIP_SEED_ERASED_OLD;
    return;
  }
  // visibility (:221-232)
  if (!a.match_only) {
    const Se3 T_cur_ref0 = se3_inverse(T_ref_cur);
    const double k = 1.0 / (double)mu;
    const double pr[3] = {k * f[0], k * f[1], k * f[2]};
    double xyz_f[3];
    se3_apply(T_cur_ref0, pr, xyz_f);
    if (xyz_f[2] < 0.0) {
      w.status[s] = SVO_HIP_SEED_BEHIND;
      return;
    }
    double pxp[2];
    world2cam(a.cam, xyz_f, pxp);
    if (!is_in_frame(a.cam, cast_int(pxp[0]), cast_int(pxp[1]), 0)) {
      w.status[s] = SVO_HIP_SEED_NOT_IN_FRAME;
      return;
    }
  }
  // inverse depth interval (:234-236)
  const float sq = sqrtf(sigma2);
  const float z_inv_min = mu + sq;
  const float zlo = mu - sq;
  const float z_inv_max = (zlo < 0.00000001f) ? 0.00000001f : zlo;
  w.z_inv_min[s] = z_inv_min;
  const double d_estimate = a.match_only ? a.d_est[s] : 1.0 / (double)mu;
  const double d_min = a.match_only ? a.d_min[s] : 1.0 / (double)z_inv_min;
  const double d_max = a.match_only ? a.d_max[s] : 1.0 / (double)z_inv_max;

  // ---- Matcher::findEpipolarMatchDirect, matcher.cpp:188-246 -------------------------
  double p[3], q[3], A[2], B[2];
  p[0] = f[0] * d_min; p[1] = f[1] * d_min; p[2] = f[2] * d_min;
  se3_apply(T_cur_ref, p, q);
  project2d(q, A);
  p[0] = f[0] * d_max; p[1] = f[1] * d_max; p[2] = f[2] * d_max;
  se3_apply(T_cur_ref, p, q);
  project2d(q, B);
  const double epi_dir[2] = {A[0] - B[0], A[1] - B[1]};
  const int rlevel = rlevel_early;
  const double rpx[2] = {rpx0_early, rpx1_early};
  double Am[4];
  warp_matrix_affine(a.cam, rpx, f, d_estimate, T_cur_ref, rlevel, Am);
  if (a.ftr.d_type && type_early == SVO_HIP_FTR_EDGELET && a.opt.epi_search_edgelet_filtering) {
    const double gx = gx_early, gy = gy_early;
    double g[2] = {Am[0] * gx + Am[1] * gy, Am[2] * gx + Am[3] * gy};
    const double gn = norm2(g);
    g[0] /= gn;
    g[1] /= gn;
    double e[2] = {epi_dir[0], epi_dir[1]};
    const double en = norm2(e);
    e[0] /= en;
    e[1] /= en;
    const double cosangle = fabs(g[0] * e[0] + g[1] * e[1]);
    if (cosangle < a.opt.epi_search_edgelet_max_angle) {
      w.status[s] = SVO_HIP_SEED_NO_MATCH;  // reject_ = true
      return;
    }
  }
  const int sl = best_search_level(Am, a.opt.n_pyr_levels - 1);
  w.search_level[s] = sl;
  double px_A[2], px_B[2];
  world2cam_uv(a.cam, A, px_A);
  world2cam_uv(a.cam, B, px_B);
  const double dAB[2] = {px_A[0] - px_B[0], px_A[1] - px_B[1]};
  const double epi_length = norm2(dAB) * pow2_inv_f64(sl);  // (/ 2^sl, same bits)
  // warp set-up (matcher.cpp:221-224)
  double Ainv[4];
  inv2<double>(Am, Ainv);
  w.A_ref_cur[4 * s] = (float)Ainv[0];
  w.A_ref_cur[4 * s + 1] = (float)Ainv[1];
  w.A_ref_cur[4 * s + 2] = (float)Ainv[2];
  w.A_ref_cur[4 * s + 3] = (float)Ainv[3];
  w.px_ref_pyr[2 * s] = (float)rpx[0] * pow2_inv_f32(rlevel);  // (/ 2^level, same bits)
  w.px_ref_pyr[2 * s + 1] = (float)rpx[1] * pow2_inv_f32(rlevel);
  w.ref_slot[s] = ref_slot_early;
  w.ref_level[s] = rlevel;
  {  // (px_A-px_B).cast<float>().normalized()
    float d0 = (float)dAB[0], d1 = (float)dAB[1];
    const float n = sqrtf(d0 * d0 + d1 * d1);
    w.dir[2 * s] = d0 / n;
    w.dir[2 * s + 1] = d1 / n;
  }
  if (epi_length < 2.0) {
    const double pc[2] = {(px_A[0] + px_B[0]) / 2.0, (px_A[1] + px_B[1]) / 2.0};
    w.px_cur[2 * s] = pc[0];
    w.px_cur[2 * s + 1] = pc[1];
    w.px_scaled[2 * s] = pc[0] * pow2_inv_f64(sl);
    w.px_scaled[2 * s + 1] = pc[1] * pow2_inv_f64(sl);
    w.mode[s] = MODE_SHORT;
    w.align_active[s] = 1;
    return;
  }
  const unsigned long long n_steps = (unsigned long long)(epi_length / 0.7);
  if (n_steps > (unsigned long long)a.opt.max_epi_search_steps) {
    w.status[s] = SVO_HIP_SEED_NO_MATCH;  // "skip epipolar search"
    return;
  }
  w.n_steps[s] = (int)n_steps;
  w.step[2 * s] = epi_dir[0] / (double)n_steps;
  w.step[2 * s + 1] = epi_dir[1] / (double)n_steps;
  w.B[2 * s] = B[0];
  w.B[2 * s + 1] = B[1];
  w.mode[s] = MODE_SCAN;
}

// A workgroup owns SCAN_CHUNK consecutive seeds.  Scan lengths differ by two orders of magnitude between seeds (a
// seed that has been matched a few times searches 2-6 positions, one that never was and sees a long baseline
// hundreds), and a wave runs as long as the longest scan among the seeds it holds: taken in list order the lanes
// idle two thirds of the time.  So the workgroup first sorts its chunk by scan length (counting sort over
// buckets: up to 8 positions, then powers of two of ceil(positions / 16); in LDS), longest first, and its waves then fetch groups of
// 8 neighbouring entries of that order from an LDS counter: the seeds a wave holds at a time need about
// the same number of passes, and no wave waits for another.  Seeds that neither warp nor scan (not visible,
// rejected) never enter the order; a seed with a segment too short to scan comes for its warp only.  Results do not
// depend on the order.
// Four waves per SIMD: the scan waits on its box fetch once per pass.  The undistorted instantiation fits (127 VGPRs, no
// scratch, 35.4 KB of LDS per workgroup); the distorted cameras' one carries the models' f64 code in the loop and spills
// outside it.  (What registers cost here: nine spilled dwords per seed took 10 % of the kernel, profiles/r05e_*.)
constexpr int SCAN_MINW = 4;
#ifdef SCAN_PROFILE
__device__ unsigned long long g_scan_prof[8];
#endif
template <bool PINHOLE>
__global__ void __launch_bounds__(SCAN_BLOCK, PINHOLE ? SCAN_MINW : SCAN_MINW - 1) epi_scan_kernel(const SeedArgs a) {
  __shared__ uint16_t s_order[SCAN_CHUNK];
  __shared__ int s_hist[SCAN_BUCKETS], s_off[SCAN_BUCKETS], s_next, s_n;
  __shared__ __attribute__((aligned(16))) uint32_t s_box[SCAN_BLOCK / SCAN_G][SCAN_BOX_DWORDS + 0];
  constexpr int PER_LANE = SCAN_CHUNK / SCAN_BLOCK;
  constexpr int GROUPS = 64 / SCAN_G;  // seeds a wave scans at a time
  const int chunk = a.scan_chunk;      // SCAN_CHUNK, or SCAN_CHUNK_SMALL for a small batch
  const int base = blockIdx.x * chunk;
  const SeedWs& w = a.ws;
  if (threadIdx.x < SCAN_BUCKETS) s_hist[threadIdx.x] = 0;
  __syncthreads();
  int bucket[PER_LANE], rank[PER_LANE];
#pragma unroll
  for (int k = 0; k < PER_LANE; ++k) {
    const int in_chunk = threadIdx.x + SCAN_BLOCK * k;
    const int s = base + in_chunk;
    bucket[k] = -1;
    rank[k] = 0;
    const int md = (in_chunk < chunk && s < a.S) ? w.mode[s] : MODE_NONE;
    if (md != MODE_NONE) {
      // n_steps + 1 positions (matcher.cpp:264): lines of up to SCAN_G positions (one pass of one position per lane) in
      // bucket 0, then power-of-two buckets of the passes of 2 SCAN_G positions.  A seed whose segment is too short to be
      // scanned (MODE_SHORT) is here for its warp alone: bucket 0
      const int n_pos = md == MODE_SCAN ? w.n_steps[s] + 1 : 1;
      const int passes = (n_pos + SCAN_PP - 1) / SCAN_PP;
      bucket[k] = n_pos <= SCAN_G ? 0 : min(SCAN_BUCKETS - 1, 1 + (passes <= 1 ? 0 : 32 - __clz(passes - 1)));
      rank[k] = atomicAdd(&s_hist[bucket[k]], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0;
    for (int b = SCAN_BUCKETS - 1; b >= 0; --b) {  // longest first: the tail of the chunk is made of short scans
      s_off[b] = off;
      off += s_hist[b];
    }
    s_n = off;
    s_next = 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER_LANE; ++k)
    if (bucket[k] >= 0) s_order[s_off[bucket[k]] + rank[k]] = (uint16_t)(threadIdx.x + SCAN_BLOCK * k);
  __syncthreads();
  const int n_scan = s_n;
  const int wl = threadIdx.x & 63;
  const int grp = wl / SCAN_G, lane = wl % SCAN_G;
#ifdef SCAN_PROFILE
  __shared__ uint32_t s_prof[SCAN_BLOCK / 64][8];
  uint32_t* const prof = s_prof[threadIdx.x >> 6];
  if (wl < 8) prof[wl] = 0;
#endif
  for (;;) {
    int p = 0;
    if (wl == 0) p = atomicAdd(&s_next, GROUPS);
    p = __builtin_amdgcn_readfirstlane(p);
    if (p >= n_scan) break;
#ifdef SCAN_PROFILE
    if (wl == 0) prof[7] += 1;
    if (p + grp < n_scan) epi_scan_seed<PINHOLE>(a, base + (int)s_order[p + grp], lane, s_box[threadIdx.x / SCAN_G], prof);
#else
    if (p + grp < n_scan) epi_scan_seed<PINHOLE>(a, base + (int)s_order[p + grp], lane, s_box[threadIdx.x / SCAN_G]);
#endif
  }
#ifdef SCAN_PROFILE
  if (wl < 8) atomicAdd(&g_scan_prof[wl], (unsigned long long)prof[wl]);
#endif
}

#ifdef SCAN_PROFILE
// (instrumentation build only: reads and clears the region totals)
extern "C" int svo_hip_scan_profile_read(unsigned long long out[8]) {
  if (hipDeviceSynchronize() != hipSuccess) return SVO_HIP_EHIP;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_prof), 8 * sizeof(unsigned long long)) != hipSuccess) return SVO_HIP_EHIP;
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_prof), zero, sizeof(zero)) != hipSuccess) return SVO_HIP_EHIP;
  return SVO_HIP_OK;
}
#endif

#define SEED_FINISH_BOUNDS __launch_bounds__(64, 4)  // (small batches only since round 6: occupancy does not matter, scratch does)
__global__ void SEED_FINISH_BOUNDS seed_finish_kernel(const SeedArgs a) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.S) return;
  seed_finish_seed<false>(a, s, false, 0, 0.0, 0.0);
}

struct SeedOnlyArgs {
  int S;
  const float* x;
  const float* tau2;
  svo_hip_seeds seeds;
};
__global__ void __launch_bounds__(256) update_seed_kernel(const SeedOnlyArgs a) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.S) return;
  float sa = a.seeds.d_a[s], sb = a.seeds.d_b[s], smu = a.seeds.d_mu[s], ssig = a.seeds.d_sigma2[s];
  update_seed(a.x[s], a.tau2[s], sa, sb, smu, a.seeds.d_z_range[s], ssig);
  a.seeds.d_a[s] = sa;
  a.seeds.d_b[s] = sb;
  a.seeds.d_mu[s] = smu;
  a.seeds.d_sigma2[s] = ssig;
}

struct TauArgs {
  int S;
  const double* t_ref_cur;
  const double* f;
  const double* z;
  double px_error_angle;
  TauConsts tau_k;
  double* tau;
};
__global__ void __launch_bounds__(256) compute_tau_kernel(const TauArgs a) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.S) return;
  Se3 T;
  T.q[0] = 1.0; T.q[1] = T.q[2] = T.q[3] = 0.0;  // only the translation enters computeTau
  for (int k = 0; k < 3; ++k) T.t[k] = a.t_ref_cur[3 * s + k];
  const double f[3] = {a.f[3 * s], a.f[3 * s + 1], a.f[3 * s + 2]};
  a.tau[s] = compute_tau(T, f, a.z[s], a.tau_k);
}

}  // namespace

extern "C" int svo_hip_compute_tau_batch(int S, const double* d_t_ref_cur, const double* d_f, const double* d_z,
                                         double px_error_angle, double* d_tau, void* stream) {
  if (S < 0) return SVO_HIP_EINVAL;
  if (S == 0) return SVO_HIP_OK;
  if (!d_t_ref_cur || !d_f || !d_z || !d_tau) return SVO_HIP_EINVAL;
  TauArgs a;
  a.S = S; a.t_ref_cur = d_t_ref_cur; a.f = d_f; a.z = d_z; a.px_error_angle = px_error_angle; a.tau = d_tau;
  a.tau_k = tau_consts(px_error_angle);
  hipLaunchKernelGGL(compute_tau_kernel, dim3((S + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

static int run_seed_chain(const svo_hip_pyr_layout* layout, const uint8_t* d_store, SeedArgs& a, int S, void* d_workspace,
                          size_t workspace_bytes, hipStream_t st);
static std::atomic<bool> g_count_evaluations{false};  // svo_hip_update_seeds_count_evaluations

extern "C" int svo_hip_find_epipolar_match_direct(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                                  const svo_hip_camera* cam, const svo_hip_frames* frames, int S,
                                                  const int32_t* d_cur_frame, const svo_hip_features* ftr,
                                                  const double* d_d_estimate, const double* d_d_min, const double* d_d_max,
                                                  const svo_hip_depth_filter_options* opt, int32_t* d_ok, double* d_depth,
                                                  double* d_px_cur, int32_t* d_search_level, void* d_workspace,
                                                  size_t workspace_bytes, void* stream) {
  if (!layout_ok(layout) || !d_store || !cam || !cam_model_ok(cam) || !frames || !ftr || !opt || S < 0) return SVO_HIP_EINVAL;
  if (S == 0) return SVO_HIP_OK;
  if (!d_cur_frame || !d_ok || !d_depth || !d_d_estimate || !d_d_min || !d_d_max || !frames->d_slot || !frames->d_T_f_w ||
      !ftr->d_frame || !ftr->d_level || !ftr->d_px || !ftr->d_f)
    return SVO_HIP_EINVAL;
  if (ftr->d_type && !ftr->d_grad) return SVO_HIP_EINVAL;
  if (opt->n_pyr_levels < 1 || opt->n_pyr_levels > layout->n_levels || opt->align_max_iter < 0 || opt->max_epi_search_steps < 0)
    return SVO_HIP_EINVAL;
  if (!d_workspace || workspace_bytes < svo_hip_match_workspace_bytes(S)) return SVO_HIP_ERANGE;
  SeedArgs a;
  a.L = *layout;
  a.store = d_store;
  a.cam = make_cam(cam);
  a.S = S;
  a.frame_slot = frames->d_slot;
  a.frame_T = frames->d_T_f_w;
  a.cur_frame = d_cur_frame;
  a.cur_index = 0;
  a.cur_T_set = 0;
  a.slot_of = nullptr;
  a.state_out = nullptr;
  a.ftr = *ftr;
  a.seeds.d_a = a.seeds.d_b = a.seeds.d_mu = a.seeds.d_z_range = a.seeds.d_sigma2 = nullptr;
  a.seeds.d_batch_id = nullptr;
  a.opt = *opt;
  a.status_out = nullptr;
  a.xyz_world = nullptr;
  a.px_cur_out = d_px_cur;
  a.match_only = 1;
  a.d_est = d_d_estimate;
  a.d_min = d_d_min;
  a.d_max = d_d_max;
  a.depth_out = d_depth;
  a.ok_out = d_ok;
  a.search_level_out = d_search_level;
  return run_seed_chain(layout, d_store, a, S, d_workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int svo_hip_update_seeds(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                    const svo_hip_camera* cam, const svo_hip_frames* frames, int S,
                                    const int32_t* d_cur_frame, const svo_hip_features* ftr,
                                    const svo_hip_seeds* seeds, const svo_hip_depth_filter_options* opt,
                                    int32_t* d_status, double* d_xyz_world, double* d_px_cur, void* d_workspace,
                                    size_t workspace_bytes, void* stream) {
  if (!layout_ok(layout) || !d_store || !cam || !cam_model_ok(cam) || !frames || !ftr || !seeds || !opt || S < 0) return SVO_HIP_EINVAL;
  if (S == 0) return SVO_HIP_OK;
  if (!d_cur_frame || !d_status || !frames->d_slot || !frames->d_T_f_w || !ftr->d_frame || !ftr->d_level ||
      !ftr->d_px || !ftr->d_f || !seeds->d_a || !seeds->d_b || !seeds->d_mu || !seeds->d_z_range ||
      !seeds->d_sigma2 || !seeds->d_batch_id)
    return SVO_HIP_EINVAL;
  if (ftr->d_type && !ftr->d_grad) return SVO_HIP_EINVAL;
  if (opt->n_pyr_levels < 1 || opt->n_pyr_levels > layout->n_levels || opt->align_max_iter < 0 ||
      opt->max_epi_search_steps < 0)
    return SVO_HIP_EINVAL;
  if (!d_workspace || workspace_bytes < svo_hip_match_workspace_bytes(S)) return SVO_HIP_ERANGE;
  SeedArgs a;
  a.L = *layout;
  a.store = d_store;
  a.cam = make_cam(cam);
  a.S = S;
  a.frame_slot = frames->d_slot;
  a.frame_T = frames->d_T_f_w;
  a.cur_frame = d_cur_frame;
  a.cur_index = 0;
  a.cur_T_set = 0;
  a.slot_of = nullptr;
  a.state_out = nullptr;
  a.ftr = *ftr;
  a.seeds = *seeds;
  a.opt = *opt;
  a.status_out = d_status;
  a.xyz_world = d_xyz_world;
  a.px_cur_out = d_px_cur;
  a.match_only = 0;
  a.d_est = a.d_min = a.d_max = nullptr;
  a.depth_out = nullptr;
  a.ok_out = nullptr;
  a.search_level_out = nullptr;
  return run_seed_chain(layout, d_store, a, S, d_workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

// ---- row N2, seeds: the resident store (include/svo_hip.h) ------------------------------------------------------------
namespace {
struct SeedPatchArgs {
  svo_hip_seed_patch p;
  svo_hip_features ftr;
  svo_hip_seeds seeds;
};
__global__ void __launch_bounds__(256) seed_patch_kernel(const SeedPatchArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.p.n) return;
  const int q = a.p.d_slot[i];
  const svo_hip_features& sf = a.p.src_ftr;
  const svo_hip_seeds& ss = a.p.src_seeds;
  // (every field read before the first store: the arrays cannot be proven distinct)
  const int32_t fr = sf.d_frame[i], lv = sf.d_level[i], bid = ss.d_batch_id[i];
  const uint8_t ty = sf.d_type ? sf.d_type[i] : (uint8_t)SVO_HIP_FTR_CORNER;
  const double px0 = sf.d_px[2 * i], px1 = sf.d_px[2 * i + 1];
  const double f0 = sf.d_f[3 * i], f1 = sf.d_f[3 * i + 1], f2 = sf.d_f[3 * i + 2];
  const double g0 = sf.d_grad ? sf.d_grad[2 * i] : 1.0, g1 = sf.d_grad ? sf.d_grad[2 * i + 1] : 0.0;
  const float sa = ss.d_a[i], sb = ss.d_b[i], smu = ss.d_mu[i], szr = ss.d_z_range[i], ss2 = ss.d_sigma2[i];
  const_cast<int32_t*>(a.ftr.d_frame)[q] = fr;
  const_cast<int32_t*>(a.ftr.d_level)[q] = lv;
  const_cast<uint8_t*>(a.ftr.d_type)[q] = ty;
  const_cast<double*>(a.ftr.d_px)[2 * q] = px0; const_cast<double*>(a.ftr.d_px)[2 * q + 1] = px1;
  const_cast<double*>(a.ftr.d_f)[3 * q] = f0; const_cast<double*>(a.ftr.d_f)[3 * q + 1] = f1; const_cast<double*>(a.ftr.d_f)[3 * q + 2] = f2;
  const_cast<double*>(a.ftr.d_grad)[2 * q] = g0; const_cast<double*>(a.ftr.d_grad)[2 * q + 1] = g1;
  a.seeds.d_a[q] = sa; a.seeds.d_b[q] = sb; a.seeds.d_mu[q] = smu; a.seeds.d_z_range[q] = szr; a.seeds.d_sigma2[q] = ss2;
  const_cast<int32_t*>(a.seeds.d_batch_id)[q] = bid;
}
}  // namespace

extern "C" int svo_hip_seed_store_patch(const svo_hip_seed_patch* patch, const svo_hip_features* store_ftr,
                                        const svo_hip_seeds* store_seeds, void* stream) {
  if (!patch || !store_ftr || !store_seeds || patch->n < 0) return SVO_HIP_EINVAL;
  if (patch->n == 0) return SVO_HIP_OK;
  const svo_hip_features& sf = patch->src_ftr;
  const svo_hip_seeds& ss = patch->src_seeds;
  if (!patch->d_slot || !sf.d_frame || !sf.d_level || !sf.d_px || !sf.d_f || !ss.d_a || !ss.d_b || !ss.d_mu || !ss.d_z_range ||
      !ss.d_sigma2 || !ss.d_batch_id)
    return SVO_HIP_EINVAL;
  if (!store_ftr->d_frame || !store_ftr->d_level || !store_ftr->d_type || !store_ftr->d_px || !store_ftr->d_f || !store_ftr->d_grad ||
      !store_seeds->d_a || !store_seeds->d_b || !store_seeds->d_mu || !store_seeds->d_z_range || !store_seeds->d_sigma2 ||
      !store_seeds->d_batch_id)
    return SVO_HIP_EINVAL;  // the store carries every column (type and gradient included)
  SeedPatchArgs a;
  a.p = *patch;
  a.ftr = *store_ftr;
  a.seeds = *store_seeds;
  hipLaunchKernelGGL(seed_patch_kernel, dim3((patch->n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}

static int update_seeds_resident_impl(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                      const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame, const double* T_cur_host, int S,
                                      const int32_t* d_slot_of, const svo_hip_features* ftr,
                                      const svo_hip_seeds* seeds, const svo_hip_depth_filter_options* opt,
                                      int32_t* d_status, double* d_xyz_world, double* d_px_cur, float* d_state_out,
                                      void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!layout_ok(layout) || !d_store || !cam || !cam_model_ok(cam) || !frames || !ftr || !seeds || !opt || S < 0) return SVO_HIP_EINVAL;
  if (S == 0) return SVO_HIP_OK;
  if (!d_slot_of || !d_status || !frames->d_slot || !frames->d_T_f_w || cur_frame < 0 || cur_frame >= frames->n_frames ||
      !ftr->d_frame || !ftr->d_level || !ftr->d_px || !ftr->d_f || !seeds->d_a || !seeds->d_b || !seeds->d_mu ||
      !seeds->d_z_range || !seeds->d_sigma2 || !seeds->d_batch_id)
    return SVO_HIP_EINVAL;
  if (ftr->d_type && !ftr->d_grad) return SVO_HIP_EINVAL;
  if (opt->n_pyr_levels < 1 || opt->n_pyr_levels > layout->n_levels || opt->align_max_iter < 0 ||
      opt->max_epi_search_steps < 0)
    return SVO_HIP_EINVAL;
  if (!d_workspace || workspace_bytes < svo_hip_match_workspace_bytes(S)) return SVO_HIP_ERANGE;
  SeedArgs a;
  a.L = *layout;
  a.store = d_store;
  a.cam = make_cam(cam);
  a.S = S;
  a.frame_slot = frames->d_slot;
  a.frame_T = frames->d_T_f_w;
  a.cur_frame = nullptr;
  a.cur_index = cur_frame;
  a.cur_T_set = T_cur_host != nullptr;
  for (int k = 0; k < 12; ++k) a.cur_T[k] = T_cur_host ? T_cur_host[k] : 0.0;
  a.slot_of = d_slot_of;
  a.state_out = d_state_out;
  a.ftr = *ftr;
  a.seeds = *seeds;
  a.opt = *opt;
  a.status_out = d_status;
  a.xyz_world = d_xyz_world;
  a.px_cur_out = d_px_cur;
  a.match_only = 0;
  a.d_est = a.d_min = a.d_max = nullptr;
  a.depth_out = nullptr;
  a.ok_out = nullptr;
  a.search_level_out = nullptr;
  return run_seed_chain(layout, d_store, a, S, d_workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int svo_hip_update_seeds_resident(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                             const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame, int S,
                                             const int32_t* d_slot_of, const svo_hip_features* ftr,
                                             const svo_hip_seeds* seeds, const svo_hip_depth_filter_options* opt,
                                             int32_t* d_status, double* d_xyz_world, double* d_px_cur, float* d_state_out,
                                             void* d_workspace, size_t workspace_bytes, void* stream) {
  return update_seeds_resident_impl(layout, d_store, cam, frames, cur_frame, nullptr, S, d_slot_of, ftr, seeds, opt, d_status, d_xyz_world,
                                    d_px_cur, d_state_out, d_workspace, workspace_bytes, stream);
}

extern "C" int svo_hip_update_seeds_resident_pose(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                                  const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame,
                                                  const double* T_cur_f_w, int S, const int32_t* d_slot_of, const svo_hip_features* ftr,
                                                  const svo_hip_seeds* seeds, const svo_hip_depth_filter_options* opt,
                                                  int32_t* d_status, double* d_xyz_world, double* d_px_cur, float* d_state_out,
                                                  void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!T_cur_f_w) return SVO_HIP_EINVAL;
  return update_seeds_resident_impl(layout, d_store, cam, frames, cur_frame, T_cur_f_w, S, d_slot_of, ftr, seeds, opt, d_status, d_xyz_world,
                                    d_px_cur, d_state_out, d_workspace, workspace_bytes, stream);
}

// seed_prepare -> (affine warp +) epipolar scan -> sub-pixel alignment -> seed_finish on one stream
static int run_seed_chain(const svo_hip_pyr_layout* layout, const uint8_t* d_store, SeedArgs& a, int S, void* d_workspace,
                          size_t workspace_bytes, hipStream_t st) {
  Carver c(d_workspace, workspace_bytes);
  const size_t n = (size_t)S;
  {
    const double focal_length = fabs(a.cam.fx), px_noise = 1.0;
    a.px_error_angle = atan(px_noise / (2.0 * focal_length)) * 2.0;
    a.tau_k = tau_consts(a.px_error_angle);
  }
  a.scan_chunk = S <= SCAN_SMALL_S ? SCAN_CHUNK_SMALL : SCAN_CHUNK;
  SeedWs& w = a.ws;
  w.n_steps = c.take<int32_t>(n);  // first array of the workspace: svo_hip_update_seeds_scan_steps
  int32_t* const align_evals = c.take<int32_t>(n);  // second: svo_hip_update_seeds_align_evaluations
  w.pwb = c.take<uint8_t>(n * 100);
  w.align_active = c.take<uint8_t>(n);
  w.use_1d = c.take<uint8_t>(n);
  w.status = c.take<int32_t>(n);
  w.mode = c.take<int32_t>(n);
  w.ref_slot = c.take<int32_t>(n);
  w.ref_level = c.take<int32_t>(n);
  w.cur_slot = c.take<int32_t>(n);
  w.search_level = c.take<int32_t>(n);
  w.align_ok = c.take<int32_t>(n);
  w.accepted_raw = c.take<uint8_t>(n);
  w.A_ref_cur = c.take<float>(4 * n);
  w.px_ref_pyr = c.take<float>(2 * n);
  w.dir = c.take<float>(2 * n);
  w.z_inv_min = c.take<float>(n);
  w.B = c.take<double>(2 * n);
  w.step = c.take<double>(2 * n);
  w.px_scaled = c.take<double>(2 * n);
  w.px_cur = c.take<double>(2 * n);
  w.uv_best = c.take<double>(2 * n);
  if (!c.ok) return SVO_HIP_ERANGE;
  const size_t phase_bytes = align_phase_workspace_bytes(S);
  void* phase_ws = phase_bytes ? c.take<uint8_t>(phase_bytes) : nullptr;
  if (!c.ok) phase_ws = nullptr;  // (a caller with an older, smaller workspace: single-launch alignment)
  // the poses of a (reference, current) pair, formed once per run of seeds (seed_prepare_kernel)
  c.ok = true;
  w.pair_index = c.take<int32_t>(n);
  w.pair_T = c.take<double>(14 * n);
  if (!c.ok) return SVO_HIP_ERANGE;  // (cannot happen: the entry points check svo_hip_match_workspace_bytes)
  hipLaunchKernelGGL(seed_prepare_kernel, dim3((S + PREP_BLOCK - 1) / PREP_BLOCK), dim3(PREP_BLOCK), 0, st, a);
  int rc = check_launch();
  if (rc) return rc;
  // (the affine warp of the reference patch is the scan kernel's first step: warp_group.h)
  if (a.cam.model == SVO_HIP_CAM_PINHOLE)
    hipLaunchKernelGGL(epi_scan_kernel<true>, dim3((S + a.scan_chunk - 1) / a.scan_chunk), dim3(SCAN_BLOCK), 0, st, a);
  else
    hipLaunchKernelGGL(epi_scan_kernel<false>, dim3((S + a.scan_chunk - 1) / a.scan_chunk), dim3(SCAN_BLOCK), 0, st, a);
  rc = check_launch();
  if (rc) return rc;
  AlignArgs al;
  al.L = *layout;
  al.store = d_store;
  al.M = S;
  al.slot = w.cur_slot;
  al.level = w.search_level;
  al.pwb = w.pwb;
  al.dir = w.dir;
  al.use_1d = w.use_1d;
  al.active = w.align_active;
  al.n_iter = a.opt.align_max_iter;
  al.px_in = w.px_scaled;
  al.px_out = w.px_cur;
  al.scale_out = 1;
  al.ok = w.align_ok;
  al.h_inv = nullptr;
  if (g_count_evaluations.load(std::memory_order_relaxed)) {
    SVO_HIP_TRY(hipMemsetAsync(align_evals, 0, n * sizeof(int32_t), st));  // (a seed that never reaches the alignment: 0)
    al.iters = align_evals;
  }
  // large batches: the last step (seed_finish.h) is the epilogue of every seed's last alignment launch
  if (align_takes_finish(S)) return launch_align(al, st, phase_ws, phase_bytes, &a);
  rc = launch_align(al, st, phase_ws, phase_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(seed_finish_kernel, dim3((S + 63) / 64), dim3(64), 0, st, a);
  return check_launch();
}

extern "C" const int32_t* svo_hip_update_seeds_scan_steps(const void* d_workspace) {
  return static_cast<const int32_t*>(d_workspace);
}

extern "C" int svo_hip_update_seeds_count_evaluations(int on) { return g_count_evaluations.exchange(on != 0) ? 1 : 0; }

extern "C" const int32_t* svo_hip_update_seeds_align_evaluations(const void* d_workspace, int S) {
  if (!d_workspace || S < 0) return nullptr;
  return reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(d_workspace) + Carver::round((size_t)S * sizeof(int32_t)));
}

extern "C" int svo_hip_update_seed_batch(int S, const float* d_x, const float* d_tau2, const svo_hip_seeds* seeds,
                                         void* stream) {
  if (S < 0 || !seeds) return SVO_HIP_EINVAL;
  if (S == 0) return SVO_HIP_OK;
  if (!d_x || !d_tau2 || !seeds->d_a || !seeds->d_b || !seeds->d_mu || !seeds->d_z_range || !seeds->d_sigma2)
    return SVO_HIP_EINVAL;
  SeedOnlyArgs a;
  a.S = S;
  a.x = d_x;
  a.tau2 = d_tau2;
  a.seeds = *seeds;
  hipLaunchKernelGGL(update_seed_kernel, dim3((S + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch();
}
