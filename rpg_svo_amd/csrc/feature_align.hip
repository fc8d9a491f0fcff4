// feature_align.hip -- K3: batched feature_alignment::align2D / align1D for gfx950.
//
// Replaces svo::feature_alignment::align2D (svo/src/feature_alignment.cpp:149-277) and
// align1D (:30-147): inverse-compositional Lucas-Kanade of an 8x8 template (cut from the
// 10x10 Matcher::patch_with_border_) against one pyramid level of the current frame.
//
// Mapping: ONE LANE PER TRIAL.  A trial is 64 pixels x <=10 iterations of strictly
// sequential float accumulation in the reference (Jres, and for align1D also H and chi2);
// giving a trial to one lane keeps that order, so with contraction off the result is
// bit-identical to the reference's, and a wave works on 64 independent trials (a frame has
// ~200 of them, a replay batch millions).  The 100 template bytes live in 25 VGPRs; the
// template gradients are re-formed from them with byte-extract converts (v_cvt_f32_ubyteN)
// instead of being cached; each 9-byte row of the current image is 3 aligned dwords +
// v_alignbyte.  No LDS, no cross-lane traffic.
//
// Batches of up to 8192 trials -- a camera frame's -- take align_wave_kernel instead: one WAVE per trial, lane = template
// pixel, the reference's accumulation order kept by an ordered sum through LDS (align_wave.h); same bits, 11 us instead
// of 27 for the ~130 trials of a frame.
#pragma clang fp contract(off)
#include <type_traits>

#include "track_kernels.h"
#include "track_math.h"
#include "align_lanes.h"
#include "align_wave.h"
#include "seed_finish.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_track;

namespace {

#ifndef ALIGN_BLOCK_VALUE
#define ALIGN_BLOCK_VALUE 128
#endif
// Trials per workgroup (a multiple of 64: every wave is on its own).  The launches after the first are sized for FULL queues
// -- how many trials survive is known on the device only -- and a workgroup past its queue's end leaves at once: with 64
// trials per workgroup the depth filter's 13 M seeds meant 205 k such workgroups per launch, ~90 + ~45 us of dispatcher time
// for the 2.7 % and 0.3 % of the seeds that get that far (profiles/r06r_*); 128: update_seeds -1.4 % / -1.1 % on two boxes,
// 256 no better (profiles/r06s_*, r06t_*).
constexpr int ALIGN_BLOCK = ALIGN_BLOCK_VALUE;

// Two waves per SIMD (<= 256 registers): with the bare __launch_bounds__(64) the compiler took 256 VGPRs plus 15-20
// AGPRs, i.e. ONE wave per SIMD, and nothing hid the round trip of an iteration's window fetch.
constexpr int ALIGN_MINW = 3;
struct NoFinish {};
// FINISH: the depth filter's call -- trial t is seed t, and a trial that ends in this launch (not active, converged, failed,
// out of iterations) goes straight on to seed_finish_seed with its verdict and pixel in registers (the gradient registers
// are dead by then); `fin` is the depth filter's argument block.
template <bool COUNT, bool FINISH>
__global__ void __launch_bounds__(ALIGN_BLOCK, ALIGN_MINW) align_kernel(const AlignArgs a, const typename std::conditional<FINISH, SeedArgs, NoFinish>::type fin) {
  // which trial: lane order in the first launch of a run, the queues filled by the previous launch afterwards
  // (workgroup b drains queue b % ALIGN_NQ, 64 entries at a time)
  int t;
  // (The 64 templates of a workgroup read as the contiguous 6400-byte block they are and handed out through LDS measured
  // slower twice, 12.25 against 11.90 ms per full-track step: every lane then waits for the whole block before its set-up
  // arithmetic can start, and the per-lane gathers of neighbouring words hit the same line in L1 anyway.)
  if (a.queue_in) {
    const int q = blockIdx.x % ALIGN_NQ, i = (blockIdx.x / ALIGN_NQ) * ALIGN_BLOCK + threadIdx.x;
    if (i >= a.n_in[q]) return;
    t = a.queue_in[(size_t)q * a.queue_cap + i];
  } else {
    t = blockIdx.x * ALIGN_BLOCK + threadIdx.x;
    if (t >= (a.M_dev ? min(*a.M_dev, a.M) : a.M)) return;
  }
  // Everything the trial needs before its first window -- the active flag, slot and level, the 25 template dwords, the
  // start pixel or the parked state, the 1-D flag and direction -- is requested before the first of these values is looked
  // at.  Testing the flag, then reading slot / level, then the template, then the pixel, then the 1-D flag, each behind
  // its own wait, was six memory round trips before the first of ~3 iterations of a phase, which is one round trip each
  // (285 -> 279 us per launch, profiles/r05a_queue_drain.txt).  Lanes that leave at once have read 130 bytes for nothing.
  // (a load under a condition -- even a uniform one -- is waited for inside its branch, where its value is turned into
  // a mask: the optional arrays are read through a stand-in pointer to memory that is always there instead)
  const bool first = a.it0 == 0;
  const bool has_act = first && a.active != nullptr;
  const uint8_t act_raw = (has_act ? a.active : a.pwb)[t];
  const int level = a.level[t];
  const int slot_t = a.slot[t];
  uint32_t g[25];
  {
    const uint32_t* gp = reinterpret_cast<const uint32_t*>(a.pwb + (size_t)t * 100);
#pragma unroll
    for (int k = 0; k < 25; ++k) g[k] = gp[k];
  }
  const double pin0 = a.px_in[2 * t], pin1 = a.px_in[2 * t + 1];
  // (the parked state of a resumed trial; in the first phase six floats of the template array, read and not used)
  const float* const sp = (first ? reinterpret_cast<const float*>(a.pwb) : a.state) + 6 * (size_t)t;
  const float s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5];
  const uint8_t u1d_raw = (a.use_1d ? a.use_1d : a.pwb)[t];
  const bool has_dir = a.use_1d != nullptr && a.dir != nullptr;
  const float* const dirp = has_dir ? a.dir : reinterpret_cast<const float*>(a.px_in);  // ([M][2] doubles: floats 2t, 2t + 1 exist)
  const float dir0 = dirp[2 * t], dir1 = dirp[2 * t + 1];
  const uint8_t u1d = a.use_1d ? u1d_raw : (uint8_t)0;
  if (has_act && act_raw == 0) {
    if (COUNT) a.iters[t] = 0;
    if constexpr (FINISH) {
      seed_finish_seed<true>(fin, t, false, 0, 0.0, 0.0);
    } else {
      a.ok[t] = 0;  // px_out is left as it is (findMatchDirect returns before touching px_cur)
    }
    return;
  }
  AlignState st;
  st.u = first ? (float)pin0 : s0;
  st.v = first ? (float)pin1 : s1;
  st.mean_diff = first ? 0.f : s2; st.chi2 = first ? 0.f : s3; st.up0 = first ? 0.f : s4; st.up1 = first ? 0.f : s5;
  const uint8_t* img = a.store + (int64_t)slot_t * a.L.slot_bytes + a.L.offset[level];
  const int cols = a.L.w[level], rows = a.L.h[level], pitch = a.L.pitch[level];
  bool wrote = true, ok = false, more;
  int n_eval = 0;
  const bool one_d = u1d != 0;
  if (one_d) {
    double h_inv = 0;
    more = align1d_lane(img, cols, rows, pitch, g, dir0, dir1, a.n_iter, a.it0, a.it1, st, h_inv, ok, wrote, n_eval);
    if (a.h_inv) a.h_inv[t] = h_inv;
  } else {
    more = align2d_lane(img, cols, rows, pitch, g, a.n_iter, a.it0, a.it1, st, ok, wrote, n_eval);
  }
  if (COUNT) a.iters[t] = (first ? 0 : a.iters[t]) + n_eval;
  if (a.queue_out) {
    // still iterating: park the loop state, append the trial to this workgroup's queue (one atomic per wave)
    const uint64_t going = __builtin_amdgcn_ballot_w64(more);
    if (more) {
      float* sp = a.state + 6 * (size_t)t;
      sp[0] = st.u; sp[1] = st.v; sp[2] = st.mean_diff; sp[3] = st.chi2; sp[4] = st.up0; sp[5] = st.up1;
      const int q = blockIdx.x % ALIGN_NQ;
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(going >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)going, 0u));
      int base = 0;
      if (rank == 0) base = atomicAdd(a.n_out + q, (int)__popcll(going));
      base = __builtin_amdgcn_readfirstlane(base);
      a.queue_out[(size_t)q * a.queue_cap + base + rank] = t;
      return;
    }
  }
  double ou = wrote ? (double)st.u : a.px_in[2 * t];
  double ov = wrote ? (double)st.v : a.px_in[2 * t + 1];
  if (a.scale_out) {
    ou = ou * (double)(1 << level);
    ov = ov * (double)(1 << level);
  }
  if constexpr (FINISH) {
    seed_finish_seed<true>(fin, t, true, ok ? 1 : 0, ou, ov);
  } else {
    a.ok[t] = ok ? 1 : 0;
    a.px_out[2 * t] = ou;
    a.px_out[2 * t + 1] = ov;
  }
}

// ---- small batches: one WAVE per trial (align_wave.h) -------------------------------------------------------------
// A camera frame's trials (~130 of the reprojector, ~330 seeds of the depth filter) are a handful of waves of the kernel
// above, each running as long as its slowest lane: ~27 us per launch in the single-stream drop-in.  With a wave per trial
// and a lane per template pixel an iteration is a fraction of that, and a frame's trials are spread over as many CUs as
// there are trials.  Same bits (align_wave.h says why); used below ALIGN_WAVE_MAX_M trials, where it is faster: the
// waves of 8192 trials are all resident at once, beyond ~16 k the lane kernel's 64 trials per wave win.
constexpr int ALIGNW_WAVES = 4;  // trials (= waves) per workgroup
template <bool COUNT>
__global__ void __launch_bounds__(64 * ALIGNW_WAVES) align_wave_kernel(const AlignArgs a) {
  __shared__ __attribute__((aligned(16))) float s_acc[ALIGNW_WAVES][3 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * ALIGNW_WAVES + wave;
  if (t >= (a.M_dev ? min(*a.M_dev, a.M) : a.M)) return;  // (the whole wave: no workgroup barrier below)
  // the trial's parameters, all requested before the first is looked at (every lane reads the same words)
  const bool has_act = a.active != nullptr;
  const uint8_t act_raw = (has_act ? a.active : a.pwb)[t];
  const int level_raw = a.level[t];
  const int slot_raw = a.slot[t];
  const double pin0 = a.px_in[2 * t], pin1 = a.px_in[2 * t + 1];
  const uint8_t u1d_raw = (a.use_1d ? a.use_1d : a.pwb)[t];
  const bool has_dir = a.use_1d != nullptr && a.dir != nullptr;
  const float* const dirp = has_dir ? a.dir : reinterpret_cast<const float*>(a.px_in);
  const float dir0 = dirp[2 * t], dir1 = dirp[2 * t + 1];
  const uint8_t* const tpl = a.pwb + (size_t)t * 100;
  if (has_act && uniform_int(act_raw) == 0) {
    if (lane == 0) {
      a.ok[t] = 0;  // px_out is left as it is (findMatchDirect returns before touching px_cur)
      if (COUNT) a.iters[t] = 0;
    }
    return;
  }
  const int level = uniform_int(level_raw), slot_t = uniform_int(slot_raw);
  const bool one_d = a.use_1d != nullptr && uniform_int(u1d_raw) != 0;
  AlignState st;
  st.u = (float)pin0;
  st.v = (float)pin1;
  st.mean_diff = 0.f; st.chi2 = 0.f; st.up0 = 0.f; st.up1 = 0.f;
  const uint8_t* img = a.store + (int64_t)slot_t * a.L.slot_bytes + a.L.offset[level];
  const int cols = a.L.w[level], rows = a.L.h[level], pitch = a.L.pitch[level];
  bool wrote = true, ok = false;
  int n_eval = 0;
  if (one_d) {
    double h_inv = 0;
    align1d_wave(img, cols, rows, pitch, tpl, dir0, dir1, a.n_iter, lane, s_acc[wave], st, h_inv, ok, wrote, n_eval);
    if (a.h_inv && lane == 0) a.h_inv[t] = h_inv;
  } else {
    align2d_wave(img, cols, rows, pitch, tpl, a.n_iter, lane, s_acc[wave], st, ok, wrote, n_eval);
  }
  if (lane != 0) return;
  if (COUNT) a.iters[t] = n_eval;
  a.ok[t] = ok ? 1 : 0;
  double ou = wrote ? (double)st.u : pin0;
  double ov = wrote ? (double)st.v : pin1;
  if (a.scale_out) {
    ou = ou * (double)(1 << level);
    ov = ov * (double)(1 << level);
  }
  a.px_out[2 * t] = ou;
  a.px_out[2 * t + 1] = ov;
}

}  // namespace

namespace svo_track {

// Alignment in phases.  A wave of one-lane-per-trial alignment runs as long as its slowest trial (on the
// representative full-track workload: 4.3 evaluations per trial, 9.8 per wave of 64), and the arithmetic order
// inside a trial must stay the reference's.  So the iterations are split over three launches -- 0..2, 3..5, 6.. --
// and between launches the trials still iterating are compacted: a launch parks the loop state of every unfinished
// trial (six floats, exactly the loop's locals) and appends its index to one of ALIGN_NQ queues; the next launch
// runs dense waves over the queues.  A resumed trial rebuilds H^-1 from its template (same instructions, same
// bits) and continues where it stopped: results are identical to the single launch.
#ifndef ALIGN_PHASE_MIN_M_VALUE  // (the CPU emulation of the test suite lowers it to reach the phased path with small batches)
#define ALIGN_PHASE_MIN_M_VALUE (1 << 16)
#endif
constexpr int ALIGN_PHASE_MIN_M = ALIGN_PHASE_MIN_M_VALUE;  // below this the extra launches cost more than the idle lanes
// (other boundaries measured slower on the representative workload -- {0, 2, 4}: +2.5 % on update_seeds, {0, 1, 3}: +6 %,
// four launches {0, 2, 4, 7}: +4 %, {0, 1, 2, 4}: +12 %: a launch re-reads the template and rebuilds H^-1; profiles/r06k_*)
constexpr int ALIGN_PHASE_START[] = {0, 3, 6};  // first iteration of every launch
constexpr int ALIGN_N_PHASES = (int)(sizeof(ALIGN_PHASE_START) / sizeof(int));
constexpr int ALIGN_PHASE_ITERS = ALIGN_PHASE_START[1];
#ifndef ALIGN_WAVE_MAX_M_VALUE  // (the CPU emulation of the test suite has a build with 0: every batch on the lane kernel)
#define ALIGN_WAVE_MAX_M_VALUE 8192
#endif
constexpr int ALIGN_WAVE_MAX_M = ALIGN_WAVE_MAX_M_VALUE;  // batches up to this many trials take the wave-per-trial kernel

static int phase_queue_cap(int M) { return (M + ALIGN_NQ - 1) / ALIGN_NQ + 2 * ALIGN_BLOCK; }

size_t align_phase_workspace_bytes(int M) {
  if (M < ALIGN_PHASE_MIN_M) return 0;
  const size_t cap = (size_t)phase_queue_cap(M);
  return 2 * Carver::round(ALIGN_NQ * cap * sizeof(int32_t)) + Carver::round((ALIGN_N_PHASES - 1) * ALIGN_NQ * sizeof(int32_t)) +
         Carver::round((size_t)M * 6 * sizeof(float));
}

static int launch_wave(const AlignArgs& a, hipStream_t s) {
  const dim3 grid((a.M + ALIGNW_WAVES - 1) / ALIGNW_WAVES), blk(64 * ALIGNW_WAVES);
  if (a.iters) hipLaunchKernelGGL(align_wave_kernel<true>, grid, blk, 0, s, a);
  else hipLaunchKernelGGL(align_wave_kernel<false>, grid, blk, 0, s, a);
  return check_launch();
}

static int launch_one(const AlignArgs& a, int n_blocks, hipStream_t s, const SeedArgs* fin) {
  const dim3 grid(n_blocks), blk(ALIGN_BLOCK);
  if (fin) {
    if (a.iters) hipLaunchKernelGGL((align_kernel<true, true>), grid, blk, 0, s, a, *fin);
    else hipLaunchKernelGGL((align_kernel<false, true>), grid, blk, 0, s, a, *fin);
  } else {
    if (a.iters) hipLaunchKernelGGL((align_kernel<true, false>), grid, blk, 0, s, a, NoFinish{});  // instrumented: also counts evaluations
    else hipLaunchKernelGGL((align_kernel<false, false>), grid, blk, 0, s, a, NoFinish{});
  }
  return check_launch();
}

bool align_takes_finish(int M) { return M > ALIGN_WAVE_MAX_M; }

int launch_align(const AlignArgs& a0, hipStream_t s, void* d_phase_ws, size_t phase_ws_bytes, const SeedArgs* fin) {
  if (a0.M <= 0) return SVO_HIP_OK;
  if (fin && !align_takes_finish(a0.M)) return SVO_HIP_EINVAL;
  const int all_blocks = (a0.M + ALIGN_BLOCK - 1) / ALIGN_BLOCK;
  const size_t need = align_phase_workspace_bytes(a0.M);
  const bool phased = d_phase_ws && need != 0 && phase_ws_bytes >= need && a0.n_iter > ALIGN_PHASE_ITERS;
  if (!phased) return a0.M <= ALIGN_WAVE_MAX_M ? launch_wave(a0, s) : launch_one(a0, all_blocks, s, fin);
  Carver c(d_phase_ws, phase_ws_bytes);
  const int cap = phase_queue_cap(a0.M);
  int32_t* queue[2] = {c.take<int32_t>((size_t)ALIGN_NQ * cap), c.take<int32_t>((size_t)ALIGN_NQ * cap)};
  int32_t* count = c.take<int32_t>((ALIGN_N_PHASES - 1) * ALIGN_NQ);
  float* state = c.take<float>((size_t)a0.M * 6);
  if (!c.ok) return launch_one(a0, all_blocks, s, fin);
  SVO_HIP_TRY(hipMemsetAsync(count, 0, (ALIGN_N_PHASES - 1) * ALIGN_NQ * sizeof(int32_t), s));
  // a queue holds the survivors of the workgroups b % ALIGN_NQ == q: at most cap entries; every launch after the first
  // is sized for full queues (workgroups past a queue's end leave at once)
  const int queue_blocks = ALIGN_NQ * ((cap + ALIGN_BLOCK - 1) / ALIGN_BLOCK);
  AlignArgs a = a0;
  a.state = state;
  a.queue_cap = cap;
  int rc = SVO_HIP_OK;
  for (int phase = 0; phase < ALIGN_N_PHASES && rc == SVO_HIP_OK; ++phase) {
    const bool last = phase == ALIGN_N_PHASES - 1;
    a.it0 = ALIGN_PHASE_START[phase];
    a.it1 = last ? (1 << 30) : ALIGN_PHASE_START[phase + 1];
    a.queue_in = phase == 0 ? nullptr : queue[(phase - 1) & 1];
    a.n_in = phase == 0 ? nullptr : count + (phase - 1) * ALIGN_NQ;
    a.queue_out = last ? nullptr : queue[phase & 1];
    a.n_out = last ? nullptr : count + phase * ALIGN_NQ;
    rc = launch_one(a, phase == 0 ? all_blocks : queue_blocks, s, fin);
  }
  return rc;
}
}  // namespace svo_track

static int align_batch(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M, const int32_t* d_slot,
                       const int32_t* d_level, const uint8_t* d_patch_with_border, const float* d_dir,
                       const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok, double* d_h_inv,
                       int32_t* d_iters, void* stream, void* d_phase_ws = nullptr, size_t phase_ws_bytes = 0) {
  if (!layout_ok(layout) || !d_store || M < 0 || n_iter < 0) return SVO_HIP_EINVAL;
  if (M == 0) return SVO_HIP_OK;
  if (!d_slot || !d_level || !d_patch_with_border || !d_px || !d_ok) return SVO_HIP_EINVAL;
  if (d_use_1d && !d_dir) return SVO_HIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(d_patch_with_border) & 3) != 0) return SVO_HIP_EINVAL;
  AlignArgs a;
  a.L = *layout;
  a.store = d_store;
  a.M = M;
  a.slot = d_slot;
  a.level = d_level;
  a.pwb = d_patch_with_border;
  a.dir = d_dir;
  a.use_1d = d_use_1d;
  a.active = nullptr;
  a.n_iter = n_iter;
  a.px_in = d_px;
  a.px_out = d_px;
  a.scale_out = 0;
  a.ok = d_ok;
  a.h_inv = d_h_inv;
  a.iters = d_iters;
  return launch_align(a, static_cast<hipStream_t>(stream), d_phase_ws, phase_ws_bytes);
}

extern "C" size_t svo_hip_align_workspace_bytes(int M) { return M < 0 ? 0 : align_phase_workspace_bytes(M); }

extern "C" int svo_hip_align_batch_phased(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M,
                                          const int32_t* d_slot, const int32_t* d_level,
                                          const uint8_t* d_patch_with_border, const float* d_dir, const uint8_t* d_use_1d,
                                          int n_iter, double* d_px, int32_t* d_ok, double* d_h_inv, int32_t* d_evaluations,
                                          void* d_workspace, size_t workspace_bytes, void* stream) {
  if (M > 0 && workspace_bytes < align_phase_workspace_bytes(M)) return SVO_HIP_ERANGE;
  if ((reinterpret_cast<uintptr_t>(d_workspace) & 255) != 0) return SVO_HIP_EINVAL;
  return align_batch(layout, d_store, M, d_slot, d_level, d_patch_with_border, d_dir, d_use_1d, n_iter, d_px, d_ok, d_h_inv,
                     d_evaluations, stream, d_workspace, workspace_bytes);
}

extern "C" int svo_hip_align_batch(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M,
                                   const int32_t* d_slot, const int32_t* d_level,
                                   const uint8_t* d_patch_with_border, const float* d_dir, const uint8_t* d_use_1d,
                                   int n_iter, double* d_px, int32_t* d_ok, double* d_h_inv, void* stream) {
  return align_batch(layout, d_store, M, d_slot, d_level, d_patch_with_border, d_dir, d_use_1d, n_iter, d_px, d_ok, d_h_inv,
                     nullptr, stream);
}

extern "C" int svo_hip_align_batch_counted(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M,
                                           const int32_t* d_slot, const int32_t* d_level,
                                           const uint8_t* d_patch_with_border, const float* d_dir,
                                           const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok,
                                           double* d_h_inv, int32_t* d_evaluations, void* stream) {
  if (!d_evaluations) return SVO_HIP_EINVAL;
  return align_batch(layout, d_store, M, d_slot, d_level, d_patch_with_border, d_dir, d_use_1d, n_iter, d_px, d_ok, d_h_inv,
                     d_evaluations, stream);
}
