// sparse_align.hip -- K1: batched sparse image alignment for gfx950.
//
// Replaces svo::SparseImgAlign::run and the vk::NLLSSolver Gauss-Newton loop
// it drives (svo/src/sparse_img_align.cpp:43-258).  One workgroup owns one
// (reference frame, current frame) problem for the whole coarse-to-fine
// schedule -- no relaunch per iteration or level.  One lane owns one 4x4 patch:
//
//   per level   (precomputeReferencePatches, :84-145)
//     lane reads its 7x7 u8 window of the reference level (7 rows x 3 aligned
//     dwords), keeps the 16 interpolated intensities and the 16 (dx,dy)
//     gradients of the interpolated image in registers, plus the two rows
//     a = jac.row(0)*f/2^l, b = jac.row(1)*f/2^l of the 2x6 projection
//     Jacobian (frame.h:116-138).  The per-pixel Jacobian of the reference,
//     J = dx*a + dy*b, is never materialised.
//   per iteration (computeResiduals, :147-243)
//     project (f64), floor, border test, 5 rows x 2 aligned dwords of the
//     current level, bilinear warp in registers, res = I - ref.  Because
//     J = dx*a + dy*b the normal equations factor per patch:
//        Jres -= (sum res*dx) a + (sum res*dy) b
//        H    += Sxx aa' + Sxy (ab'+ba') + Syy bb'        (constant per level)
//     so an iteration reduces only 8 numbers (6 Jres, chi2, #meas) over the
//     workgroup; the 21 unique entries of H and its LDL' factors are rebuilt
//     only when the set of patches inside the current image changes.
//   serial point (solve/update, :245-258 + vk::NLLSSolver::optimizeGaussNewton)
//     one wave of the workgroup (rotating with the workgroup id) sums the per-wave
//     partials from LDS, multiplies by the cached H^-1, applies the stop / rollback
//     rules and publishes the new pose through LDS; the others wait at a barrier.
//
// Numerics: pixel math in f32 (like the reference); projection, the Jacobian rows, the Jres / H partials and their
// reductions, the solve and the pose in f64 (like the reference).  The 16 per-pixel products of a patch and their sums
// are f32 (the reference forms them in f64): exact products of 24-bit factors, summed 16 at a time.  Sums are tree-
// instead of sequentially reduced, so results agree to rounding, not bit-for-bit.
// -DSIA_F64_PARTIALS builds the REFERENCE-WIDTH variant of this kernel: also the per-pixel products and SE3::exp in f64
// like H_, Jres_ and jacobian_cache_ of the reference (sparse_img_align.cpp:228-230,253-258).  bench.py times it next to
// the default (`f64_partials` / `roofline_f64_build` in the JSON line) so the price of that width is a number.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "sia_common.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_sia;

namespace {

// Waves per SIMD asked of the register allocator.  With the Jacobian products kept inside the iteration
// loop (see the opaque copies in the H rebuild) and the Gauss-Jordan rebuild the 256-lane instantiation
// fits 128 VGPRs = 4 waves per SIMD WITH the window cache, i.e. four frames per CU instead of three
// (1.40 against 1.50 ms on the headline batch); one 8-byte value is spilled once per level.  64/128-lane
// workgroups are not limited by registers.  The distorted-camera instantiations (no window cache, the model's
// world2cam in the loop) stay at 3 waves per SIMD where the workgroup size allows: at 4 they spill 22 dwords.
// (Five frames per CU -- the reference tile cut to 208 patches, 96 VGPRs -- measured 24 % slower in round 4:
// profiles/r04k_k1_five_frames_per_cu.txt.)
#define MINW(BLOCK) ((BLOCK) >= 256 ? 4 : 3)

// SIA_PROFILE: per-phase shader-clock totals of wave 0 of every workgroup, written over H_out[b][0..7]
// (pixel work, reduce, barrier 1, H rebuild, solve + barrier 2, per-level precompute, total, #iterations).
// Instrumentation build only (scripts/k1_phase_profile.py); costs ~10 % itself.
#ifdef SIA_PROFILE
#define SIA_T() ((long long)__builtin_readcyclecounter())
#define SIA_ACC(k, t0, t1) prof[k] += (t1) - (t0)
#else
#define SIA_T() 0ll
#define SIA_ACC(k, t0, t1)
#endif

constexpr int MAX_WAVES = SVO_HIP_MAX_PATCHES / 64;

// Arithmetic widths.  sia_pix: the per-pixel products res*dx, res*dy, dx*dx ... and their 16-term sums over a patch;
// sia_acc: everything from the patch upwards -- the Jacobian rows a / b, the per-lane Jres and H partials, the wave
// reductions, the per-wave partials in LDS.  The reference keeps all of it in f64 (jacobian_cache_, H_, Jres_:
// sparse_img_align.cpp:139-140,228-230).  Default: products of two f32 values summed 16 at a time in f32, the rest in
// f64 -- 13.5 M frames/s against 14.9 M with the rows and partials in f32 as in rounds 2-4, and the same agreement with the
// reference as the all-f64 build below (max 1.0e-5 / 1.4e-5, 99.5 % identical iteration counts over 8192 frames:
// profiles/r05b_k1_width_ab.txt).
//   -DSIA_F64_PARTIALS   the reference's width throughout (also SE3::exp in f64): three waves per SIMD, 10.5 M frames/s
#if defined(SIA_F64_PARTIALS)
using sia_pix = double;
using sia_acc = double;
#undef MINW
#define MINW(BLOCK) 3  // the f64 accumulators do not fit 128 VGPRs
#else
using sia_pix = float;
using sia_acc = double;
#endif

// Gauss-Newton state (quaternion form, as Sophus stores it), shared by the TWO waves of a workgroup that run the serial
// solve / update step of an iteration between them (one solver wave: 1.139 against 1.119 ms per 16384-frame launch,
// profiles/r04j_k1_split_solve_ab.txt).  q and t are double-buffered by update parity: an update reads buffer `cur` and
// writes buffer `cur ^ 1`, so the wave that composes the translation can read the quaternion the other wave is
// replacing, and old_model (the rollback of vk::NLLSSolver) is simply the buffer the last update came from -- no copies.
struct SplitModel {
  double q[2][4], t[2][3];
};

struct SiaLds {
  SplitModel sm;
  double Rt[12];                 // pose published by the solver wave of the iteration
  int sw_done, sw_stop, sw_nmeas, sw_cur, sw_old;
  double sw_chi2;
  double H[21];                  // H_ of the last evaluated iteration (packed upper triangle)
  double Hinv[36];               // its inverse, row-major
  double A[36];                  // Gauss-Jordan scratch
  sia_acc part[2][MAX_WAVES][8];  // per-wave partials of Jres[6], chi2, n_meas (double-buffered)
  int chg[2][MAX_WAVES];          // per wave: some patch entered or left the current image (same buffering)
  int xfail;                      // ... some lane of the workgroup gave up waiting for a sibling part
  sia_acc Hpart[MAX_WAVES][24];   // per-wave partials of H (21 used)
  long long lo[SVO_HIP_MAX_LEVELS];  // pyramid geometry per level (copied from the kernel
  int lw[SVO_HIP_MAX_LEVELS];        // arguments so the level loop can index it dynamically)
  int lh[SVO_HIP_MAX_LEVELS];
  int lp[SVO_HIP_MAX_LEVELS];
};
__shared__ SiaLds g_s;


// H = sum of the per-wave partials, then H^-1 by Gauss-Jordan on a 6x6 tile held one
// element per lane (36 lanes), LDS as the row/column exchange.  Wave 0 only; runs
// once per level (or when the set of patches inside the current image changes).
// Same-wave LDS traffic: DS instructions of one wave execute in order; the
// wavefront-scope fences stop the compiler from moving accesses across the exchanges.
// PARTS > 1: the workgroup's sums are one part of the frame's; the parts exchange them (xh: the frame's H chunks of this
// exchange's buffer half, [PARTS][SIA_XH_SLOTS]) and every part adds them up in the same order.
template <int PARTS>
__device__ __forceinline__ void sia_rebuild_hinv(int lane, int nw, [[maybe_unused]] XChunk* xh, [[maybe_unused]] int part,
                                                 [[maybe_unused]] unsigned gen, [[maybe_unused]] int& xfail) {
  asm volatile("" : "+v"(lane));  // keep this cold block's address math out of the caller's loops
#ifndef SVO_HOST_MATH_TEST
  if constexpr (PARTS > 1) {
    // lane = (j-th of the OTHER parts, element k): the 3 x 21 chunks are polled side by side -- one memory round trip, where
    // a lane per element polling its three chunks one after the other paid three (lane 63 idles)
    static_assert((PARTS - 1) * 21 <= 64, "one lane per chunk of the other parts");
    const int j = lane / 21, k = lane - 21 * j;
    double v = 0.0;
    for (int w = 0; w < nw; ++w) v += (double)g_s.Hpart[w][k];
    if (j == 0) sia_xstore(xh + part * SIA_XH_SLOTS + k, v, gen);
    const int other = j < part ? j : j + 1;
    double got = 0.0;
    if (j < PARTS - 1) got = sia_xpoll(xh + other * SIA_XH_SLOTS + k, gen, xfail);
    if (xfail) g_s.xfail = 1;
    // the parts in a fixed order: every part forms the same sums
    double tot = 0.0;
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
      const double theirs = __shfl(got, (p < part ? p : p - 1) * 21 + k, 64);  // (p == part: not used)
      const double term = p == part ? v : theirs;
      tot = p == 0 ? term : tot + term;
    }
    if (lane < 21) g_s.H[lane] = tot;
  } else
#endif
  if (lane < 21) {
    double v = 0.0;
    for (int w = 0; w < nw; ++w) v += (double)g_s.Hpart[w][lane];
    g_s.H[lane] = v;
  }
  SVO_WAVE_LDS_FENCE();
  const int gi = lane / 6, gj = lane - 6 * gi;
  if (lane < 36) {
    g_s.A[lane] = g_s.H[sym6_rt(gi, gj)];
    g_s.Hinv[lane] = (gi == gj) ? 1.0 : 0.0;
  }
  for (int k = 0; k < 6; ++k) {
    SVO_WAVE_LDS_FENCE();
    if (lane < 36) {
      const double p = g_s.A[k * 6 + k];
      const double aik = g_s.A[gi * 6 + k];
      const double akj = g_s.A[k * 6 + gj];
      const double bkj = g_s.Hinv[k * 6 + gj];
      const double aij = g_s.A[lane];
      const double bij = g_s.Hinv[lane];
      // a zero pivot contributes nothing, like the D^-1 step of Eigen's LDLT::solve
      // (sia_rcp, not 1.0 / p: six dependent divisions per rebuild of H^-1 -- K1 1.220 -> 1.207 ms together with the patch's
      // normalised coordinates, profiles/r06ah_*)
      const double ip = (fabs(p) > 2.2250738585072014e-308) ? sia_rcp(p) : 0.0;
      const double na = (gi == k) ? akj * ip : aij - aik * (akj * ip);
      const double nb = (gi == k) ? bkj * ip : bij - aik * (bkj * ip);
      SVO_LANES_LDS_FENCE();
      g_s.A[lane] = na;
      g_s.Hinv[lane] = nb;
    }
  }
}

// (A fetch-ahead of the next level's windows -- global_load_lds touches of their corner tiles during iteration 0 -- measured
// worse, 1.30 against 1.21 ms: the extra gathers cost more L1 look-ups and in-order vmcnt waiting than warm lines give back.)

// DIST: the camera is a distorted model (radial-tangential pinhole or ATAN); the undistorted pinhole
// keeps its own instantiation so that its inner loop carries no model dispatch.
// The bilinear sample, with the order of its one multiplication and three fused multiply-adds spelled out: the template
// (precompute block) and the warped patch (every evaluation, scalar or packed) must round alike -- a frame aligned
// against itself has residuals that are exactly zero -- and left to itself the compiler may fuse a*b + c*d either way
// round, differently for scalars and for register pairs.
__device__ __forceinline__ float sia_bilerp(float wtl, float wtr, float wbl, float wbr, float tl, float tr, float bl, float br) {
  return __builtin_fmaf(wbr, br, __builtin_fmaf(wbl, bl, __builtin_fmaf(wtr, tr, wtl * tl)));
}
typedef float sia_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sia_f2 sia_bilerp2(sia_f2 wtl, sia_f2 wtr, sia_f2 wbl, sia_f2 wbr, sia_f2 tl, sia_f2 tr, sia_f2 bl,
                                              sia_f2 br) {
  return __builtin_elementwise_fma(wbr, br, __builtin_elementwise_fma(wbl, bl, __builtin_elementwise_fma(wtr, tr, wtl * tl)));
}

// SIA_PACKED: the pixel loop of an evaluation on v_pk_*_f32, two pixels per instruction; the reference-width build
// keeps its f64 sums and the scalar loop.
#ifdef SIA_F64_PARTIALS
#define SIA_PACKED 0
#else
#define SIA_PACKED 1
#endif
// PARTS > 1 (round 6): a frame of more than 512 patches in a batch too small to fill the GPU -- BASELINE configs[3]: 1000
// patches, 64 frames, i.e. 64 of 256 CUs busy with 16 waves each -- is split over PARTS workgroups of BLOCK lanes, part p
// owning patches [p BLOCK, (p + 1) BLOCK).  An iteration's eight sums (and the "membership changed" flag, and H when it is
// rebuilt) are exchanged between the parts as write-through 16-byte chunks (sia_common.h: XChunk); every part then
// solves and updates for itself -- the same sums in the same order, hence the same pose -- so one exchange per iteration
// is all the parts ever say to each other.  Part 0 writes the results.
template <int BLOCK, bool WC, bool DIST, int PARTS = 1>
__global__ void __launch_bounds__(BLOCK, PARTS > 1 ? 1 : ((DIST && BLOCK < 1024) ? 3 : MINW(BLOCK))) sia_kernel(const SiaArgs a) {
  constexpr int NW = BLOCK / 64;
  // XCD-aware problem order (capi_common.h): in a replay batch consecutive problems share a frame
  // (frame b+1 is the current image of problem b and the reference image of problem b+1)
  int b, part = 0;
  if constexpr (PARTS > 1) {
    // the parts of a frame 8 workgroup ids apart: usually the same XCD (speed only: the exchange does not rely on it)
    b = (int)(blockIdx.x / (8 * PARTS)) * 8 + (int)(blockIdx.x % 8);
    part = (int)(blockIdx.x / 8) % PARTS;
    if (b >= a.B) return;
  } else {
    b = (int)xcd_contiguous_block();
  }
  [[maybe_unused]] XChunk* const xbase = PARTS > 1 ? static_cast<XChunk*>(a.xw) + (size_t)b * SIA_X_BLOCK : nullptr;
  // epoch of the last exchange (uniform over the frame's parts): continues from the block's last user (sia_common.h).  No
  // part can overwrite the word before all have read it: part 0 writes it at the end, behind exchanges every part joined.
  [[maybe_unused]] unsigned xseq = 0;
#ifndef SVO_HOST_MATH_TEST
  if constexpr (PARTS > 1) xseq = (unsigned)__builtin_amdgcn_readfirstlane((int)sia_xload(xbase + SIA_X_CHUNKS).lo);
#endif
  [[maybe_unused]] int xfail = 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // Interpolated reference image around this lane's patch, staged in LDS: the
  // bilinear sample at window pixel (r,c) of the 6x6 neighbourhood (corners unused
  // -> 32 floats = 8 float4 per lane, laid out [8][BLOCK] so a wave reads 1 KiB
  // contiguous per ds_read_b128).  ref_patch_cache_(y,x) = Bt[y+1][x+1]; the gradients
  // dx,dy (:133-136) are central differences of Bt, rebuilt in flight.
  //   q0 = r0 c1..4 | q1 = r1 c0..3 | q2 = r1 c4,5 r2 c0,1 | q3 = r2 c2..5
  //   q4 = r3 c0..3 | q5 = r3 c4,5 r4 c0,1 | q6 = r4 c2..5 | q7 = r5 c1..4
  __shared__ float4 s_bt[8][BLOCK];

  int n = a.n[b];
  n = n > a.n_stride ? a.n_stride : n;  // contract: n <= n_stride (svo_hip.h); never read the next problem's rows
  const svo_hip_sia_params P = a.P;

  if (n <= 0) {  // sparse_img_align.cpp:47-51: nothing to track, pose untouched
    if (tid == 0 && part == 0) {
      for (int k = 0; k < 12; ++k) a.T_out[12 * b + k] = a.T_in[12 * b + k];
      if (a.H_out)
        for (int k = 0; k < 36; ++k) a.H_out[36 * b + k] = 0.0;
      a.n_tracked[b] = 0;
      if (a.iters)
        for (int k = 0; k < SVO_HIP_MAX_LEVELS; ++k) a.iters[SVO_HIP_MAX_LEVELS * b + k] = 0;
      if (a.chi2) a.chi2[b] = 1e10;
      if (a.status) a.status[b] = 0;
    }
    return;
  }

  // ---- per-lane geometry (Feature::px is re-read per level, f*depth kept) --
  const int pid = part * BLOCK + tid;  // this lane's patch
  const size_t fo = (size_t)b * a.n_stride + pid;
  const bool has = (pid < n) && (a.valid ? a.valid[fo] != 0 : true);
  double X = 0, Y = 0, Z = 1;
  if (has) {
    X = a.xyz[3 * fo];
    Y = a.xyz[3 * fo + 1];
    Z = a.xyz[3 * fo + 2];
  }
  // Feature::px kept in four registers instead of re-read at the top of every level: the re-read is a memory round trip
  // the level's reference-window gather has to wait for, i.e. two dependent round trips per level where the gathers could
  // leave together (1.122 -> 1.101 ms per 16 384 frames, profiles/r05a_queue_drain.txt)
  double px_kept0 = 0, px_kept1 = 0;
  if (has) {
    px_kept0 = a.px[2 * fo];
    px_kept1 = a.px[2 * fo + 1];
  }
  // normalised coordinates of xyz_ref: all of Frame::jacobian_xyz2uv (frame.h:116-138)
  // is a function of (x/z, y/z, 1/z)
  const double rz = sia_rcp(Z);
  const sia_acc zi = (sia_acc)rz;
  const sia_acc xn = (sia_acc)(X * rz);
  const sia_acc yn = (sia_acc)(Y * rz);
  const uint8_t* ref_base = a.store + (int64_t)a.ref_slot[b] * a.L.slot_bytes;
  const uint8_t* cur_base = a.store + (int64_t)a.cur_slot[b] * a.L.slot_bytes;

  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < SVO_HIP_MAX_LEVELS; ++k) {
      g_s.lw[k] = a.L.w[k];
      g_s.lh[k] = a.L.h[k];
      g_s.lp[k] = a.L.pitch[k];
      g_s.lo[k] = a.L.offset[k];
    }
    for (int k = 0; k < 21; ++k) g_s.H[k] = 0.0;
    g_s.xfail = 0;
    if (a.iters && part == 0)
      for (int k = 0; k < SVO_HIP_MAX_LEVELS; ++k) a.iters[SVO_HIP_MAX_LEVELS * b + k] = 0;
  }
  // model of this wave: every lane computes the same values, lane 0 stores them
  double R[9], tr[3];
  {
    double q[4];
    for (int k = 0; k < 9; ++k) R[k] = a.T_in[12 * b + k];
    for (int k = 0; k < 3; ++k) tr[k] = a.T_in[12 * b + 9 + k];
    quat_from_R(R, q);
    quat_to_R(q, R);
    if (tid == 0) {
      for (int k = 0; k < 4; ++k) g_s.sm.q[0][k] = q[k];
      for (int k = 0; k < 3; ++k) g_s.sm.t[0][k] = tr[k];
    }
  }
  int m_cur = 0, m_old = 0;  // parity buffer holding the model / old_model (uniform over the workgroup)
  // vk::NLLSSolver::reset(): wave-uniform solver state
  double chi2_prev = 1e10;
  int stop = 0;
  int n_meas_last = 0;
  int buf = 0;  // which half of g_s.part this iteration writes

  // which wave runs the serial solve/update step (see below)
  const int sw = (NW > 1) ? (int)(blockIdx.x % NW) : 0;
  sia_pix Sxx = 0, Sxy = 0, Syy = 0;
  sia_acc gmask = 0;  // 0 while this lane's Jacobian columns are zero at this level
  bool vis = false;   // visible_fts_[i]; never cleared between levels (:57)

  __syncthreads();

#ifdef SIA_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = SIA_T();
#endif
  for (int level = P.max_level; level >= P.min_level; --level) {
    [[maybe_unused]] const long long tl0 = SIA_T();
    const int cols = g_s.lw[level], rows = g_s.lh[level], pitch = g_s.lp[level];
    const uint8_t* ref_img = ref_base + g_s.lo[level];
    const uint8_t* cur_img = cur_base + g_s.lo[level];
    const float scale = pow2_inv_f32(level);  // == 1.0f / (float)(1 << level), from exponent bits (no division sequence)
    // focal_length / 2^level (:139-140), folded into the per-patch sums
    const sia_acc fl = (sia_acc)(fabs(P.fx) * pow2_inv_f64(level));  // == / 2^level, bit for bit

    uint32_t wc[WC ? 7 : 1][3];
    int wc_u0 = 0, wc_v0 = -100000;  // cached columns [wc_u0, wc_u0+11], rows [wc_v0, wc_v0+6]
    if (WC) {
#pragma unroll
      for (int r = 0; r < (WC ? 7 : 1); ++r) wc[r][0] = wc[r][1] = wc[r][2] = 0u;
    }
    // ---- precomputeReferencePatches (:84-145) ----------------------------
    {
      const double pxx = px_kept0, pxy = px_kept1;
      const float u_ref = (float)(pxx * (double)scale);
      const float v_ref = (float)(pxy * (double)scale);
      const int u_i = (int)floorf(u_ref);
      const int v_i = (int)floorf(v_ref);
      const bool inb = has && !(u_i - 3 < 0 || v_i - 3 < 0 || u_i + 3 >= cols || v_i + 3 >= rows);
      Sxx = Sxy = Syy = 0;
      if (inb) {
        vis = true;
        gmask = 1;
        const float su = u_ref - (float)u_i, sv = v_ref - (float)v_i;
        // The reference forms these products in double and rounds to float (:118-121).  u >= 3 here, so su
        // and sv are multiples of 2^-22: 1-su and 1-sv are exact in f32, and the f32 product of two f32
        // values is the correctly rounded exact product, like the rounded double product -- bit-identical.
        const float wtl = (1.f - su) * (1.f - sv);
        const float wtr = su * (1.f - sv);
        const float wbl = (1.f - su) * sv;
        const float wbr = su * sv;
        float Bt[6][6];
        float Wp[7], Wc[7];
        // the 7 x 7 window as 7 runs of 12 bytes, placed inside one tile row of the store where the 7 bytes allow it
        uint32_t rw[7][3];
        const int rxa = run_start(u_i - 3, 7);
        const uint32_t rbo = (uint32_t)(u_i - 3 - rxa);  // 0..5
        load_window12<7>(ref_img, pitch, rxa, v_i - 3, rw);
        cut_row7(rw[0], rbo, Wp);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          cut_row7(rw[r + 1], rbo, Wc);
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const bool need = ((r >= 1 && r <= 4)) || ((c >= 1 && c <= 4));
            if (need) Bt[r][c] = sia_bilerp(wtl, wtr, wbl, wbr, Wp[c], Wp[c + 1], Wc[c], Wc[c + 1]);
          }
#pragma unroll
          for (int c = 0; c < 7; ++c) Wp[c] = Wc[c];
        }
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float dx = 0.5f * (Bt[y + 1][x + 2] - Bt[y + 1][x]);
            const float dy = 0.5f * (Bt[y + 2][x + 1] - Bt[y][x + 1]);
            Sxx += (sia_pix)dx * (sia_pix)dx;
            Sxy += (sia_pix)dx * (sia_pix)dy;
            Syy += (sia_pix)dy * (sia_pix)dy;
          }
#if SIA_PACKED
        // columns in the order the packed pixel loop pairs them: (0,2) (1,3) (4,5) per row, (1,3) (2,4) in rows 0 and 5
        s_bt[0][tid] = make_float4(Bt[0][1], Bt[0][3], Bt[0][2], Bt[0][4]);
        s_bt[1][tid] = make_float4(Bt[1][0], Bt[1][2], Bt[1][1], Bt[1][3]);
        s_bt[2][tid] = make_float4(Bt[1][4], Bt[1][5], Bt[2][0], Bt[2][2]);
        s_bt[3][tid] = make_float4(Bt[2][1], Bt[2][3], Bt[2][4], Bt[2][5]);
        s_bt[4][tid] = make_float4(Bt[3][0], Bt[3][2], Bt[3][1], Bt[3][3]);
        s_bt[5][tid] = make_float4(Bt[3][4], Bt[3][5], Bt[4][0], Bt[4][2]);
        s_bt[6][tid] = make_float4(Bt[4][1], Bt[4][3], Bt[4][4], Bt[4][5]);
        s_bt[7][tid] = make_float4(Bt[5][1], Bt[5][3], Bt[5][2], Bt[5][4]);
#else
        s_bt[0][tid] = make_float4(Bt[0][1], Bt[0][2], Bt[0][3], Bt[0][4]);
        s_bt[1][tid] = make_float4(Bt[1][0], Bt[1][1], Bt[1][2], Bt[1][3]);
        s_bt[2][tid] = make_float4(Bt[1][4], Bt[1][5], Bt[2][0], Bt[2][1]);
        s_bt[3][tid] = make_float4(Bt[2][2], Bt[2][3], Bt[2][4], Bt[2][5]);
        s_bt[4][tid] = make_float4(Bt[3][0], Bt[3][1], Bt[3][2], Bt[3][3]);
        s_bt[5][tid] = make_float4(Bt[3][4], Bt[3][5], Bt[4][0], Bt[4][1]);
        s_bt[6][tid] = make_float4(Bt[4][2], Bt[4][3], Bt[4][4], Bt[4][5]);
        s_bt[7][tid] = make_float4(Bt[5][1], Bt[5][2], Bt[5][3], Bt[5][4]);
#endif
      } else {
        // jacobian_cache_.setZero() (:64): the J columns of a feature skipped here
        // stay zero; a stale ref_patch_cache_ row (if any) is kept
        gmask = 0;
      }
    }

    // ---- vk::NLLSSolver::optimizeGaussNewton ------------------------------
    // old_model = model at the start of every optimize() call: a stop at the first evaluation of
    // this level (NaN solve, stop_ carried over) keeps the pose the previous level ended with
    m_old = m_cur;
    int inH = -1;  // membership of this lane in the sum that g_s.H currently holds
    int evals = 0;
    SIA_ACC(5, tl0, SIA_T());
    for (int iter = 0; iter < P.n_iter; ++iter) {
      [[maybe_unused]] const long long tp0 = SIA_T();
      // -- computeResiduals (:147-243): this lane's patch -------------------
      bool m = false;
      sia_pix gx = 0, gy = 0;
      float c2 = 0.f;
      if (vis) {
        const double xc = R[0] * X + R[1] * Y + R[2] * Z + tr[0];
        const double yc = R[3] * X + R[4] * Y + R[5] * Z + tr[1];
        const double zc = R[6] * X + R[7] * Y + R[8] * Z + tr[2];
        // cam_->world2cam(project2d(xyz)) (:183), one reciprocal (v_rcp_f64 + two Newton steps: the
        // IEEE division sequence is twice as long and sits on the critical path of every iteration)
        double izc = __builtin_amdgcn_rcp(zc);
        izc = fma(fma(-zc, izc, 1.0), izc, izc);
        izc = fma(fma(-zc, izc, 1.0), izc, izc);
        double pu, pv;
        if (DIST) {
          Cam cm;
          cm.fx = P.fx; cm.fy = P.fy; cm.cx = P.cx; cm.cy = P.cy;
          cm.width = 0; cm.height = 0;
          cm.model = P.cam_model;
#pragma unroll
          for (int k = 0; k < 5; ++k) cm.d[k] = P.d[k];
          const double uvn[2] = {xc * izc, yc * izc};
          double pxd[2];
          world2cam_uv(cm, uvn, pxd);
          pu = pxd[0];
          pv = pxd[1];
        } else {
          pu = P.fx * (xc * izc) + P.cx;
          pv = P.fy * (yc * izc) + P.cy;
        }
        const float u_cur = (float)pu * scale;
        const float v_cur = (float)pv * scale;
        const float fu = floorf(u_cur), fv = floorf(v_cur);
        // NaN / huge coordinates fail the comparisons below like the int tests do
        if (fu - 3.f >= 0.f && fv - 3.f >= 0.f && fu + 3.f < (float)cols && fv + 3.f < (float)rows) {
          m = true;
          const int u_i = (int)fu, v_i = (int)fv;
          const float su = u_cur - fu, sv = v_cur - fv;
          const float wtl = (1.f - su) * (1.f - sv);  // == the reference's rounded double products (:200-203), see above
          const float wtr = su * (1.f - sv);
          const float wbl = (1.f - su) * sv;
          const float wbr = su * sv;
          {
#if SIA_PACKED
          // Two pixels per instruction (v_pk_mul/fma/add_f32): columns x = 0, 2 in one register pair, x = 1, 3 in
          // another.  Every pixel sees the same operations in the same order as in the scalar loop (its intensity and
          // residual are the same bits); the sums over the patch are formed as two interleaved partial sums.
          // Rows slide: window row y+2 is cut and tile row y+3 is fetched from LDS while rows y, y+1 are used, so that a
          // third of the 5x5 window and half of the 6x6 tile are live at a time (all of both at once does not fit 128
          // VGPRs: 13 spills).  The empty asm statements keep the compiler from gathering all rows up front again.
          typedef float f2 __attribute__((ext_vector_type(2)));
          uint32_t cw[5][3];
          uint32_t cbo = 0;
          uint64_t k0 = 0, k1 = 0, kup = 0;
          int bo = 0;
          if (WC) {
            int r0 = (v_i - 2) - wc_v0;  // first cached row needed
            bo = (u_i - 2) - wc_u0;      // first cached byte needed
            if (!(r0 >= 0 && r0 <= 2 && bo >= 0 && bo <= 7)) {
              wc_v0 = v_i - 3;
              wc_u0 = run_start(u_i - 3, 7);
              load_window12<(WC ? 7 : 1)>(cur_img, pitch, wc_u0, wc_v0, wc);
              r0 = 1;
              bo = (u_i - 2) - wc_u0;
            }
            k0 = SVO_BALLOT_ACTIVE(r0 == 0);
            k1 = SVO_BALLOT_ACTIVE(r0 == 1);
            kup = SVO_BALLOT_ACTIVE(bo >= 4);
          } else {
            const int cxa = run_start(u_i - 2, 5);
            cbo = (uint32_t)(u_i - 2 - cxa);  // 0..7
            load_window12<5>(cur_img, pitch, cxa, v_i - 2, cw);
          }
          // window row r as the three column pairs (0,2) (1,3) (2,4)
#define SIA_WIN_ROW(r, p02, p13, p24)                                                                                   \
  do {                                                                                                                  \
    float o_[5];                                                                                                        \
    if (WC) {                                                                                                           \
      constexpr int S_ = WC ? 1 : 0;                                                                                    \
      const uint32_t d0_ = sel_e64(k0, wc[(r) * S_][0], sel_e64(k1, wc[((r) + 1) * S_][0], wc[((r) + 2) * S_][0]));     \
      const uint32_t d1_ = sel_e64(k0, wc[(r) * S_][1], sel_e64(k1, wc[((r) + 1) * S_][1], wc[((r) + 2) * S_][1]));     \
      const uint32_t d2_ = sel_e64(k0, wc[(r) * S_][2], sel_e64(k1, wc[((r) + 1) * S_][2], wc[((r) + 2) * S_][2]));     \
      cut_row5(d0_, d1_, d2_, bo, kup, o_);                                                                             \
    } else {                                                                                                            \
      cut_row5_plain(cw[WC ? 0 : (r)], cbo, o_);                                                                        \
    }                                                                                                                   \
    p02 = f2{o_[0], o_[2]};                                                                                             \
    p13 = f2{o_[1], o_[3]};                                                                                             \
    p24 = f2{o_[2], o_[4]};                                                                                             \
  } while (0)
          // tile rows from LDS as float2 (layout written by the precompute block): row 0 / 5: (1,3) (2,4);
          // rows 1-4: (0,2) (1,3) (4,5), from which (2,4) and (3,5) are put together
          const float2* const bt2 = reinterpret_cast<const float2*>(&s_bt[0][0]);
          // float2 index of (quad q, half h) of this lane
#define SIA_BT2(q, h) bt2[((q) * BLOCK + tid) * 2 + (h)]
          const f2 vtl = f2{wtl, wtl}, vtr = f2{wtr, wtr}, vbl = f2{wbl, wbl}, vbr = f2{wbr, wbr};
          f2 c2v = f2{0.f, 0.f}, gxv = f2{0.f, 0.f}, gyv = f2{0.f, 0.f};
          f2 t02, t13, t24, b02, b13, b24;  // window rows y and y+1
          SIA_WIN_ROW(0, t02, t13, t24);
          SIA_WIN_ROW(1, b02, b13, b24);
          // tile rows y (a), y+1 (b), y+2 (c)
          f2 a13, a24, b02t, b13t, b45t, c02, c13, c45;
          { const float2 v0 = SIA_BT2(0, 0), v1 = SIA_BT2(0, 1); a13 = f2{v0.x, v0.y}; a24 = f2{v1.x, v1.y}; }
          { const float2 v0 = SIA_BT2(1, 0), v1 = SIA_BT2(1, 1), v2 = SIA_BT2(2, 0);
            b02t = f2{v0.x, v0.y}; b13t = f2{v1.x, v1.y}; b45t = f2{v2.x, v2.y}; }
          { const float2 v0 = SIA_BT2(2, 1), v1 = SIA_BT2(3, 0), v2 = SIA_BT2(3, 1);
            c02 = f2{v0.x, v0.y}; c13 = f2{v1.x, v1.y}; c45 = f2{v2.x, v2.y}; }
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            // next rows on their way: tile row y+3, window row y+2
            f2 n02 = f2{0.f, 0.f}, n13 = f2{0.f, 0.f}, n45 = f2{0.f, 0.f};
            if (y == 0) { const float2 v0 = SIA_BT2(4, 0), v1 = SIA_BT2(4, 1), v2 = SIA_BT2(5, 0);
                          n02 = f2{v0.x, v0.y}; n13 = f2{v1.x, v1.y}; n45 = f2{v2.x, v2.y}; }
            if (y == 1) { const float2 v0 = SIA_BT2(5, 1), v1 = SIA_BT2(6, 0), v2 = SIA_BT2(6, 1);
                          n02 = f2{v0.x, v0.y}; n13 = f2{v1.x, v1.y}; n45 = f2{v2.x, v2.y}; }
            if (y == 2) { const float2 v0 = SIA_BT2(7, 0), v1 = SIA_BT2(7, 1);
                          n13 = f2{v0.x, v0.y}; n45 = f2{v1.x, v1.y}; }  // row 5: (1,3) and (2,4)
            const f2 b24t = f2{b02t.y, b45t.x}, b35t = f2{b13t.y, b45t.y};
            // row y+2's (2,4): put together for rows 2-4, stored as such for row 5
            const f2 c24 = (y < 3) ? f2{c02.y, c45.x} : c45;
            const f2 Ia = sia_bilerp2(vtl, vtr, vbl, vbr, t02, t13, b02, b13);  // pixels x = 0, 2
            const f2 Ib = sia_bilerp2(vtl, vtr, vbl, vbr, t13, t24, b13, b24);  // pixels x = 1, 3
            const f2 ra = Ia - b13t, rb = Ib - b24t;
            c2v += ra * ra;
            c2v += rb * rb;
            gxv += ra * (b24t - b02t);
            gxv += rb * (b35t - b13t);
            gyv += ra * (c13 - a13);
            gyv += rb * (c24 - a24);
            if (y < 3) {
              f2 w02, w13, w24;
              SIA_WIN_ROW(y + 2, w02, w13, w24);
              t02 = b02; t13 = b13; t24 = b24;
              b02 = w02; b13 = w13; b24 = w24;
              a13 = b13t; a24 = b24t;
              b02t = c02; b13t = c13; b45t = c45;
              c02 = n02; c13 = n13; c45 = n45;
              asm volatile("" ::: "memory");
            }
          }
#undef SIA_WIN_ROW
#undef SIA_BT2
          c2 += c2v.x + c2v.y;
          gx += gxv.x + gxv.y;
          gy += gyv.x + gyv.y;
#else
          float W[5][5];
          if (WC) {
            int r0 = (v_i - 2) - wc_v0;  // first cached row needed
            int bo = (u_i - 2) - wc_u0;  // first cached byte needed
            if (!(r0 >= 0 && r0 <= 2 && bo >= 0 && bo <= 7)) {
              wc_v0 = v_i - 3;
              wc_u0 = run_start(u_i - 3, 7);
              load_window12<(WC ? 7 : 1)>(cur_img, pitch, wc_u0, wc_v0, wc);
              r0 = 1;
              bo = (u_i - 2) - wc_u0;
            }
            const uint64_t k0 = SVO_BALLOT_ACTIVE(r0 == 0), k1 = SVO_BALLOT_ACTIVE(r0 == 1);
            const uint64_t kup = SVO_BALLOT_ACTIVE(bo >= 4);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              constexpr int S = WC ? 1 : 0;  // keeps the indices in range when the cache is compiled out
              const uint32_t d0 = sel_e64(k0, wc[r * S][0], sel_e64(k1, wc[(r + 1) * S][0], wc[(r + 2) * S][0]));
              const uint32_t d1 = sel_e64(k0, wc[r * S][1], sel_e64(k1, wc[(r + 1) * S][1], wc[(r + 2) * S][1]));
              const uint32_t d2 = sel_e64(k0, wc[r * S][2], sel_e64(k1, wc[(r + 1) * S][2], wc[(r + 2) * S][2]));
              cut_row5(d0, d1, d2, bo, kup, W[r]);
            }
          } else {
            // 5 rows x 5 bytes [u_i-2, u_i+2] as 12-byte runs (one three-way branch on the tile position for all rows)
            uint32_t cw[5][3];
            const int cxa = run_start(u_i - 2, 5);
            const uint32_t cbo = (uint32_t)(u_i - 2 - cxa);  // 0..7
            load_window12<5>(cur_img, pitch, cxa, v_i - 2, cw);
#pragma unroll
            for (int r = 0; r < 5; ++r) cut_row5_plain(cw[r], cbo, W[r]);
          }
          float Bt[6][6];
          {
            const float4 q0 = s_bt[0][tid], q1 = s_bt[1][tid], q2 = s_bt[2][tid], q3 = s_bt[3][tid];
            const float4 q4 = s_bt[4][tid], q5 = s_bt[5][tid], q6 = s_bt[6][tid], q7 = s_bt[7][tid];
            Bt[0][0] = Bt[0][5] = Bt[5][0] = Bt[5][5] = 0.f;
            Bt[0][1] = q0.x; Bt[0][2] = q0.y; Bt[0][3] = q0.z; Bt[0][4] = q0.w;
            Bt[1][0] = q1.x; Bt[1][1] = q1.y; Bt[1][2] = q1.z; Bt[1][3] = q1.w;
            Bt[1][4] = q2.x; Bt[1][5] = q2.y; Bt[2][0] = q2.z; Bt[2][1] = q2.w;
            Bt[2][2] = q3.x; Bt[2][3] = q3.y; Bt[2][4] = q3.z; Bt[2][5] = q3.w;
            Bt[3][0] = q4.x; Bt[3][1] = q4.y; Bt[3][2] = q4.z; Bt[3][3] = q4.w;
            Bt[3][4] = q5.x; Bt[3][5] = q5.y; Bt[4][0] = q5.z; Bt[4][1] = q5.w;
            Bt[4][2] = q6.x; Bt[4][3] = q6.y; Bt[4][4] = q6.z; Bt[4][5] = q6.w;
            Bt[5][1] = q7.x; Bt[5][2] = q7.y; Bt[5][3] = q7.z; Bt[5][4] = q7.w;
          }
#pragma unroll
          for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const float I = sia_bilerp(wtl, wtr, wbl, wbr, W[y][x], W[y][x + 1], W[y + 1][x], W[y + 1][x + 1]);
              const float res = I - Bt[y + 1][x + 1];
              c2 += res * res;
              gx += (sia_pix)res * (sia_pix)(Bt[y + 1][x + 2] - Bt[y + 1][x]);
              gy += (sia_pix)res * (sia_pix)(Bt[y + 2][x + 1] - Bt[y][x + 1]);
            }
#endif
          }
        }
      }
      // -- workgroup reduction of Jres, chi2, n_meas ------------------------
      // Jres -= J res with J = dx*a + dy*b, a = fl*jac.row(0), b = fl*jac.row(1)
      {
        // dx, dy carry a factor 0.5 (central difference)
        const sia_acc gsc = (sia_acc)0.5 * fl * gmask;
        const sia_acc gxf = m ? (sia_acc)gx * gsc : (sia_acc)0, gyf = m ? (sia_acc)gy * gsc : (sia_acc)0;
        sia_acc part[8];
        sia_acc zi_ = zi, xn_ = xn, yn_ = yn;  // opaque, as in the H rebuild below: keeps xn*yn, 1+xn^2, ... inside the loop
        asm volatile("" : "+v"(zi_), "+v"(xn_), "+v"(yn_));
        part[0] = zi_ * gxf;
        part[1] = zi_ * gyf;
        part[2] = -zi_ * (xn_ * gxf + yn_ * gyf);
        part[3] = -(xn_ * yn_ * gxf + ((sia_acc)1 + yn_ * yn_) * gyf);
        part[4] = ((sia_acc)1 + xn_ * xn_) * gxf + xn_ * yn_ * gyf;
        part[5] = xn_ * gyf - yn_ * gxf;
        part[6] = (sia_acc)c2;
        part[7] = m ? (sia_acc)16 : (sia_acc)0;
        [[maybe_unused]] const long long tp1 = SIA_T();
        SIA_ACC(0, tp0, tp1);
        const sia_acc tot = wave_reduce8(part, lane);
        if ((lane & 7) == 0) g_s.part[buf][wave][lane >> 3] = tot;
        // "did the set of patches inside the current image change" travels with the partials: one barrier,
        // where __syncthreads_or costs three and an LDS atomic
        const int mine = __builtin_amdgcn_ballot_w64((int)m != inH) != 0ull;
        if (lane == 0) g_s.chg[buf][wave] = mine;
        SIA_ACC(1, tp1, SIA_T());
      }
      // the one workgroup barrier of an iteration
      [[maybe_unused]] const long long tb0 = SIA_T();
      __syncthreads();
      int changed = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) changed |= g_s.chg[buf][w];
      [[maybe_unused]] double xtot = 0.0;  // PARTS > 1: lane k < 9 of every wave holds the FRAME's sum k (k = 8: "changed" anywhere)
#ifndef SVO_HOST_MATH_TEST
      if constexpr (PARTS > 1) {
        // every wave takes part (no second barrier): lane = (part xp, sum xk); the workgroup's own sums come from LDS, wave
        // `sw` publishes them, the other parts' are polled.  (One polling wave per workgroup that hands the sums to the others
        // through LDS behind a second barrier measured the same: 0.0938 against 0.0949 ms, profiles/r06j_*.)
        ++xseq;
        const int xp = lane >> 4, xk = lane & 15;
        double own = 0.0;
        if (xk < 8) {
#pragma unroll
          for (int w = 0; w < NW; ++w) own += (double)g_s.part[buf][w][xk];
        } else if (xk == 8) {
          own = (double)changed;
        }
        XChunk* const slot = xbase + (size_t)(xseq & 1) * (SIA_X_MAX_PARTS * SIA_X_SLOTS);
        if (wave == sw && xp == part && xk < 9) sia_xstore(slot + part * SIA_X_SLOTS + xk, own, xseq);
        double val = own;
        if (xp != part && xp < PARTS && xk < 9) val = sia_xpoll(slot + xp * SIA_X_SLOTS + xk, xseq, xfail);
        if (xfail) g_s.xfail = 1;
        // the parts in a fixed order: every part forms the same sums
        xtot = __shfl(val, xk, 64);
#pragma unroll
        for (int p = 1; p < PARTS; ++p) xtot += __shfl(val, 16 * p + xk, 64);
        changed = readlane_f64<8>(xtot) != 0.0;
      }
#endif
      [[maybe_unused]] const long long tb1 = SIA_T();
      SIA_ACC(2, tb0, tb1);
      if (changed) {
        // the set of patches inside the current image changed: rebuild H.
        // H += J J' summed over the patch = Sxx aa' + Sxy (ab'+ba') + Syy bb'
        // (opaque copies: otherwise the ~60 products below, all invariant in the iteration loop, are hoisted
        // out of it and stay live across it -- 35 VGPRs, the difference between 3 and 4 waves per SIMD)
        sia_acc zi_ = zi, xn_ = xn, yn_ = yn;
        asm volatile("" : "+v"(zi_), "+v"(xn_), "+v"(yn_));
        const sia_acc one = 1, zero = 0;
        const sia_acc ja[6] = {-zi_ * fl, zero, xn_ * zi_ * fl, xn_ * yn_ * fl, -(one + xn_ * xn_) * fl, yn_ * fl};
        const sia_acc jb[6] = {zero, -zi_ * fl, yn_ * zi_ * fl, (one + yn_ * yn_) * fl, -xn_ * yn_ * fl, -xn_ * fl};
        sia_acc hp[24];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) {
            const sia_acc v = (sia_acc)Sxx * (ja[i] * ja[j]) + (sia_acc)Sxy * (ja[i] * jb[j] + jb[i] * ja[j]) + (sia_acc)Syy * (jb[i] * jb[j]);
            hp[sym6(i, j)] = m ? v : zero;
          }
        hp[21] = hp[22] = hp[23] = zero;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const sia_acc tot = wave_reduce8(hp + 8 * g, lane);
          if ((lane & 7) == 0) g_s.Hpart[wave][8 * g + (lane >> 3)] = tot;
        }
        inH = (int)m;
        __syncthreads();
        if constexpr (PARTS > 1) {
          // (its own generation: the H exchange of this iteration follows the iteration's sum exchange)
          ++xseq;
          XChunk* const xh = xbase + 2 * SIA_X_MAX_PARTS * SIA_X_SLOTS + (size_t)(xseq & 1) * (SIA_X_MAX_PARTS * SIA_XH_SLOTS);
          if (wave == 0) sia_rebuild_hinv<PARTS>(lane, NW, xh, part, xseq, xfail);
        } else {
          if (wave == 0) sia_rebuild_hinv<1>(lane, NW, nullptr, 0, 0u, xfail);
        }
        __syncthreads();
      }
      ++evals;
      [[maybe_unused]] const long long ts0 = SIA_T();
      SIA_ACC(3, tb1, ts0);
#ifdef SIA_PROFILE
      prof[7] += 1;
#endif

      // -- solve() / update() and the stop / rollback rules of
      //    vk::NLLSSolver::optimizeGaussNewton (:245-258) --
      int done = 0;
      // The serial step is ~300 VALU instructions against ~250 for a wave's pixel work: running it in all waves redundantly
      // spent more than half of the issue slots on it.  Which waves run it rotates with the workgroup id so that the solver
      // waves of the workgroups sharing a CU do not pile up on one SIMD; the others take the second barrier and pick the
      // new pose up from LDS.
      // TWO waves share the serial step: wave `sw` composes the rotation (quaternion part of SE3::exp, quaternion product,
      // normalisation, q -> R), wave `sw + 1` the translation (V(omega) upsilon, rotated by the OLD quaternion, added to
      // the old translation).  Each forms the totals, x = H^-1 Jres and the stop / rollback decision for itself (the
      // same ~40 instructions on the same LDS words, hence the same decision) and then runs its half: the dependent
      // chain between the two barriers is ~130 instructions instead of ~310, on two SIMDs.
      const int swb = (NW > 1) ? (sw + 1) % NW : sw;
      const bool rot_wave = wave == sw, trans_wave = wave == swb;
      int m_next = m_cur;  // (both solver waves compute it; the others read it from LDS behind the barrier)
      if (rot_wave || trans_wave) {
        __builtin_amdgcn_s_setprio(3);  // the other waves of the workgroup wait for these two: let them win the issue arbitration
        double colsum = 0.0;
        if constexpr (PARTS > 1) {
          colsum = xtot;  // (lanes 0..7: the frame's sums)
        } else {
          const int k = lane & 7;
#pragma unroll
          for (int w = 0; w < NW; ++w) colsum += (double)g_s.part[buf][w][k];
        }
        const double b0 = readlane_f64<0>(colsum), b1 = readlane_f64<1>(colsum), b2 = readlane_f64<2>(colsum);
        const double b3 = readlane_f64<3>(colsum), b4 = readlane_f64<4>(colsum), b5 = readlane_f64<5>(colsum);
        const float chi2_sum = (float)readlane_f64<6>(colsum);
        const int n_meas = (int)readlane_f64<7>(colsum);
        double xi;
        {
          const double* row = &g_s.Hinv[6 * (lane < 6 ? lane : 0)];
          xi = row[0] * b0 + row[1] * b1 + row[2] * b2 + row[3] * b3 + row[4] * b4 + row[5] * b5;
        }
        const double x0 = readlane_f64<0>(xi), x1 = readlane_f64<1>(xi), x2 = readlane_f64<2>(xi);
        const double x3 = readlane_f64<3>(xi), x4 = readlane_f64<4>(xi), x5 = readlane_f64<5>(xi);
        n_meas_last = n_meas;
        const double new_chi2 = (double)(chi2_sum / (float)n_meas);
        if (isnan(x0)) stop = 1;  // solve(), :248-249
        const SplitModel& sm = g_s.sm;
        const bool rollback = (iter > 0 && new_chi2 > chi2_prev) || stop;
        if (rollback) {
          m_next = m_old;  // model = old_model: the buffer the last update came from
          done = 1;
        } else {
          m_next = m_cur ^ 1;
          m_old = m_cur;
          chi2_prev = new_chi2;
          const double nm = fmax(fmax(fmax(fabs(x0), fabs(x1)), fmax(fabs(x2), fabs(x3))), fmax(fabs(x4), fabs(x5)));
          if (nm <= P.eps) done = 1;
        }
        // update(): T_new = T_old * SE3::exp(-x_)  (:253-258), each wave its half
        const float mx[6] = {-(float)x0, -(float)x1, -(float)x2, -(float)x3, -(float)x4, -(float)x5};
        if (rot_wave) {
          double q[4];
          if (rollback) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = sm.q[m_next][k];
          } else {
#ifdef SIA_F64_PARTIALS
            const double mxd[6] = {-x0, -x1, -x2, -x3, -x4, -x5};
            double eq[4], et_unused[3];
            se3_exp(mxd, eq, et_unused);  // Sophus SE3::exp in f64, as the reference evaluates it
#else
            float eqf[4];
            se3_exp_rot_f32(mx, eqf);
            const double eq[4] = {eqf[0], eqf[1], eqf[2], eqf[3]};
#endif
            double oq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) oq[k] = sm.q[m_cur][k];
            quat_mul(oq, eq, q);
            quat_normalize_fast(q);
            if (lane == 0) {
#pragma unroll
              for (int k = 0; k < 4; ++k) g_s.sm.q[m_next][k] = q[k];
            }
          }
          quat_to_R(q, R);
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) g_s.Rt[k] = R[k];
            g_s.sw_done = done;
            g_s.sw_stop = stop;
            g_s.sw_nmeas = n_meas_last;
            g_s.sw_chi2 = chi2_prev;
            g_s.sw_cur = m_next;
            g_s.sw_old = m_old;
          }
        }
        if (trans_wave) {
          if (rollback) {
#pragma unroll
            for (int k = 0; k < 3; ++k) tr[k] = sm.t[m_next][k];
          } else {
#ifdef SIA_F64_PARTIALS
            const double mxd[6] = {-x0, -x1, -x2, -x3, -x4, -x5};
            double eq_unused[4], et[3];
            se3_exp(mxd, eq_unused, et);
#else
            float etf[3];
            se3_exp_trans_f32(mx, etf);
            const double et[3] = {etf[0], etf[1], etf[2]};
#endif
            double oq[4], rt[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) oq[k] = sm.q[m_cur][k];
            quat_rot(oq, et, rt);
#pragma unroll
            for (int k = 0; k < 3; ++k) tr[k] = sm.t[m_cur][k] + rt[k];
            if (lane == 0) {
#pragma unroll
              for (int k = 0; k < 3; ++k) g_s.sm.t[m_next][k] = tr[k];
            }
          }
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) g_s.Rt[9 + k] = tr[k];
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      m_cur = m_next;
      if (NW > 1) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = g_s.Rt[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tr[k] = g_s.Rt[9 + k];
        done = g_s.sw_done;
        stop = g_s.sw_stop;
        n_meas_last = g_s.sw_nmeas;
        chi2_prev = g_s.sw_chi2;
        m_cur = g_s.sw_cur;
        m_old = g_s.sw_old;
      }
      buf ^= 1;
      SIA_ACC(4, ts0, SIA_T());
      if (done) break;
    }
    if (tid == 0 && part == 0 && a.iters) a.iters[SVO_HIP_MAX_LEVELS * b + level] = evals;
  }

  if (tid == 0 && part == 0) {
    for (int k = 0; k < 9; ++k) a.T_out[12 * b + k] = R[k];
    for (int k = 0; k < 3; ++k) a.T_out[12 * b + 9 + k] = tr[k];
    if (a.H_out)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) a.H_out[36 * b + i * 6 + j] = g_s.H[sym6_rt(i, j)];
#ifdef SIA_PROFILE
    prof[6] = SIA_T() - t_begin;
    if (a.H_out)
      for (int k = 0; k < 8; ++k) a.H_out[36 * b + k] = (double)prof[k];
#endif
    a.n_tracked[b] = n_meas_last / 16;
    if (a.chi2) a.chi2[b] = chi2_prev;
    if constexpr (PARTS > 1) {
      // the block's next user starts behind this frame's last epoch (far behind it after a time-out, whose parts may not
      // have counted alike)
#ifndef SVO_HOST_MATH_TEST
      sia_xstore_epoch(xbase + SIA_X_CHUNKS, xseq + (g_s.xfail ? (1u << 16) : 0u));
#endif
    }
    // (g_s.xfail: written by whichever lane gave up, ahead of the barrier that ends its iteration)
    if (a.status) a.status[b] = (stop ? SVO_HIP_SIA_STOP : 0) | ((PARTS > 1 && g_s.xfail) ? SVO_HIP_SIA_EXCHANGE_TIMEOUT : 0);
  }
}

// the workgroup size that carries the window cache (256 lanes: +25 % at 4 levels x 3.4 iterations; 64/128 lanes run
// few iterations per level at the default 4->2 schedule; 512/1024 lanes measured the same with and without it
// and spill less without)

template <int BLOCK>
int launch(const SiaArgs& args, int B, hipStream_t s) {
  // window cache where the workgroup is large enough for the extra registers to pay (see MINW)
  constexpr bool WC = BLOCK == 256;
  if (args.P.cam_model == SVO_HIP_CAM_PINHOLE)
    hipLaunchKernelGGL((sia_kernel<BLOCK, WC, false>), dim3(B), dim3(BLOCK), 0, s, args);
  else
    hipLaunchKernelGGL((sia_kernel<BLOCK, false, true>), dim3(B), dim3(BLOCK), 0, s, args);  // atan / radtan code: no room for the window cache
  return check_launch();
}

}  // namespace

namespace {

// validates the arguments of the two entry points and fills the kernels' argument block
int sia_prepare(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B, const int32_t* d_ref_slot,
                const int32_t* d_cur_slot, const int32_t* d_n, int n_stride, const double* d_px, const double* d_xyz_ref,
                const uint8_t* d_valid, const svo_hip_sia_params* params, const double* d_T_in, double* d_T_out,
                double* d_H_out, int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2, int32_t* d_status, SiaArgs* out) {
  if (!layout_ok(layout) || !d_store || !params || B < 0) return SVO_HIP_EINVAL;
  if (B == 0) return SVO_HIP_OK;
  if (!d_ref_slot || !d_cur_slot || !d_n || !d_px || !d_xyz_ref || !d_T_in || !d_T_out || !d_n_tracked)
    return SVO_HIP_EINVAL;
  if (n_stride < 1) return SVO_HIP_EINVAL;
  if (n_stride > SVO_HIP_MAX_PATCHES) return SVO_HIP_ERANGE;
  if (params->min_level < 0 || params->max_level < params->min_level || params->max_level >= layout->n_levels ||
      params->n_iter < 0)
    return SVO_HIP_EINVAL;
  if (params->cam_model != SVO_HIP_CAM_PINHOLE && params->cam_model != SVO_HIP_CAM_PINHOLE_RADTAN &&
      params->cam_model != SVO_HIP_CAM_ATAN)
    return SVO_HIP_EINVAL;
  SiaArgs& args = *out;
  args.L = *layout;
  args.store = d_store;
  args.ref_slot = d_ref_slot;
  args.cur_slot = d_cur_slot;
  args.n = d_n;
  args.n_stride = n_stride;
  args.px = d_px;
  args.xyz = d_xyz_ref;
  args.valid = d_valid;
  args.P = *params;
  args.T_in = d_T_in;
  args.T_out = d_T_out;
  args.H_out = d_H_out;
  args.n_tracked = d_n_tracked;
  args.iters = d_iters;
  args.chi2 = d_chi2;
  args.status = d_status;
  return 1;  // launch
}

// ---- a frame split over four workgroups (PARTS = 4) -------------------------------------------------------------------
// For frames of more than 512 patches in batches that leave most of the GPU idle (4 B workgroups of 256 lanes must be
// RESIDENT AT ONCE -- the parts spin on each other --: B <= SIA_SPLIT_MAX_B).  The exchange blocks belong to the (device,
// stream) pair: allocated and zeroed once, on the pair's first split launch, and never cleared again -- a block remembers
// the last epoch used in it (sia_common.h), launches on one stream follow each other, launches on different streams use
// different blocks.  No per-launch allocation, memset or salt argument: a captured launch replays correctly.  A stream
// that is being captured when its blocks would have to be created keeps the frame on one workgroup (an allocation is not
// a capturable operation).  SVO_HIP_K1_SPLIT=0 keeps every frame on one workgroup.
constexpr int SIA_SPLIT_PARTS = 4;
constexpr int SIA_SPLIT_MAX_B = 128;
#ifndef SVO_HOST_MATH_TEST
bool sia_split_applies(const SiaArgs& args, int B) {
  static const bool on = [] { const char* v = std::getenv("SVO_HIP_K1_SPLIT"); return !(v && v[0] == '0'); }();
  return on && args.n_stride > 512 && B <= SIA_SPLIT_MAX_B;
}
// the exchange blocks of (current device, stream); NULL: none can be had right now
void* sia_split_blocks(hipStream_t s) {
  static std::mutex mut;
  static std::map<std::pair<int, hipStream_t>, void*> blocks;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> g(mut);
  void*& p = blocks[std::make_pair(dev, s)];
  if (p) return p;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  const size_t bytes = (size_t)SIA_SPLIT_MAX_B * SIA_X_BLOCK * sizeof(XChunk);
  void* q = nullptr;
  if (hipMalloc(&q, bytes) != hipSuccess) return nullptr;
  if (hipMemset(q, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(q);
    return nullptr;
  }
  p = q;
  return p;
}
int launch_split(SiaArgs args, int B, void* xw, hipStream_t s) {
  args.B = B;
  args.xw = xw;
  const dim3 grid((unsigned)((B + 7) / 8) * 8 * SIA_SPLIT_PARTS), blk(256);
  if (args.P.cam_model == SVO_HIP_CAM_PINHOLE)
    hipLaunchKernelGGL((sia_kernel<256, true, false, SIA_SPLIT_PARTS>), grid, blk, 0, s, args);
  else
    hipLaunchKernelGGL((sia_kernel<256, false, true, SIA_SPLIT_PARTS>), grid, blk, 0, s, args);
  return check_launch();
}
#endif

int launch_workgroup(const SiaArgs& args, int B, hipStream_t s) {
  if (args.n_stride <= 64) return launch<64>(args, B, s);
  if (args.n_stride <= 128) return launch<128>(args, B, s);
  if (args.n_stride <= 256) return launch<256>(args, B, s);
  if (args.n_stride <= 512) return launch<512>(args, B, s);
  return launch<1024>(args, B, s);
}

}  // namespace

extern "C" int svo_hip_sparse_align(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B,
                                    const int32_t* d_ref_slot, const int32_t* d_cur_slot, const int32_t* d_n,
                                    int n_stride, const double* d_px, const double* d_xyz_ref,
                                    const uint8_t* d_valid, const svo_hip_sia_params* params,
                                    const double* d_T_in, double* d_T_out, double* d_H_out,
                                    int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2, int32_t* d_status,
                                    void* stream) {
  SiaArgs args;
  const int rc = sia_prepare(layout, d_store, B, d_ref_slot, d_cur_slot, d_n, n_stride, d_px, d_xyz_ref, d_valid, params,
                             d_T_in, d_T_out, d_H_out, d_n_tracked, d_iters, d_chi2, d_status, &args);
  if (rc <= 0) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // large batches of frames with up to 192 patches: one wave per frame (sparse_align_wave.hip); else one workgroup
  if (sia_wave_applies(args, B)) return launch_sia_wave(args, B, s);
#ifndef SVO_HOST_MATH_TEST
  if (sia_split_applies(args, B))
    if (void* xw = sia_split_blocks(s)) return launch_split(args, B, xw, s);
#endif
  return launch_workgroup(args, B, s);
}

// The workgroup-per-frame kernel for any patch count: the path svo_hip_sparse_align takes above 256 patches
// per frame, exposed so that tests and the bench can run both kernels on the same problems.
extern "C" int svo_hip_sparse_align_workgroup(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B,
                                              const int32_t* d_ref_slot, const int32_t* d_cur_slot, const int32_t* d_n,
                                              int n_stride, const double* d_px, const double* d_xyz_ref,
                                              const uint8_t* d_valid, const svo_hip_sia_params* params,
                                              const double* d_T_in, double* d_T_out, double* d_H_out,
                                              int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2,
                                              int32_t* d_status, void* stream) {
  SiaArgs args;
  const int rc = sia_prepare(layout, d_store, B, d_ref_slot, d_cur_slot, d_n, n_stride, d_px, d_xyz_ref, d_valid, params,
                             d_T_in, d_T_out, d_H_out, d_n_tracked, d_iters, d_chi2, d_status, &args);
  if (rc <= 0) return rc;
  return launch_workgroup(args, B, static_cast<hipStream_t>(stream));
}
