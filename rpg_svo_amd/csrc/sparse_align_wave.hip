// sparse_align_wave.hip -- K1, one WAVE per problem: taken by svo_hip_sparse_align for batches of >= 1024 problems
// with at most 192 patches (64 under a distorted camera model); everything else -- single frames, the 200-patch
// headline workload -- runs the workgroup-per-problem kernel of sparse_align.hip (sia_wave_applies() below).
//
// Same algorithm and numerics as sparse_align.hip (svo::SparseImgAlign::run + the Gauss-Newton loop of
// vk::NLLSSolver, svo/src/sparse_img_align.cpp:43-258); what changes is who does the work.  There a
// workgroup of four waves owns a frame, one lane per patch, and every iteration pays two workgroup
// barriers and a serial solve that three of the four waves sit out.  Here a frame is ONE wave:
//
//   * a lane carries PPL = ceil(n/64) <= 3 patches, each with its own window cache, and runs them one after
//     the other; the 8 sums of an iteration are accumulated over the lane's patches before ONE transposing
//     wave reduction (wave_reduce.h);
//   * the solve runs in the same wave straight after the reduction: no barrier, no LDS exchange of
//     partials, no idle waves; the wave totals are read out of the lanes with v_readlane (SGPRs);
//   * waves of different frames share nothing, so the two waves a SIMD holds (<= 256 VGPRs each) always
//     have something to issue: one frame's serial solve overlaps the other's pixel work.
//   A fourth patch per lane does not fit 256 VGPRs with its window cache (PPL = 4 spills in the loop): at 200
//   patches the workgroup kernel is the faster one (1.39 against 1.63 ms, round 2), at 192 this one (1.26 / 1.36).
//
// LDS holds only the interpolated reference patches (128 bytes per patch: 24.6 KB for 192 patches, six
// frames per CU) and a few hundred bytes of per-frame scalars.
#include "sia_common.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_sia;

namespace {

// Waves per SIMD asked of the register allocator (2 -> at most 256 VGPRs per wave)
constexpr int SIAW_MINW = 2;

// One 4x4 reference patch as a lane carries it through the coarse-to-fine schedule.
struct Patch {
  double X, Y, Z;        // xyz_ref = f * depth (:107-108)
  float zi, xn, yn;      // 1/z, x/z, y/z of xyz_ref: all Frame::jacobian_xyz2uv needs (frame.h:116-138)
  float Sxx, Sxy, Syy;   // sums of dx*dx, dx*dy, dy*dy over the patch at this level
  float gmask;           // 0 while the Jacobian columns of this patch are zero at this level (:64)
  bool has, vis;         // carries a point / visible_fts_ (never cleared between levels, :57)
  int inH;               // membership in the sum H currently holds (-1: none yet at this level)
  int wc_u0, wc_v0;      // window cache: columns [wc_u0, wc_u0+11], rows [wc_v0, wc_v0+6] of the current level
  uint32_t wc[7][3];
};

// Per-frame scalars (the frame's only wave reads and writes them: DS operations of one wave execute in order).
struct WaveLds {
  double q[4], t[3];     // model: T_cur_from_ref as Sophus stores it (unit quaternion + translation)
  double oq[4], ot[3];   // old_model (rollback)
  double H[21];          // H_ of the last evaluated iteration (packed upper triangle)
  float red[24];         // wave totals of the 21 H entries, handed from the lanes that hold them to lanes 0..20
  double Hinv[36];       // H^-1, row-major
  double A[36];          // Gauss-Jordan scratch
  long long lo[SVO_HIP_MAX_LEVELS];
  int lw[SVO_HIP_MAX_LEVELS], lh[SVO_HIP_MAX_LEVELS], lp[SVO_HIP_MAX_LEVELS];
};

__device__ __forceinline__ float readlane_f32(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

template <int PPL, bool DIST>
__global__ void __launch_bounds__(64, SIAW_MINW) sia_wave_kernel(const SiaArgs a, const int n_slots) {
  // XCD-aware problem order (capi_common.h): consecutive problems of a replay batch share a frame
  const int b = (int)xcd_contiguous_block();
  const int lane = threadIdx.x;

  // Interpolated reference image around every patch: the bilinear sample at window pixel (r,c) of the 6x6
  // neighbourhood, corners unused -> 32 floats = 8 float4 per patch, laid out [8][n_slots] so a wave reads
  // 1 KiB contiguous per ds_read_b128 (same packing as sparse_align.hip):
  //   q0 = r0 c1..4 | q1 = r1 c0..3 | q2 = r1 c4,5 r2 c0,1 | q3 = r2 c2..5
  //   q4 = r3 c0..3 | q5 = r3 c4,5 r4 c0,1 | q6 = r4 c2..5 | q7 = r5 c1..4
  // Patches 64k..64k+63 (the lanes' k-th patches) form one block [8][64] float4, so that every address is
  // the lane's base plus a compile-time offset; the last block is [8][last_n], last_n = n_slots - 64 (PPL-1).
  SVO_DYNAMIC_LDS(float4, s_bt);
  __shared__ WaveLds g;
  const int last_n = n_slots - 64 * (PPL - 1);
#define SIA_BT(k, q) s_bt[((k) < PPL - 1) ? ((k) * 512 + (q) * 64 + lane) : ((PPL - 1) * 512 + (q) * last_n + lane)]

  int n = a.n[b];
  n = n > a.n_stride ? a.n_stride : n;  // contract: n <= n_stride (svo_hip.h)
  const svo_hip_sia_params P = a.P;

  if (n <= 0) {  // sparse_img_align.cpp:47-51: nothing to track, pose untouched
    if (lane == 0) {
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

  Patch pt[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    Patch& p = pt[k];
    const int i = lane + 64 * k;
    const size_t fo = (size_t)b * a.n_stride + i;
    p.has = (i < n) && (a.valid ? a.valid[fo] != 0 : true);
    p.X = 0; p.Y = 0; p.Z = 1;
    if (p.has) {
      p.X = a.xyz[3 * fo];
      p.Y = a.xyz[3 * fo + 1];
      p.Z = a.xyz[3 * fo + 2];
    }
    const double rz = sia_rcp(p.Z);  // (as sparse_align.hip: the two kernels keep the same per-patch arithmetic)
    p.zi = (float)rz;
    p.xn = (float)(p.X * rz);
    p.yn = (float)(p.Y * rz);
    p.Sxx = p.Sxy = p.Syy = 0.f;
    p.gmask = 0.f;
    p.vis = false;
    p.inH = -1;
    p.wc_u0 = 0;
    p.wc_v0 = -100000;
#pragma unroll
    for (int r = 0; r < 7; ++r) p.wc[r][0] = p.wc[r][1] = p.wc[r][2] = 0u;
  }
  const uint8_t* ref_base = a.store + (int64_t)a.ref_slot[b] * a.L.slot_bytes;
  const uint8_t* cur_base = a.store + (int64_t)a.cur_slot[b] * a.L.slot_bytes;

  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < SVO_HIP_MAX_LEVELS; ++k) {
      g.lw[k] = a.L.w[k];
      g.lh[k] = a.L.h[k];
      g.lp[k] = a.L.pitch[k];
      g.lo[k] = a.L.offset[k];
    }
    if (a.iters)
      for (int k = 0; k < SVO_HIP_MAX_LEVELS; ++k) a.iters[SVO_HIP_MAX_LEVELS * b + k] = 0;
  }
  if (lane < 21) g.H[lane] = 0.0;
  // the model: every lane computes the same values, lane 0 stores them; the rotation the pixel work uses
  // lives in SGPRs (wave-uniform, v_readfirstlane)
  double R[9], tr[3];
  {
    double q[4], Rm[9];
    for (int k = 0; k < 9; ++k) Rm[k] = a.T_in[12 * b + k];
    for (int k = 0; k < 3; ++k) tr[k] = sia_uni(a.T_in[12 * b + 9 + k]);
    quat_from_R(Rm, q);
    quat_to_R(q, Rm);
    for (int k = 0; k < 9; ++k) R[k] = sia_uni(Rm[k]);
    if (lane == 0) {
      for (int k = 0; k < 4; ++k) g.q[k] = g.oq[k] = q[k];
      for (int k = 0; k < 3; ++k) g.t[k] = g.ot[k] = tr[k];
    }
  }
  // vk::NLLSSolver::reset(): wave-uniform solver state
  double chi2_prev = 1e10;
  int stop = 0;
  int n_meas_last = 0;
  __syncthreads();  // a single wave: orders the LDS writes above before the reads below

  for (int level = P.max_level; level >= P.min_level; --level) {
    const int cols = g.lw[level], rows = g.lh[level], pitch = g.lp[level];
    const uint8_t* ref_img = ref_base + g.lo[level];
    const uint8_t* cur_img = cur_base + g.lo[level];
    const float scale = pow2_inv_f32(level);  // == 1.0f / (float)(1 << level), from exponent bits (no division sequence)
    // focal_length / 2^level (:139-140), folded into the per-patch sums
    const float fl = (float)(fabs(P.fx) * pow2_inv_f64(level));  // == / 2^level, bit for bit

    // ---- precomputeReferencePatches (:84-145) --------------------------------------------------------
    // Stage A, all of the lane's patches: positions, then EVERY global load of the level in one go -- the
    // 7 x 3 dwords of the reference window and, with the pose the level starts from, the 7 x 3 dwords of the
    // current image that iteration 0 will cut its 5x5 window from (the window cache is filled here, so that
    // iteration finds it valid).  A single wave has nobody to hide a round trip behind: taken patch by patch
    // these are 2 x PPL dependent round trips per level, taken together one.  Lanes without a usable patch load
    // from a clamped position and throw the bytes away (keeps the block free of branches).
    uint32_t rw[PPL][7][3];
    int sel_ref[PPL];
    bool inb_k[PPL];
    float su_k[PPL], sv_k[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      Patch& p = pt[k];
      const int slot = lane + 64 * k;
      const size_t fo = (size_t)b * a.n_stride + slot;
      double pxx = 0, pxy = 0;
      if (p.has) {
        pxx = a.px[2 * fo];
        pxy = a.px[2 * fo + 1];
      }
      const float u_ref = (float)(pxx * (double)scale);
      const float v_ref = (float)(pxy * (double)scale);
      const int u_i = (int)floorf(u_ref);
      const int v_i = (int)floorf(v_ref);
      const bool inb = p.has && !(u_i - 3 < 0 || v_i - 3 < 0 || u_i + 3 >= cols || v_i + 3 >= rows);
      inb_k[k] = inb;
      su_k[k] = u_ref - (float)u_i;
      sv_k[k] = v_ref - (float)v_i;
      const int cu = inb ? u_i : 3, cv = inb ? v_i : 3;
      const int rxa = run_start(cu - 3, 7);
      sel_ref[k] = cu - 3 - rxa;  // 0..5
      load_window12<7>(ref_img, pitch, rxa, cv - 3, rw[k]);
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      Patch& p = pt[k];
      p.wc_v0 = -100000;  // the cache holds rows of the previous level
      const bool will_see = p.vis || inb_k[k];
      const double xc = R[0] * p.X + R[1] * p.Y + R[2] * p.Z + tr[0];
      const double yc = R[3] * p.X + R[4] * p.Y + R[5] * p.Z + tr[1];
      const double zc = R[6] * p.X + R[7] * p.Y + R[8] * p.Z + tr[2];
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
        for (int j = 0; j < 5; ++j) cm.d[j] = P.d[j];
        const double uvn[2] = {xc * izc, yc * izc};
        double pxd[2];
        world2cam_uv(cm, uvn, pxd);
        pu = pxd[0];
        pv = pxd[1];
      } else {
        pu = P.fx * (xc * izc) + P.cx;
        pv = P.fy * (yc * izc) + P.cy;
      }
      const float fu = floorf((float)pu * scale), fv = floorf((float)pv * scale);
      const bool okc = will_see && fu - 3.f >= 0.f && fv - 3.f >= 0.f && fu + 3.f < (float)cols && fv + 3.f < (float)rows;
      const int cu = okc ? (int)fu : 3, cv = okc ? (int)fv : 3;
      const int v0 = cv - 3, u0 = run_start(cu - 3, 7);
      load_window12<7>(cur_img, pitch, u0, v0, p.wc);
      p.wc_u0 = u0;
      p.wc_v0 = okc ? v0 : -100000;
    }
    // Stage B: the interpolated reference tiles
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      Patch& p = pt[k];
      p.Sxx = p.Sxy = p.Syy = 0.f;
      p.inH = -1;
      if (inb_k[k]) {
        p.vis = true;
        p.gmask = 1.f;
        const float su = su_k[k], sv = sv_k[k];
        // == the reference's rounded double products (:118-121): u >= 3, so su, sv are multiples of 2^-22,
        // 1-su and 1-sv are exact in f32 and an f32 product is the correctly rounded exact product
        const float wtl = (1.f - su) * (1.f - sv);
        const float wtr = su * (1.f - sv);
        const float wbl = (1.f - su) * sv;
        const float wbr = su * sv;
        float Bt[6][6];
        float Wp[7], Wc[7];
        cut_row7(rw[k][0], (uint32_t)sel_ref[k], Wp);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          cut_row7(rw[k][r + 1], (uint32_t)sel_ref[k], Wc);
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const bool need = ((r >= 1 && r <= 4)) || ((c >= 1 && c <= 4));
            if (need) Bt[r][c] = wtl * Wp[c] + wtr * Wp[c + 1] + wbl * Wc[c] + wbr * Wc[c + 1];
          }
#pragma unroll
          for (int c = 0; c < 7; ++c) Wp[c] = Wc[c];
        }
        float Sxx = 0.f, Sxy = 0.f, Syy = 0.f;
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float dx = 0.5f * (Bt[y + 1][x + 2] - Bt[y + 1][x]);
            const float dy = 0.5f * (Bt[y + 2][x + 1] - Bt[y][x + 1]);
            Sxx += dx * dx;
            Sxy += dx * dy;
            Syy += dy * dy;
          }
        p.Sxx = Sxx; p.Sxy = Sxy; p.Syy = Syy;
        SIA_BT(k, 0) = make_float4(Bt[0][1], Bt[0][2], Bt[0][3], Bt[0][4]);
        SIA_BT(k, 1) = make_float4(Bt[1][0], Bt[1][1], Bt[1][2], Bt[1][3]);
        SIA_BT(k, 2) = make_float4(Bt[1][4], Bt[1][5], Bt[2][0], Bt[2][1]);
        SIA_BT(k, 3) = make_float4(Bt[2][2], Bt[2][3], Bt[2][4], Bt[2][5]);
        SIA_BT(k, 4) = make_float4(Bt[3][0], Bt[3][1], Bt[3][2], Bt[3][3]);
        SIA_BT(k, 5) = make_float4(Bt[3][4], Bt[3][5], Bt[4][0], Bt[4][1]);
        SIA_BT(k, 6) = make_float4(Bt[4][2], Bt[4][3], Bt[4][4], Bt[4][5]);
        SIA_BT(k, 7) = make_float4(Bt[5][1], Bt[5][2], Bt[5][3], Bt[5][4]);
      } else {
        // jacobian_cache_.setZero() (:64): the J columns of a feature skipped here stay zero; a stale
        // ref_patch_cache_ row (if any) is kept
        p.gmask = 0.f;
      }
    }

    // ---- vk::NLLSSolver::optimizeGaussNewton ------------------------------------------------------------
    // old_model = model at the start of every optimize() call
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) g.oq[k] = g.q[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) g.ot[k] = g.t[k];
    }
    int evals = 0;
    for (int iter = 0; iter < P.n_iter; ++iter) {
      // -- computeResiduals (:147-243): this lane's patches, one after the other --------------------------
      float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      unsigned mbits = 0u;   // bit k: patch k lies inside the current image
      bool changed = false;  // ... and that differs from what H holds
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        Patch& p = pt[k];
        bool m = false;
        float gx = 0.f, gy = 0.f, c2 = 0.f;
        if (p.vis) {
          const double xc = R[0] * p.X + R[1] * p.Y + R[2] * p.Z + tr[0];
          const double yc = R[3] * p.X + R[4] * p.Y + R[5] * p.Z + tr[1];
          const double zc = R[6] * p.X + R[7] * p.Y + R[8] * p.Z + tr[2];
          // cam_->world2cam(project2d(xyz)) (:183), one reciprocal: v_rcp_f64 + two Newton steps
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
            for (int j = 0; j < 5; ++j) cm.d[j] = P.d[j];
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
            const float wtl = (1.f - su) * (1.f - sv);  // == the reference's rounded double products (:200-203)
            const float wtr = su * (1.f - sv);
            const float wbl = (1.f - su) * sv;
            const float wbr = su * sv;
            float W[5][5];
            int r0 = (v_i - 2) - p.wc_v0;  // first cached row needed
            int bo = (u_i - 2) - p.wc_u0;  // first cached byte needed
            if (!(r0 >= 0 && r0 <= 2 && bo >= 0 && bo <= 7)) {
              p.wc_v0 = v_i - 3;
              p.wc_u0 = run_start(u_i - 3, 7);
              load_window12<7>(cur_img, pitch, p.wc_u0, p.wc_v0, p.wc);
              r0 = 1;
              bo = (u_i - 2) - p.wc_u0;
            }
            const uint64_t k0 = SVO_BALLOT_ACTIVE(r0 == 0), k1 = SVO_BALLOT_ACTIVE(r0 == 1);
            const uint64_t kup = SVO_BALLOT_ACTIVE(bo >= 4);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              const uint32_t d0 = sel_e64(k0, p.wc[r][0], sel_e64(k1, p.wc[r + 1][0], p.wc[r + 2][0]));
              const uint32_t d1 = sel_e64(k0, p.wc[r][1], sel_e64(k1, p.wc[r + 1][1], p.wc[r + 2][1]));
              const uint32_t d2 = sel_e64(k0, p.wc[r][2], sel_e64(k1, p.wc[r + 1][2], p.wc[r + 2][2]));
              cut_row5(d0, d1, d2, bo, kup, W[r]);
            }
            float Bt[6][6];
            {
              const float4 q0 = SIA_BT(k, 0), q1 = SIA_BT(k, 1), q2 = SIA_BT(k, 2);
              const float4 q3 = SIA_BT(k, 3), q4 = SIA_BT(k, 4), q5 = SIA_BT(k, 5);
              const float4 q6 = SIA_BT(k, 6), q7 = SIA_BT(k, 7);
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
                const float I = wtl * W[y][x] + wtr * W[y][x + 1] + wbl * W[y + 1][x] + wbr * W[y + 1][x + 1];
                const float res = I - Bt[y + 1][x + 1];
                c2 += res * res;
                gx += res * (Bt[y + 1][x + 2] - Bt[y + 1][x]);
                gy += res * (Bt[y + 2][x + 1] - Bt[y][x + 1]);
              }
          }
        }
        // Jres -= J res with J = dx*a + dy*b, a = fl*jac.row(0), b = fl*jac.row(1); dx, dy carry a
        // factor 0.5 (central difference)
        // (the empty asm keeps the compiler from hoisting xn*yn, 1+xn^2, ... of every patch out of the
        // iteration loop: a dozen more registers live across it for a handful of multiplies)
        float zi = p.zi, xn = p.xn, yn = p.yn;
        asm volatile("" : "+v"(zi), "+v"(xn), "+v"(yn));
        const float gsc = 0.5f * fl * p.gmask;
        const float gxf = m ? gx * gsc : 0.f, gyf = m ? gy * gsc : 0.f;
        part[0] += zi * gxf;
        part[1] += zi * gyf;
        part[2] += -zi * (xn * gxf + yn * gyf);
        part[3] += -(xn * yn * gxf + (1.f + yn * yn) * gyf);
        part[4] += (1.f + xn * xn) * gxf + xn * yn * gyf;
        part[5] += xn * gyf - yn * gxf;
        part[6] += c2;
        part[7] += m ? 16.f : 0.f;
        mbits |= (m ? 1u : 0u) << k;
        changed |= ((int)m != p.inH);
      }
      // every lane of the 8-lane group j holds the wave total of part[j]
      const float tot = wave_reduce8(part, lane);

      if (__builtin_amdgcn_ballot_w64(changed) != 0ull) {
        // the set of patches inside the current image changed: rebuild H.
        // H += J J' summed over a patch = Sxx aa' + Sxy (ab'+ba') + Syy bb'
        float hp[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) hp[j] = 0.f;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          Patch& p = pt[k];
          const bool m = (mbits >> k) & 1u;
          float zi = p.zi, xn = p.xn, yn = p.yn;
          asm volatile("" : "+v"(zi), "+v"(xn), "+v"(yn));  // or the ~60 products below are hoisted out of the loop, per patch
          const float ja[6] = {-zi * fl, 0.f, xn * zi * fl, xn * yn * fl, -(1.f + xn * xn) * fl, yn * fl};
          const float jb[6] = {0.f, -zi * fl, yn * zi * fl, (1.f + yn * yn) * fl, -xn * yn * fl, -xn * fl};
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
              const float v = p.Sxx * (ja[i] * ja[j]) + p.Sxy * (ja[i] * jb[j] + jb[i] * ja[j]) + p.Syy * (jb[i] * jb[j]);
              hp[sym6(i, j)] += m ? v : 0.f;
            }
          p.inH = (int)m;
        }
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) {
          const float t8 = wave_reduce8(hp + 8 * gq, lane);
          if ((lane & 7) == 0) g.red[8 * gq + (lane >> 3)] = t8;
        }
        SVO_WAVE_LDS_FENCE();  // same-wave LDS hand-over: keep the order
        // H^-1 by Gauss-Jordan on a 6x6 tile held one element per lane (36 lanes), LDS as the row/column
        // exchange: a few registers instead of the ~90 a register LDL^T keeps live, which matters here
        // because the lane also carries the state of up to four patches.  A zero pivot contributes nothing,
        // like the D^-1 step of Eigen's LDLT::solve.  Runs when the set of patches inside the image changes
        // (about once per level).
        if (lane < 21) g.H[lane] = (double)g.red[lane];
        SVO_WAVE_LDS_FENCE();
        const int gi = lane / 6, gj = lane - 6 * gi;
        if (lane < 36) {
          g.A[lane] = g.H[sym6_rt(gi, gj)];
          g.Hinv[lane] = (gi == gj) ? 1.0 : 0.0;
        }
        for (int kk = 0; kk < 6; ++kk) {
          SVO_WAVE_LDS_FENCE();
          if (lane < 36) {
            const double pv = g.A[kk * 6 + kk];
            const double aik = g.A[gi * 6 + kk];
            const double akj = g.A[kk * 6 + gj];
            const double bkj = g.Hinv[kk * 6 + gj];
            const double aij = g.A[lane];
            const double bij = g.Hinv[lane];
            const double ip = (fabs(pv) > 2.2250738585072014e-308) ? sia_rcp(pv) : 0.0;
            const double na = (gi == kk) ? akj * ip : aij - aik * (akj * ip);
            const double nb = (gi == kk) ? bkj * ip : bij - aik * (bkj * ip);
            SVO_LANES_LDS_FENCE();
            g.A[lane] = na;
            g.Hinv[lane] = nb;
          }
        }
        SVO_WAVE_LDS_FENCE();
      }
      ++evals;

      // -- solve() / update() and the stop / rollback rules of vk::NLLSSolver::optimizeGaussNewton
      //    (:245-258), in this same wave --
      int done = 0;
      {
        const double b0 = (double)readlane_f32(tot, 0), b1 = (double)readlane_f32(tot, 8), b2 = (double)readlane_f32(tot, 16);
        const double b3 = (double)readlane_f32(tot, 24), b4 = (double)readlane_f32(tot, 32), b5 = (double)readlane_f32(tot, 40);
        const float chi2_sum = readlane_f32(tot, 48);
        const int n_meas = (int)readlane_f32(tot, 56);
        // x_ = H_.ldlt().solve(Jres_) (:247), here x = H^-1 Jres: lane i (<6) owns row i
        double xi;
        {
          const double* row = &g.Hinv[6 * (lane < 6 ? lane : 0)];
          xi = row[0] * b0 + row[1] * b1 + row[2] * b2 + row[3] * b3 + row[4] * b4 + row[5] * b5;
        }
        const double x0 = readlane_f64<0>(xi), x1 = readlane_f64<1>(xi), x2 = readlane_f64<2>(xi);
        const double x3 = readlane_f64<3>(xi), x4 = readlane_f64<4>(xi), x5 = readlane_f64<5>(xi);
        n_meas_last = n_meas;
        // return chi2/n_meas_  (float / size_t -> float), :242
        const double new_chi2 = (double)(chi2_sum / (float)n_meas);
        if (isnan(x0)) stop = 1;  // solve(), :248-249
        double q[4];
        if ((iter > 0 && new_chi2 > chi2_prev) || stop) {
          // rollback: model = old_model
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = g.oq[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) tr[k] = g.ot[k];
          done = 1;
        } else {
          // update(): T_new = T_old * SE3::exp(-x_)  (:253-258)
          const float mx[6] = {-(float)x0, -(float)x1, -(float)x2, -(float)x3, -(float)x4, -(float)x5};
          float eqf[4], etf[3];
          se3_exp_f32(mx, eqf, etf);
          const double eq[4] = {eqf[0], eqf[1], eqf[2], eqf[3]};
          const double et[3] = {etf[0], etf[1], etf[2]};
          double oq[4], ot[3], rt[3];
#pragma unroll
          for (int k = 0; k < 4; ++k) oq[k] = g.q[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) ot[k] = g.t[k];
          quat_rot(oq, et, rt);
          quat_mul(oq, eq, q);
          quat_normalize_fast(q);
#pragma unroll
          for (int k = 0; k < 3; ++k) tr[k] = ot[k] + rt[k];
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g.oq[k] = oq[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) g.ot[k] = ot[k];
          }
          chi2_prev = new_chi2;
          // vk::norm_max(x_) <= eps_
          const double nm = fmax(fmax(fmax(fabs(x0), fabs(x1)), fmax(fabs(x2), fabs(x3))), fmax(fabs(x4), fabs(x5)));
          if (nm <= P.eps) done = 1;
        }
        double Rm[9];
        quat_to_R(q, Rm);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) g.q[k] = q[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) g.t[k] = tr[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = sia_uni(Rm[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) tr[k] = sia_uni(tr[k]);
      }
      if (done) break;
    }
    if (lane == 0 && a.iters) a.iters[SVO_HIP_MAX_LEVELS * b + level] = evals;
  }

  if (lane == 0) {
    for (int k = 0; k < 9; ++k) a.T_out[12 * b + k] = R[k];
    for (int k = 0; k < 3; ++k) a.T_out[12 * b + 9 + k] = tr[k];
    a.n_tracked[b] = n_meas_last / 16;
    if (a.chi2) a.chi2[b] = chi2_prev;
    if (a.status) a.status[b] = stop ? SVO_HIP_SIA_STOP : 0;
  }
  if (a.H_out && lane < 36) {
    const int i = lane / 6, j = lane - 6 * i;
    a.H_out[36 * b + lane] = g.H[sym6_rt(i, j)];
  }
}

#undef SIA_BT

template <int PPL, bool DIST>
int launch_ppl(const SiaArgs& args, int B, hipStream_t s) {
  const int n_slots = args.n_stride;  // one 128-byte reference-patch slot per patch
  const size_t lds = (size_t)8 * n_slots * sizeof(float4);
  hipLaunchKernelGGL((sia_wave_kernel<PPL, DIST>), dim3(B), dim3(64), lds, s, args, n_slots);
  return check_launch();
}

template <bool DIST>
int launch_dist(const SiaArgs& args, int B, hipStream_t s) {
  const int ppl = (args.n_stride + 63) / 64;
  switch (ppl) {
    case 1: return launch_ppl<1, DIST>(args, B, s);
    case 2: return launch_ppl<2, DIST>(args, B, s);
    case 3: return launch_ppl<3, DIST>(args, B, s);
    default: return launch_ppl<4, DIST>(args, B, s);
  }
}

}  // namespace

namespace svo_sia {

// Where one wave per frame pays: the lane's patches run one after the other, so a frame takes longer than
// with a workgroup of waves working side by side, and only a batch large enough to give every SIMD its two
// waves turns the saved barriers and the idle solver-wave partners into throughput.  Measured on the 640x480
// batch of 16384 frames: 1.29 against 1.36 ms at 192 patches per frame (3 per lane); at 200 (4 per lane, the
// fourth pass for 8 patches) the workgroup kernel wins, 1.41 against 1.62 ms.
constexpr int SIAW_MAX_PATCHES = 192;
// Distorted cameras: the model's world2cam in the loop costs registers (26 / 89 spilled dwords at 2 / 3 patches
// per lane), so they take this kernel only with one patch per lane.
bool sia_wave_applies(const SiaArgs& args, int B) {
  const int max_patches = args.P.cam_model == SVO_HIP_CAM_PINHOLE ? SIAW_MAX_PATCHES : 64;
  return args.n_stride <= max_patches && B >= 1024;
}

int launch_sia_wave(const SiaArgs& args, int B, hipStream_t s) {
  return args.P.cam_model == SVO_HIP_CAM_PINHOLE ? launch_dist<false>(args, B, s) : launch_dist<true>(args, B, s);
}

}  // namespace svo_sia
