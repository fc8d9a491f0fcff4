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
//     wave 0 sums the per-wave partials from LDS, back-substitutes through the
//     cached factors, applies the stop / rollback rules and writes the new
//     pose (quaternion + t, as Sophus stores it) to LDS for everyone.
//
// Numerics: pixel math in f32 (like the reference), projection, Jacobian rows,
// H, Jres and the pose in f64 (like the reference); sums are tree- instead of
// sequentially reduced, so results agree to rounding, not bit-for-bit.
#include "capi_common.h"
#include "device_math.h"

using namespace svo_capi;
using namespace svo_dev;

namespace {

struct SiaArgs {
  svo_hip_pyr_layout L;
  const uint8_t* store;
  const int32_t* ref_slot;
  const int32_t* cur_slot;
  const int32_t* n;
  int n_stride;
  const double* px;
  const double* xyz;
  const uint8_t* valid;
  svo_hip_sia_params P;
  const double* T_in;
  double* T_out;
  double* H_out;
  int32_t* n_tracked;
  int32_t* iters;
  double* chi2;
  int32_t* status;
};

// bytes [x0, x0+4] of a row (x0 = first column, any alignment) as floats
__device__ __forceinline__ void load_row5(const uint8_t* __restrict__ row, int x0, float out[5]) {
  const int xa = x0 & ~3;
  const uint32_t sel = (uint32_t)(x0 & 3);
  const uint32_t* p = reinterpret_cast<const uint32_t*>(row + xa);
  const uint32_t d0 = p[0], d1 = p[1];
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sel);  // bytes x0..x0+3
  const uint32_t hi = d1 >> (8 * sel);                          // byte x0+4 in bits 0..7
  out[0] = (float)(lo & 0xffu);
  out[1] = (float)((lo >> 8) & 0xffu);
  out[2] = (float)((lo >> 16) & 0xffu);
  out[3] = (float)(lo >> 24);
  out[4] = (float)(hi & 0xffu);
}

// bytes [x0, x0+6] of a row as floats
__device__ __forceinline__ void load_row7(const uint8_t* __restrict__ row, int x0, float out[7]) {
  const int xa = x0 & ~3;
  const uint32_t sel = (uint32_t)(x0 & 3);
  const uint32_t* p = reinterpret_cast<const uint32_t*>(row + xa);
  const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sel);  // bytes 0..3
  const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sel);  // bytes 4..7
  out[0] = (float)(lo & 0xffu);
  out[1] = (float)((lo >> 8) & 0xffu);
  out[2] = (float)((lo >> 16) & 0xffu);
  out[3] = (float)(lo >> 24);
  out[4] = (float)(hi & 0xffu);
  out[5] = (float)((hi >> 8) & 0xffu);
  out[6] = (float)((hi >> 16) & 0xffu);
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) sia_kernel(const SiaArgs a) {
  constexpr int NW = BLOCK / 64;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  __shared__ double s_q[4], s_t[3];      // model: T_cur_from_ref as Sophus stores it
  __shared__ double s_R[9];              // rotation matrix of s_q (for the lanes)
  __shared__ double s_oq[4], s_ot[3];    // old_model (rollback)
  __shared__ double s_part[NW][8];       // per-wave partials: Jres[6], chi2, n_meas
  __shared__ double s_Hpart[NW][21];     // per-wave partials of H (packed upper)
  __shared__ double s_H[21];             // H_ of the last evaluated iteration
  __shared__ double s_LD[21];            // its LDL' factors
  __shared__ int s_done;                 // level finished flag

  const int n = a.n[b];
  const svo_hip_sia_params& P = a.P;

  if (n <= 0) {  // sparse_img_align.cpp:47-51: nothing to track, pose untouched
    if (tid == 0) {
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

  // ---- per-lane geometry (Feature::px, f*depth) --------------------------
  const size_t fo = (size_t)b * a.n_stride + tid;
  const bool has = (tid < n) && (a.valid ? a.valid[fo] != 0 : true);
  double pxx = 0, pxy = 0, X = 0, Y = 0, Z = 1;
  if (has) {
    pxx = a.px[2 * fo];
    pxy = a.px[2 * fo + 1];
    X = a.xyz[3 * fo];
    Y = a.xyz[3 * fo + 1];
    Z = a.xyz[3 * fo + 2];
  }
  const uint8_t* ref_base = a.store + (int64_t)a.ref_slot[b] * a.L.slot_bytes;
  const uint8_t* cur_base = a.store + (int64_t)a.cur_slot[b] * a.L.slot_bytes;

  if (tid == 0) {
    double R[9], q[4];
    for (int k = 0; k < 9; ++k) R[k] = a.T_in[12 * b + k];
    quat_from_R(R, q);
    quat_to_R(q, R);
    for (int k = 0; k < 4; ++k) s_q[k] = s_oq[k] = q[k];
    for (int k = 0; k < 3; ++k) s_t[k] = s_ot[k] = a.T_in[12 * b + 9 + k];
    for (int k = 0; k < 9; ++k) s_R[k] = R[k];
    for (int k = 0; k < 21; ++k) s_H[k] = 0.0;
    if (a.iters)
      for (int k = 0; k < SVO_HIP_MAX_LEVELS; ++k) a.iters[SVO_HIP_MAX_LEVELS * b + k] = 0;
  }

  // NLLSSolver state after reset(); authoritative copy lives in wave 0 (uniform)
  double chi2_prev = 1e10;
  int stop = 0;
  int n_meas_last = 0;

  float refv[16], dxv[16], dyv[16];  // ref_patch_cache_ row + gradients of this lane's patch
#pragma unroll
  for (int k = 0; k < 16; ++k) refv[k] = dxv[k] = dyv[k] = 0.f;
  double ja[6], jb[6];
  double Sxx = 0, Sxy = 0, Syy = 0;
  bool vis = false;  // visible_fts_[i]; never cleared between levels (:57)

  __syncthreads();

  for (int level = P.max_level; level >= P.min_level; --level) {
    const int cols = a.L.w[level], rows = a.L.h[level], pitch = a.L.pitch[level];
    const uint8_t* ref_img = ref_base + a.L.offset[level];
    const uint8_t* cur_img = cur_base + a.L.offset[level];
    const float scale = 1.0f / (float)(1 << level);

    // ---- precomputeReferencePatches (:84-145) ----------------------------
    {
      const float u_ref = (float)(pxx * (double)scale);
      const float v_ref = (float)(pxy * (double)scale);
      const int u_i = (int)floorf(u_ref);
      const int v_i = (int)floorf(v_ref);
      const bool inb = has && !(u_i - 3 < 0 || v_i - 3 < 0 || u_i + 3 >= cols || v_i + 3 >= rows);
      // jacobian_cache_.setZero() (:64): features skipped below keep J = 0
#pragma unroll
      for (int k = 0; k < 6; ++k) ja[k] = jb[k] = 0.0;
      Sxx = Sxy = Syy = 0.0;
      if (inb) {
        vis = true;
        // Frame::jacobian_xyz2uv(xyz_ref) scaled by focal_length / 2^level (:139-140)
        const double fl = fabs(P.fx) / (double)(1 << level);
        const double z_inv = 1.0 / Z;
        const double z_inv_2 = z_inv * z_inv;
        const double j02 = X * z_inv_2, j12 = Y * z_inv_2;
        ja[0] = -z_inv * fl;
        ja[1] = 0.0;
        ja[2] = j02 * fl;
        ja[3] = (Y * j02) * fl;
        ja[4] = -(1.0 + X * j02) * fl;
        ja[5] = (Y * z_inv) * fl;
        jb[0] = 0.0;
        jb[1] = -z_inv * fl;
        jb[2] = j12 * fl;
        jb[3] = (1.0 + Y * j12) * fl;
        jb[4] = -(Y * j02) * fl;
        jb[5] = -(X * z_inv) * fl;

        const float su = u_ref - (float)u_i, sv = v_ref - (float)v_i;
        const float wtl = (float)((1.0 - su) * (1.0 - sv));
        const float wtr = (float)(su * (1.0 - sv));
        const float wbl = (float)((1.0 - su) * sv);
        const float wbr = (float)((double)su * (double)sv);

        float W[7][7];
#pragma unroll
        for (int r = 0; r < 7; ++r) load_row7(ref_img + (int64_t)(v_i - 3 + r) * pitch, u_i - 3, W[r]);
        // bilinear image B(r,c) = interpolated reference at window pixel (r,c), r,c in 0..5
        // patch pixel (y,x) sits at window (y+1, x+1); gradients are central
        // differences of the interpolated image (:133-136)
        float Bi[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const bool need = ((r >= 1 && r <= 4) && (c <= 5)) || ((c >= 1 && c <= 4) && (r <= 5));
            Bi[r][c] = need ? (wtl * W[r][c] + wtr * W[r][c + 1] + wbl * W[r + 1][c] + wbr * W[r + 1][c + 1]) : 0.f;
          }
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int k = y * 4 + x;
            refv[k] = Bi[y + 1][x + 1];
            dxv[k] = 0.5f * (Bi[y + 1][x + 2] - Bi[y + 1][x]);
            dyv[k] = 0.5f * (Bi[y + 2][x + 1] - Bi[y][x + 1]);
          }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          Sxx += (double)dxv[k] * (double)dxv[k];
          Sxy += (double)dxv[k] * (double)dyv[k];
          Syy += (double)dyv[k] * (double)dyv[k];
        }
      } else {
        // J column stays zero; a stale ref_patch_cache_ row (if any) is kept
#pragma unroll
        for (int k = 0; k < 16; ++k) dxv[k] = dyv[k] = 0.f;
      }
    }

    // ---- vk::NLLSSolver::optimizeGaussNewton ------------------------------
    int inH = -1;  // membership of this lane in the sum that s_H currently holds
    int evals = 0;
    for (int iter = 0; iter < P.n_iter; ++iter) {
      // -- computeResiduals (:147-243): this lane's patch -------------------
      bool m = false;
      float gx = 0.f, gy = 0.f, c2 = 0.f;
      if (vis) {
        const double xc = s_R[0] * X + s_R[1] * Y + s_R[2] * Z + s_t[0];
        const double yc = s_R[3] * X + s_R[4] * Y + s_R[5] * Z + s_t[1];
        const double zc = s_R[6] * X + s_R[7] * Y + s_R[8] * Z + s_t[2];
        // vk::PinholeCamera::world2cam(project2d(xyz))
        const double pu = P.fx * (xc / zc) + P.cx;
        const double pv = P.fy * (yc / zc) + P.cy;
        const float u_cur = (float)pu * scale;
        const float v_cur = (float)pv * scale;
        const float fu = floorf(u_cur), fv = floorf(v_cur);
        // NaN / huge coordinates fail the comparisons below like the int tests do
        if (fu - 3.f >= 0.f && fv - 3.f >= 0.f && fu + 3.f < (float)cols && fv + 3.f < (float)rows) {
          m = true;
          const int u_i = (int)fu, v_i = (int)fv;
          const float su = u_cur - fu, sv = v_cur - fv;
          const float wtl = (float)((1.0 - su) * (1.0 - sv));
          const float wtr = (float)(su * (1.0 - sv));
          const float wbl = (float)((1.0 - su) * sv);
          const float wbr = (float)((double)su * (double)sv);
          float W[5][5];
#pragma unroll
          for (int r = 0; r < 5; ++r) load_row5(cur_img + (int64_t)(v_i - 2 + r) * pitch, u_i - 2, W[r]);
#pragma unroll
          for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int k = y * 4 + x;
              const float I = wtl * W[y][x] + wtr * W[y][x + 1] + wbl * W[y + 1][x] + wbr * W[y + 1][x + 1];
              const float res = I - refv[k];
              c2 += res * res;
              gx += res * dxv[k];
              gy += res * dyv[k];
            }
        }
      }
      // -- workgroup reduction of Jres, chi2, n_meas ------------------------
      double part[8];
#pragma unroll
      for (int k = 0; k < 6; ++k) part[k] = m ? -((double)gx * ja[k] + (double)gy * jb[k]) : 0.0;
      part[6] = (double)c2;
      part[7] = m ? 16.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) part[k] = wave_sum(part[k]);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_part[wave][k] = part[k];
      }
      const int changed = __syncthreads_or((int)m != inH);
      if (changed) {
        // the set of patches inside the current image changed: rebuild H
        double hp[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) {
            const double v = Sxx * (ja[i] * ja[j]) + Sxy * (ja[i] * jb[j] + jb[i] * ja[j]) + Syy * (jb[i] * jb[j]);
            hp[sym6(i, j)] = m ? v : 0.0;
          }
#pragma unroll
        for (int k = 0; k < 21; ++k) hp[k] = wave_sum(hp[k]);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 21; ++k) s_Hpart[wave][k] = hp[k];
        }
        inH = (int)m;
        __syncthreads();
      }
      ++evals;

      // -- solve / update / stop rules: wave 0, all lanes redundantly -------
      if (wave == 0) {
        double tot[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          double v = s_part[0][k];
#pragma unroll
          for (int w = 1; w < NW; ++w) v += s_part[w][k];
          tot[k] = v;
        }
        double LD[21];
        if (changed) {
          double H[21];
#pragma unroll
          for (int k = 0; k < 21; ++k) {
            double v = s_Hpart[0][k];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += s_Hpart[w][k];
            H[k] = v;
          }
          ldlt6_factor(H, LD);
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 21; ++k) {
              s_H[k] = H[k];
              s_LD[k] = LD[k];
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 21; ++k) LD[k] = s_LD[k];
        }
        double x[6];
        ldlt6_solve(LD, tot, x);
        const int n_meas = (int)tot[7];
        n_meas_last = n_meas;
        // return chi2/n_meas_  (float / size_t -> float), :242
        const double new_chi2 = (double)((float)tot[6] / (float)n_meas);
        if (isnan(x[0])) stop = 1;  // solve(), :248-249
        int done = 0;
        if ((iter > 0 && new_chi2 > chi2_prev) || stop) {
          // rollback: model = old_model
          if (lane == 0) {
            double q[4] = {s_oq[0], s_oq[1], s_oq[2], s_oq[3]};
            double R[9];
            quat_to_R(q, R);
#pragma unroll
            for (int k = 0; k < 4; ++k) s_q[k] = q[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) s_t[k] = s_ot[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) s_R[k] = R[k];
          }
          done = 1;
        } else {
          // update(): T_new = T_old * SE3::exp(-x_)  (:253-258)
          double mx[6], eq[4], et[3];
#pragma unroll
          for (int k = 0; k < 6; ++k) mx[k] = -x[k];
          se3_exp(mx, eq, et);
          double q[4] = {s_q[0], s_q[1], s_q[2], s_q[3]};
          double t[3] = {s_t[0], s_t[1], s_t[2]};
          double rt[3], nq[4], R[9];
          quat_rot(q, et, rt);
          quat_mul(q, eq, nq);
          quat_normalize(nq);
          quat_to_R(nq, R);
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              s_oq[k] = q[k];
              s_q[k] = nq[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              s_ot[k] = t[k];
              s_t[k] = t[k] + rt[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) s_R[k] = R[k];
          }
          chi2_prev = new_chi2;
          double nm = 0.0;  // vk::norm_max(x_) <= eps_
#pragma unroll
          for (int k = 0; k < 6; ++k) nm = fmax(nm, fabs(x[k]));
          if (nm <= P.eps) done = 1;
        }
        if (lane == 0) s_done = done;
      }
      __syncthreads();
      if (s_done) break;
    }
    if (tid == 0 && a.iters) a.iters[SVO_HIP_MAX_LEVELS * b + level] = evals;
    __syncthreads();  // s_done / model are re-used by the next level
  }

  if (tid == 0) {
    double q[4] = {s_q[0], s_q[1], s_q[2], s_q[3]};
    double R[9];
    quat_to_R(q, R);
    for (int k = 0; k < 9; ++k) a.T_out[12 * b + k] = R[k];
    for (int k = 0; k < 3; ++k) a.T_out[12 * b + 9 + k] = s_t[k];
    if (a.H_out)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) a.H_out[36 * b + i * 6 + j] = s_H[sym6(i, j)];
    a.n_tracked[b] = n_meas_last / 16;
    if (a.chi2) a.chi2[b] = chi2_prev;
    if (a.status) a.status[b] = stop ? SVO_HIP_SIA_STOP : 0;
  }
}

template <int BLOCK>
int launch(const SiaArgs& args, int B, hipStream_t s) {
  hipLaunchKernelGGL(sia_kernel<BLOCK>, dim3(B), dim3(BLOCK), 0, s, args);
  return check_launch();
}

}  // namespace

extern "C" int svo_hip_sparse_align(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B,
                                    const int32_t* d_ref_slot, const int32_t* d_cur_slot, const int32_t* d_n,
                                    int n_stride, const double* d_px, const double* d_xyz_ref,
                                    const uint8_t* d_valid, const svo_hip_sia_params* params,
                                    const double* d_T_in, double* d_T_out, double* d_H_out,
                                    int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2, int32_t* d_status,
                                    void* stream) {
  if (!layout_ok(layout) || !d_store || !params || B < 0) return SVO_HIP_EINVAL;
  if (B == 0) return SVO_HIP_OK;
  if (!d_ref_slot || !d_cur_slot || !d_n || !d_px || !d_xyz_ref || !d_T_in || !d_T_out || !d_n_tracked)
    return SVO_HIP_EINVAL;
  if (n_stride < 1) return SVO_HIP_EINVAL;
  if (n_stride > SVO_HIP_MAX_PATCHES) return SVO_HIP_ERANGE;
  if (params->min_level < 0 || params->max_level < params->min_level || params->max_level >= layout->n_levels ||
      params->n_iter < 0)
    return SVO_HIP_EINVAL;
  SiaArgs args;
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_stride <= 64) return launch<64>(args, B, s);
  if (n_stride <= 128) return launch<128>(args, B, s);
  if (n_stride <= 256) return launch<256>(args, B, s);
  if (n_stride <= 512) return launch<512>(args, B, s);
  return launch<1024>(args, B, s);
}
