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
#pragma clang fp contract(off)
#include "track_kernels.h"
#include "track_math.h"

using namespace svo_capi;
using namespace svo_dev;
using namespace svo_track;

namespace {

// byte i (compile-time) of the 100-byte template held in 25 dwords
#define PWB(i) ((int)((g[(i) >> 2] >> (8 * ((i)&3))) & 0xffu))

// ALIGN_REMAT_TEMPLATE keeps the byte extractions and the template gradients derived from g[] inside the
// iteration that uses them (the compiler otherwise hoists them out of the loop as 192 floats: 255 VGPRs, 2 waves
// per SIMD).  Measured: 165 VGPRs / 3 waves per SIMD but +50 % instructions per iteration -- no faster (0.82 ms
// either way on 3.3 M trials), so the hoisted form stays the default.  Emits no instruction.
#ifdef ALIGN_REMAT_TEMPLATE
#define ALIGN_OPAQUE_TEMPLATE(g)                                                                                   \
  asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),  \
               "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]));                                        \
  asm volatile("" : "+v"(g[13]), "+v"(g[14]), "+v"(g[15]), "+v"(g[16]), "+v"(g[17]), "+v"(g[18]), "+v"(g[19]),       \
               "+v"(g[20]), "+v"(g[21]), "+v"(g[22]), "+v"(g[23]), "+v"(g[24]))
#else
#define ALIGN_OPAQUE_TEMPLATE(g)
#endif

__device__ __forceinline__ void cut_row9(const uint32_t d[3], uint32_t sel, float out[9]);
// bytes [x0, x0+8] of the image row at byte offset ro (svo_pyr::row_off) as floats: one 12-byte run of three aligned
// dwords starting at xa = x0 & ~3 (one load when it lies inside a tile row of the store, two otherwise), sel = x0 & 3
__device__ __forceinline__ void load_row9(const uint8_t* __restrict__ img, uint32_t ro, int xa, uint32_t sel, float out[9]) {
  uint32_t d[3];
  svo_pyr::load_run12(img, ro, xa, d);
  cut_row9(d, sel, out);
}
// bytes [sel, sel+8] of three consecutive dwords as floats
__device__ __forceinline__ void cut_row9(const uint32_t d[3], uint32_t sel, float out[9]) {
  const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sel);
  const uint32_t mid = __builtin_amdgcn_alignbyte(d2, d1, sel);
  const uint32_t hi = d2 >> (8 * sel);
  out[0] = (float)(lo & 0xffu);
  out[1] = (float)((lo >> 8) & 0xffu);
  out[2] = (float)((lo >> 16) & 0xffu);
  out[3] = (float)(lo >> 24);
  out[4] = (float)(mid & 0xffu);
  out[5] = (float)((mid >> 8) & 0xffu);
  out[6] = (float)((mid >> 16) & 0xffu);
  out[7] = (float)(mid >> 24);
  out[8] = (float)(hi & 0xffu);
}

// float -> int like the x86 build (cvttss2si of floor(x)): NaN / out of range -> INT_MIN
__device__ __forceinline__ int floor_int(float x) {
  const float f = floorf(x);
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int)0x80000000;
  return (int)f;
}

// align2D, feature_alignment.cpp:149-277.  Returns converged; (u,v) in/out.
__device__ __forceinline__ bool align2d_lane(const uint8_t* __restrict__ img, int cols, int rows, int pitch,
                                             uint32_t g[25], int n_iter, float& u, float& v, bool& wrote,
                                             int& n_eval) {
  bool converged = false;
  wrote = true;
  n_eval = 0;  // residual evaluations (9x9 windows read); dead code unless the caller stores it
  // H = sum J J', J = (dx, dy, 1) (:166-181).  dx, dy are half-integers (byte differences / 2) and every
  // partial sum of the reference's float accumulation is a multiple of 0.25 below 2^22: no rounding ever
  // happens, so the sums can be formed in any order -- here as integers of the doubled gradients
  // (|2dx| <= 255, sum of squares <= 64 * 255^2 < 2^23), five v_mad_i32_i24 per pixel instead of nine
  // multiply-adds, and converted once.  Same bits.
  float H[9];
  {
    int sxx = 0, sxy = 0, syy = 0, sx = 0, sy = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int c = (y + 1) * 10 + x + 1;
        const int gx2 = PWB(c + 1) - PWB(c - 1);    // 2 dx
        const int gy2 = PWB(c + 10) - PWB(c - 10);  // 2 dy
        sxx += gx2 * gx2;
        sxy += gx2 * gy2;
        syy += gy2 * gy2;
        sx += gx2;
        sy += gy2;
      }
    H[0] = 0.25f * (float)sxx;
    H[1] = H[3] = 0.25f * (float)sxy;
    H[4] = 0.25f * (float)syy;
    H[2] = H[6] = 0.5f * (float)sx;
    H[5] = H[7] = 0.5f * (float)sy;
    H[8] = 64.f;
  }
  float Hinv[9];
  inv3f(H, Hinv);
  float mean_diff = 0;
  const float min_update_squared = (float)(0.03 * 0.03);
  for (int iter = 0; iter < n_iter; ++iter) {
    ALIGN_OPAQUE_TEMPLATE(g);
    const int u_r = floor_int(u);
    const int v_r = floor_int(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (isnan(u) || isnan(v)) {  // unreachable after the bounds test, kept for the record (:209)
      wrote = false;
      return false;
    }
    ++n_eval;
    const float subpix_x = u - (float)u_r;
    const float subpix_y = v - (float)v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    float Jres0 = 0, Jres1 = 0, Jres2 = 0;
    const int wxa = (u_r - 4) & ~3;
    const uint32_t wsel = (uint32_t)((u_r - 4) & 3);
    float P0[9], P1[9];
#ifndef ALIGN_ROW_LOADS  // all nine rows fetched up front, ONE three-way branch on the tile position for the window
    // (a branch per row measured 2.15 against 1.68 ms for findMatchDirect on 3.3 M trials)
    uint32_t win[9][3];
    svo_pyr::load_window12<9>(img, pitch, wxa, v_r - 4, win);
    cut_row9(win[0], wsel, P0);
#else
    load_row9(img, svo_pyr::row_off(v_r - 4, pitch), wxa, wsel, P0);
#endif
#pragma unroll
    for (int y = 0; y < 8; ++y) {
#ifndef ALIGN_ROW_LOADS
      cut_row9(win[y + 1], wsel, P1);
#else
      load_row9(img, svo_pyr::row_off(v_r - 3 + y, pitch), wxa, wsel, P1);
#endif
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int c = (y + 1) * 10 + x + 1;
        const float search_pixel = wTL * P0[x] + wTR * P0[x + 1] + wBL * P1[x] + wBR * P1[x + 1];
        const float res = search_pixel - (float)PWB(c) + mean_diff;
        Jres0 -= res * (0.5f * (float)(PWB(c + 1) - PWB(c - 1)));
        Jres1 -= res * (0.5f * (float)(PWB(c + 10) - PWB(c - 10)));
        Jres2 -= res;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) P0[k] = P1[k];
    }
    const float up0 = Hinv[0] * Jres0 + Hinv[1] * Jres1 + Hinv[2] * Jres2;
    const float up1 = Hinv[3] * Jres0 + Hinv[4] * Jres1 + Hinv[5] * Jres2;
    const float up2 = Hinv[6] * Jres0 + Hinv[7] * Jres1 + Hinv[8] * Jres2;
    u += up0;
    v += up1;
    mean_diff += up2;
    if (up0 * up0 + up1 * up1 < min_update_squared) {
      converged = true;
      break;
    }
  }
  return converged;
}

// align1D, feature_alignment.cpp:30-147
__device__ __forceinline__ bool align1d_lane(const uint8_t* __restrict__ img, int cols, int rows, int pitch,
                                             uint32_t g[25], float dir0, float dir1, int n_iter, float& u,
                                             float& v, double& h_inv, bool& wrote, int& n_eval) {
  bool converged = false;
  wrote = true;
  n_eval = 0;
  float H[4] = {0, 0, 0, 0};
#pragma unroll
  for (int y = 0; y < 8; ++y)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int c = (y + 1) * 10 + x + 1;
      // J[0] = 0.5*(dir[0]*(it[1]-it[-1]) + dir[1]*(it[ref_step]-it[-ref_step]))  (double 0.5 * float)
      const float s = dir0 * (float)(PWB(c + 1) - PWB(c - 1)) + dir1 * (float)(PWB(c + 10) - PWB(c - 10));
      const float J0 = (float)(0.5 * (double)s);
      H[0] += J0 * J0;
      H[1] += J0 * 1.f;
      H[2] += 1.f * J0;
      H[3] += 1.f * 1.f;
    }
  h_inv = 1.0 / (double)H[0] * 8 * 8;
  float Hinv[4];
  inv2<float>(H, Hinv);
  float mean_diff = 0;
  const float min_update_squared = (float)(0.03 * 0.03);
  float chi2 = 0;
  float up0 = 0, up1 = 0;
  for (int iter = 0; iter < n_iter; ++iter) {
    ALIGN_OPAQUE_TEMPLATE(g);
    const int u_r = floor_int(u);
    const int v_r = floor_int(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (isnan(u) || isnan(v)) {
      wrote = false;
      return false;
    }
    ++n_eval;
    const float subpix_x = u - (float)u_r;
    const float subpix_y = v - (float)v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    float new_chi2 = 0, Jres0 = 0, Jres1 = 0;
    const int wxa = (u_r - 4) & ~3;
    const uint32_t wsel = (uint32_t)((u_r - 4) & 3);
    float P0[9], P1[9];
#ifndef ALIGN_ROW_LOADS  // all nine rows fetched up front, ONE three-way branch on the tile position for the window
    // (a branch per row measured 2.15 against 1.68 ms for findMatchDirect on 3.3 M trials)
    uint32_t win[9][3];
    svo_pyr::load_window12<9>(img, pitch, wxa, v_r - 4, win);
    cut_row9(win[0], wsel, P0);
#else
    load_row9(img, svo_pyr::row_off(v_r - 4, pitch), wxa, wsel, P0);
#endif
#pragma unroll
    for (int y = 0; y < 8; ++y) {
#ifndef ALIGN_ROW_LOADS
      cut_row9(win[y + 1], wsel, P1);
#else
      load_row9(img, svo_pyr::row_off(v_r - 3 + y, pitch), wxa, wsel, P1);
#endif
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int c = (y + 1) * 10 + x + 1;
        const float search_pixel = wTL * P0[x] + wTR * P0[x + 1] + wBL * P1[x] + wBR * P1[x + 1];
        const float res = search_pixel - (float)PWB(c) + mean_diff;
        const float s = dir0 * (float)(PWB(c + 1) - PWB(c - 1)) + dir1 * (float)(PWB(c + 10) - PWB(c - 10));
        Jres0 -= res * (float)(0.5 * (double)s);
        Jres1 -= res;
        new_chi2 += res * res;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) P0[k] = P1[k];
    }
    if (iter > 0 && new_chi2 > chi2) {
      u -= up0;  // sic (:116-117)
      v -= up1;
      break;
    }
    chi2 = new_chi2;
    up0 = Hinv[0] * Jres0 + Hinv[1] * Jres1;
    up1 = Hinv[2] * Jres0 + Hinv[3] * Jres1;
    u += up0 * dir0;
    v += up0 * dir1;
    mean_diff += up1;
    if (up0 * up0 + up1 * up1 < min_update_squared) {
      converged = true;
      break;
    }
  }
  return converged;
}

constexpr int ALIGN_BLOCK = 64;

template <bool COUNT>
__global__ void __launch_bounds__(ALIGN_BLOCK) align_kernel(const AlignArgs a) {
  const int t = blockIdx.x * ALIGN_BLOCK + threadIdx.x;
  // ALIGN_TEMPLATE_LDS: the 64 templates of the workgroup are 6400 contiguous bytes.  Read per lane they are 25
  // dword gathers with a 100-byte lane stride (64 cache lines per instruction); read as the contiguous block
  // they are and handed out through LDS they are 7 coalesced 16-byte loads (a lane's 25 LDS words sit 25 banks
  // apart from its neighbour's: conflict-free).  Measured: no difference (5.96 against 5.93 ms for the full-track
  // step) -- the kernel is bound by its per-trial dependent arithmetic, not by these loads; off by default.
#ifdef ALIGN_TEMPLATE_LDS
  __shared__ uint32_t s_tpl[ALIGN_BLOCK * 25];
  {
    const long long first = (long long)blockIdx.x * ALIGN_BLOCK;               // first trial of the workgroup
    const long long left = (long long)a.M - first;
    const int n_dw = 25 * (int)(left < ALIGN_BLOCK ? left : ALIGN_BLOCK);      // dwords present (multiple of 25)
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.pwb + (size_t)first * 100);
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int q = 4 * (threadIdx.x + ALIGN_BLOCK * k);  // first dword of this lane's quad
        if (q + 3 < n_dw) {
          *reinterpret_cast<uint4*>(&s_tpl[q]) = *reinterpret_cast<const uint4*>(src + q);
        } else {
          for (int j = q; j < n_dw && j < q + 4; ++j) s_tpl[j] = src[j];
        }
      }
    } else {  // the caller's buffer starts off a 16-byte boundary: dwords
      for (int j = threadIdx.x; j < n_dw; j += ALIGN_BLOCK) s_tpl[j] = src[j];
    }
  }
  __syncthreads();
#endif
  if (t >= a.M) return;
  if (a.active && !a.active[t]) {
    a.ok[t] = 0;  // px_out is left as it is (findMatchDirect returns before touching px_cur)
    if (COUNT) a.iters[t] = 0;
    return;
  }
  const int level = a.level[t];
  const uint8_t* img = a.store + (int64_t)a.slot[t] * a.L.slot_bytes + a.L.offset[level];
  const int cols = a.L.w[level], rows = a.L.h[level], pitch = a.L.pitch[level];
  uint32_t g[25];
  {
#ifdef ALIGN_TEMPLATE_LDS
    const uint32_t* gp = &s_tpl[threadIdx.x * 25];
#else
    const uint32_t* gp = reinterpret_cast<const uint32_t*>(a.pwb + (size_t)t * 100);
#endif
#pragma unroll
    for (int k = 0; k < 25; ++k) g[k] = gp[k];
  }
  float u = (float)a.px_in[2 * t];
  float v = (float)a.px_in[2 * t + 1];
  bool wrote = true;
  bool ok;
  int n_eval = 0;
  const bool one_d = a.use_1d && a.use_1d[t];
  if (one_d) {
    double h_inv = 0;
    ok = align1d_lane(img, cols, rows, pitch, g, a.dir[2 * t], a.dir[2 * t + 1], a.n_iter, u, v, h_inv, wrote, n_eval);
    if (a.h_inv) a.h_inv[t] = h_inv;
  } else {
    ok = align2d_lane(img, cols, rows, pitch, g, a.n_iter, u, v, wrote, n_eval);
  }
  if (COUNT) a.iters[t] = n_eval;
  a.ok[t] = ok ? 1 : 0;
  double ou = wrote ? (double)u : a.px_in[2 * t];
  double ov = wrote ? (double)v : a.px_in[2 * t + 1];
  if (a.scale_out) {
    ou = ou * (double)(1 << level);
    ov = ov * (double)(1 << level);
  }
  a.px_out[2 * t] = ou;
  a.px_out[2 * t + 1] = ov;
}

}  // namespace

namespace svo_track {
int launch_align(const AlignArgs& a, hipStream_t s) {
  if (a.M <= 0) return SVO_HIP_OK;
  const dim3 grid((a.M + ALIGN_BLOCK - 1) / ALIGN_BLOCK), blk(ALIGN_BLOCK);
  if (a.iters) hipLaunchKernelGGL(align_kernel<true>, grid, blk, 0, s, a);  // instrumented: also counts evaluations
  else hipLaunchKernelGGL(align_kernel<false>, grid, blk, 0, s, a);
  return check_launch();
}
}  // namespace svo_track

static int align_batch(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M, const int32_t* d_slot,
                       const int32_t* d_level, const uint8_t* d_patch_with_border, const float* d_dir,
                       const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok, double* d_h_inv,
                       int32_t* d_iters, void* stream) {
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
  return launch_align(a, static_cast<hipStream_t>(stream));
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
