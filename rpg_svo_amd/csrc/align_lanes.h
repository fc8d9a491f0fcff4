// align_lanes.h -- the per-lane bodies of K3: align2D (feature_alignment.cpp:149-277) and align1D (:30-147) of ONE trial,
// iterations [it0, it1) of the reference's loop, on a level of the tiled pyramid store.  Device functions of
// feature_align.hip's kernels; also compiled for the CPU by the test suite (SVO_HOST_MATH_TEST, see device_math.h: the
// register-allocation hints become no-ops and v_alignbyte a shift) and compared with the oracle bit for bit there.
// The including translation unit sets `#pragma clang fp contract(off)` first: every multiply and add rounds on its own,
// like the x86 build of the reference.
#pragma once
#include "track_math.h"
#include "pyr_addr.h"

#if defined(SVO_HOST_MATH_TEST) && !defined(__builtin_amdgcn_alignbyte)
#define __builtin_amdgcn_alignbyte(hi, lo, sel) \
  ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint64_t)(uint32_t)(lo)) >> (8u * ((sel)&3u))))
#endif

namespace svo_track {
using namespace svo_dev;

// byte i (compile-time) of the 100-byte template held in 25 dwords
#define PWB(i) ((int)((g[(i) >> 2] >> (8 * ((i)&3))) & 0xffu))

// ALIGN_OPAQUE_TEMPLATE(g) at the top of an iteration keeps whatever is derived from g[] after it inside the
// iteration (the compiler otherwise hoists all 192 template-derived floats out of the loop and needs more than 256
// registers: one wave per SIMD).  align2D hoists the 64 gradient pairs itself (they feed the packed accumulation) and
// re-derives the 64 template intensities from the bytes in flight: one v_cvt_f32_ubyte per pixel.  Emits no instruction.
#ifdef SVO_HOST_MATH_TEST
#define ALIGN_OPAQUE_TEMPLATE(g) ((void)0)
#else
#define ALIGN_OPAQUE_TEMPLATE(g)                                                                                   \
  asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),  \
               "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]));                                        \
  asm volatile("" : "+v"(g[13]), "+v"(g[14]), "+v"(g[15]), "+v"(g[16]), "+v"(g[17]), "+v"(g[18]), "+v"(g[19]),       \
               "+v"(g[20]), "+v"(g[21]), "+v"(g[22]), "+v"(g[23]), "+v"(g[24]))
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

// Everything the iteration loops of align2D / align1D carry from one iteration to the next (a phased run parks it
// between launches; f32 like the reference's locals, so a resumed trial continues bit for bit)
struct AlignState {
  float u, v, mean_diff, chi2, up0, up1;
};

// align2D, feature_alignment.cpp:149-277: iterations [it0, min(it1, n_iter)) of the loop.  Returns true when the trial
// has to go on (the range ended before n_iter, without convergence or failure); else `converged` is the verdict.
__device__ __forceinline__ bool align2d_lane(const uint8_t* __restrict__ img, int cols, int rows, int pitch,
                                             uint32_t g[25], int n_iter, int it0, int it1, AlignState& st,
                                             bool& converged, bool& wrote, int& n_eval) {
  converged = false;
  wrote = true;
  n_eval = 0;  // residual evaluations (9x9 windows read); dead code unless the caller stores it
  float u = st.u, v = st.v;
  // H = sum J J', J = (dx, dy, 1) (:166-181).  dx, dy are half-integers (byte differences / 2) and every
  // partial sum of the reference's float accumulation is a multiple of 0.25 below 2^22: no rounding ever
  // happens, so the sums can be formed in any order -- here as integers of the doubled gradients
  // (|2dx| <= 255, sum of squares <= 64 * 255^2 < 2^23).  Same bits.
  // Here: in f32 with explicit fma -- the doubled gradients are differences of byte values (exact), their products
  // are below 2^17 and the running sums below 2^23, so every fma is exact too: one v_cvt_f32_ubyte per template byte,
  // then subtractions and fused multiply-adds, no integer extraction.
  typedef float f2 __attribute__((ext_vector_type(2)));
  float H[9];
  // Round 6: the 64 gradient pairs are NOT kept (128 registers: two waves per SIMD).  The pixel loop re-forms them row by
  // row from the template rows it converts anyway (see there); here only their sums are needed.
  {
    float sxx = 0.f, sxy = 0.f, syy = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int c = (y + 1) * 10 + x + 1;
        const float gx2 = (float)PWB(c + 1) - (float)PWB(c - 1);    // 2 dx
        const float gy2 = (float)PWB(c + 10) - (float)PWB(c - 10);  // 2 dy
        sxx = __builtin_fmaf(gx2, gx2, sxx);
        sxy = __builtin_fmaf(gx2, gy2, sxy);
        syy = __builtin_fmaf(gy2, gy2, syy);
        sx += gx2;
        sy += gy2;
      }
    H[0] = 0.25f * sxx;
    H[1] = H[3] = 0.25f * sxy;
    H[4] = 0.25f * syy;
    H[2] = H[6] = 0.5f * sx;
    H[5] = H[7] = 0.5f * sy;
    H[8] = 64.f;
  }
  float Hinv[9];
  inv3f(H, Hinv);
  float mean_diff = st.mean_diff;
  const float min_update_squared = (float)(0.03 * 0.03);
  const int it_end = it1 < n_iter ? it1 : n_iter;
  bool left = false;  // the loop was left by a break
  for (int iter = it0; iter < it_end; ++iter) {
    ALIGN_OPAQUE_TEMPLATE(g);
    const int u_r = floor_int(u);
    const int v_r = floor_int(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) {
      left = true;
      break;
    }
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
    // (a branch per row measured 2.15 against 1.68 ms for findMatchDirect on 3.3 M trials)
    uint32_t win[9][3];
    svo_pyr::load_window12<9>(img, pitch, wxa, v_r - 4, win);
    cut_row9(win[0], wsel, P0);
    // The pixel loop in packed f32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per issue slot; every product and
    // sum is rounded on its own, in the reference's order, so the bits do not change).  Pixels x and x+4 of a row
    // form a pair for the interpolation and the residual -- the row is kept as Q[k] = {P[k], P[k+4]}, which serves
    // as left AND right neighbour pair -- and the two gradient products of a pixel form a pair for the accumulation:
    // {Jres0, Jres1} -= {res, res} * {dx, dy}, pixel after pixel in raster order (the third sum stays scalar).
    const f2 wTL2 = {wTL, wTL}, wTR2 = {wTR, wTR}, wBL2 = {wBL, wBL}, wBR2 = {wBR, wBR}, md2 = {mean_diff, mean_diff};
    // The gradients of a template row are differences of the rows the loop converts anyway: three rows of the 10 x 10
    // template as floats, rolling (Tm: above, Tc: the row under the window row, Tp: below), one conversion per byte as
    // before plus the two border bytes of a row, and 16 subtractions per row.  They are used DOUBLED (2 dx, 2 dy: exact byte
    // differences): scaling every term of the reference's sequential sum by two scales every partial sum by exactly two
    // (binary floating point, no overflow or underflow anywhere near), so {2 Jres0, 2 Jres1} come out with the reference's
    // mantissas and are halved once at the end -- the same bits, without the 128 registers of a gradient cache.
    f2 J01 = {0.f, 0.f};
    f2 Q0[5], Q1[5];
    float Tm[10], Tc[10], Tp[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      Tm[k] = (float)PWB(k);
      Tc[k] = (float)PWB(10 + k);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) Q0[k] = (f2){P0[k], P0[k + 4]};
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      cut_row9(win[y + 1], wsel, P1);
#pragma unroll
      for (int k = 0; k < 5; ++k) Q1[k] = (f2){P1[k], P1[k + 4]};
#pragma unroll
      for (int k = 0; k < 10; ++k) Tp[k] = (float)PWB((y + 2) * 10 + k);
      f2 res2[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const f2 sp = wTL2 * Q0[x] + wTR2 * Q0[x + 1] + wBL2 * Q1[x] + wBR2 * Q1[x + 1];
        const f2 ref2 = {Tc[x + 1], Tc[x + 5]};
        res2[x] = sp - ref2 + md2;
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) {  // raster order: x = 0..3 are the low halves, 4..7 the high halves
        const float res = (x < 4) ? res2[x].x : res2[x - 4].y;
        const f2 g2 = {Tc[x + 2] - Tc[x], Tp[x + 1] - Tm[x + 1]};  // {2 dx, 2 dy} of template pixel (x, y)
        J01 -= (f2){res, res} * g2;
        Jres2 -= res;
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) Q0[k] = Q1[k];
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        Tm[k] = Tc[k];
        Tc[k] = Tp[k];
      }
    }
    Jres0 = 0.5f * J01.x;
    Jres1 = 0.5f * J01.y;
    const float up0 = Hinv[0] * Jres0 + Hinv[1] * Jres1 + Hinv[2] * Jres2;
    const float up1 = Hinv[3] * Jres0 + Hinv[4] * Jres1 + Hinv[5] * Jres2;
    const float up2 = Hinv[6] * Jres0 + Hinv[7] * Jres1 + Hinv[8] * Jres2;
    u += up0;
    v += up1;
    mean_diff += up2;
    if (up0 * up0 + up1 * up1 < min_update_squared) {
      converged = true;
      left = true;
      break;
    }
  }
  st.u = u; st.v = v; st.mean_diff = mean_diff;
  return !left && it_end < n_iter;
}

// align1D, feature_alignment.cpp:30-147
__device__ __forceinline__ bool align1d_lane(const uint8_t* __restrict__ img, int cols, int rows, int pitch,
                                             uint32_t g[25], float dir0, float dir1, int n_iter, int it0, int it1,
                                             AlignState& st, double& h_inv, bool& converged, bool& wrote, int& n_eval) {
  converged = false;
  wrote = true;
  n_eval = 0;
  float u = st.u, v = st.v;
  float H[4] = {0, 0, 0, 0};
  // (the directional derivative of every template pixel, :53-56, is not kept -- 64 registers -- but re-formed in the pixel
  // loop from the template rows converted there, as align2D does with its gradients)
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
  float mean_diff = st.mean_diff;
  const float min_update_squared = (float)(0.03 * 0.03);
  float chi2 = st.chi2;
  float up0 = st.up0, up1 = st.up1;
  const int it_end = it1 < n_iter ? it1 : n_iter;
  bool left = false;
  for (int iter = it0; iter < it_end; ++iter) {
    ALIGN_OPAQUE_TEMPLATE(g);
    const int u_r = floor_int(u);
    const int v_r = floor_int(v);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) {
      left = true;
      break;
    }
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
    // (a branch per row measured 2.15 against 1.68 ms for findMatchDirect on 3.3 M trials)
    uint32_t win[9][3];
    svo_pyr::load_window12<9>(img, pitch, wxa, v_r - 4, win);
    cut_row9(win[0], wsel, P0);
    // three rows of the template as floats, rolling (above / under the window row / below): the byte differences of
    // :53-56 are differences of these (exact), and 0.5 * (double)s rounded to float is 0.5f * s
    float Tm[10], Tc[10], Tp[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      Tm[k] = (float)PWB(k);
      Tc[k] = (float)PWB(10 + k);
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      cut_row9(win[y + 1], wsel, P1);
#pragma unroll
      for (int k = 0; k < 10; ++k) Tp[k] = (float)PWB((y + 2) * 10 + k);
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const float search_pixel = wTL * P0[x] + wTR * P0[x + 1] + wBL * P1[x] + wBR * P1[x + 1];
        const float res = search_pixel - Tc[x + 1] + mean_diff;
        const float s = dir0 * (Tc[x + 2] - Tc[x]) + dir1 * (Tp[x + 1] - Tm[x + 1]);
        Jres0 -= res * (0.5f * s);
        Jres1 -= res;
        new_chi2 += res * res;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) P0[k] = P1[k];
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        Tm[k] = Tc[k];
        Tc[k] = Tp[k];
      }
    }
    if (iter > 0 && new_chi2 > chi2) {
      u -= up0;  // sic (:116-117)
      v -= up1;
      left = true;
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
      left = true;
      break;
    }
  }
  st.u = u; st.v = v; st.mean_diff = mean_diff; st.chi2 = chi2; st.up0 = up0; st.up1 = up1;
  return !left && it_end < n_iter;
}

}  // namespace svo_track
