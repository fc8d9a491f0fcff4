// align_wave.h -- K3 for SMALL batches: align2D (feature_alignment.cpp:149-277) and align1D (:30-147) of one trial by ONE
// WAVE, lane = template pixel (x = lane & 7, y = lane >> 3).
//
// align_lanes.h gives a trial to one lane: 64 trials per wave, the right mapping for a replay batch of millions, and the
// worst one for a single camera frame -- its ~130 trials are three waves that each run ~850 dependent instructions plus a
// window fetch per iteration, ten iterations for the slowest trial: 27 us of a 300 us frame, twice (the reprojector's
// trials and the depth filter's).  Here the 64 pixels of an iteration are 64 lanes.  What keeps the reference's bits:
//
//   * per pixel everything is the reference's expression, rounded operation by operation (contraction is off);
//   * the reference ACCUMULATES Jres (and chi2; align1D also H) over the 64 pixels in raster order in f32, and f32
//     addition does not associate: the 64 products go to LDS and every lane adds them up in that order (64 dependent
//     subtractions per accumulator -- two accumulators as a packed pair, the third beside it -- read as broadcast
//     float4): the sums are the reference's, and every lane holds them, so the update and the loop's decisions need
//     no broadcast;
//   * align2D's H is a sum of exact terms (see align_lanes.h): a butterfly over the wave gives the same bits.
//
// An iteration is then ~40 instructions of pixel work, 4 byte loads that hit L1 after the first iteration, and a chain of
// 128 additions, instead of ~850 instructions and a 9 x 12-byte window per lane.  Measured on 330 trials
// (scripts/align_small_bench.py, profiles/r05m_align_wave_variants.txt): 15.1 us with three scalar chains, 13.1 us with
// the packed pair; the window kept in LDS instead of four L1 hits per iteration: 14.2 us alone, 14.0 us with the packed
// pair -- not kept.  The floor of an iteration is the 64-deep dependent chain itself.
// The including translation unit sets `#pragma clang fp contract(off)` first.
#pragma once
#include "align_lanes.h"
#include "wave_reduce.h"

namespace svo_track {

// three ordered accumulations over the 64 lanes' values: r[j] = (((init - or + v_j[0]) ...) v_j[63]) in lane order.
// s: the wave's [3][64] floats of LDS.  SUB0/1/2: subtract (Jres -= x) or add (chi2 += x).
// values 0 and 1 always share their sign (Jres0 / Jres1, H0 / H1): they are summed as a PAIR (v_pk_add_f32: two IEEE
// additions per instruction, each rounded on its own), the third on its own: 128 dependent instructions instead of 192
template <bool SUB0, bool SUB1, bool SUB2>
__device__ __forceinline__ void ordered_sums3(float* s, int lane, float v0, float v1, float v2, float& r0, float& r1, float& r2) {
  static_assert(SUB0 == SUB1, "the pair shares its sign");
  typedef float f2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<f2*>(s + 2 * lane) = (f2){v0, v1};
  s[128 + lane] = v2;
  SVO_WAVE_LDS_HANDOVER();
  const float4* q01 = reinterpret_cast<const float4*>(s);
  const float4* q2 = reinterpret_cast<const float4*>(s + 128);
  f2 a01 = {0.f, 0.f};
  float a2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float4 A = q01[2 * k], B = q01[2 * k + 1], C = q2[k];
    const f2 p0 = {A.x, A.y}, p1 = {A.z, A.w}, p2 = {B.x, B.y}, p3 = {B.z, B.w};
    a01 = SUB0 ? a01 - p0 : a01 + p0; a2 = SUB2 ? a2 - C.x : a2 + C.x;
    a01 = SUB0 ? a01 - p1 : a01 + p1; a2 = SUB2 ? a2 - C.y : a2 + C.y;
    a01 = SUB0 ? a01 - p2 : a01 + p2; a2 = SUB2 ? a2 - C.z : a2 + C.z;
    a01 = SUB0 ? a01 - p3 : a01 + p3; a2 = SUB2 ? a2 - C.w : a2 + C.w;
  }
  SVO_WAVE_LDS_HANDOVER();  // (the next iteration's stores stay behind these reads)
  r0 = a01.x; r1 = a01.y; r2 = a2;
}

// the lane's four neighbours (x0, y0), (x0+1, y0), (x0, y0+1), (x0+1, y0+1) of a tiled level as floats
__device__ __forceinline__ void load_quad(const uint8_t* __restrict__ img, int pitch, int x0, int y0, float& p00, float& p01,
                                          float& p10, float& p11) {
  const uint32_t r0 = svo_pyr::row_off(y0, pitch), r1 = svo_pyr::row_off(y0 + 1, pitch);
  const uint32_t c0 = svo_pyr::col_off(x0), c1 = svo_pyr::col_off(x0 + 1);
  const uint8_t b00 = img[r0 + c0], b01 = img[r0 + c1], b10 = img[r1 + c0], b11 = img[r1 + c1];
  p00 = (float)b00; p01 = (float)b01; p10 = (float)b10; p11 = (float)b11;
}


// the value every lane holds, as a wave-uniform one (lane 0's copy: scalar registers, scalar branches)
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

// align2D by a wave.  tpl: the trial's 100 template bytes (Matcher::patch_with_border_).  s: [3][64] floats of LDS.
__device__ __forceinline__ void align2d_wave(const uint8_t* __restrict__ img, int cols, int rows, int pitch,
                                             const uint8_t* __restrict__ tpl, int n_iter, int lane, float* s, AlignState& st,
                                             bool& converged, bool& wrote, int& n_eval) {
  converged = false;
  wrote = true;
  n_eval = 0;
  const int x = lane & 7, y = lane >> 3;
  const int c = (y + 1) * 10 + x + 1;
  const uint8_t bc = tpl[c], bl = tpl[c - 1], br = tpl[c + 1], bu = tpl[c - 10], bd = tpl[c + 10];
  const float ref = (float)bc;
  const float gx2 = (float)br - (float)bl;  // 2 dx
  const float gy2 = (float)bd - (float)bu;  // 2 dy
  const float dx = 0.5f * gx2, dy = 0.5f * gy2;
  // H = sum J J' (:166-181): sums of exact terms below 2^23 (align_lanes.h) -- any order gives the reference's bits
  float H[9];
  {
    const float part[8] = {gx2 * gx2, gx2 * gy2, gy2 * gy2, gx2, gy2, 0.f, 0.f, 0.f};
    const float tot = svo_dev::wave_reduce8(part, lane);  // lanes 8g .. 8g+7 hold the total of part[g]
    const float sxx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 0));
    const float sxy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 8));
    const float syy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 16));
    const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 24));
    const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 32));
    H[0] = 0.25f * sxx;
    H[1] = H[3] = 0.25f * sxy;
    H[4] = 0.25f * syy;
    H[2] = H[6] = 0.5f * sx;
    H[5] = H[7] = 0.5f * sy;
    H[8] = 64.f;
  }
  float Hinv[9];
  inv3f(H, Hinv);
  float u = st.u, v = st.v, mean_diff = st.mean_diff;
  const float min_update_squared = (float)(0.03 * 0.03);
  for (int iter = 0; iter < n_iter; ++iter) {
    const int u_r = uniform_int(floor_int(u));
    const int v_r = uniform_int(floor_int(v));
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (uniform_int((isnan(u) || isnan(v)) ? 1 : 0)) {  // unreachable after the bounds test, kept for the record (:209)
      wrote = false;
      return;
    }
    ++n_eval;
    const float subpix_x = u - (float)u_r;
    const float subpix_y = v - (float)v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    float p00, p01, p10, p11;
    load_quad(img, pitch, u_r - 4 + x, v_r - 4 + y, p00, p01, p10, p11);
    const float search_pixel = wTL * p00 + wTR * p01 + wBL * p10 + wBR * p11;
    const float res = search_pixel - ref + mean_diff;
    float Jres0, Jres1, Jres2;
    ordered_sums3<true, true, true>(s, lane, res * dx, res * dy, res, Jres0, Jres1, Jres2);
    const float up0 = Hinv[0] * Jres0 + Hinv[1] * Jres1 + Hinv[2] * Jres2;
    const float up1 = Hinv[3] * Jres0 + Hinv[4] * Jres1 + Hinv[5] * Jres2;
    const float up2 = Hinv[6] * Jres0 + Hinv[7] * Jres1 + Hinv[8] * Jres2;
    u += up0;
    v += up1;
    mean_diff += up2;
    if (uniform_int(up0 * up0 + up1 * up1 < min_update_squared ? 1 : 0)) {
      converged = true;
      break;
    }
  }
  st.u = u; st.v = v; st.mean_diff = mean_diff;
}

// align1D by a wave
__device__ __forceinline__ void align1d_wave(const uint8_t* __restrict__ img, int cols, int rows, int pitch,
                                             const uint8_t* __restrict__ tpl, float dir0, float dir1, int n_iter, int lane,
                                             float* s, AlignState& st, double& h_inv, bool& converged, bool& wrote, int& n_eval) {
  converged = false;
  wrote = true;
  n_eval = 0;
  const int x = lane & 7, y = lane >> 3;
  const int c = (y + 1) * 10 + x + 1;
  const uint8_t bc = tpl[c], bl = tpl[c - 1], br = tpl[c + 1], bu = tpl[c - 10], bd = tpl[c + 10];
  const float ref = (float)bc;
  // J[0] = 0.5*(dir[0]*(it[1]-it[-1]) + dir[1]*(it[ref_step]-it[-ref_step]))  (double 0.5 * float; :53-56)
  const float sd = dir0 * (float)((int)br - (int)bl) + dir1 * (float)((int)bd - (int)bu);
  const float J0 = (float)(0.5 * (double)sd);
  // H += J J' with J = (J0, 1): f32 sums of inexact terms, in raster order like the reference's loop
  float H[4];
  {
    float h0, h1, h3;
    ordered_sums3<false, false, false>(s, lane, J0 * J0, J0 * 1.f, 1.f * 1.f, h0, h1, h3);
    H[0] = h0; H[1] = h1; H[2] = h1; H[3] = h3;
  }
  h_inv = 1.0 / (double)H[0] * 8 * 8;
  float Hinv[4];
  inv2<float>(H, Hinv);
  float u = st.u, v = st.v, mean_diff = st.mean_diff;
  const float min_update_squared = (float)(0.03 * 0.03);
  float chi2 = st.chi2;
  float up0 = st.up0, up1 = st.up1;
  for (int iter = 0; iter < n_iter; ++iter) {
    const int u_r = uniform_int(floor_int(u));
    const int v_r = uniform_int(floor_int(v));
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (uniform_int((isnan(u) || isnan(v)) ? 1 : 0)) {
      wrote = false;
      return;
    }
    ++n_eval;
    const float subpix_x = u - (float)u_r;
    const float subpix_y = v - (float)v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    float p00, p01, p10, p11;
    load_quad(img, pitch, u_r - 4 + x, v_r - 4 + y, p00, p01, p10, p11);
    const float search_pixel = wTL * p00 + wTR * p01 + wBL * p10 + wBR * p11;
    const float res = search_pixel - ref + mean_diff;
    float Jres0, Jres1, new_chi2;
    ordered_sums3<true, true, false>(s, lane, res * J0, res, res * res, Jres0, Jres1, new_chi2);
    if (uniform_int((iter > 0 && new_chi2 > chi2) ? 1 : 0)) {
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
    if (uniform_int(up0 * up0 + up1 * up1 < min_update_squared ? 1 : 0)) {
      converged = true;
      break;
    }
  }
  st.u = u; st.v = v; st.mean_diff = mean_diff; st.chi2 = chi2; st.up0 = up0; st.up1 = up1;
}

}  // namespace svo_track
