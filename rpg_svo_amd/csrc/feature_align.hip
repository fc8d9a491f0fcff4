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

// ALIGN_OPAQUE_TEMPLATE(g) at the top of an iteration keeps whatever is derived from g[] after it inside the
// iteration (the compiler otherwise hoists all 192 template-derived floats out of the loop and needs more than 256
// registers: one wave per SIMD).  align2D hoists the 64 gradient pairs itself (they feed the packed accumulation) and
// re-derives the 64 template intensities from the bytes in flight: one v_cvt_f32_ubyte per pixel.  Emits no instruction.
#define ALIGN_OPAQUE_TEMPLATE(g)                                                                                   \
  asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),  \
               "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]));                                        \
  asm volatile("" : "+v"(g[13]), "+v"(g[14]), "+v"(g[15]), "+v"(g[16]), "+v"(g[17]), "+v"(g[18]), "+v"(g[19]),       \
               "+v"(g[20]), "+v"(g[21]), "+v"(g[22]), "+v"(g[23]), "+v"(g[24]))

// the same for the 64 gradient words of the ALIGN_G_F16 build: their conversions to f32 stay inside the iteration
#define ALIGN_OPAQUE_G(G)                                         \
  asm volatile("" : "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3]), "+v"(G[4]), "+v"(G[5]), "+v"(G[6]), "+v"(G[7]), "+v"(G[8]), "+v"(G[9]), "+v"(G[10]), "+v"(G[11]), "+v"(G[12]), "+v"(G[13]), "+v"(G[14]), "+v"(G[15]));  \
  asm volatile("" : "+v"(G[16]), "+v"(G[17]), "+v"(G[18]), "+v"(G[19]), "+v"(G[20]), "+v"(G[21]), "+v"(G[22]), "+v"(G[23]), "+v"(G[24]), "+v"(G[25]), "+v"(G[26]), "+v"(G[27]), "+v"(G[28]), "+v"(G[29]), "+v"(G[30]), "+v"(G[31]));  \
  asm volatile("" : "+v"(G[32]), "+v"(G[33]), "+v"(G[34]), "+v"(G[35]), "+v"(G[36]), "+v"(G[37]), "+v"(G[38]), "+v"(G[39]), "+v"(G[40]), "+v"(G[41]), "+v"(G[42]), "+v"(G[43]), "+v"(G[44]), "+v"(G[45]), "+v"(G[46]), "+v"(G[47]));  \
  asm volatile("" : "+v"(G[48]), "+v"(G[49]), "+v"(G[50]), "+v"(G[51]), "+v"(G[52]), "+v"(G[53]), "+v"(G[54]), "+v"(G[55]), "+v"(G[56]), "+v"(G[57]), "+v"(G[58]), "+v"(G[59]), "+v"(G[60]), "+v"(G[61]), "+v"(G[62]), "+v"(G[63]))
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
#ifdef ALIGN_G_F16
  // (round-5 queue, UNMEASURED: the gradients are half-integers of magnitude <= 127.5 -- EXACT in f16 -- so the 64 {dx, dy}
  // pairs fit 64 registers instead of 128 and the kernel three waves per SIMD instead of two (it waits for an iteration's
  // window fetch 48 % of its wave cycles); two conversions per pixel and iteration bring them back, same values.)
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  uint32_t G[64];  // {dx, dy} as two f16
#else
  f2 G[64];  // {dx, dy} of every template pixel (:176-181), formed once
#endif
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
#ifdef ALIGN_G_F16
        G[8 * y + x] = __builtin_bit_cast(uint32_t, (h2){(_Float16)(0.5f * gx2), (_Float16)(0.5f * gy2)});
#else
        G[8 * y + x] = (f2){0.5f * gx2, 0.5f * gy2};  // == 0.5f * (float)(int difference): the difference is exact either way
#endif
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
#ifdef ALIGN_G_F16
    ALIGN_OPAQUE_G(G);
#endif
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
#ifndef ALIGN_ROW_LOADS  // all nine rows fetched up front, ONE three-way branch on the tile position for the window
    // (a branch per row measured 2.15 against 1.68 ms for findMatchDirect on 3.3 M trials)
    uint32_t win[9][3];
    svo_pyr::load_window12<9>(img, pitch, wxa, v_r - 4, win);
    cut_row9(win[0], wsel, P0);
#else
    load_row9(img, svo_pyr::row_off(v_r - 4, pitch), wxa, wsel, P0);
#endif
#ifndef ALIGN_NO_PACKED
    // The pixel loop in packed f32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per issue slot; every product and
    // sum is rounded on its own, in the reference's order, so the bits do not change).  Pixels x and x+4 of a row
    // form a pair for the interpolation and the residual -- the row is kept as Q[k] = {P[k], P[k+4]}, which serves
    // as left AND right neighbour pair -- and the two gradient products of a pixel form a pair for the accumulation:
    // {Jres0, Jres1} -= {res, res} * {dx, dy}, pixel after pixel in raster order (the third sum stays scalar).
    const f2 wTL2 = {wTL, wTL}, wTR2 = {wTR, wTR}, wBL2 = {wBL, wBL}, wBR2 = {wBR, wBR}, md2 = {mean_diff, mean_diff};
    f2 J01 = {0.f, 0.f};
    f2 Q0[5], Q1[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) Q0[k] = (f2){P0[k], P0[k + 4]};
#pragma unroll
    for (int y = 0; y < 8; ++y) {
#ifndef ALIGN_ROW_LOADS
      cut_row9(win[y + 1], wsel, P1);
#else
      load_row9(img, svo_pyr::row_off(v_r - 3 + y, pitch), wxa, wsel, P1);
#endif
#pragma unroll
      for (int k = 0; k < 5; ++k) Q1[k] = (f2){P1[k], P1[k + 4]};
      f2 res2[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = (y + 1) * 10 + x + 1;
        const f2 sp = wTL2 * Q0[x] + wTR2 * Q0[x + 1] + wBL2 * Q1[x] + wBR2 * Q1[x + 1];
        const f2 ref2 = {(float)PWB(c), (float)PWB(c + 4)};
        res2[x] = sp - ref2 + md2;
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) {  // raster order: x = 0..3 are the low halves, 4..7 the high halves
        const float res = (x < 4) ? res2[x].x : res2[x - 4].y;
#ifdef ALIGN_G_F16
        {
          const h2 gh = __builtin_bit_cast(h2, G[8 * y + x]);
          J01 -= (f2){res, res} * (f2){(float)gh.x, (float)gh.y};
        }
#else
        J01 -= (f2){res, res} * G[8 * y + x];
#endif
        Jres2 -= res;
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) Q0[k] = Q1[k];
    }
    Jres0 = J01.x;
    Jres1 = J01.y;
#else
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
#endif
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
  float Jd[64];  // the directional derivative of every template pixel (:53-56), formed once
#pragma unroll
  for (int y = 0; y < 8; ++y)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int c = (y + 1) * 10 + x + 1;
      // J[0] = 0.5*(dir[0]*(it[1]-it[-1]) + dir[1]*(it[ref_step]-it[-ref_step]))  (double 0.5 * float)
      const float s = dir0 * (float)(PWB(c + 1) - PWB(c - 1)) + dir1 * (float)(PWB(c + 10) - PWB(c - 10));
      const float J0 = (float)(0.5 * (double)s);
      Jd[8 * y + x] = J0;
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
        Jres0 -= res * Jd[8 * y + x];
        Jres1 -= res;
        new_chi2 += res * res;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) P0[k] = P1[k];
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

constexpr int ALIGN_BLOCK = 64;

// Two waves per SIMD (<= 256 registers): with the bare __launch_bounds__(64) the compiler took 256 VGPRs plus 15-20
// AGPRs, i.e. ONE wave per SIMD, and nothing hid the round trip of an iteration's window fetch.
#ifndef ALIGN_MINW
#ifdef ALIGN_G_F16
#define ALIGN_MINW 3
#else
#define ALIGN_MINW 2
#endif
#endif
template <bool COUNT>
__global__ void __launch_bounds__(ALIGN_BLOCK, ALIGN_MINW) align_kernel(const AlignArgs a) {
  // which trial: lane order in the first launch of a run, the queues filled by the previous launch afterwards
  // (workgroup b drains queue b % ALIGN_NQ, 64 entries at a time)
  int t;
  // ALIGN_TEMPLATE_LDS (off: measured SLOWER twice, rounds 2 and 3 -- 12.25 against 11.90 ms for the full-track step).
  // In list order the 64 templates of a workgroup are 6400 contiguous bytes; read per lane they are 25 dword gathers
  // with a 100-byte lane stride, read as the block they are (seven coalesced 16-byte loads per lane) and handed out
  // through LDS they are far fewer cache-line look-ups -- but the hand-over makes every lane wait for the whole block
  // before its set-up arithmetic can start, and the gathers of neighbouring words hit the same line in L1 anyway.
#ifdef ALIGN_TEMPLATE_LDS
  __shared__ __attribute__((aligned(16))) uint32_t s_tpl[ALIGN_BLOCK * 25];
  bool tpl_in_lds = false;
  if (!a.queue_in) {
    const long long first_t = (long long)blockIdx.x * ALIGN_BLOCK;  // wave-uniform: one wave per workgroup
    const long long left = (long long)a.M - first_t;
    const int n_dw = 25 * (int)(left < ALIGN_BLOCK ? left : ALIGN_BLOCK);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.pwb + (size_t)first_t * 100);
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      tpl_in_lds = true;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int q = 4 * ((int)threadIdx.x + ALIGN_BLOCK * k);  // first dword of this lane's quad
        if (q + 3 < n_dw) *reinterpret_cast<uint4*>(&s_tpl[q]) = *reinterpret_cast<const uint4*>(src + q);
        else
          for (int j = q; j < n_dw && j < q + 4; ++j) s_tpl[j] = src[j];
      }
      // one wave: DS operations execute in order; keep the compiler from moving the reads below above the writes
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
#endif
  if (a.queue_in) {
    const int q = blockIdx.x % ALIGN_NQ, i = (blockIdx.x / ALIGN_NQ) * ALIGN_BLOCK + threadIdx.x;
    if (i >= a.n_in[q]) return;
    t = a.queue_in[(size_t)q * a.queue_cap + i];
  } else {
    t = blockIdx.x * ALIGN_BLOCK + threadIdx.x;
    if (t >= (a.M_dev ? min(*a.M_dev, a.M) : a.M)) return;
  }
  const bool first = a.it0 == 0;
  if (first && a.active && !a.active[t]) {
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
    const uint32_t* gp = tpl_in_lds ? &s_tpl[threadIdx.x * 25] : reinterpret_cast<const uint32_t*>(a.pwb + (size_t)t * 100);
#else
    const uint32_t* gp = reinterpret_cast<const uint32_t*>(a.pwb + (size_t)t * 100);
#endif
#pragma unroll
    for (int k = 0; k < 25; ++k) g[k] = gp[k];
  }
  AlignState st;
  if (first) {
    st.u = (float)a.px_in[2 * t];
    st.v = (float)a.px_in[2 * t + 1];
    st.mean_diff = 0.f; st.chi2 = 0.f; st.up0 = 0.f; st.up1 = 0.f;
  } else {
    const float* sp = a.state + 6 * (size_t)t;
    st.u = sp[0]; st.v = sp[1]; st.mean_diff = sp[2]; st.chi2 = sp[3]; st.up0 = sp[4]; st.up1 = sp[5];
  }
  bool wrote = true, ok = false, more;
  int n_eval = 0;
  const bool one_d = a.use_1d && a.use_1d[t];
  if (one_d) {
    double h_inv = 0;
    more = align1d_lane(img, cols, rows, pitch, g, a.dir[2 * t], a.dir[2 * t + 1], a.n_iter, a.it0, a.it1, st, h_inv, ok, wrote, n_eval);
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
  a.ok[t] = ok ? 1 : 0;
  double ou = wrote ? (double)st.u : a.px_in[2 * t];
  double ov = wrote ? (double)st.v : a.px_in[2 * t + 1];
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
constexpr int ALIGN_PHASE_MIN_M = 1 << 16;  // below this the extra launches cost more than the idle lanes
#ifndef ALIGN_PHASE_ITERS
#define ALIGN_PHASE_ITERS 3
#endif

static int phase_queue_cap(int M) { return (M + ALIGN_NQ - 1) / ALIGN_NQ + 2 * ALIGN_BLOCK; }

size_t align_phase_workspace_bytes(int M) {
  if (M < ALIGN_PHASE_MIN_M) return 0;
  const size_t cap = (size_t)phase_queue_cap(M);
  return 2 * Carver::round(ALIGN_NQ * cap * sizeof(int32_t)) + Carver::round(2 * ALIGN_NQ * sizeof(int32_t)) +
         Carver::round((size_t)M * 6 * sizeof(float));
}

static int launch_one(const AlignArgs& a, int n_blocks, hipStream_t s) {
  const dim3 grid(n_blocks), blk(ALIGN_BLOCK);
  if (a.iters) hipLaunchKernelGGL(align_kernel<true>, grid, blk, 0, s, a);  // instrumented: also counts evaluations
  else hipLaunchKernelGGL(align_kernel<false>, grid, blk, 0, s, a);
  return check_launch();
}

int launch_align(const AlignArgs& a0, hipStream_t s, void* d_phase_ws, size_t phase_ws_bytes) {
  if (a0.M <= 0) return SVO_HIP_OK;
  const int all_blocks = (a0.M + ALIGN_BLOCK - 1) / ALIGN_BLOCK;
  const size_t need = align_phase_workspace_bytes(a0.M);
  if (!d_phase_ws || need == 0 || phase_ws_bytes < need || a0.n_iter <= ALIGN_PHASE_ITERS) return launch_one(a0, all_blocks, s);
  Carver c(d_phase_ws, phase_ws_bytes);
  const int cap = phase_queue_cap(a0.M);
  int32_t* queue[2] = {c.take<int32_t>((size_t)ALIGN_NQ * cap), c.take<int32_t>((size_t)ALIGN_NQ * cap)};
  int32_t* count = c.take<int32_t>(2 * ALIGN_NQ);
  float* state = c.take<float>((size_t)a0.M * 6);
  if (!c.ok) return launch_one(a0, all_blocks, s);
  SVO_HIP_TRY(hipMemsetAsync(count, 0, 2 * ALIGN_NQ * sizeof(int32_t), s));
  // a queue holds the survivors of the workgroups b % ALIGN_NQ == q: at most cap entries; every launch after the first
  // is sized for full queues (workgroups past a queue's end leave at once)
  const int queue_blocks = ALIGN_NQ * ((cap + ALIGN_BLOCK - 1) / ALIGN_BLOCK);
  AlignArgs a = a0;
  a.state = state;
  a.queue_cap = cap;
  int rc = SVO_HIP_OK;
  for (int phase = 0; phase < 3 && rc == SVO_HIP_OK; ++phase) {
    a.it0 = phase * ALIGN_PHASE_ITERS;
    a.it1 = phase == 2 ? (1 << 30) : (phase + 1) * ALIGN_PHASE_ITERS;
    a.queue_in = phase == 0 ? nullptr : queue[(phase - 1) & 1];
    a.n_in = phase == 0 ? nullptr : count + ((phase - 1) & 1) * ALIGN_NQ;
    a.queue_out = phase == 2 ? nullptr : queue[phase & 1];
    a.n_out = phase == 2 ? nullptr : count + (phase & 1) * ALIGN_NQ;
    rc = launch_one(a, phase == 0 ? all_blocks : queue_blocks, s);
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
