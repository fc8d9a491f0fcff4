// warp_sample.h -- the sample arithmetic of warp::warpAffine (matcher.cpp:72-105) as the lanes of warp_group.h run it: a
// sample of a trial's 10 x 10 patch read from a copy of the source region laid out in rows of 48 bytes (reg_o[48 yi + xi]
// is pixel (xi, yi) of the level).  Device functions of matcher.hip; also
// compiled for the CPU by the test suite (SVO_HOST_MATH_TEST, see device_math.h) and compared with the oracle bit for bit.
// The including translation unit sets `#pragma clang fp contract(off)`: every product and sum rounds on its own.
#pragma once
#include "track_math.h"

#ifdef SVO_HOST_MATH_TEST
#define __builtin_amdgcn_fractf(x) ((x)-floorf(x))  // (exact for x >= 0: the only arguments it sees here)
#define __mul24(a, b) ((a) * (b))
#endif

namespace svo_track {

// CHECK: the per-sample bounds test of vk::interpolateMat_8u (a sample outside the image is 0); without it the caller
// has shown that every sample lies inside (the box of the four corner samples does).
// floor and fraction of a coordinate are one instruction each on the device (v_cvt_flr_i32_f32, v_fract_f32: u - floor(u)
// is exact for u >= 0, so the fraction has the same bits), the sample's address one 24-bit multiply and one add.

// The four weights of vk::interpolateMat_8u and their blend (every product and sum rounded on its own, in this order).
__device__ __forceinline__ float warp_blend(const float sx, const float sy, const float p00, const float p10, const float p01,
                                            const float p11) {
  const float w00 = (1.0f - sx) * (1.0f - sy);
  const float w01 = (1.0f - sx) * sy;
  const float w10 = sx * (1.0f - sy);
  const float w11 = 1.0f - w00 - w01 - w10;
  return w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11;
}

// The arithmetic with the two coordinates, two of the weights and the four products as the halves of v_pk_*_f32
// operations (each half rounds on its own, like the scalar instruction): 26 VALU instructions per sample where the scalar
// form took 31.5 (round 6: update_seeds 7.36 -> 7.12 ms per 13.2 M seeds, profiles/r06k_*; an earlier pairing of two output
// ROWS of a lane cost more in register moves than it saved).
typedef float wf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float warp_blend_pk(const float sx, const float sy, const float p00, const float p10, const float p01,
                                               const float p11) {
  const wf2 a = wf2{1.0f - sx, sx}, b = wf2{1.0f - sy, sy};
  const wf2 w0 = wf2{a.x, a.x} * b;                        // {w00, w01}
  const float w10 = a.y * b.x;
  const float w11 = 1.0f - w0.x - w0.y - w10;
  const wf2 pa = w0 * wf2{p00, p01};                       // {w00 p00, w01 p01}
  const wf2 pb = wf2{w10, w11} * wf2{p10, p11};            // {w10 p10, w11 p11}
  return ((pa.x + pa.y) + pb.x) + pb.y;
}

// Sample (x, y) of the 10 x 10 patch: px_patch = (x - 5, y - 5) * 2^search_level, px = A_ref_cur * px_patch + px_ref_pyr
// (matcher.cpp:90-93, Eigen's 2 x 2 product: (a * b + c * d) + e), read from a copy of the source region laid out in rows of
// 48 bytes: reg_o[48 yi + xi] is pixel (xi, yi) of the level.
template <bool CHECK>
__device__ __forceinline__ uint8_t warp_sample(const float Ax, const float Ay, const float Az, const float Aw, const float pyrx,
                                               const float pyry, const float sc, const int x, const int y, const int cols,
                                               const int rows, const int xlo, const int ylo, const uint8_t* const reg_o) {
  float pp0 = (float)(x - 5), pp1 = (float)(y - 5);
  pp0 *= sc;
  pp1 *= sc;
  // px = (A.col(0) * pp0 + A.col(1) * pp1) + px_ref_pyr, both components at once
  const wf2 pxy = (wf2{Ax, Az} * wf2{pp0, pp0} + wf2{Ay, Aw} * wf2{pp1, pp1}) + wf2{pyrx, pyry};
  const float px0 = pxy.x, px1 = pxy.y;
  const bool in = !CHECK || !(px0 < 0 || px1 < 0 || px0 >= (float)(cols - 1) || px1 >= (float)(rows - 1));
  // vk::interpolateMat_8u (a sample outside the image is 0)
  const float u = in ? px0 : (float)xlo, v = in ? px1 : (float)ylo;
  const int xi = svo_dev::floor_to_int(u), yi = svo_dev::floor_to_int(v);
  const float sx = __builtin_amdgcn_fractf(u), sy = __builtin_amdgcn_fractf(v);
  const uint8_t* q = reg_o + (__mul24(yi, 48) + xi);
  // (Four 1-byte reads.  The two pixels of a row as ONE 2-byte read at any byte address -- the DS unit takes it, the
  // compiler emits ds_read_u16 -- halves the LDS instructions of a sample and measured update_seeds 6.95 -> 9.17 ms,
  // find_match_direct 1.60 -> 2.31: an unaligned DS access stalls the unit for far longer than the read it saves,
  // profiles/r06ac_*.)
  const float val = warp_blend_pk(sx, sy, (float)q[0], (float)q[1], (float)q[48], (float)q[49]);
  return in ? (uint8_t)val : (uint8_t)0;
}

// One output COLUMN x of a trial's patch, ten samples down the rows.
template <bool CHECK>
__device__ __forceinline__ void warp_column(const float Ax, const float Ay, const float Az, const float Aw, const float pyrx,
                                            const float pyry, const float sc, const int x, const int cols, const int rows,
                                            const int xlo, const int ylo, const uint8_t* const reg_o, uint8_t out[10]) {
#pragma unroll
  for (int y = 0; y < 10; ++y) out[y] = warp_sample<CHECK>(Ax, Ay, Az, Aw, pyrx, pyry, sc, x, y, cols, rows, xlo, ylo, reg_o);
}


}  // namespace svo_track
