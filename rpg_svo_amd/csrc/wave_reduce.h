// wave_reduce.h -- multi-value wave64 reductions without LDS traffic (gfx950).
//
// Summing K independent values over the 64 lanes with a butterfly costs
// 6*K cross-lane ops.  Here the first three butterfly levels "transpose"
// instead: at each level a lane gives away half of its values and receives the
// partner's copy of the half it keeps, so 8 values cost 4+2+1 exchanges plus
// 3 in-group steps.  Exchanges use v_permlane32_swap / v_permlane16_swap
// (gfx950) and DPP row operations -- no ds_bpermute, no LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace svo_dev {

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int DPP_ROW_ROR8 = 0x128;        // lane i <-> i^8 inside each 16 lanes

// Sums v[0..7] over the wave.  On return every lane of the 8-lane group
// g = lane>>3 holds the wave total of v[g].
__device__ __forceinline__ float wave_reduce8(const float v[8], int lane) {
  float r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // lanes 32..63 of the first operand <-> lanes 0..31 of the second
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 4]), false, false);
    r[k] = __uint_as_float(s[0]) + __uint_as_float(s[1]);  // lanes<32: v[k], lanes>=32: v[k+4]
  }
  float q[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    // odd 16-lane rows of the first operand <-> even rows of the second
    auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[k]), __float_as_uint(r[k + 2]), false, false);
    q[k] = __uint_as_float(s[0]) + __uint_as_float(s[1]);  // row parity 0: r[k], parity 1: r[k+2]
  }
  const bool hi = (lane & 8) != 0;
  const float mine = hi ? q[1] : q[0];
  const float send = hi ? q[0] : q[1];
  float t = mine + dpp_f32<DPP_ROW_ROR8>(send);
  t += dpp_f32<DPP_QUAD_XOR1>(t);
  t += dpp_f32<DPP_QUAD_XOR2>(t);
  t += dpp_f32<DPP_ROW_HALF_MIRROR>(t);
  return t;
}

}  // namespace svo_dev
