// wave_reduce.h -- multi-value wave64 reductions without LDS traffic (gfx950).
//
// Summing K independent values over the 64 lanes with a butterfly costs
// 6*K cross-lane ops.  Here the first three butterfly levels "transpose"
// instead: at each level a lane gives away half of its values and receives the
// partner's copy of the half it keeps, so 8 values cost 4+2+1 exchanges plus
// 3 in-group steps.  Exchanges use v_permlane32_swap / v_permlane16_swap
// (gfx950) and DPP row operations -- no ds_bpermute, no LDS.
#pragma once
#ifndef SVO_HOST_MATH_TEST  // (tests/host/hip_emu.h serves the cross-lane builtins when the kernels are compiled for the CPU tests)
#include <hip/hip_runtime.h>
#endif

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

// ---- the same for doubles (a 64-bit value is exchanged as two dwords) ----------------------------------
__device__ __forceinline__ double mk_f64(uint32_t lo, uint32_t hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lo32(double v) { return (uint32_t)__double_as_longlong(v); }
__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32); }

// lanes < 32: a[l] + a[l+32];  lanes >= 32: b[l-32] + b[l]
__device__ __forceinline__ double swap32_add(double a, double b) {
  auto s0 = __builtin_amdgcn_permlane32_swap(lo32(a), lo32(b), false, false);
  auto s1 = __builtin_amdgcn_permlane32_swap(hi32(a), hi32(b), false, false);
  return mk_f64(s0[0], s1[0]) + mk_f64(s0[1], s1[1]);
}
// even 16-lane rows: a[l] + a[l+16];  odd rows: b[l-16] + b[l]
__device__ __forceinline__ double swap16_add(double a, double b) {
  auto s0 = __builtin_amdgcn_permlane16_swap(lo32(a), lo32(b), false, false);
  auto s1 = __builtin_amdgcn_permlane16_swap(hi32(a), hi32(b), false, false);
  return mk_f64(s0[0], s1[0]) + mk_f64(s0[1], s1[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo32(v), CTRL, 0xf, 0xf, true);
  const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi32(v), CTRL, 0xf, 0xf, true);
  return mk_f64(l, h);
}

// wave_reduce8 for doubles: same exchange pattern, same result placement (group lane>>3 holds the total of v[g])
__device__ __forceinline__ double wave_reduce8(const double v[8], int lane) {
  double r[4], q[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = swap32_add(v[k], v[k + 4]);
#pragma unroll
  for (int k = 0; k < 2; ++k) q[k] = swap16_add(r[k], r[k + 2]);
  const bool hi = (lane & 8) != 0;
  const double mine = hi ? q[1] : q[0];
  const double send = hi ? q[0] : q[1];
  double t = mine + dpp_f64<DPP_ROW_ROR8>(send);
  t += dpp_f64<DPP_QUAD_XOR1>(t);
  t += dpp_f64<DPP_QUAD_XOR2>(t);
  t += dpp_f64<DPP_ROW_HALF_MIRROR>(t);
  return t;
}

}  // namespace svo_dev
