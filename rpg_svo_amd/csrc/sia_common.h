// sia_common.h -- pieces shared by the two K1 kernels (sparse_align.hip: one workgroup of several waves per
// problem; sparse_align_wave.hip: one wave per problem): the argument block, the unaligned row loads of the
// 8-bit pyramid levels and the window-cache cuts.
#pragma once
#include "capi_common.h"
#include "device_math.h"
#include "track_math.h"
#include "wave_reduce.h"

namespace svo_sia {

using namespace svo_capi;
using namespace svo_dev;

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

using svo_pyr::load_window12;
using svo_pyr::run_start;

// Window rows are fetched as 12-byte runs of three aligned dwords (pyr_addr.h: run_start, load_run12, load_window12)
// and cut to the bytes wanted in registers.

// bytes [bo, bo+6] (bo in 0..5) of three consecutive dwords as floats
__device__ __forceinline__ void cut_row7(const uint32_t d[3], uint32_t bo, float out[7]) {
  const bool up = bo >= 4u;
  const uint32_t a = up ? d[1] : d[0], b = up ? d[2] : d[1], c = up ? 0u : d[2];
  const uint32_t sel = bo & 3u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, sel);  // bytes 0..3
  const uint32_t hi = __builtin_amdgcn_alignbyte(c, b, sel);  // bytes 4..7
  out[0] = (float)(lo & 0xffu);
  out[1] = (float)((lo >> 8) & 0xffu);
  out[2] = (float)((lo >> 16) & 0xffu);
  out[3] = (float)(lo >> 24);
  out[4] = (float)(hi & 0xffu);
  out[5] = (float)((hi >> 8) & 0xffu);
  out[6] = (float)((hi >> 16) & 0xffu);
}

// bytes [bo, bo+4] (bo in 0..7) of three consecutive dwords as floats (the kernels without a window cache)
__device__ __forceinline__ void cut_row5_plain(const uint32_t d[3], uint32_t bo, float out[5]) {
  const bool up = bo >= 4u;
  const uint32_t a = up ? d[1] : d[0], b = up ? d[2] : d[1];
  const uint32_t sel = bo & 3u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, sel);  // bytes bo..bo+3
  const uint32_t hi = b >> (8 * sel);                         // byte bo+4 in bits 0..7
  out[0] = (float)(lo & 0xffu);
  out[1] = (float)((lo >> 8) & 0xffu);
  out[2] = (float)((lo >> 16) & 0xffu);
  out[3] = (float)(lo >> 24);
  out[4] = (float)(hi & 0xffu);
}

// ---- window cache (template parameter WC) --------------------------------------------------------------
// The 5x5 window of the current image moves by a fraction of a pixel per Gauss-Newton iteration,
// yet re-fetching it every iteration misses L2 (128 resident problems per XCD x ~64 KB of touched
// sectors) and puts an HBM round trip on the critical path of every iteration.  With the cache a
// lane fetches 7 rows x 3 aligned dwords around the patch (49+ bytes, once) and later iterations
// cut their 5x5 window out of those 21 registers as long as the integer position stays within
// +-1 row and the 12 cached columns; only then is nothing loaded at all.
// v_cndmask_b32 with the lane mask in an SGPR pair.  The VOP2 form the compiler prefers reads VCC and
// issues ~3.6x slower on gfx950 (scripts/valu_ubench.hip: 10.7 against 2.95 SIMD cycles per
// wave-instruction); K1 spends 45 selects per patch and iteration on its window cache.
__device__ __forceinline__ uint32_t sel_e64(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
#ifdef SIA_VCC_SELECT
  return ((mask >> (threadIdx.x & 63)) & 1) ? if_set : if_clear;
#else
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
#endif
}
// bytes [bo, bo+4] (bo in 0..7) of three consecutive dwords as floats; up = lanes with bo >= 4
__device__ __forceinline__ void cut_row5(uint32_t a, uint32_t b, uint32_t c, int bo, uint64_t up, float out[5]) {
  const uint32_t d0 = sel_e64(up, b, a), d1 = sel_e64(up, c, b);
  const uint32_t sel = (uint32_t)(bo & 3);
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sel);
  const uint32_t hi = d1 >> (8 * sel);
  out[0] = (float)(lo & 0xffu);
  out[1] = (float)((lo >> 8) & 0xffu);
  out[2] = (float)((lo >> 16) & 0xffu);
  out[3] = (float)(lo >> 24);
  out[4] = (float)(hi & 0xffu);
}

// a wave-uniform value read back through v_readfirstlane: its dwords stay in SGPRs instead of VGPRs
__device__ __forceinline__ double sia_uni(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}


__device__ __forceinline__ int sym6_rt(int i, int j) {
  const int a = i < j ? i : j, b = i < j ? j : i;
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

// sparse_align_wave.hip
int launch_sia_wave(const SiaArgs& args, int B, hipStream_t s);
bool sia_wave_applies(const SiaArgs& args, int B);

}  // namespace svo_sia
