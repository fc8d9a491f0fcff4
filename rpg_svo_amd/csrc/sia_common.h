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
  // A frame split over several workgroups (sparse_align.hip, PARTS > 1): the number of frames (the grid is padded), the
  // exchange blocks (one SIA_X_BLOCK-chunk block per frame).  NULL / 0 otherwise.
  int B = 0;
  void* xw = nullptr;
};

// ---- the exchange between the workgroups of a split frame -------------------------------------------------------------
// One 16-byte chunk = a partial sum and the epoch of the exchange it belongs to, written and read as ONE access: data and
// flag arrive together, no counter, no atomic, no fence.  PLACEMENT-INDEPENDENT: a part stores its chunks write-through
// (sc1: the line leaves the XCD's L2 for memory) and every wave polls the other parts' chunks with sc1 loads (which miss
// the polling CU's vector L1 and are served below it) until they carry this exchange's epoch -- the agent-scope forms,
// correct wherever the dispatcher puts the four workgroups (it promises nothing; ids 8 apart usually share an XCD, which
// is a matter of speed only).  Round 6 first built this with sc0 stores kept in a shared L2: it never saw its siblings
// inside the real kernel although a microbenchmark of the same instructions did (profiles/r06f_*): placement is not a
// contract.  Each 8-byte half of a chunk carries the epoch, so that a reader can tell a torn 16-byte read (never
// observed on gfx950) from a whole one.  Epochs never repeat in a block: a frame's block ends with a word holding the last
// epoch used in it (zero when the block is new), every part reads it when it starts and part 0 writes it back when the
// frame is done, so a launch continues counting where the block's last user stopped and no chunk an earlier launch (or an
// earlier replay of a captured one) left behind can carry an epoch this launch waits for -- nothing is cleared per launch.
struct XChunk {
  unsigned lo, tag0, hi, tag1;
};
static_assert(sizeof(XChunk) == 16, "one 16-byte access");
constexpr int SIA_X_SLOTS = 16;   // chunks per part and buffer half of the per-iteration exchange (9 used: 8 sums + "changed")
constexpr int SIA_XH_SLOTS = 24;  // ... of the H exchange (21 used)
constexpr int SIA_X_MAX_PARTS = 4;
constexpr int SIA_X_CHUNKS = 2 * SIA_X_MAX_PARTS * (SIA_X_SLOTS + SIA_XH_SLOTS);  // exchange chunks per frame ...
constexpr int SIA_X_BLOCK = SIA_X_CHUNKS + 1;                                     // ... + the chunk that holds the block's last epoch

#ifndef SVO_HOST_MATH_TEST
__device__ __forceinline__ void sia_xstore(XChunk* p, double v, unsigned epoch) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  u4 w;
  w.x = (unsigned)b; w.y = epoch; w.z = (unsigned)(b >> 32); w.w = epoch;
  // The wait states behind the store are part of it: a VMEM store of more than 8 bytes reads its data registers for a few
  // cycles after issue, and the compiler's hazard recogniser does not look inside an asm statement -- it placed a VALU
  // write to the first data register pair right behind this store, and the quad of lanes the store had not read yet
  // (lanes 12-15 of 21) sent a chunk with a clobbered epoch that no reader ever accepted (profiles/r06h_*).
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ XChunk sia_xload(const XChunk* p) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  XChunk c;
  c.lo = w.x; c.tag0 = w.y; c.hi = w.z; c.tag1 = w.w;
  return c;
}
// the word at the end of a frame's block: the last epoch used in it
__device__ __forceinline__ void sia_xstore_epoch(XChunk* p, unsigned epoch) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 w;
  w.x = epoch; w.y = 0u; w.z = 0u; w.w = 0u;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(p), "v"(w) : "memory");
}
// the value of chunk p once both its halves carry `epoch`; NaN (and failed = 1) when they never do: a part that is not
// running (the launcher only splits grids that are resident at once) must not hang the device
__device__ __forceinline__ double sia_xpoll(const XChunk* p, unsigned epoch, int& failed) {
  XChunk c = sia_xload(p);
  unsigned spins = 0;
  while (c.tag0 != epoch || c.tag1 != epoch) {
    if (++spins > (1u << 16)) {
      failed = 1;
      return __longlong_as_double(0x7ff8000000000000ll);
    }
    __builtin_amdgcn_s_sleep(1);
    c = sia_xload(p);
  }
  return __longlong_as_double((long long)(((unsigned long long)c.hi << 32) | c.lo));
}
#endif

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

// 1 / z as v_rcp_f64 + two Newton steps (the IEEE division sequence is three times as long; K1 is a tolerance-mode kernel
// and its agreement with the reference's translation unit is the same to six digits either way: profiles/r06ah_*)
__device__ __forceinline__ double sia_rcp(double z) {
  double r = __builtin_amdgcn_rcp(z);
  r = fma(fma(-z, r, 1.0), r, r);
  r = fma(fma(-z, r, 1.0), r, r);
  return r;
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
