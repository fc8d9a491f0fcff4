// capi_common.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#ifndef SVO_HOST_MATH_TEST  // (the CPU tests compile kernels for the host through tests/host/hip_emu.h)
#include <hip/hip_runtime.h>
#endif

#include "pyr_addr.h"
#include "svo_hip.h"

namespace svo_capi {

// last raw HIP error seen by any entry point (per host thread: the library may
// be driven from the tracking and the mapping thread concurrently)
extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = static_cast<int>(e);
  return SVO_HIP_EHIP;
}

#define SVO_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return ::svo_capi::hip_fail(_e);  \
  } while (0)

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e);
  return SVO_HIP_OK;
}

inline bool layout_ok(const svo_hip_pyr_layout* L) {
  if (!L || L->n_levels < 1 || L->n_levels > SVO_HIP_MAX_LEVELS) return false;
  if (L->tile != SVO_HIP_PYR_TILED) return false;  // the only layout the kernels address (pyr_addr.h)
  for (int i = 0; i < L->n_levels; ++i)
    if (L->w[i] < 1 || L->h[i] < 1 || L->pitch[i] < L->w[i] || (L->pitch[i] & 15) || (L->offset[i] & 127)) return false;
  return L->slot_bytes > 0;
}

// XCD-aware workgroup order.  Workgroup ids are dealt round-robin to the 8 XCDs of an MI355X, each
// with its own L2.  Work items that are neighbours in memory (the trials of one frame, the
// problems that share a frame) are neighbours in the batch, so every XCD is handed a CONTIGUOUS
// eighth of the grid instead of every eighth workgroup.  Identity when the grid is not a multiple of 8.
__device__ __forceinline__ unsigned xcd_contiguous_block() {
  const unsigned nb = gridDim.x, bid = blockIdx.x;
  return (nb % 8u == 0u) ? (bid % 8u) * (nb / 8u) + bid / 8u : bid;
}

}  // namespace svo_capi
