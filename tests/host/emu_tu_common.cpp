// TEST INFRASTRUCTURE: what capi_util.hip defines for the C-ABI translation units, for the host-emulated builds
#include "hip_emu.h"
#include "capi_common.h"
namespace svo_capi {
thread_local int g_last_hip_error = 0;
}

// The wave-per-frame form of K1 (sparse_align_wave.hip) is not part of the emulated build: svo_hip_sparse_align takes the
// workgroup-per-frame kernel for every batch here.
#include "sia_common.h"
namespace svo_sia {
bool sia_wave_applies(const SiaArgs&, int) { return false; }
int launch_sia_wave(const SiaArgs&, int, hipStream_t) { return SVO_HIP_EINVAL; }
}  // namespace svo_sia
