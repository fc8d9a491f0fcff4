// TEST INFRASTRUCTURE: what capi_util.hip defines for the C-ABI translation units, for the host-emulated builds
#include "hip_emu.h"
#include "capi_common.h"
namespace svo_capi {
thread_local int g_last_hip_error = 0;
}

