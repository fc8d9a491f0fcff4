// map_mirror_emulated.cpp -- TEST INFRASTRUCTURE.  rpg_svo_amd/csrc/map_mirror.hip (row N2: reprojectMap on the resident
// map mirror) compiled for the CPU through tests/host/hip_emu.h: svo_hip_reproject_map, the C-ABI entry point itself, then
// runs its one-workgroup kernel with 1024 host threads.  tests/test_map_mirror_emulated.py calls it on host arrays and
// compares with the oracle, like the GPU test does on the device.
#include "hip_emu.h"

namespace svo_capi {
thread_local int g_last_hip_error = 0;
}

#include "../../rpg_svo_amd/csrc/map_mirror.hip"
