// pyramid_emulated.cpp -- TEST INFRASTRUCTURE.  rpg_svo_amd/csrc/pyramid.hip (K0 / row N1: the image pyramid built in one
// fused pass into the tiled store) compiled for the CPU through tests/host/hip_emu.h; its C-ABI entry points then run
// their kernels with host threads.  tests/test_pyramid_emulated.py compares the levels with the oracle's halfSample.
#include "hip_emu.h"

namespace svo_capi {
thread_local int g_last_hip_error = 0;
}

#include "../../rpg_svo_amd/csrc/pyramid.hip"
