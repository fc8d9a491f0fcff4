// TEST INFRASTRUCTURE: rpg_svo_amd/csrc/fast_detect.hip compiled for the host (tests/host/hip_emu.h); part of the emulated build of
// the C-ABI library that tests/emu_build.py links.
#include "hip_emu.h"
#include "../../rpg_svo_amd/csrc/fast_detect.hip"
