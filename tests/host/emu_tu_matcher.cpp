// TEST INFRASTRUCTURE: rpg_svo_amd/csrc/matcher.hip compiled for the host (tests/host/hip_emu.h)
#include "hip_emu.h"
#include "../../rpg_svo_amd/csrc/matcher.hip"
