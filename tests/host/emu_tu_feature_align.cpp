// TEST INFRASTRUCTURE: rpg_svo_amd/csrc/feature_align.hip compiled for the host (tests/host/hip_emu.h).  The phased
// alignment (three launches, survivors compacted through queues in between) starts at 2048 trials here instead of 65536,
// so that the emulated tests reach it.
#include "hip_emu.h"
#define ALIGN_PHASE_MIN_M_VALUE 2048
#include "../../rpg_svo_amd/csrc/feature_align.hip"
