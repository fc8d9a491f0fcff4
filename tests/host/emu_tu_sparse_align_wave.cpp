// TEST INFRASTRUCTURE: rpg_svo_amd/csrc/sparse_align_wave.hip (K1, the barrier-free form: a frame per wave, four lanes per
// patch at B = 1) compiled for the host (tests/host/hip_emu.h); part of the emulated build that tests/emu_build.py links.
#include "hip_emu.h"
#define SIA_VCC_SELECT  // (sel_e64 in C instead of the v_cndmask_b32_e64 form)
#include "../../rpg_svo_amd/csrc/sparse_align_wave.hip"
