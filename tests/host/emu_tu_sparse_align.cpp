// TEST INFRASTRUCTURE: rpg_svo_amd/csrc/sparse_align.hip (K1) compiled for the host (tests/host/hip_emu.h); part of the
// emulated build of the C-ABI library that tests/emu_build.py links.  The wave-per-frame kernel (sparse_align_wave.hip) is
// not part of it: svo_hip_sparse_align always takes the workgroup-per-frame kernel here.
#include "hip_emu.h"
#define SIA_VCC_SELECT  // (sel_e64 in C instead of the v_cndmask_b32_e64 form)
#include "../../rpg_svo_amd/csrc/sparse_align.hip"
