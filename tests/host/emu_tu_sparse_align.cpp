// TEST INFRASTRUCTURE: rpg_svo_amd/csrc/sparse_align.hip (K1) compiled for the host (tests/host/hip_emu.h); part of the
// emulated build of the C-ABI library that tests/emu_build.py links.
#include "hip_emu.h"
#define SIA_VCC_SELECT  // (sel_e64 in C instead of the v_cndmask_b32_e64 form)
#include "../../rpg_svo_amd/csrc/sparse_align.hip"

// Test-only entry: the wave-per-frame kernel (sparse_align_wave.hip) on a batch of any size -- svo_hip_sparse_align hands
// it batches of >= 1024 frames only, which is more than the emulation should be asked to run.
extern "C" int emu_sparse_align_wave(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B, const int32_t* d_ref_slot,
                                     const int32_t* d_cur_slot, const int32_t* d_n, int n_stride, const double* d_px,
                                     const double* d_xyz_ref, const uint8_t* d_valid, const svo_hip_sia_params* params,
                                     const double* d_T_in, double* d_T_out, double* d_H_out, int32_t* d_n_tracked, int32_t* d_iters,
                                     double* d_chi2, int32_t* d_status, void* stream) {
  SiaArgs args;
  const int rc = sia_prepare(layout, d_store, B, d_ref_slot, d_cur_slot, d_n, n_stride, d_px, d_xyz_ref, d_valid, params, d_T_in, d_T_out,
                             d_H_out, d_n_tracked, d_iters, d_chi2, d_status, &args);
  if (rc <= 0) return rc;
  if (!sia_wave_applies(args, 1024)) return SVO_HIP_EINVAL;
  return launch_sia_wave(args, B, static_cast<hipStream_t>(stream));
}
