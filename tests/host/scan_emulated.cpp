// scan_emulated.cpp -- TEST INFRASTRUCTURE.  The epipolar ZMSSD scan of epi_scan_kernel (rpg_svo_amd/csrc/epi_scan.h) run on
// the CPU through tests/host/hip_emu.h: tests/test_scan_emulated.py compares what it writes with a sequential numpy scan.
#include "hip_emu.h"

#include <vector>

#include "epi_scan.h"

extern "C" {

// Warps and scans seeds [0, S) (`form` is unused: there is one form of the scan).  One level-`n_levels` store of one slot;
// workspace arrays as SeedWs names them (only what the scan kernel reads and writes).
int scan_emulated(int form, int S, const uint8_t* store, long long slot_bytes, int n_levels, const long long* level_offset,
                  const int* level_w, const int* level_h, const int* level_pitch, const double cam_k[4], int width, int height,
                  int subpix_refinement, const int32_t* search_level, const int32_t* cur_slot, const int32_t* n_steps, const double* B,
                  const double* step, uint8_t* pwb, const float* A_ref_cur, const float* px_ref_pyr, const int32_t* ref_slot,
                  const int32_t* ref_level, const int32_t* mode, double* uv_best, double* px_cur, double* px_scaled, uint8_t* align_active,
                  uint8_t* accepted_raw, int32_t* status) {
  SeedArgs a;
  std::memset(&a, 0, sizeof(a));
  for (int l = 0; l < n_levels && l < SVO_HIP_MAX_LEVELS; ++l) {
    a.L.offset[l] = level_offset[l];
    a.L.w[l] = level_w[l];
    a.L.h[l] = level_h[l];
    a.L.pitch[l] = level_pitch[l];
  }
  a.L.n_levels = n_levels;
  a.L.slot_bytes = slot_bytes;
  a.store = store;
  a.cam.fx = cam_k[0]; a.cam.fy = cam_k[1]; a.cam.cx = cam_k[2]; a.cam.cy = cam_k[3];
  a.cam.width = width; a.cam.height = height; a.cam.model = SVO_HIP_CAM_PINHOLE;
  a.S = S;
  a.opt.subpix_refinement = subpix_refinement;
  a.ws.search_level = const_cast<int32_t*>(search_level);
  a.ws.cur_slot = const_cast<int32_t*>(cur_slot);
  a.ws.n_steps = const_cast<int32_t*>(n_steps);
  a.ws.B = const_cast<double*>(B);
  a.ws.step = const_cast<double*>(step);
  a.ws.pwb = pwb;  // written: the warped patch of a seed that goes on to the alignment
  a.ws.A_ref_cur = const_cast<float*>(A_ref_cur);
  a.ws.px_ref_pyr = const_cast<float*>(px_ref_pyr);
  a.ws.ref_slot = const_cast<int32_t*>(ref_slot);
  a.ws.ref_level = const_cast<int32_t*>(ref_level);
  a.ws.mode = const_cast<int32_t*>(mode);
  a.ws.uv_best = uv_best;
  a.ws.px_cur = px_cur;
  a.ws.px_scaled = px_scaled;
  a.ws.align_active = align_active;
  a.ws.accepted_raw = accepted_raw;
  a.ws.status = status;
  // a wave of 8 groups at a time, as epi_scan_kernel hands seeds to a wave: group g scans seed s0 + g
  for (int s0 = 0; s0 < S; s0 += 8) {
    const int n_groups = S - s0 < 8 ? S - s0 : 8;
    std::vector<std::vector<uint32_t>> boxes(8, std::vector<uint32_t>(SCAN_BOX_DWORDS + 8, 0u));
    svo_emu::launch(dim3(1), dim3(64), [&] {
      const int grp = (int)threadIdx.x / 8, lane = (int)threadIdx.x % 8;
      if (grp >= n_groups) return;
      (void)form;
      epi_scan_seed<true>(a, s0 + grp, lane, boxes[grp].data());
    });
  }
  return 0;
}

}  // extern "C"
