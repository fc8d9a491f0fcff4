// epi_scan.h -- the depth filter's workspace layout and the ZMSSD scan along the epipolar line (matcher.cpp:248-291) of ONE
// seed by a group of SCAN_LANES lanes, as epi_scan_kernel (depth_filter.hip) runs it.  Also compiled for the CPU by the
// test suite: there the cross-lane moves are served by a small SIMT emulation (tests/host/hip_emu.h: one fiber per lane,
// tests/test_scan_emulated.py).
// (Requesting the NEXT pass's box before this pass is scored -- 16 registers in flight, three waves per SIMD -- measured
// 8 % slower, and 42 % slower held to four waves: profiles/r05a_queue_drain.txt.)
#pragma once
#include "track_math.h"
#include "pyr_addr.h"
#include "matcher_device.h"
#include "seed_math.h"
#include "wave_reduce.h"

using namespace svo_dev;
using namespace svo_track;

// (an unnamed namespace, as in the translation unit these definitions come from: the kernels' symbols keep their names)
namespace {

constexpr int ZMSSD_THRESHOLD = 2000 * 64;  // vk::patch_score::ZMSSD<4>::threshold()

enum : int { MODE_NONE = 0, MODE_SHORT = 1, MODE_SCAN = 2 };

struct SeedWs {
  uint8_t* warp_active;   // [S]
  uint8_t* align_active;  // [S]
  uint8_t* use_1d;        // [S]
  int32_t* status;        // [S] preliminary status (0 = still running)
  int32_t* mode;          // [S]
  int32_t* ref_slot;
  int32_t* ref_level;
  int32_t* cur_slot;
  int32_t* search_level;
  int32_t* n_steps;
  float* A_ref_cur;   // [S][4]
  float* px_ref_pyr;  // [S][2]
  float* dir;         // [S][2]
  float* z_inv_min;   // [S]
  double* B;          // [S][2] epipolar start (unit plane)
  double* step;       // [S][2]
  double* px_scaled;  // [S][2] align start, level coordinates
  double* px_cur;     // [S][2] Matcher::px_cur_
  double* uv_best;    // [S][2]
  uint8_t* pwb;       // [S][100]
  int32_t* align_ok;  // [S] written by the alignment kernel only
  uint8_t* accepted_raw;  // [S] 1: scan match accepted without sub-pixel refinement (subpix_refinement == false)
};

struct SeedArgs {
  svo_hip_pyr_layout L;
  const uint8_t* store;
  Cam cam;
  int S;
  const int32_t* frame_slot;
  const double* frame_T;
  const int32_t* cur_frame;
  svo_hip_features ftr;
  svo_hip_seeds seeds;
  svo_hip_depth_filter_options opt;
  int32_t* status_out;
  double* xyz_world;
  double* px_cur_out;
  // Matcher::findEpipolarMatchDirect on its own (svo_hip_find_epipolar_match_direct): the depth interval
  // is given, nothing of DepthFilter::updateSeeds runs around it
  int match_only;
  const double* d_est;
  const double* d_min;
  const double* d_max;
  double* depth_out;
  int32_t* ok_out;
  int32_t* search_level_out;
  double px_error_angle;  // atan(1 / (2 |fx|)) * 2
  TauConsts tau_k;         // its sines and cosines (seed_math.h)
  SeedWs ws;
};

// bytes [bo, bo+7] (bo in 0..7) of a 12-byte run of three aligned dwords, as two dwords
__device__ __forceinline__ void cut_row8(const uint32_t d[3], uint32_t bo, uint32_t& lo, uint32_t& hi) {
  const bool up = bo >= 4u;  // then bo == 4: the run was moved one dword to the left to stay inside a tile row
  const uint32_t a = up ? d[1] : d[0], b = up ? d[2] : d[1], c = up ? 0u : d[2];
  const uint32_t sel = bo & 3u;
  lo = __builtin_amdgcn_alignbyte(b, a, sel);
  hi = __builtin_amdgcn_alignbyte(c, b, sel);
}

constexpr int SCAN_BLOCK = 256;
// Lanes per seed in the epipolar scan.  A line has a few tens of steps: with a whole wave per seed most
// lanes had no step of their own, and every lane still replays the chain of additions.  8 lanes per seed
// share the chain replay between eight seeds of a wave and keep the lanes busy (update_seeds on 3.3 M seeds:
// 3.24 ms at 64 lanes per seed, 2.74 at 32, 2.47 at 16, 2.33 at 8).
constexpr int SCAN_LANES = 8;
constexpr int SCAN_G = SCAN_LANES;
static_assert(SCAN_G == 8, "the box fetch and the DPP minima below are written for groups of 8 lanes");

// Seeds per workgroup of the scan.  The seeds of a chunk are ordered by scan length inside the workgroup (below).
constexpr int SCAN_CHUNK = 1024;
constexpr int SCAN_BUCKETS = 8;

// The 8 lanes of a group look at 8 consecutive positions of an epipolar line, 0.7 px apart: their 8 x 8 windows
// overlap almost completely, and what the scan costs is the number of cache-line look-ups its gathers make (8 rows x
// 1-2 look-ups per position, ~12; the arithmetic of a position is ~100 instructions).  With SCAN_BOX the group fetches
// the bounding box of its windows ONCE, as 16-byte tile rows of the store (one look-up each, at most 16 rows x 2
// columns of tiles = 4 per lane), parks it in LDS and every lane cuts its window out of that; a group whose windows do
// not fit a 16 x 32 box (never on an undistorted line) falls back to fetching per lane.
constexpr int SCAN_BOX_DWORDS = 16 * 8 + 4;  // 16 rows x 32 bytes (+ one dword: the cut reads three dwords per row)

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// minimum / maximum over the 8 lanes of a group (quad butterflies + mirror inside the 8 lanes: no LDS)
__device__ __forceinline__ int group8_min(int v) {
  v = min(v, dpp_i32<svo_dev::DPP_QUAD_XOR1>(v));
  v = min(v, dpp_i32<svo_dev::DPP_QUAD_XOR2>(v));
  return min(v, dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(v));
}
__device__ __forceinline__ int group8_max(int v) {
  v = max(v, dpp_i32<svo_dev::DPP_QUAD_XOR1>(v));
  v = max(v, dpp_i32<svo_dev::DPP_QUAD_XOR2>(v));
  return max(v, dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(v));
}

// ZMSSD scan of one seed, matcher.cpp:248-291, by the SCAN_LANES lanes of a group (lane = position in the group).
// box: the group's SCAN_BOX_DWORDS dwords of LDS.
__device__ __forceinline__ void epi_scan_seed(const SeedArgs& a, const int s, const int lane, uint32_t* box) {
  const SeedWs& w = a.ws;
  const int sl = w.search_level[s];
  const uint8_t* img = a.store + (int64_t)w.cur_slot[s] * a.L.slot_bytes + a.L.offset[sl];
  const int pitch = a.L.pitch[sl];
  // reference patch: interior of patch_with_border (createPatchFromPatchWithBorder), 8 rows of 8
  uint32_t ra[16];
  {
    const uint8_t* pw = w.pwb + (size_t)s * 100;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const uint8_t* r = pw + (y + 1) * 10 + 1;
      // one unaligned 8-byte load per row (gfx950 global memory takes any alignment)
      uint32_t lo, hi;
      __builtin_memcpy(&lo, r, 4);
      __builtin_memcpy(&hi, r + 4, 4);
      ra[2 * y] = lo;
      ra[2 * y + 1] = hi;
    }
  }
  uint32_t sumA_u = 0, sumAA_u = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    sumA_u = __builtin_amdgcn_udot4(ra[k], 0x01010101u, sumA_u, false);
    sumAA_u = __builtin_amdgcn_udot4(ra[k], ra[k], sumAA_u, false);
  }
  const int sumA = (int)sumA_u, sumAA = (int)sumAA_u;
  const double step0 = w.step[2 * s], step1 = w.step[2 * s + 1];
  double uv0 = w.B[2 * s] - step0, uv1 = w.B[2 * s + 1] - step1;
  const int n_total = w.n_steps[s] + 1;
  int best = ZMSSD_THRESHOLD;
  int best_i = 0x7fffffff;
  double best_uv0 = 0, best_uv1 = 0;
  const double lvl = (double)(1 << sl);
  // Dividing by 2^level is exact, so multiplying by 2^-level gives the same bits without f64 division sequences.
  const double inv_lvl = 1.0 / lvl;
  // The reference walks the line sequentially (matcher.cpp:268: uv += step, in f64) and skips a step whose
  // integer pixel equals the previous step's.  Lane l of the seed's group takes the steps l, l+SCAN_G, ...: it replays only the CHAIN of
  // additions up to its step (two v_add_f64 per step, so the positions carry the reference's rounding), keeps the
  // position of the step before, and does the expensive part -- camera model, rounding, the 8x8 ZMSSD -- for its
  // own steps only, all lanes of the group at once.  "Same pixel as the last step looked at" is "same pixel as step i-1":
  // last_x/last_y are overwritten by every step that differs from them, so they always hold step i-1's pixel.
  // The pixel of step i-1 is the pixel the lane to the left computed in this pass (the same chain of additions, the same
  // arithmetic: the same bits), and for the group's first lane the pixel its last lane computed in the pass before:
  // one DPP move per coordinate instead of a second camera projection and rounding per position (round 4).
  {
    const int lead = lane < n_total ? lane : n_total;
    for (int j = 0; j < lead; ++j) {
      uv0 += step0; uv1 += step1;
    }
  }
  int carry0 = 0, carry1 = 0;  // pixel of the last step of the pass before (last_x, last_y before the first step: 0, 0)
  for (int base = 0; base < n_total; base += SCAN_G) {
    const int i = base + lane;
    bool want = false;  // this lane's position is new (not the pixel of the step before) and its patch lies inside the frame
    int pxi0 = 0, pxi1 = 0;
    if (i < n_total) {
      double pxs[2];
      const double uvs[2] = {uv0, uv1};
      world2cam_uv(a.cam, uvs, pxs);  // cur_frame.cam_->world2cam(uv), matcher.cpp:269
      pxi0 = cast_int(pxs[0] * inv_lvl + 0.5);
      pxi1 = cast_int(pxs[1] * inv_lvl + 0.5);
    }
    {
      // (every lane takes part in the cross-lane moves; a lane whose step does not exist is never anybody's i-1)
      // (row_shr:1 stays inside a row of 16 lanes: groups of more lanes -- experimental builds -- take a shuffle)
      const int left0 = SCAN_G <= 16 ? __builtin_amdgcn_update_dpp(0, pxi0, 0x111 /* row_shr:1 */, 0xf, 0xf, true) : __shfl_up(pxi0, 1, 64);
      const int left1 = SCAN_G <= 16 ? __builtin_amdgcn_update_dpp(0, pxi1, 0x111, 0xf, 0xf, true) : __shfl_up(pxi1, 1, 64);
      const int prv0 = lane == 0 ? carry0 : left0, prv1 = lane == 0 ? carry1 : left1;
      want = i < n_total && !(pxi0 == prv0 && pxi1 == prv1) && is_in_frame_level(a.cam, pxi0, pxi1, 8, sl);
      const int last = (int)(threadIdx.x & 63u & ~(unsigned)(SCAN_G - 1)) + SCAN_G - 1;  // the group's last lane
      carry0 = __shfl(pxi0, last, 64);
      carry1 = __shfl(pxi1, last, 64);
    }
    uint32_t sumB = 0, sumBB = 0, sumAB = 0;
    bool boxed = false;
    {
      // bounding box of the group's windows [px-4, px+3]^2 (every lane of the group takes part, wanted or not)
      const int x_lo = group8_min(want ? pxi0 - 4 : 0x7fffffff), x_hi = group8_max(want ? pxi0 + 3 : -1);
      const int y_lo = group8_min(want ? pxi1 - 4 : 0x7fffffff), y_hi = group8_max(want ? pxi1 + 3 : -1);
      const int cx0 = x_lo & ~15;                 // first tile column
      const int n_rows = y_hi - y_lo + 1;
      const bool two = x_hi - cx0 >= 16;          // a second tile column
      boxed = x_hi >= 0 && n_rows <= 16 && x_hi - cx0 < 32;  // (uniform over the group)
      if (boxed) {
        const int n_chunks = two ? 2 * n_rows : n_rows;  // 16-byte tile rows to fetch: <= 32, four per lane
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = lane + 8 * k;
          const int row = two ? (c >> 1) : c, cc = two ? (c & 1) : 0;
          v[k] = make_uint4(0, 0, 0, 0);
          if (c < n_chunks) v[k] = *reinterpret_cast<const uint4*>(img + (svo_pyr::row_off(y_lo + row, pitch) + svo_pyr::col_off(cx0 + 16 * cc)));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = lane + 8 * k;
          const int row = two ? (c >> 1) : c, cc = two ? (c & 1) : 0;
          if (c < n_chunks) *reinterpret_cast<uint4*>(box + row * 8 + cc * 4) = v[k];
        }
        // hand-over inside the wave: DS operations of one wave execute in order
        SVO_LANES_LDS_HANDOVER();
        if (want) {
          const int bx = pxi0 - 4 - cx0;             // 0..24: first byte of the window inside the 32-byte box row
          const uint32_t sel = (uint32_t)(bx & 3);
          const uint32_t* r = box + (pxi1 - 4 - y_lo) * 8 + (bx >> 2);
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            const uint32_t d0 = r[8 * y], d1 = r[8 * y + 1], d2 = r[8 * y + 2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sel), hi = __builtin_amdgcn_alignbyte(d2, d1, sel);
            sumB = __builtin_amdgcn_udot4(lo, 0x01010101u, sumB, false);
            sumB = __builtin_amdgcn_udot4(hi, 0x01010101u, sumB, false);
            sumBB = __builtin_amdgcn_udot4(lo, lo, sumBB, false);
            sumBB = __builtin_amdgcn_udot4(hi, hi, sumBB, false);
            sumAB = __builtin_amdgcn_udot4(lo, ra[2 * y], sumAB, false);
            sumAB = __builtin_amdgcn_udot4(hi, ra[2 * y + 1], sumAB, false);
          }
        }
        SVO_LANES_LDS_HANDOVER();
      }
    }
    if (want) {
      if (!boxed) {
        // 8 rows x 8 bytes [pxi0-4, pxi0+3]: 12-byte runs, inside one tile row of the store where the 8 bytes are
        const int wxa = svo_pyr::run_start(pxi0 - 4, 8);
        const uint32_t wbo = (uint32_t)(pxi0 - 4 - wxa);  // 0..4
        uint32_t win[8][3];
        svo_pyr::load_window12<8>(img, pitch, wxa, pxi1 - 4, win);
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          uint32_t lo, hi;
          cut_row8(win[y], wbo, lo, hi);
          sumB = __builtin_amdgcn_udot4(lo, 0x01010101u, sumB, false);
          sumB = __builtin_amdgcn_udot4(hi, 0x01010101u, sumB, false);
          sumBB = __builtin_amdgcn_udot4(lo, lo, sumBB, false);
          sumBB = __builtin_amdgcn_udot4(hi, hi, sumBB, false);
          sumAB = __builtin_amdgcn_udot4(lo, ra[2 * y], sumAB, false);
          sumAB = __builtin_amdgcn_udot4(hi, ra[2 * y + 1], sumAB, false);
        }
      }
      const int sB = (int)sumB, sBB = (int)sumBB, sAB = (int)sumAB;
      const int zmssd = sumAA - 2 * sAB + sBB - (sumA * sumA - 2 * sumA * sB + sB * sB) / 64;
      if (zmssd < best) {  // the lane's steps come in increasing order: keeps its first minimum
        best = zmssd;
        best_i = i;
        best_uv0 = uv0;
        best_uv1 = uv1;
      }
    }
    if (base + SCAN_G < n_total) {  // on to this lane's next step: SCAN_G more additions
      for (int j = 0; j < SCAN_G; ++j) {
        uv0 += step0; uv1 += step1;
      }
    }
  }
  // first strictly smaller score along the line == lexicographic minimum of (score, step)
  unsigned long long key = ((unsigned long long)(unsigned)best << 32) | (unsigned)best_i;
#pragma unroll
  for (int off = SCAN_G / 2; off > 0; off >>= 1) {  // within the seed's lane group
    const unsigned long long o = __shfl_xor(key, off, 64);
    key = o < key ? o : key;
  }
  const int win_score = (int)(key >> 32);
  const int win_i = (int)(key & 0xffffffffu);
  if (win_score < ZMSSD_THRESHOLD && win_i == best_i && best < ZMSSD_THRESHOLD) {
    // exactly one lane owns step win_i
    w.uv_best[2 * s] = best_uv0;
    w.uv_best[2 * s + 1] = best_uv1;
    double pcs[2];
    {
      const double uvb[2] = {best_uv0, best_uv1};
      world2cam_uv(a.cam, uvb, pcs);  // px_cur_ = cur_frame.cam_->world2cam(uv_best), matcher.cpp:297,316
    }
    const double pc0 = pcs[0], pc1 = pcs[1];
    w.px_cur[2 * s] = pc0;
    w.px_cur[2 * s + 1] = pc1;
    w.px_scaled[2 * s] = pc0 / lvl;
    w.px_scaled[2 * s + 1] = pc1 / lvl;
    // matcher.cpp:293-318: refine with align1D/2D, or triangulate straight from uv_best.  The
    // second case is flagged in its own array: the alignment kernel clears ok[] of inactive trials.
    w.align_active[s] = a.opt.subpix_refinement ? 1 : 0;
    w.accepted_raw[s] = a.opt.subpix_refinement ? 0 : 1;
  }
  if (lane == 0 && !(win_score < ZMSSD_THRESHOLD)) w.status[s] = SVO_HIP_SEED_NO_MATCH;
}


}  // namespace
