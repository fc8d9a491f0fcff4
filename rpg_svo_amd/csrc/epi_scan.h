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
#include "warp_group.h"
#include "seed_finish.h"

using namespace svo_dev;
using namespace svo_track;

// (an unnamed namespace, as in the translation unit these definitions come from: the kernels' symbols keep their names)
namespace {

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
constexpr int SCAN_G = 8;
// Positions a group looks at per pass on a line of more than SCAN_G positions: two per lane, CONSECUTIVE ones (lane l: steps 2l
// and 2l+1 of the pass); a shorter line takes one per lane in its only pass.
constexpr int SCAN_PP = 2 * SCAN_G;

// Seeds per workgroup of the scan.  The seeds of a chunk are ordered by scan length inside the workgroup (below).
#ifndef SCAN_CHUNK_VALUE
#define SCAN_CHUNK_VALUE 1024
#endif
constexpr int SCAN_CHUNK = SCAN_CHUNK_VALUE;
// ... of a batch of up to SCAN_SMALL_S seeds (a single camera frame: ~330): one round of the workgroup's 32 groups.  With
// the warp inside the scan kernel (round 6) a frame's seeds on ONE workgroup took 39.6 us, eleven rounds one after the
// other, where the separate warp and scan kernels of round 5 had taken 5 + 6.5 (profiles/r06l_dropin_frame_timeline_600.txt).
constexpr int SCAN_CHUNK_SMALL = 32, SCAN_SMALL_S = 8192;
constexpr int SCAN_BUCKETS = 8;

// The lanes of a group look at 16 consecutive positions of an epipolar line, 0.7 px apart: their 8 x 8 windows overlap
// almost completely, and what the scan costs is (a) the number of cache-line look-ups its gathers make and (b) the memory
// round trip a pass waits for.  The group therefore fetches the bounding box of its windows ONCE per pass, as 16-byte
// tile rows of the store (one look-up each), parks it in LDS and every lane cuts its windows out of that.  Round 5: the
// box covers 16 positions instead of 8 (half the round trips and half the box bookkeeping per position: the kernel
// waited 57 % of its wave cycles), is up to 20 rows x 3 tile columns (16 positions span 10.5 px: 19 rows on a vertical
// line, 19 columns on a horizontal one; at most 48 of the 60 tile rows are ever needed, six per lane) of 48 bytes (a row
// of the box is 12 dwords, which spreads the rows of a window over the banks -- rows r and r + 4 of the 8-dword rows of
// round 3 shared theirs), and the 8 x 8 template lives in LDS next to it instead of in 16 registers of every lane.  (16
// box rows and a second round for steeper lines measured only -5 %: a wave holds eight lines, nearly always a steep one
// among them, and every group of the wave then sits through both rounds.)  A pass whose windows do not fit (a distorted
// line) is done as two half passes, the box of lanes 0-3 and then the box of lanes 4-7; a group whose half box does not fit
// either fetches per lane.
// Round 6: the group also WARPS its seed's reference patch (warp_group.h) -- into the same box, before the scan needs it --
// and keeps the 10 x 10 patch in LDS: the template is cut from there, and the 100 bytes travel to HBM only for a seed that
// goes on to the sub-pixel alignment (warp_kernel's launch over the seeds, its 100-byte write and the scan's read of it
// are gone).
constexpr int SCAN_BOX_ROWS = WG_BOX_ROWS, SCAN_BOX_ROW_DWORDS = WG_ROW_DWORDS;
constexpr int SCAN_PATCH_OFF = WG_BOX_DWORDS;                          // box (+ 4 dwords: the cut reads three dwords per row)
constexpr int SCAN_TPL_OFF = SCAN_PATCH_OFF + WG_PATCH_DWORDS;         // ... the 10 x 10 patch_with_border
constexpr int SCAN_BOX_DWORDS = SCAN_TPL_OFF + 16;                     // ... the template's 8 rows of 2 dwords
static_assert(SCAN_BOX_DWORDS % 4 == 0 && SCAN_TPL_OFF % 4 == 0 && SCAN_PATCH_OFF % 4 == 0, "16-byte LDS stores");

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// minimum / maximum over the 4 lanes of a half group (a quad: two butterflies), then over the group (mirror inside 8 lanes)
__device__ __forceinline__ int quad_min(int v) {
  v = min(v, dpp_i32<svo_dev::DPP_QUAD_XOR1>(v));
  return min(v, dpp_i32<svo_dev::DPP_QUAD_XOR2>(v));
}
__device__ __forceinline__ int quad_max(int v) {
  v = max(v, dpp_i32<svo_dev::DPP_QUAD_XOR1>(v));
  return max(v, dpp_i32<svo_dev::DPP_QUAD_XOR2>(v));
}
__device__ __forceinline__ int group8_sum(int v) {
  v += dpp_i32<svo_dev::DPP_QUAD_XOR1>(v);
  v += dpp_i32<svo_dev::DPP_QUAD_XOR2>(v);
  return v + dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(v);
}

// the three sums of a ZMSSD over the 8 x 8 window whose top-left byte is (bx, row r0) of the box, against the template
struct ScanSums {
  uint32_t B, BB, AB;
};
__device__ __forceinline__ void scan_row(const uint32_t* __restrict__ r, const uint32_t sel, const uint32_t t0, const uint32_t t1, ScanSums& s) {
  const uint32_t d0 = r[0], d1 = r[1], d2 = r[2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sel), hi = __builtin_amdgcn_alignbyte(d2, d1, sel);
  s.B = __builtin_amdgcn_udot4(lo, 0x01010101u, s.B, false);
  s.B = __builtin_amdgcn_udot4(hi, 0x01010101u, s.B, false);
  s.BB = __builtin_amdgcn_udot4(lo, lo, s.BB, false);
  s.BB = __builtin_amdgcn_udot4(hi, hi, s.BB, false);
  s.AB = __builtin_amdgcn_udot4(lo, t0, s.AB, false);
  s.AB = __builtin_amdgcn_udot4(hi, t1, s.AB, false);
}

// cur_frame.cam_->world2cam(uv).  PINHOLE: the undistorted model on its own (the same two expressions as
// world2cam_uv's first branch), so that the instantiation the undistorted camera runs carries neither the code nor the
// registers of the radial-tangential and ATAN models.
template <bool PINHOLE>
__device__ __forceinline__ void scan_world2cam(const Cam& c, const double uv[2], double px[2]) {
  if (PINHOLE) {
    px[0] = c.fx * uv[0] + c.cx;
    px[1] = c.fy * uv[1] + c.cy;
  } else {
    world2cam_uv(c, uv, px);
  }
}

// SCAN_PROFILE (instrumentation build only, scripts/scan_phase_profile.py): shader-clock totals of every wave per region of
// epi_scan_seed -- 0 parameters + affine warp, 1 template + set-up, 2 a pass's positions (camera model, pixels, boxes),
// 3 box fetch, 4 scoring from the box, 5 per-lane fallback + scores + advancing the chain, 6 reduction + results,
// 7 wave iterations -- summed into g_scan_prof by the kernel.
#ifdef SCAN_PROFILE
#define SCAN_T() ((long long)__builtin_readcyclecounter())
// (one of the lanes that are active at the point adds the wave's delta to the wave's totals in LDS)
#define SCAN_ACC(k, t0, t1)                                                                                   \
  do {                                                                                                        \
    const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true);                                          \
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m_)) atomicAdd(prof + (k), (uint32_t)((t1) - (t0)));       \
  } while (0)
#define SCAN_PROF_PARAM , uint32_t* prof
#else
#define SCAN_T() 0ll
#define SCAN_ACC(k, t0, t1)
#define SCAN_PROF_PARAM
#endif

// ZMSSD scan of one seed, matcher.cpp:248-291, by the SCAN_G lanes of a group (lane = position in the group).
// box: the group's SCAN_BOX_DWORDS dwords of LDS (16-byte aligned).
template <bool PINHOLE>
__device__ __forceinline__ void epi_scan_seed(const SeedArgs& a, const int s, const int lane, uint32_t* box SCAN_PROF_PARAM) {
  const SeedWs& w = a.ws;
  [[maybe_unused]] long long tp0 = SCAN_T(), tp1;
  // everything the warp and the scan set-up read, requested together (seed_prepare_kernel wrote it)
  const int sl = w.search_level[s];
  const int mode = w.mode[s];
  const int cur_slot = w.cur_slot[s];
  const int ref_slot = w.ref_slot[s], ref_level = w.ref_level[s] & (SVO_HIP_MAX_LEVELS - 1);
  const float4 A = *reinterpret_cast<const float4*>(w.A_ref_cur + 4 * (size_t)s);
  const float2 pyr = *reinterpret_cast<const float2*>(w.px_ref_pyr + 2 * (size_t)s);
  // ---- warp::warpAffine of the seed's reference patch (matcher.cpp:221-224), into LDS ------------------------------------
  uint32_t* const patch = box + SCAN_PATCH_OFF;
  {
    const uint8_t* ref_img = a.store + (int64_t)ref_slot * a.L.slot_bytes + a.L.offset[ref_level];
    warp_patch_group8(ref_img, a.L.w[ref_level], a.L.h[ref_level], a.L.pitch[ref_level], A.x, A.y, A.z, A.w, pyr.x, pyr.y, sl, lane,
                      box, patch);
  }
  tp1 = SCAN_T(); SCAN_ACC(0, tp0, tp1); tp0 = tp1;
  if (mode != MODE_SCAN) {
    // a segment shorter than two pixels (matcher.cpp:226-238): straight to the sub-pixel alignment, which reads the patch
    warp_patch_store_group8(patch, lane, w.pwb + (size_t)s * 100);
    return;
  }
  // (the segment: requested here, used after the template's sums)
  const double step0 = w.step[2 * s], step1 = w.step[2 * s + 1];
  const double B0 = w.B[2 * s], B1 = w.B[2 * s + 1];
  const int n_total = w.n_steps[s] + 1;
  const uint8_t* img = a.store + (int64_t)cur_slot * a.L.slot_bytes + a.L.offset[sl];
  const int pitch = a.L.pitch[sl];
  // reference patch: interior of patch_with_border (createPatchFromPatchWithBorder), 8 rows of 8; lane y parks row y
  uint32_t* const tpl = box + SCAN_TPL_OFF;
  int sumA, sumAA;
  {
    // bytes 10 (y + 1) + 1 .. + 8 of the patch: three dwords and a byte shift
    const uint32_t b = 10u * (uint32_t)lane + 11u;
    const uint32_t* r = patch + (b >> 2);
    const uint32_t d0 = r[0], d1 = r[1], d2 = r[2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, b & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, b & 3u);
    tpl[2 * lane] = lo;
    tpl[2 * lane + 1] = hi;
    SVO_LANES_LDS_HANDOVER();
    uint32_t sa = __builtin_amdgcn_udot4(lo, 0x01010101u, 0u, false), saa = __builtin_amdgcn_udot4(lo, lo, 0u, false);
    sa = __builtin_amdgcn_udot4(hi, 0x01010101u, sa, false);
    saa = __builtin_amdgcn_udot4(hi, hi, saa, false);
    sumA = group8_sum((int)sa);
    sumAA = group8_sum((int)saa);
  }
  double uv0 = B0 - step0, uv1 = B1 - step1;
  int best = ZMSSD_THRESHOLD;
  int best_i = 0x7fffffff;
  double best_uv0 = 0, best_uv1 = 0;
  const double lvl = (double)(1 << sl);
  // Dividing by 2^level is exact, so multiplying by 2^-level gives the same bits without f64 division sequences.
  const double inv_lvl = pow2_inv_f64(sl);
  // The reference walks the line sequentially (matcher.cpp:268: uv += step, in f64) and skips a step whose
  // integer pixel equals the previous step's.  Lane l of the seed's group takes the steps 2l, 2l+1, 2l+16, 2l+17, ...: it
  // replays only the CHAIN of additions up to its steps (two v_add_f64 per step, so the positions carry the reference's
  // rounding) and does the expensive part -- camera model, rounding, the 8x8 ZMSSD -- for its own steps only, all lanes
  // of the group at once.  "Same pixel as the last step looked at" is "same pixel as step i-1": last_x/last_y are
  // overwritten by every step that differs from them, so they always hold step i-1's pixel.  The pixel of step i-1 is the
  // lane's own first pixel (for its second step), the second pixel of the lane to the left (the same chain of additions,
  // the same arithmetic: the same bits), and for the group's first lane the second pixel of its last lane in the pass
  // before: DPP moves, no second camera projection and no LDS.
  // A line of up to SCAN_G positions (half of the scanning seeds search 2-8) takes ONE position per lane -- lane l = step l,
  // no second position; the second position's work sits in branches of its own, which a wave whose eight lines are all
  // short jumps over (the seeds of a wave are neighbours in the order by length).  One code path for both: a second
  // instantiation next to this one cost nine spilled registers and two scratch round trips per seed (profiles/r05e_*).
  const bool two = n_total > SCAN_G;  // (uniform over the group)
  const int PP = two ? 2 : 1;
  for (int j = 0; j < PP * lane; ++j) {
    uv0 += step0; uv1 += step1;
  }
  int carry0 = 0, carry1 = 0;  // pixel of the last step of the pass before (last_x, last_y before the first step: 0, 0)
  tp1 = SCAN_T(); SCAN_ACC(1, tp0, tp1); tp0 = tp1;
  for (int base = 0; base < n_total; base += PP * SCAN_G) {
    const int ia = base + PP * lane, ib = ia + 1;
    const double ub0 = uv0 + step0, ub1 = uv1 + step1;  // the lane's second step
    int xa = 0, ya = 0, xb = 0, yb = 0;
    if (ia < n_total) {
      double pxs[2];
      const double uvs[2] = {uv0, uv1};
      scan_world2cam<PINHOLE>(a.cam, uvs, pxs);  // cur_frame.cam_->world2cam(uv), matcher.cpp:269
      xa = cast_int(pxs[0] * inv_lvl + 0.5);
      ya = cast_int(pxs[1] * inv_lvl + 0.5);
    }
    if (two && ib < n_total) {
      double pxs[2];
      const double uvs[2] = {ub0, ub1};
      scan_world2cam<PINHOLE>(a.cam, uvs, pxs);
      xb = cast_int(pxs[0] * inv_lvl + 0.5);
      yb = cast_int(pxs[1] * inv_lvl + 0.5);
    }
    bool want_a, want_b;  // the position is new (not the pixel of the step before) and its patch lies inside the frame
    {
      // (every lane takes part in the cross-lane moves; a lane whose step does not exist is never anybody's i-1;
      // row_shr:1 stays inside a row of 16 lanes, and the first lane of a group takes the carry instead)
      const int xl = two ? xb : xa, yl = two ? yb : ya;  // the lane's LAST pixel of the pass
      const int left0 = dpp_i32<0x111 /* row_shr:1 */>(xl), left1 = dpp_i32<0x111>(yl);
      const int prv0 = lane == 0 ? carry0 : left0, prv1 = lane == 0 ? carry1 : left1;
      want_a = ia < n_total && !(xa == prv0 && ya == prv1) && is_in_frame_level(a.cam, xa, ya, 8, sl);
      want_b = two && ib < n_total && !(xb == xa && yb == ya) && is_in_frame_level(a.cam, xb, yb, 8, sl);
      carry0 = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(xl);  // lane 0 <- lane 7 (the only lane that looks at it)
      carry1 = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(yl);
    }
    // bounding boxes of the windows [px-4, px+3]^2: of this lane's two, of its half group (a quad), of the group
    constexpr int BIG = 0x3fffffff;
    int x_lo = min(want_a ? xa - 4 : BIG, want_b ? xb - 4 : BIG), x_hi = max(want_a ? xa + 3 : -1, want_b ? xb + 3 : -1);
    int y_lo = min(want_a ? ya - 4 : BIG, want_b ? yb - 4 : BIG), y_hi = max(want_a ? ya + 3 : -1, want_b ? yb + 3 : -1);
    x_lo = quad_min(x_lo); x_hi = quad_max(x_hi); y_lo = quad_min(y_lo); y_hi = quad_max(y_hi);
    // (ox, oy: the bounds of the OTHER half group, from the mirror lane)
    const int ox_lo = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(x_lo), ox_hi = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(x_hi);
    const int oy_lo = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(y_lo), oy_hi = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(y_hi);
    const int gx_lo = min(x_lo, ox_lo), gx_hi = max(x_hi, ox_hi), gy_lo = min(y_lo, oy_lo), gy_hi = max(y_hi, oy_hi);
    // one box for the whole group if it fits (uniform over the group); else a box per half group, one after the other
    const bool whole = gy_hi - gy_lo < SCAN_BOX_ROWS && gx_hi - (gx_lo & ~15) < 48 &&
                       (gy_hi - gy_lo + 1) * (((gx_hi - (gx_lo & ~15)) >> 4) + 1) <= 48;
    tp1 = SCAN_T(); SCAN_ACC(2, tp0, tp1); tp0 = tp1;
    if (gx_hi < 0) {
      // nobody wants anything in this pass (uniform over the group)
    } else {
      ScanSums sa = {0, 0, 0}, sb = {0, 0, 0};
      bool scored = false;
      const int n_rounds = whole ? 1 : 2;
      for (int h = 0; h < n_rounds; ++h) {
        // bounds of this round's box: the group's, or (all lanes compute them alike) those of the half group h
        int bx_lo = gx_lo, bx_hi = gx_hi, by_lo = gy_lo, by_hi = gy_hi;
        if (!whole) {
          const bool own = (lane >> 2) == h;
          bx_lo = own ? x_lo : ox_lo; bx_hi = own ? x_hi : ox_hi; by_lo = own ? y_lo : oy_lo; by_hi = own ? y_hi : oy_hi;
        }
        const bool mine = whole || (lane >> 2) == h;
        const int cx0 = bx_lo & ~15;                 // first tile column
        const int n_rows = by_hi - by_lo + 1;
        const int n_cols = ((bx_hi - cx0) >> 4) + 1;  // tile columns
        const bool boxed = bx_hi >= 0 && n_rows <= SCAN_BOX_ROWS && n_cols <= 3 && n_rows * n_cols <= 48;  // (uniform over the group)
        if (bx_hi < 0) continue;  // nothing wanted in this half
        if (!boxed) continue;     // (left to the per-lane path below)
        // the box: 16-byte tile rows, chunk c = row * n_cols + column, lane l takes the chunks l, l + 8, ... (at most 48:
        // a box of many rows is narrow) -- all requested, then all parked
        const int n_chunks = n_rows * n_cols;
        const uint32_t inv = n_cols == 1 ? 65536u : (n_cols == 2 ? 32768u : 21846u);  // c / n_cols for c < 128
        uint4 v[6];
        int dst[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int c = lane + 8 * k;
          const int row = (int)(((uint32_t)c * inv) >> 16), cc = c - row * n_cols;
          dst[k] = row * SCAN_BOX_ROW_DWORDS + cc * 4;
          v[k] = make_uint4(0, 0, 0, 0);
          if (c < n_chunks) v[k] = *reinterpret_cast<const uint4*>(img + (svo_pyr::row_off(by_lo + row, pitch) + svo_pyr::col_off(cx0 + 16 * cc)));
        }
        // (the reads of the round before are done: DS operations of one wave execute in order)
        SVO_LANES_LDS_HANDOVER();
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (lane + 8 * k < n_chunks) *reinterpret_cast<uint4*>(box + dst[k]) = v[k];
        // hand-over inside the wave
        SVO_LANES_LDS_HANDOVER();
        tp1 = SCAN_T(); SCAN_ACC(3, tp0, tp1); tp0 = tp1;
        // (each position in a branch of its own: a wave in which no lane has a second position jumps over that block)
        const uint2* t2 = reinterpret_cast<const uint2*>(tpl);
        if (mine && (want_a || want_b)) scored = true;
        if (mine && want_a) {
          const int ba = xa - 4 - cx0;  // first byte of the window inside the 48-byte box row
          const uint32_t sela = (uint32_t)(ba & 3);
          const uint32_t* ra = box + (ya - 4 - by_lo) * SCAN_BOX_ROW_DWORDS + (ba >> 2);
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            const uint2 t = t2[y];
            scan_row(ra + SCAN_BOX_ROW_DWORDS * y, sela, t.x, t.y, sa);
          }
        }
        if (mine && want_b) {
          const int bb = xb - 4 - cx0;
          const uint32_t selb = (uint32_t)(bb & 3);
          const uint32_t* rb = box + (yb - 4 - by_lo) * SCAN_BOX_ROW_DWORDS + (bb >> 2);
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            const uint2 t = t2[y];
            scan_row(rb + SCAN_BOX_ROW_DWORDS * y, selb, t.x, t.y, sb);
          }
        }
        tp1 = SCAN_T(); SCAN_ACC(4, tp0, tp1); tp0 = tp1;
      }
      if (!scored && (want_a || want_b)) {
        // 8 rows x 8 bytes [px-4, px+3]: 12-byte runs, inside one tile row of the store where the 8 bytes are
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bool wq = q == 0 ? want_a : want_b;
          if (!wq) continue;
          const int px = q == 0 ? xa : xb, py = q == 0 ? ya : yb;
          ScanSums& sq = q == 0 ? sa : sb;
          const int wxa = svo_pyr::run_start(px - 4, 8);
          const uint32_t wbo = (uint32_t)(px - 4 - wxa);  // 0..4
          uint32_t win[8][3];
          svo_pyr::load_window12<8>(img, pitch, wxa, py - 4, win);
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            uint32_t lo, hi;
            cut_row8(win[y], wbo, lo, hi);
            const uint32_t t0 = tpl[2 * y], t1 = tpl[2 * y + 1];
            sq.B = __builtin_amdgcn_udot4(lo, 0x01010101u, sq.B, false);
            sq.B = __builtin_amdgcn_udot4(hi, 0x01010101u, sq.B, false);
            sq.BB = __builtin_amdgcn_udot4(lo, lo, sq.BB, false);
            sq.BB = __builtin_amdgcn_udot4(hi, hi, sq.BB, false);
            sq.AB = __builtin_amdgcn_udot4(lo, t0, sq.AB, false);
            sq.AB = __builtin_amdgcn_udot4(hi, t1, sq.AB, false);
          }
        }
      }
      // the lane's steps come in increasing order: it keeps its first minimum
      if (want_a) {
        const int sB = (int)sa.B, sBB = (int)sa.BB, sAB = (int)sa.AB;
        const int zmssd = sumAA - 2 * sAB + sBB - (sumA * sumA - 2 * sumA * sB + sB * sB) / 64;
        if (zmssd < best) {
          best = zmssd; best_i = ia; best_uv0 = uv0; best_uv1 = uv1;
        }
      }
      if (want_b) {
        const int sB = (int)sb.B, sBB = (int)sb.BB, sAB = (int)sb.AB;
        const int zmssd = sumAA - 2 * sAB + sBB - (sumA * sumA - 2 * sumA * sB + sB * sB) / 64;
        if (zmssd < best) {
          best = zmssd; best_i = ib; best_uv0 = ub0; best_uv1 = ub1;
        }
      }
    }
    if (base + PP * SCAN_G < n_total) {  // on to this lane's next step(s): PP * SCAN_G more additions
      for (int j = 0; j < PP * SCAN_G; ++j) {
        uv0 += step0; uv1 += step1;
      }
    }
    tp1 = SCAN_T(); SCAN_ACC(5, tp0, tp1); tp0 = tp1;
  }
  // first strictly smaller score along the line == lexicographic minimum of (score, step)
  unsigned long long key = ((unsigned long long)(unsigned)best << 32) | (unsigned)best_i;
  {  // minimum over the seed's eight lanes: two butterflies inside the quads, then the mirror lane (DPP moves, no LDS)
    auto other = [&](const int which) {
      const int lo = (int)(unsigned)key, hi = (int)(unsigned)(key >> 32);
      int olo, ohi;
      if (which == 0) { olo = dpp_i32<svo_dev::DPP_QUAD_XOR1>(lo); ohi = dpp_i32<svo_dev::DPP_QUAD_XOR1>(hi); }
      else if (which == 1) { olo = dpp_i32<svo_dev::DPP_QUAD_XOR2>(lo); ohi = dpp_i32<svo_dev::DPP_QUAD_XOR2>(hi); }
      else { olo = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(lo); ohi = dpp_i32<svo_dev::DPP_ROW_HALF_MIRROR>(hi); }
      return ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo;
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long o = other(k);
      key = o < key ? o : key;
    }
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
      scan_world2cam<PINHOLE>(a.cam, uvb, pcs);  // px_cur_ = cur_frame.cam_->world2cam(uv_best), matcher.cpp:297,316
    }
    const double pc0 = pcs[0], pc1 = pcs[1];
    w.px_cur[2 * s] = pc0;
    w.px_cur[2 * s + 1] = pc1;
    w.px_scaled[2 * s] = pc0 * inv_lvl;  // == pc0 / lvl, bit for bit (a power of two)
    w.px_scaled[2 * s + 1] = pc1 * inv_lvl;
    // matcher.cpp:293-318: refine with align1D/2D, or triangulate straight from uv_best.  The
    // second case is flagged in its own array: the alignment kernel clears ok[] of inactive trials.
    w.align_active[s] = a.opt.subpix_refinement ? 1 : 0;
    w.accepted_raw[s] = a.opt.subpix_refinement ? 0 : 1;
  }
  // (every lane holds the group's minimum) a match that goes on to align1D / align2D: its patch_with_border to HBM
  if (win_score < ZMSSD_THRESHOLD && a.opt.subpix_refinement) warp_patch_store_group8(patch, lane, w.pwb + (size_t)s * 100);
  if (lane == 0 && !(win_score < ZMSSD_THRESHOLD)) w.status[s] = SVO_HIP_SEED_NO_MATCH;
  tp1 = SCAN_T(); SCAN_ACC(6, tp0, tp1);
}

}  // namespace
