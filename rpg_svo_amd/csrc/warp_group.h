// warp_group.h -- warp::warpAffine (matcher.cpp:72-105), the 10 x 10 patch_with_border_ of ONE trial, by a GROUP OF 8 LANES.
//
// Rounds 2-5 gave a trial 10 lanes (lane = output column, 6 trials per wave, 60 of 64 lanes) in a kernel of its own whose
// 100 bytes went to HBM and were read back by the epipolar scan and by the alignment.  Per wave iteration that kernel
// spent as many instructions around the samples -- parameters, the box of the source region, the fetch's address
// arithmetic, assembling and storing the output -- as on the 60 samples per lane group themselves (ISA: ~470 + ~150
// against ~355 of ~1050).  Here 8 lanes own a trial: 8 trials per wave share that fixed part, every lane is busy, and
// the patch is produced IN LDS, where the depth filter's scan -- the same 8 lanes per seed (epi_scan.h) -- takes its
// 8 x 8 template from without a round trip through HBM; the matcher's warp_kernel (matcher.hip) uses the same function
// and stores the patch as the C ABI's patch_with_border.
//
// Lane l computes column l of the patch down its ten rows, then its share of columns 8 and 9 (20 samples over 8 lanes:
// sample e = l, l + 8, l + 16 of the 20, column 8 + e / 10, row e % 10): 13 sample slots for 12.5 samples per lane.
// The source region -- the bounding box of the four corner samples, at most WG_BOX_ROWS rows of 48 bytes -- is fetched
// once as 16-byte tile rows (six per lane) into LDS rows of 12 dwords; a trial whose box is larger (strong down-scaling,
// rotation) gathers its samples from the store instead.  Arithmetic per sample: warp_sample.h, bit for bit the reference's.
#pragma once
#include "track_math.h"
#include "pyr_addr.h"
#include "warp_sample.h"

namespace svo_track {

constexpr int WG_LANES = 8;
constexpr int WG_BOX_ROWS = 20, WG_ROW_DWORDS = 12;
constexpr int WG_BOX_DWORDS = WG_BOX_ROWS * WG_ROW_DWORDS + 4;  // (+ 4: readers cut 12-byte runs out of a row)
constexpr int WG_PATCH_DWORDS = 28;                             // 100 bytes, padded to 16
static_assert(WG_BOX_DWORDS % 4 == 0 && WG_PATCH_DWORDS % 4 == 0, "16-byte LDS stores");

// (x, y) of the lane's k-th sample slot, k = 0..12; false when the slot is empty (k = 12 of lanes 4..7)
__device__ __forceinline__ bool wg_slot(const int lane, const int k, int& x, int& y) {
  if (k < 10) {
    x = lane; y = k;
    return true;
  }
  const int e = lane + 8 * (k - 10);  // 0..19 (+ 4 empty)
  x = e >= 10 ? 9 : 8;
  y = e >= 10 ? e - 10 : e;
  return e < 20;
}

// img / cols / rows / pitch: the reference level; A: A_ref_cur row-major; pyr: px_ref / 2^level_ref; slev: search level.
// region: WG_BOX_DWORDS dwords of LDS (16-byte aligned), patch: WG_PATCH_DWORDS dwords of LDS, both the group's own.
// On return (after the hand-over inside) every lane of the group may read the 100 bytes at `patch`.
__device__ __forceinline__ void warp_patch_group8(const uint8_t* __restrict__ img, const int cols, const int rows, const int pitch,
                                                  const float Ax, const float Ay, const float Az, const float Aw, const float pyrx,
                                                  const float pyry, const int slev, int lane, uint32_t* const region,
                                                  uint32_t* const patch) {
  // (opaque: what the lane index turns into -- (float)(lane - 5), the extra slots' coordinates and byte offsets, chunk
  // indices: ~25 registers -- is invariant in the caller's loop over trials, where the compiler would park it all)
  asm volatile("" : "+v"(lane));
  uint8_t* const pb = reinterpret_cast<uint8_t*>(patch);
  const float sc = (float)(1 << slev);
  // The 100 samples lie in the parallelogram spanned by the four corner samples (the map is affine and every rounding in
  // it is monotone, so the extremes ARE the corners).
  float bx0 = 3.0e38f, bx1 = -3.0e38f, by0 = 3.0e38f, by1 = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float pp0 = (float)((k & 1) ? 4 : -5), pp1 = (float)((k & 2) ? 4 : -5);
    pp0 *= sc;
    pp1 *= sc;
    const float q0 = (Ax * pp0 + Ay * pp1) + pyrx;
    const float q1 = (Az * pp0 + Aw * pp1) + pyry;
    bx0 = fminf(bx0, q0); bx1 = fmaxf(bx1, q0);
    by0 = fminf(by0, q1); by1 = fmaxf(by1, q1);
  }
  // 0: zeros ("Affine warp is NaN": the reference leaves the previous patch in place; or every sample outside the image),
  // 1: samples from the LDS copy of the box, 2: samples gathered from the store
  int how = 2;
  int xlo = 0, ylo = 0, cx0 = 0, nch = 0, nrow = 0;
  if (isnan(Ax)) {
    how = 0;
  } else if (bx0 > -1.0e6f && bx1 < 1.0e6f && by0 > -1.0e6f && by1 < 1.0e6f && bx0 <= bx1 && by0 <= by1) {
    // (comparisons are false for NaN: such a trial is not boxed)
    int xhi = (int)floorf(bx1) + 1, yhi = (int)floorf(by1) + 1;
    xlo = max((int)floorf(bx0), 0); ylo = max((int)floorf(by0), 0);
    xhi = min(xhi, cols - 1); yhi = min(yhi, rows - 1);
    cx0 = xlo & ~15;
    nch = xhi >= cx0 ? ((xhi - cx0) >> 4) + 1 : 0;
    nrow = yhi - ylo + 1;
    if (xhi < xlo || yhi < ylo) how = 0;
    else if (nch <= 3 && nrow <= WG_BOX_ROWS && nrow * nch <= 6 * WG_LANES) how = 1;
  }
  // (the group's earlier readers of `region` and `patch` are done: DS operations of a wave execute in order)
  SVO_LANES_LDS_HANDOVER();
  if (how == 1) {
    // the box: 16-byte tile rows, chunk c = row * nch + column, lane l takes the chunks l, l + 8, ...: all requested, then
    // all parked (a loop of load-then-store pays one memory round trip per chunk)
    const int n_chunks = nrow * nch;
    const uint32_t inv = nch == 1 ? 65536u : (nch == 2 ? 32768u : 21846u);  // c / nch for c < 128
    uint4 v[6];
    int dst[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int c = lane + WG_LANES * k;
      const int row = (int)(((uint32_t)c * inv) >> 16), cc = c - row * nch;
      dst[k] = row * WG_ROW_DWORDS + cc * 4;
      v[k] = make_uint4(0, 0, 0, 0);
      if (c < n_chunks) v[k] = *reinterpret_cast<const uint4*>(img + (svo_pyr::row_off(ylo + row, pitch) + svo_pyr::col_off(cx0 + 16 * cc)));
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (lane + WG_LANES * k < n_chunks) *reinterpret_cast<uint4*>(region + dst[k]) = v[k];
    SVO_LANES_LDS_HANDOVER();
    // The samples lie between the corner samples, so when the box of the corners is inside the image every sample is: the
    // usual trial skips the four comparisons and three selects per sample (same values: `in` would be true everywhere).
    const bool all_in = bx0 >= 0.f && by0 >= 0.f && bx1 < (float)(cols - 1) && by1 < (float)(rows - 1);
    const uint8_t* const reg_o = reinterpret_cast<const uint8_t*>(region) - (__mul24(ylo, 48) + cx0);  // pixel (xi, yi) is reg_o[48 yi + xi]
    if (all_in) {
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        int x, y;
        if (wg_slot(lane, k, x, y)) pb[y * 10 + x] = warp_sample<false>(Ax, Ay, Az, Aw, pyrx, pyry, sc, x, y, cols, rows, xlo, ylo, reg_o);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        int x, y;
        if (wg_slot(lane, k, x, y)) pb[y * 10 + x] = warp_sample<true>(Ax, Ay, Az, Aw, pyrx, pyry, sc, x, y, cols, rows, xlo, ylo, reg_o);
      }
    }
  } else if (how == 2) {
    // a box too large for the LDS copy: the four pixels of a sample straight from the tiled store (rare: not unrolled)
    for (int k = 0; k < 13; ++k) {
      int x, y;
      if (!wg_slot(lane, k, x, y)) continue;
      float pp0 = (float)(x - 5), pp1 = (float)(y - 5);
      pp0 *= sc;
      pp1 *= sc;
      const float px0 = (Ax * pp0 + Ay * pp1) + pyrx;
      const float px1 = (Az * pp0 + Aw * pp1) + pyry;
      const bool in = !(px0 < 0 || px1 < 0 || px0 >= (float)(cols - 1) || px1 >= (float)(rows - 1));
      uint8_t o = 0;
      if (in) {
        const int xi = (int)floorf(px0), yi = (int)floorf(px1);
        const float sx = px0 - (float)xi, sy = px1 - (float)yi;
        const uint32_t rt = svo_pyr::row_off(yi, pitch), rb = svo_pyr::row_off(yi + 1, pitch);
        const uint32_t cl = svo_pyr::col_off(xi), cr = svo_pyr::col_off(xi + 1);
        o = (uint8_t)warp_blend(sx, sy, (float)img[rt + cl], (float)img[rt + cr], (float)img[rb + cl], (float)img[rb + cr]);
      }
      pb[y * 10 + x] = o;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      int x, y;
      if (wg_slot(lane, k, x, y)) pb[y * 10 + x] = 0;
    }
  }
  SVO_LANES_LDS_HANDOVER();
}

// the group's 25 patch dwords to 100 bytes of global memory (4-byte aligned): lane l stores dwords l, l + 8, l + 16 and,
// lane 0, dword 24
__device__ __forceinline__ void warp_patch_store_group8(const uint32_t* const patch, const int lane, uint8_t* const dst) {
  uint32_t* const d = reinterpret_cast<uint32_t*>(dst);
  const uint32_t a = patch[lane], b = patch[lane + 8], c = patch[lane + 16], e = patch[24];
  d[lane] = a;
  d[lane + 8] = b;
  d[lane + 16] = c;
  if (lane == 0) d[24] = e;
}

}  // namespace svo_track
