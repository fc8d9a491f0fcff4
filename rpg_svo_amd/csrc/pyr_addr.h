// pyr_addr.h -- where pixel (x, y) of a pyramid level lives inside the level's bytes.
//
// The pyramid store is OURS (K0 writes it, nothing of the reference sees it), so its layout follows what the
// readers do: K1, K2, K3 and K5 gather small 2-D windows (7 rows x 12 bytes, 9 x 12, 8 x 12, 2 x 2) at
// scattered positions.  In a row-major level every row of such a window is its own 128-byte line (7-9 lines per
// window); here a level is cut into TILES of 16 bytes x 8 rows = 128 bytes = one L2 line / HBM request, tiles
// of one 8-row band consecutive, so a window touches 1.5 x 1.75 = 2.6 lines on average.
//
//   byte offset of (x, y) = row_off(y, pitch) + col_off(x)
//     row_off = (y >> 3) * 8 * pitch + (y & 7) * 16        pitch = width rounded up to 16 (bytes per image row)
//     col_off = (x >> 4) * 128 + (x & 15)
//
// The offset separates into a row term and a column term in either layout, which is all the kernels rely on; a
// run of bytes may be read with one load only if it does not cross a multiple of 16 in x (aligned dwords never do).
// -DSVO_PYR_ROWMAJOR builds the row-major store of rounds 1-2 (row_off = y * pitch, col_off = x) for A/B timing;
// svo_hip_pyr_layout::tile records which one a layout was made for and layout_ok() rejects the other.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

#ifdef SVO_PYR_ROWMAJOR
#define SVO_PYR_TILE 0
#else
#define SVO_PYR_TILE 1
#endif

namespace svo_pyr {

constexpr int TILE_W = 16, TILE_H = 8;

// a * b for factors below 2^24 (rows, columns, band strides): v_mul_u32_u24 / v_mad_u32_u24 on the device,
// where a full 32-bit multiply issues at quarter rate
__host__ __device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
  return __umul24(a, b);
#else
  return a * b;
#endif
}

__host__ __device__ __forceinline__ uint32_t row_off(int y, int pitch) {
#if SVO_PYR_TILE
  // == (y >> 3) * 8 * pitch + (y & 7) * 16, as one shift and two multiply-adds (v_mad_u32_u24)
  return mul24((uint32_t)(y >> 3), ((uint32_t)pitch << 3) - 128u) + ((uint32_t)y << 4);
#else
  return mul24((uint32_t)y, (uint32_t)pitch);
#endif
}
__host__ __device__ __forceinline__ uint32_t col_off(int x) {
#if SVO_PYR_TILE
  // == (x >> 4) * 128 + (x & 15), as one shift and one multiply-add
  return mul24((uint32_t)(x >> 4), 112u) + (uint32_t)x;
#else
  return (uint32_t)x;
#endif
}
__host__ __device__ __forceinline__ uint32_t px_off(int x, int y, int pitch) { return row_off(y, pitch) + col_off(x); }

// bytes one level occupies
__host__ __device__ __forceinline__ int64_t level_bytes(int pitch, int h) {
#if SVO_PYR_TILE
  return (int64_t)pitch * ((h + 7) & ~7);
#else
  return (int64_t)pitch * h;
#endif
}

// Three consecutive ALIGNED dwords of a row (columns xa .. xa+11, xa % 4 == 0): the column terms are the same
// for every row of a window, so a window load is 3 column terms + one row term per row + one add per dword.
struct Cols3 {
  uint32_t c0, c1, c2;
};
__device__ __forceinline__ Cols3 cols3(int xa) {
  Cols3 c;
  c.c0 = col_off(xa);
  c.c1 = col_off(xa + 4);
  c.c2 = col_off(xa + 8);
  return c;
}
__device__ __forceinline__ uint32_t ld32(const uint8_t* __restrict__ lvl, uint32_t off) {
  return *reinterpret_cast<const uint32_t*>(lvl + off);
}
__device__ __forceinline__ void load3(const uint8_t* __restrict__ lvl, uint32_t ro, const Cols3& c, uint32_t d[3]) {
  d[0] = ld32(lvl, ro + c.c0);
  d[1] = ld32(lvl, ro + c.c1);
  d[2] = ld32(lvl, ro + c.c2);
}

}  // namespace svo_pyr
