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
// (The row-major store of rounds 1-2 -- row_off = y * pitch, col_off = x -- measured slower in every kernel that gathers
// windows: profiles/r03b_*.  svo_hip_pyr_layout::tile still names the layout and layout_ok() rejects anything but TILED.)
#pragma once
#include <stdint.h>

#ifndef SVO_HOST_MATH_TEST  // (the CPU tests compile the window loaders with the host compiler, see device_math.h)
#include <hip/hip_runtime.h>
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
  // == (y >> 3) * 8 * pitch + (y & 7) * 16, as one shift and two multiply-adds (v_mad_u32_u24)
  return mul24((uint32_t)(y >> 3), ((uint32_t)pitch << 3) - 128u) + ((uint32_t)y << 4);
}
__host__ __device__ __forceinline__ uint32_t col_off(int x) {
  // == (x >> 4) * 128 + (x & 15), as one shift and one multiply-add
  return mul24((uint32_t)(x >> 4), 112u) + (uint32_t)x;
}
__host__ __device__ __forceinline__ uint32_t px_off(int x, int y, int pitch) { return row_off(y, pitch) + col_off(x); }

// bytes one level occupies
__host__ __device__ __forceinline__ int64_t level_bytes(int pitch, int h) {
  return (int64_t)pitch * ((h + 7) & ~7);
}

// ---- window rows: three consecutive ALIGNED dwords ("a run": columns xa .. xa+11, xa % 4 == 0) -------------------
// What a gather costs is the number of cache-line look-ups the vector L1 performs for it: one per lane and load
// instruction (measured: K1 went from 1.30 to 1.55 ms when its 12-byte row fetches were split into three dword
// loads).  A run that lies inside one tile row (xa % 16 <= 4) is ONE 12-byte load; a run that continues in the next
// tile is two (8 + 4 or 4 + 8 bytes).  run_start() therefore places the run inside a tile row whenever the bytes the
// caller needs allow it.

// First column of the run fetched for a window row whose needed bytes are columns [c, c + need - 1] (need <= 9):
// 4-aligned, covers the needed bytes, and inside one tile row whenever the needed bytes are.
__device__ __forceinline__ int run_start(int c, int need) {
  int xa = c & ~3;
  if ((xa & 15) == 8 && (c & 15) + need <= 16) xa -= 4;  // [xa, xa+11] would cross, [c, c+need-1] does not
  return xa;
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
__device__ __forceinline__ uint32_t ld32(const uint8_t* __restrict__ lvl, uint32_t off) {
  return *reinterpret_cast<const uint32_t*>(lvl + off);
}
// 8 / 12 bytes from a 4-byte aligned address (global_load_dwordx2 / x3 need dword alignment only)
__device__ __forceinline__ u32x2_t ld64(const uint8_t* __restrict__ lvl, uint32_t off) {
  typedef u32x2_t __attribute__((aligned(4))) u32x2_a4;
  return *reinterpret_cast<const u32x2_a4*>(lvl + off);
}
__device__ __forceinline__ u32x3_t ld96(const uint8_t* __restrict__ lvl, uint32_t off) {
  typedef u32x3_t __attribute__((aligned(4))) u32x3_a4;
  return *reinterpret_cast<const u32x3_a4*>(lvl + off);
}

// The run [xa, xa+11] of the row at byte offset ro (row_off).  Per-row form: the three-way branch sits around one row.
__device__ __forceinline__ void load_run12(const uint8_t* __restrict__ lvl, uint32_t ro, int xa, uint32_t d[3]) {
  const uint32_t a = ro + col_off(xa);
  const int o = xa & 15;
  if (o <= 4) {  // one tile row
    const u32x3_t v = ld96(lvl, a);
    d[0] = v.x; d[1] = v.y; d[2] = v.z;
  } else if (o == 8) {  // bytes 8..15 here, 0..3 of the next tile (128 bytes further, 8 columns back)
    const u32x2_t v = ld64(lvl, a);
    d[0] = v.x; d[1] = v.y;
    d[2] = ld32(lvl, a + 120u);
  } else {  // bytes 12..15 here, 0..7 of the next tile
    d[0] = ld32(lvl, a);
    const u32x2_t v = ld64(lvl, a + 116u);
    d[1] = v.x; d[2] = v.y;
  }
}

// NR rows x one run from row v0, column xa of a level.  Window form: the three-way branch sits around all rows, so
// each case issues its NR (or 2 NR) loads back to back.
template <int NR>
__device__ __forceinline__ void load_window12(const uint8_t* __restrict__ lvl, int pitch, int xa, int v0,
                                              uint32_t d[][3]) {
  uint32_t a[NR];
  const uint32_t c = col_off(xa);
#pragma unroll
  for (int r = 0; r < NR; ++r) a[r] = row_off(v0 + r, pitch) + c;
  const int o = xa & 15;
  if (o <= 4) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const u32x3_t v = ld96(lvl, a[r]);
      d[r][0] = v.x; d[r][1] = v.y; d[r][2] = v.z;
    }
  } else if (o == 8) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const u32x2_t v = ld64(lvl, a[r]);
      d[r][0] = v.x; d[r][1] = v.y;
      d[r][2] = ld32(lvl, a[r] + 120u);
    }
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      d[r][0] = ld32(lvl, a[r]);
      const u32x2_t v = ld64(lvl, a[r] + 116u);
      d[r][1] = v.x; d[r][2] = v.y;
    }
  }
}

}  // namespace svo_pyr
